// Training kernels of the try-on condition generator (train_condition.py:113-286):
// batch-statistics BatchNorm forward/backward (networks.py:171-198 with tocg.train()),
// the adjoints of the bilinear resize and of the fused flow warp (F.interpolate /
// F.grid_sample backward, networks.py:130-152), grid_sample with an explicit grid
// (train_condition.py:237-245), channel softmax, 2-D cross entropy (utils.py:29-42) and
// the flow total-variation term (train_condition.py:190-199).  All HBM-bound, gfx950.
#include "hrv_common.h"

namespace hrv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int grid_for(size_t work, int block = 256) {
  size_t g = (work + block - 1) / block;
  const size_t cap = 256 * 16;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// ---------------------------------------------------------------------------
// BatchNorm2d, training mode.  Per-sample (mean, rstd) come from
// hrv_instnorm_stats_nhwc_f32 (eps_in); this folds them over the batch:
//   mean_c = avg_n mean_nc ; var_c = avg_n (var_nc + mean_nc^2) - mean_c^2   (biased, N*H*W)
// and emits the per-channel affine (scale, shift) of y = gamma*(x-mean)*rstd + beta, plus the
// running-statistics update (momentum, unbiased variance) of nn.BatchNorm2d.
// ---------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* __restrict__ mean_nc, const float* __restrict__ rstd_nc, int N, int C,
                                   int Cs, float eps_in, long long HW, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ scale,
                                   float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double sm = 0.0, sq = 0.0;
  for (int n = 0; n < N; ++n) {
    const double m = (double)mean_nc[(size_t)n * Cs + c];
    const double r = (double)rstd_nc[(size_t)n * Cs + c];
    const double v = 1.0 / (r * r) - (double)eps_in;
    sm += m;
    sq += v + m * m;
  }
  const double m = sm / N;
  double var = sq / N - m * m;
  if (var < 0.0) var = 0.0;
  const double rs = 1.0 / sqrt(var + (double)eps);
  const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)rs;
  scale[c] = (float)(g * rs);
  shift[c] = (float)(b - m * g * rs);
  if (running_mean) {
    const double cnt = (double)N * (double)HW;
    const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
  }
}

// out = act(x*scale[c] + shift[c] (+ residual))
__global__ void affine_act_kernel(const float* __restrict__ x, int xcs, int xco, int C4, size_t npix,
                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                  const float* __restrict__ res, int rcs, int rco, int act, float slope,
                                  float* __restrict__ out, int ocs, int oco) {
  const size_t total = npix * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % C4);
    const size_t pix = i / C4;
    f32x4 v = ld4(x + pix * xcs + xco + g * 4);
    const f32x4 sc = ld4(scale + g * 4), sh = ld4(shift + g * 4);
    v = v * sc + sh;
    if (res) v += ld4(res + pix * rcs + rco + g * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act, slope);
    *reinterpret_cast<f32x4*>(out + pix * ocs + oco + g * 4) = v;
  }
}

// out = x * m   (dropout: m holds 0 or 1/(1-p); also its backward d *= m)
__global__ void mul_kernel(const float* __restrict__ x, const float* __restrict__ m, size_t n4, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    *reinterpret_cast<f32x4*>(out + i * 4) = ld4(x + i * 4) * ld4(m + i * 4);
}

// BatchNorm backward, stage 1: per block partial sums over its pixel range of
//   s1[c] = sum dy, s2[c] = sum dy * (x - mean[c]) * rstd[c]
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, int dcs, int dco,
                                                             const float* __restrict__ x, int xcs, int xco, int P,
                                                             int C4, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, int NB,
                                                             float* __restrict__ part) {
  __shared__ f32x4 red1[256];
  __shared__ f32x4 red2[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const int PB = (P + NB - 1) / NB;
  const int p0 = b * PB, p1 = min(p0 + PB, P);
  const int GB = C4 < 256 ? C4 : 256;
  const int R = 256 / GB;
  const int r = t / GB, gl = t - r * GB;
  for (int g0 = 0; g0 < C4; g0 += GB) {
    const int g = g0 + gl;
    f32x4 s1 = (f32x4)(0.f), s2 = (f32x4)(0.f);
    if (r < R && g < C4) {
      const f32x4 m = ld4(mean + g * 4), rs = ld4(rstd + g * 4);
      // four independent pixel chains per thread (eight 16-byte loads in flight): one chain per thread on <= 256 blocks kept ~2 MB
      // in flight and ran at 0.50 of the achievable HBM rate; fixed summation order (a + b) + (c + d)
      f32x4 a1 = (f32x4)(0.f), a2 = (f32x4)(0.f), b1 = (f32x4)(0.f), b2 = (f32x4)(0.f), c1 = (f32x4)(0.f), c2 = (f32x4)(0.f);
      int px = p0 + r;
      for (; px + 3 * R < p1; px += 4 * R) {
        const f32x4 d0 = ld4(dy + (size_t)px * dcs + dco + g * 4), x0 = ld4(x + (size_t)px * xcs + xco + g * 4);
        const f32x4 d1 = ld4(dy + (size_t)(px + R) * dcs + dco + g * 4), x1 = ld4(x + (size_t)(px + R) * xcs + xco + g * 4);
        const f32x4 d2 = ld4(dy + (size_t)(px + 2 * R) * dcs + dco + g * 4), x2 = ld4(x + (size_t)(px + 2 * R) * xcs + xco + g * 4);
        const f32x4 d3 = ld4(dy + (size_t)(px + 3 * R) * dcs + dco + g * 4), x3 = ld4(x + (size_t)(px + 3 * R) * xcs + xco + g * 4);
        s1 += d0; s2 += d0 * ((x0 - m) * rs);
        a1 += d1; a2 += d1 * ((x1 - m) * rs);
        b1 += d2; b2 += d2 * ((x2 - m) * rs);
        c1 += d3; c2 += d3 * ((x3 - m) * rs);
      }
      for (; px < p1; px += R) {
        const f32x4 d = ld4(dy + (size_t)px * dcs + dco + g * 4);
        const f32x4 xv = ld4(x + (size_t)px * xcs + xco + g * 4);
        s1 += d;
        s2 += d * ((xv - m) * rs);
      }
      s1 = (s1 + a1) + (b1 + c1);
      s2 = (s2 + a2) + (b2 + c2);
    }
    red1[t] = s1;
    red2[t] = s2;
    __syncthreads();
    if (r == 0 && g < C4) {
      for (int rr = 1; rr < R; ++rr) {
        s1 += red1[rr * GB + gl];
        s2 += red2[rr * GB + gl];
      }
      *reinterpret_cast<f32x4*>(part + ((size_t)b * 2 * C4 + g) * 4) = s1;
      *reinterpret_cast<f32x4*>(part + ((size_t)b * 2 * C4 + C4 + g) * 4) = s2;
    }
    __syncthreads();
  }
}

// stage 2: fixed-order double-precision sum of the partials -> sums[0..Cp) = s1, sums[Cp..2Cp) = s2;
// dbeta (+)= s1, dgamma (+)= s2
__global__ void bn_bwd_final_kernel(const float* __restrict__ part, int NB, int C, int Cp, float* __restrict__ sums,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cp) return;
  double a = 0.0, b = 0.0;
  for (int i = 0; i < NB; ++i) {
    a += (double)part[(size_t)i * 2 * Cp + c];
    b += (double)part[(size_t)i * 2 * Cp + Cp + c];
  }
  sums[c] = (float)a;
  sums[Cp + c] = (float)b;
  if (c < C) {
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)a : (float)a;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)b : (float)b;
  }
}

// stage 3: dx = gamma*rstd * (dy - s1/M - xhat * s2/M), gamma*rstd = the forward's `scale`
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, int dcs, int dco, const float* __restrict__ x, int xcs,
                                    int xco, size_t npix, int C4, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ scale,
                                    const float* __restrict__ sums, float invM, float* __restrict__ dx, int ocs, int oco,
                                    int accumulate) {
  const size_t total = npix * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % C4);
    const size_t pix = i / C4;
    const f32x4 d = ld4(dy + pix * dcs + dco + g * 4);
    const f32x4 xv = ld4(x + pix * xcs + xco + g * 4);
    const f32x4 m = ld4(mean + g * 4), rs = ld4(rstd + g * 4);
    const f32x4 s1 = ld4(sums + g * 4), s2 = ld4(sums + (size_t)C4 * 4 + g * 4);
    const f32x4 sc = ld4(scale + g * 4);
    const f32x4 xh = (xv - m) * rs;
    f32x4 v = sc * (d - s1 * invM - xh * (s2 * invM));
    float* o = dx + pix * ocs + oco + g * 4;
    if (accumulate) v += ld4(o);
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

// ---------------------------------------------------------------------------
// Bilinear resize (align_corners=False) adjoint, gather form (deterministic):
// dx[n,i,j,c] (+)= sum over the output pixels (o,q) whose two source taps include (i,j).
// ---------------------------------------------------------------------------
struct LinB {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ LinB lin_src_b(int dst, int in_size, float r) {
  float s = ((float)dst + 0.5f) * r - 0.5f;
  s = s < 0.f ? 0.f : s;
  int i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  LinB o;
  o.i0 = i0;
  o.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  float l1 = s - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
  o.l1 = l1;
  o.l0 = 1.f - l1;
  return o;
}
// candidate output range [lo, hi] whose taps may touch input index i
__device__ __forceinline__ void out_range(int i, int in_size, int out_size, float r, int& lo, int& hi) {
  // src(o) = (o+0.5)*r - 0.5 in (i-1, i+1)  <=>  o in ((i-0.5)/r - 0.5, (i+1.5)/r - 0.5)
  const float a = ((float)i - 0.5f) / r - 0.5f, b = ((float)i + 1.5f) / r - 0.5f;
  lo = (int)floorf(a) - 1;
  hi = (int)ceilf(b) + 1;
  if (i == 0) lo = 0;                 // clamped sources (src < 0) all land on row 0
  if (i >= in_size - 1) hi = out_size - 1;
  lo = lo < 0 ? 0 : lo;
  hi = hi > out_size - 1 ? out_size - 1 : hi;
}

template <int VEC>
__global__ void resize_bwd_kernel(const float* __restrict__ dy, int N, int H, int W, int Cg, int Ho, int Wo, int dcs,
                                  int dco, float rh, float rw, float* __restrict__ dx, int xcs, int xco, int accumulate) {
  // Cg = channel groups (C/4 when VEC==4, else C)
  const size_t total = (size_t)N * H * W * Cg;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(idx % Cg);
    const size_t pix = idx / Cg;
    const int j = (int)(pix % W);
    const size_t t = pix / W;
    const int i = (int)(t % H);
    const int n = (int)(t / H);
    int olo, ohi, qlo, qhi;
    out_range(i, H, Ho, rh, olo, ohi);
    out_range(j, W, Wo, rw, qlo, qhi);
    f32x4 acc = (f32x4)(0.f);
    for (int o = olo; o <= ohi; ++o) {
      const LinB ly = lin_src_b(o, H, rh);
      const float wy = (ly.i0 == i ? ly.l0 : 0.f) + (ly.i1 == i ? ly.l1 : 0.f);
      if (wy == 0.f) continue;
      for (int q = qlo; q <= qhi; ++q) {
        const LinB lx = lin_src_b(q, W, rw);
        const float wx = (lx.i0 == j ? lx.l0 : 0.f) + (lx.i1 == j ? lx.l1 : 0.f);
        if (wx == 0.f) continue;
        const float* s = dy + (((size_t)n * Ho + o) * Wo + q) * dcs + dco + g * VEC;
        if (VEC == 4) acc += ld4(s) * (wy * wx);
        else acc[0] += s[0] * (wy * wx);
      }
    }
    float* d = dx + pix * xcs + xco + g * VEC;
    if (VEC == 4) {
      if (accumulate) acc += ld4(d);
      *reinterpret_cast<f32x4*>(d) = acc;
    } else {
      d[0] = accumulate ? d[0] + acc[0] : acc[0];
    }
  }
}

// ---------------------------------------------------------------------------
// Adjoint of hrv_flow_warp_nhwc_f32 through grid_sample(bilinear, border, align_corners=False):
// d_src (atomic scatter of the 4 taps) and d_flow_up[n,ho,wo,:] = d(grid)/norm, where the
// coordinate gradient is zero wherever the sample position was clamped to the border
// (torch clip_coordinates_set_grad).  One 16-lane group per output pixel, 4 per wave.
// ---------------------------------------------------------------------------
struct WarpBwdParams {
  const float* src;
  int N, H, W, C4, scs, sco;
  const float* flow_up;  // [N,Ho,Wo,2] un-normalised flow at the output resolution
  int Ho, Wo;
  float norm_x, norm_y, step_x, step_y;
  const float* dout;
  int dcs, dco;
  float* dsrc;  // or null
  int gcs, gco;
  float* dflow;  // [N,Ho,Wo,2] or null
  int dflow_accumulate;
};

__device__ __forceinline__ float lin_m1_1b(int i, int n, float step) {
  // torch.linspace(-1, 1, n): symmetric evaluation from both ends (same as sample.hip)
  return i < n / 2 ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}

__global__ __launch_bounds__(256) void flow_warp_bwd_kernel(const WarpBwdParams p) {
  const int lane16 = threadIdx.x & 15;
  const size_t npix = (size_t)p.N * p.Ho * p.Wo;
  const size_t ngroups = ((size_t)gridDim.x * blockDim.x) >> 4;
  for (size_t pix = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; pix < ((npix + 3) & ~(size_t)3) ;
       pix += ngroups) {
    const bool live = pix < npix;
    const size_t pp = live ? pix : 0;
    const int wo = (int)(pp % p.Wo);
    const size_t t = pp / p.Wo;
    const int ho = (int)(t % p.Ho);
    const int n = (int)(t / p.Ho);
    const float fx = p.flow_up[pp * 2], fy = p.flow_up[pp * 2 + 1];
    const float gx = fx / p.norm_x + lin_m1_1b(wo, p.Wo, p.step_x);
    const float gy = fy / p.norm_y + lin_m1_1b(ho, p.Ho, p.step_y);
    float ix = ((gx + 1.f) * (float)p.W - 1.f) / 2.f;
    float iy = ((gy + 1.f) * (float)p.H - 1.f) / 2.f;
    const float mx = (ix <= 0.f || ix >= (float)(p.W - 1)) ? 0.f : 1.f;
    const float my = (iy <= 0.f || iy >= (float)(p.H - 1)) ? 0.f : 1.f;
    ix = fminf(fmaxf(ix, 0.f), (float)(p.W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(p.H - 1));
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - x0f, wx0 = (x0f + 1.f) - ix;
    const float wy1 = iy - y0f, wy0 = (y0f + 1.f) - iy;
    const bool vx1 = x1 < p.W, vy1 = y1 < p.H;
    const size_t base = (size_t)n * p.H * p.W;
    const size_t o00 = (base + (size_t)y0 * p.W + x0), o01 = o00 + (vx1 ? 1 : 0);
    const size_t o10 = o00 + (vy1 ? p.W : 0), o11 = o10 + (vx1 ? 1 : 0);
    float gix = 0.f, giy = 0.f;
    if (live) {
      for (int g = lane16; g < p.C4; g += 16) {
        const f32x4 d = ld4(p.dout + pp * p.dcs + p.dco + g * 4);
        if (p.dflow) {
          const float* sb = p.src + p.sco + g * 4;
          const f32x4 v00 = ld4(sb + o00 * p.scs);
          const f32x4 v01 = vx1 ? ld4(sb + o01 * p.scs) : (f32x4)(0.f);
          const f32x4 v10 = vy1 ? ld4(sb + o10 * p.scs) : (f32x4)(0.f);
          const f32x4 v11 = (vx1 && vy1) ? ld4(sb + o11 * p.scs) : (f32x4)(0.f);
          const f32x4 tx = (v01 - v00) * wy0 + (v11 - v10) * wy1;  // d out / d ix
          const f32x4 ty = (v10 - v00) * wx0 + (v11 - v01) * wx1;  // d out / d iy
          gix += d[0] * tx[0] + d[1] * tx[1] + d[2] * tx[2] + d[3] * tx[3];
          giy += d[0] * ty[0] + d[1] * ty[1] + d[2] * ty[2] + d[3] * ty[3];
        }
        if (p.dsrc) {
          float* gb = p.dsrc + p.gco + g * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            unsafeAtomicAdd(gb + o00 * p.gcs + e, d[e] * (wx0 * wy0));
            if (vx1) unsafeAtomicAdd(gb + o01 * p.gcs + e, d[e] * (wx1 * wy0));
            if (vy1) unsafeAtomicAdd(gb + o10 * p.gcs + e, d[e] * (wx0 * wy1));
            if (vx1 && vy1) unsafeAtomicAdd(gb + o11 * p.gcs + e, d[e] * (wx1 * wy1));
          }
        }
      }
    }
    if (p.dflow) {
      // reduce over the 16 lanes of the pixel's group (fixed butterfly order: deterministic)
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        gix += __shfl_xor(gix, o, 16);
        giy += __shfl_xor(giy, o, 16);
      }
      if (live && lane16 == 0) {
        const float dfx = gix * mx * ((float)p.W * 0.5f) / p.norm_x;
        const float dfy = giy * my * ((float)p.H * 0.5f) / p.norm_y;
        float* o = p.dflow + pp * 2;
        o[0] = p.dflow_accumulate ? o[0] + dfx : dfx;
        o[1] = p.dflow_accumulate ? o[1] + dfy : dfy;
      }
    }
  }
}

// d(src) of the warp, LDS-privatised.  The scatter-add above issues 4 fp32 atomics per (output pixel, channel);
// at 8 x 512x384 x 96 channels that is 600 M L2 atomics (31 ms).  Flows are smooth, so the samples of a 16x16
// output tile land in a compact source window: a block accumulates its tile (one 16-channel chunk) into an LDS
// window of up to 24x24 source pixels with ds_add_f32, then flushes only the touched cells with one global
// atomic each -- the windows of neighbouring tiles overlap, so the flush stays atomic, but a cell receives ~1
// global atomic per tile instead of one per contributing output pixel.  A tile whose samples do not fit the
// window (large local flow gradients) falls back to direct global atomics; the result is the same sum either way.
constexpr int WT = 16;                 // output tile edge
constexpr int WB = 24;                 // source window edge
constexpr int WCH = 16;                // channels per block
constexpr int WCELLS = WB * WB + 1;    // +1: odd row stride between channel planes

__global__ __launch_bounds__(256) void flow_warp_dsrc_tiled_kernel(const WarpBwdParams p, int tiles_x, int tiles_y) {
  __shared__ float acc[WCH][WCELLS];
  __shared__ int red[4][4];
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int txi = b % tiles_x; b /= tiles_x;
  const int tyi = b % tiles_y;
  const int n = b / tiles_y;
  const int c0 = blockIdx.y * WCH;                       // first channel of this block
  const int wo = txi * WT + (tid & (WT - 1)), ho = tyi * WT + (tid >> 4);
  const bool live = wo < p.Wo && ho < p.Ho;
  const size_t pp = live ? ((size_t)n * p.Ho + ho) * p.Wo + wo : 0;
  const float fx = p.flow_up[pp * 2], fy = p.flow_up[pp * 2 + 1];
  const float gx = fx / p.norm_x + lin_m1_1b(live ? wo : 0, p.Wo, p.step_x);
  const float gy = fy / p.norm_y + lin_m1_1b(live ? ho : 0, p.Ho, p.step_y);
  float ix = ((gx + 1.f) * (float)p.W - 1.f) / 2.f;
  float iy = ((gy + 1.f) * (float)p.H - 1.f) / 2.f;
  ix = fminf(fmaxf(ix, 0.f), (float)(p.W - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(p.H - 1));
  const float x0f = floorf(ix), y0f = floorf(iy);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float wx1 = ix - x0f, wx0 = (x0f + 1.f) - ix;
  const float wy1 = iy - y0f, wy0 = (y0f + 1.f) - iy;
  const bool vx1 = x0 + 1 < p.W, vy1 = y0 + 1 < p.H;
  // window = bounding box of the tile's taps
  int mnx = live ? x0 : 0x7fffffff, mny = live ? y0 : 0x7fffffff;
  int mxx = live ? x0 + (vx1 ? 1 : 0) : -1, mxy = live ? y0 + (vy1 ? 1 : 0) : -1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mnx = min(mnx, __shfl_xor(mnx, o)); mny = min(mny, __shfl_xor(mny, o));
    mxx = max(mxx, __shfl_xor(mxx, o)); mxy = max(mxy, __shfl_xor(mxy, o));
  }
  if ((tid & 63) == 0) { red[tid >> 6][0] = mnx; red[tid >> 6][1] = mny; red[tid >> 6][2] = mxx; red[tid >> 6][3] = mxy; }
  for (int i = tid; i < WCH * WCELLS; i += 256) (&acc[0][0])[i] = 0.f;
  __syncthreads();
  const int bx0 = min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
  const int by0 = min(min(red[0][1], red[1][1]), min(red[2][1], red[3][1]));
  const int bx1 = max(max(red[0][2], red[1][2]), max(red[2][2], red[3][2]));
  const int by1 = max(max(red[0][3], red[1][3]), max(red[2][3], red[3][3]));
  const bool fits = bx1 - bx0 < WB && by1 - by0 < WB;    // block-uniform
  const int nch = min(WCH, p.C4 * 4 - c0);               // channels of this chunk (multiple of 4)
  const size_t base = (size_t)n * p.H * p.W;
  if (live) {
    const int cell = (y0 - by0) * WB + (x0 - bx0);
    const size_t o00 = base + (size_t)y0 * p.W + x0;
    const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
    for (int c = 0; c < nch; c += 4) {
      const f32x4 d = ld4(p.dout + pp * p.dcs + p.dco + c0 + c);
      if (fits) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float* a = &acc[c + e][cell];
          atomicAdd(a, d[e] * w00);
          if (vx1) atomicAdd(a + 1, d[e] * w01);
          if (vy1) atomicAdd(a + WB, d[e] * w10);
          if (vx1 && vy1) atomicAdd(a + WB + 1, d[e] * w11);
        }
      } else {
        float* gb = p.dsrc + p.gco + c0 + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsafeAtomicAdd(gb + o00 * p.gcs + e, d[e] * w00);
          if (vx1) unsafeAtomicAdd(gb + (o00 + 1) * p.gcs + e, d[e] * w01);
          if (vy1) unsafeAtomicAdd(gb + (o00 + p.W) * p.gcs + e, d[e] * w10);
          if (vx1 && vy1) unsafeAtomicAdd(gb + (o00 + p.W + 1) * p.gcs + e, d[e] * w11);
        }
      }
    }
  }
  if (!fits) return;
  __syncthreads();
  // flush: lanes run over (cell, channel) with the channel fastest -> 64-byte runs per source pixel
  const int ww = bx1 - bx0 + 1, wh = by1 - by0 + 1;
  for (int i = tid; i < ww * wh * nch; i += 256) {
    const int c = i % nch, cellc = i / nch;
    const int cy = cellc / ww, cx = cellc - cy * ww;
    const float v = acc[c][cy * WB + cx];
    if (v != 0.f)
      unsafeAtomicAdd(p.dsrc + (base + (size_t)(by0 + cy) * p.W + bx0 + cx) * p.gcs + p.gco + c0 + c, v);
  }
}

// ---------------------------------------------------------------------------
// F.grid_sample(input NCHW, grid [N,Ho,Wo,2], bilinear, border, align_corners=False) with an
// explicit grid and its backward (train_condition.py:243-245; few channels: cloth 3, mask 1).
// ---------------------------------------------------------------------------
__global__ void grid_sample_nchw_kernel(const float* __restrict__ in, int N, int C, int H, int W,
                                        const float* __restrict__ grid, int Ho, int Wo, float* __restrict__ out) {
  const size_t npix = (size_t)N * Ho * Wo;
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
    const size_t hw = pix % ((size_t)Ho * Wo);
    const int n = (int)(pix / ((size_t)Ho * Wo));
    float ix = ((grid[pix * 2] + 1.f) * (float)W - 1.f) / 2.f;
    float iy = ((grid[pix * 2 + 1] + 1.f) * (float)H - 1.f) / 2.f;
    ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const bool vx1 = x0 + 1 < W, vy1 = y0 + 1 < H;
    const float wx1 = ix - x0f, wx0 = (x0f + 1.f) - ix, wy1 = iy - y0f, wy0 = (y0f + 1.f) - iy;
    const size_t o00 = (size_t)y0 * W + x0;
    for (int c = 0; c < C; ++c) {
      const float* s = in + ((size_t)n * C + c) * H * W;
      float v = s[o00] * (wx0 * wy0);
      if (vx1) v += s[o00 + 1] * (wx1 * wy0);
      if (vy1) v += s[o00 + W] * (wx0 * wy1);
      if (vx1 && vy1) v += s[o00 + W + 1] * (wx1 * wy1);
      out[((size_t)n * C + c) * Ho * Wo + hw] = v;
    }
  }
}

__global__ void grid_sample_nchw_bwd_kernel(const float* __restrict__ in, int N, int C, int H, int W,
                                            const float* __restrict__ grid, int Ho, int Wo,
                                            const float* __restrict__ dout, float* __restrict__ din,
                                            float* __restrict__ dgrid) {
  const size_t npix = (size_t)N * Ho * Wo;
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
    const size_t hw = pix % ((size_t)Ho * Wo);
    const int n = (int)(pix / ((size_t)Ho * Wo));
    float ix = ((grid[pix * 2] + 1.f) * (float)W - 1.f) / 2.f;
    float iy = ((grid[pix * 2 + 1] + 1.f) * (float)H - 1.f) / 2.f;
    const float mx = (ix <= 0.f || ix >= (float)(W - 1)) ? 0.f : 1.f;
    const float my = (iy <= 0.f || iy >= (float)(H - 1)) ? 0.f : 1.f;
    ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const bool vx1 = x0 + 1 < W, vy1 = y0 + 1 < H;
    const float wx1 = ix - x0f, wx0 = (x0f + 1.f) - ix, wy1 = iy - y0f, wy0 = (y0f + 1.f) - iy;
    const size_t o00 = (size_t)y0 * W + x0;
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < C; ++c) {
      const float d = dout[((size_t)n * C + c) * Ho * Wo + hw];
      const float* s = in + ((size_t)n * C + c) * H * W;
      const float v00 = s[o00], v01 = vx1 ? s[o00 + 1] : 0.f, v10 = vy1 ? s[o00 + W] : 0.f,
                  v11 = (vx1 && vy1) ? s[o00 + W + 1] : 0.f;
      gix += d * ((v01 - v00) * wy0 + (v11 - v10) * wy1);
      giy += d * ((v10 - v00) * wx0 + (v11 - v01) * wx1);
      if (din) {
        float* gsrc = din + ((size_t)n * C + c) * H * W;
        unsafeAtomicAdd(gsrc + o00, d * (wx0 * wy0));
        if (vx1) unsafeAtomicAdd(gsrc + o00 + 1, d * (wx1 * wy0));
        if (vy1) unsafeAtomicAdd(gsrc + o00 + W, d * (wx0 * wy1));
        if (vx1 && vy1) unsafeAtomicAdd(gsrc + o00 + W + 1, d * (wx1 * wy1));
      }
    }
    if (dgrid) {
      dgrid[pix * 2] = gix * mx * ((float)W * 0.5f);
      dgrid[pix * 2 + 1] = giy * my * ((float)H * 0.5f);
    }
  }
}

// ---------------------------------------------------------------------------
// softmax over the channel dim of an NCHW tensor (train_condition.py:175,246,260) + backward;
// fused log-softmax + NLL (utils.cross_entropy2d) value and gradient in one pass.
// ---------------------------------------------------------------------------
constexpr int kMaxSoftmaxC = 64;

__global__ void softmax_nchw_kernel(const float* __restrict__ x, int N, int C, size_t HW, float* __restrict__ y) {
  const size_t npix = (size_t)N * HW;
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
    const size_t hw = pix % HW;
    const size_t n = pix / HW;
    const float* s = x + n * C * HW + hw;
    float m = s[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, s[(size_t)c * HW]);
    float e[kMaxSoftmaxC];
    float sum = 0.f;
    for (int c = 0; c < C; ++c) {
      e[c] = expf(s[(size_t)c * HW] - m);
      sum += e[c];
    }
    const float inv = 1.f / sum;
    float* o = y + n * C * HW + hw;
    for (int c = 0; c < C; ++c) o[(size_t)c * HW] = e[c] * inv;
  }
}

// dx = y * (dy - sum_c dy*y)
__global__ void softmax_nchw_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, int N, int C,
                                        size_t HW, float* __restrict__ dx) {
  const size_t npix = (size_t)N * HW;
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
    const size_t hw = pix % HW;
    const size_t n = pix / HW;
    const size_t b = n * C * HW + hw;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot += dy[b + (size_t)c * HW] * y[b + (size_t)c * HW];
    for (int c = 0; c < C; ++c) dx[b + (size_t)c * HW] = y[b + (size_t)c * HW] * (dy[b + (size_t)c * HW] - dot);
  }
}

// loss partials: sum over pixels of (logsumexp(x) - x[target]); grad = gscale*(softmax - onehot)
// targets outside [0,C) (ignore_index) contribute nothing; `count` partials hold the valid pixels.
__global__ __launch_bounds__(256) void ce_nchw_kernel(const float* __restrict__ x, const long long* __restrict__ target,
                                                      int N, int C, size_t HW, float gscale, float* __restrict__ grad,
                                                      float* __restrict__ part) {
  __shared__ float red[256];
  __shared__ float redc[256];
  const size_t npix = (size_t)N * HW;
  float s = 0.f, cnt = 0.f;
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
    const size_t hw = pix % HW;
    const size_t n = pix / HW;
    const size_t b = n * C * HW + hw;
    const long long t = target[pix];
    const bool valid = t >= 0 && t < C;
    float m = x[b];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[b + (size_t)c * HW]);
    float e[kMaxSoftmaxC];
    float sum = 0.f;
    for (int c = 0; c < C; ++c) {
      e[c] = expf(x[b + (size_t)c * HW] - m);
      sum += e[c];
    }
    if (valid) {
      s += logf(sum) + m - x[b + (size_t)t * HW];
      cnt += 1.f;
    }
    if (grad) {
      const float inv = 1.f / sum;
      for (int c = 0; c < C; ++c)
        grad[b + (size_t)c * HW] = valid ? gscale * (e[c] * inv - (c == (int)t ? 1.f : 0.f)) : 0.f;
    }
  }
  red[threadIdx.x] = s;
  redc[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[threadIdx.x] += red[threadIdx.x + o];
      redc[threadIdx.x] += redc[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[blockIdx.x] = red[0];
    part[gridDim.x + blockIdx.x] = redc[0];
  }
}

__global__ void ce_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0, c = 0.0;
    for (int i = 0; i < nb; ++i) {
      s += (double)part[i];
      c += (double)part[nb + i];
    }
    out[0] = (float)(c > 0.0 ? s / c : 0.0);  // mean over the valid pixels
    out[1] = (float)c;
  }
}

// TV of one [N,h,w,2] flow: loss = mean|f[:,1:]-f[:,:-1]| + mean|f[:,:,1:]-f[:,:,:-1]|; grad likewise
__global__ __launch_bounds__(256) void tv_kernel(const float* __restrict__ f, int N, int H, int W, float sy, float sx,
                                                 float* __restrict__ grad, float* __restrict__ part) {
  __shared__ float red[256];
  const size_t total = (size_t)N * H * W * 2;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i >> 1;
    const int w = (int)(pix % W);
    const int h = (int)((pix / W) % H);
    const float v = f[i];
    float g = 0.f;
    auto sgn = [](float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); };
    if (h + 1 < H) { const float d = f[i + (size_t)W * 2] - v; s += fabsf(d) * sy; g -= sgn(d) * sy; }
    if (h > 0) { const float d = v - f[i - (size_t)W * 2]; g += sgn(d) * sy; }
    if (w + 1 < W) { const float d = f[i + 2] - v; s += fabsf(d) * sx; g -= sgn(d) * sx; }
    if (w > 0) { const float d = v - f[i - 2]; g += sgn(d) * sx; }
    if (grad) grad[i] = g;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// Adjoint of hrv_tapsum_nhwc_f32 (sample.hip): dy[q][tap*Cout+co] = dout[q - off(tap)][co], zero where
// q - off(tap) leaves the image; channels >= KH*KW*Cout of dy (padding) are zeroed.
__global__ void tapsum_bwd_kernel(const float* __restrict__ dout, int N, int H, int W, int KH, int KW, int pad, int Cout,
                                  int dcs, float* __restrict__ dy, int ycs, int ycp) {
  const size_t total = (size_t)N * H * W * ycp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % ycp);
    const size_t pix = i / ycp;
    const int w = (int)(pix % W);
    const size_t t = pix / W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float v = 0.f;
    if (ch < KH * KW * Cout) {
      const int tap = ch / Cout, co = ch - tap * Cout;
      const int kh = tap / KW, kw = tap - kh * KW;
      const int hh = h - (kh - pad), ww = w - (kw - pad);
      if (hh >= 0 && hh < H && ww >= 0 && ww < W) v = dout[((size_t)(n * H + hh) * W + ww) * dcs + co];
    }
    dy[pix * ycs + ch] = v;
  }
}

__global__ void sum_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nb; ++i) s += (double)part[i];
    out[0] = (float)s;
  }
}

}  // namespace hrv

using namespace hrv;

extern "C" int hrv_bn_finalize_f32(const float* mean_nc, const float* rstd_nc, int32_t N, int32_t C, int32_t nc_stride,
                                   float eps_in, int64_t HW, const float* gamma, const float* beta, float eps,
                                   float momentum, float* running_mean, float* running_var, float* mean, float* rstd,
                                   float* scale, float* shift, hrv_stream_t stream) {
  HRV_REQUIRE(mean_nc && rstd_nc && mean && rstd && scale && shift && N > 0 && C > 0 && nc_stride >= C && HW > 0,
              "bn_finalize: bad args");
  HRV_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running_mean/var go together");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, mean_nc, rstd_nc, N, C,
                     nc_stride, eps_in, (long long)HW, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd,
                     scale, shift);
  return check_launch("bn_finalize_kernel");
}

extern "C" int hrv_affine_act_nhwc_f32(const float* x, int32_t x_cstride, int32_t x_coff, int32_t C, int64_t npix,
                                       const float* scale, const float* shift, const float* residual, int32_t res_cstride,
                                       int32_t res_coff, int32_t act, float act_slope, float* out, int32_t out_cstride,
                                       int32_t out_coff, hrv_stream_t stream) {
  HRV_REQUIRE(x && scale && shift && out && npix > 0 && C > 0 && C % 4 == 0 &&
                  ((x_cstride | x_coff | out_cstride | out_coff | res_cstride | res_coff) & 3) == 0,
              "affine_act: bad args (channels/strides must be multiples of 4)");
  HRV_REQUIRE((((uintptr_t)scale | (uintptr_t)shift) & 15) == 0, "affine_act: scale/shift must be 16-byte aligned");
  hipLaunchKernelGGL(affine_act_kernel, dim3(grid_for((size_t)npix * (C / 4))), dim3(256), 0, (hipStream_t)stream, x,
                     x_cstride, x_coff, C / 4, (size_t)npix, scale, shift, residual, res_cstride, res_coff, act,
                     act_slope, out, out_cstride, out_coff);
  return check_launch("affine_act_kernel");
}

extern "C" int64_t hrv_bn_bwd_workspace_elems(int32_t C) {
  const int Cp = (C + 3) / 4 * 4;
  return (int64_t)256 * 2 * Cp + 2 * Cp;
}

extern "C" int hrv_bn_bwd_nhwc_f32(const float* dy, int32_t dy_cstride, int32_t dy_coff, const float* x, int32_t x_cstride,
                                   int32_t x_coff, int32_t C, int64_t npix, const float* mean, const float* rstd,
                                   const float* scale, float* workspace, float* dx, int32_t dx_cstride, int32_t dx_coff,
                                   int32_t dx_accumulate, float* dgamma, float* dbeta, int32_t dgb_accumulate,
                                   hrv_stream_t stream) {
  HRV_REQUIRE(dy && x && mean && rstd && scale && workspace && dx && npix > 0 && npix < ((int64_t)1 << 31) && C > 0,
              "bn_bwd: bad args");
  HRV_REQUIRE(((dy_cstride | dy_coff | x_cstride | x_coff | dx_cstride | dx_coff) & 3) == 0, "bn_bwd: layout");
  HRV_REQUIRE((((uintptr_t)mean | (uintptr_t)rstd | (uintptr_t)scale | (uintptr_t)workspace) & 15) == 0,
              "bn_bwd: per-channel vectors must be 16-byte aligned");
  const int Cp = (C + 3) / 4 * 4;  // mean / rstd / scale hold Cp floats (zero in the pad channels)
  HRV_REQUIRE(dy_coff + Cp <= dy_cstride && x_coff + Cp <= x_cstride && dx_coff + Cp <= dx_cstride,
              "bn_bwd: padded channel group must lie inside the pixel row");
  const int C4 = Cp / 4;
  int nb = (int)((npix + 1023) / 1024);
  nb = nb < 1 ? 1 : (nb > 256 ? 256 : nb);      // (<= 256 partial rows: the fixed-order final sum walks them serially per channel)
  hipStream_t st = (hipStream_t)stream;
  float* part = workspace;
  float* sums = workspace + (size_t)256 * 2 * Cp;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nb), dim3(256), 0, st, dy, dy_cstride, dy_coff, x, x_cstride, x_coff,
                     (int)npix, C4, mean, rstd, nb, part);
  int rc = check_launch("bn_bwd_partial_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((Cp + 127) / 128), dim3(128), 0, st, part, nb, C, Cp, sums, dgamma, dbeta,
                     dgb_accumulate);
  rc = check_launch("bn_bwd_final_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for((size_t)npix * C4)), dim3(256), 0, st, dy, dy_cstride, dy_coff, x,
                     x_cstride, x_coff, (size_t)npix, C4, mean, rstd, scale, sums, 1.0f / (float)npix, dx, dx_cstride,
                     dx_coff, dx_accumulate);
  return check_launch("bn_bwd_apply_kernel");
}

extern "C" int hrv_resize_bilinear_bwd_nhwc_f32(const float* dy, int32_t N, int32_t Ho, int32_t Wo, int32_t C,
                                                int32_t dy_cstride, int32_t dy_coff, float rh, float rw, float* dx,
                                                int32_t H, int32_t W, int32_t dx_cstride, int32_t dx_coff,
                                                int32_t accumulate, hrv_stream_t stream) {
  HRV_REQUIRE(dy && dx && N > 0 && Ho > 0 && Wo > 0 && H > 0 && W > 0 && C > 0 && rh > 0.f && rw > 0.f,
              "resize_bilinear_bwd: bad args");
  HRV_REQUIRE(dy_cstride >= dy_coff + C && dx_cstride >= dx_coff + C, "resize_bilinear_bwd: channel slices out of range");
  const bool vec = C % 4 == 0 && ((dy_cstride | dy_coff | dx_cstride | dx_coff) & 3) == 0 &&
                   (((uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
  hipStream_t st = (hipStream_t)stream;
  if (vec) {
    const size_t total = (size_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(resize_bwd_kernel<4>, dim3(grid_for(total)), dim3(256), 0, st, dy, N, H, W, C / 4, Ho, Wo,
                       dy_cstride, dy_coff, rh, rw, dx, dx_cstride, dx_coff, accumulate);
  } else {
    const size_t total = (size_t)N * H * W * C;
    hipLaunchKernelGGL(resize_bwd_kernel<1>, dim3(grid_for(total)), dim3(256), 0, st, dy, N, H, W, C, Ho, Wo, dy_cstride,
                       dy_coff, rh, rw, dx, dx_cstride, dx_coff, accumulate);
  }
  return check_launch("resize_bwd_kernel");
}

// ---------------------------------------------------------------------------------------------------------------------
// F.interpolate(mode='nearest') over planes [P][H][W] -> [P][Ho][Wo] and its adjoint: train_condition.py:242 with --upsample nearest
// (the inter-flow loss resizes every intermediate flow to the image size).  torch's legacy nearest: src = min(floorf(dst * scale),
// in - 1) with scale = (float)in / out.  The adjoint is a gather (deterministic): a source pixel sums the output pixels whose
// source index -- computed with the SAME float expression -- is that pixel; candidates are bracketed by the inverse map +- 1.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
  const int s = (int)floorf((float)dst * scale);
  return s < in_size - 1 ? s : in_size - 1;
}

__global__ void resize_nearest_planes_kernel(const float* __restrict__ in, size_t total, int H, int W, int Ho, int Wo, float sh,
                                             float sw, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wo);
    const size_t t = i / Wo;
    const int y = (int)(t % Ho);
    const size_t p = t / Ho;
    out[i] = in[(p * H + nearest_src(y, sh, H)) * W + nearest_src(x, sw, W)];
  }
}

__global__ void resize_nearest_planes_bwd_kernel(const float* __restrict__ dout, size_t total, int H, int W, int Ho, int Wo, float sh,
                                                 float sw, float* __restrict__ dx) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int sx = (int)(i % W);
    const size_t t = i / W;
    const int sy = (int)(t % H);
    const size_t p = t / H;
    int y0 = (int)floorf((float)sy / sh) - 1, y1 = (int)floorf((float)(sy + 1) / sh) + 1;
    int x0 = (int)floorf((float)sx / sw) - 1, x1 = (int)floorf((float)(sx + 1) / sw) + 1;
    if (sy == H - 1) y1 = Ho - 1;          // (the clamp maps every overshooting output to the last source row / column)
    if (sx == W - 1) x1 = Wo - 1;
    y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0;
    y1 = y1 > Ho - 1 ? Ho - 1 : y1; x1 = x1 > Wo - 1 ? Wo - 1 : x1;
    float acc = 0.f;
    for (int y = y0; y <= y1; ++y) {
      if (nearest_src(y, sh, H) != sy) continue;
      for (int x = x0; x <= x1; ++x)
        if (nearest_src(x, sw, W) == sx) acc += dout[(p * Ho + y) * Wo + x];
    }
    dx[i] = acc;
  }
}

// the same selection over NHWC activations (any C, channel slices) with an optional addend: ConditionGenerator.forward(upsample='nearest'),
// networks.py:130-133,150 -- T = nearest_x2(T) + conv1x1(E), and the x2 flow upsample in front of the warp
__global__ void resize_nearest_nhwc_kernel(const float* __restrict__ in, size_t total, int H, int W, int C, int ics, int ico, int Ho, int Wo,
                                           float sh, float sw, const float* __restrict__ add, int acs, int aco, float* __restrict__ out,
                                           int ocs, int oco) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t t = i / C;
    const int x = (int)(t % Wo); t /= Wo;
    const int y = (int)(t % Ho);
    const size_t n = t / Ho;
    const size_t op = (n * Ho + y) * Wo + x;
    float v = in[((n * H + nearest_src(y, sh, H)) * W + nearest_src(x, sw, W)) * ics + ico + c];
    if (add) v += add[op * acs + aco + c];
    out[op * ocs + oco + c] = v;
  }
}

__global__ void resize_nearest_nhwc_bwd_kernel(const float* __restrict__ dout, size_t total, int H, int W, int C, int dcs, int dco, int Ho,
                                               int Wo, float sh, float sw, float* __restrict__ dx, int xcs, int xco, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t t = i / C;
    const int sx = (int)(t % W); t /= W;
    const int sy = (int)(t % H);
    const size_t n = t / H;
    int y0 = (int)floorf((float)sy / sh) - 1, y1 = (int)floorf((float)(sy + 1) / sh) + 1;
    int x0 = (int)floorf((float)sx / sw) - 1, x1 = (int)floorf((float)(sx + 1) / sw) + 1;
    if (sy == H - 1) y1 = Ho - 1;
    if (sx == W - 1) x1 = Wo - 1;
    y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0;
    y1 = y1 > Ho - 1 ? Ho - 1 : y1; x1 = x1 > Wo - 1 ? Wo - 1 : x1;
    float acc = 0.f;
    for (int y = y0; y <= y1; ++y) {
      if (nearest_src(y, sh, H) != sy) continue;
      for (int x = x0; x <= x1; ++x)
        if (nearest_src(x, sw, W) == sx) acc += dout[((n * Ho + y) * Wo + x) * dcs + dco + c];
    }
    float* o = dx + ((n * H + sy) * W + sx) * xcs + xco + c;
    *o = accumulate ? *o + acc : acc;
  }
}

extern "C" int hrv_resize_nearest_nhwc_f32(const float* in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_cstride, int32_t in_coff,
                                           int32_t Ho, int32_t Wo, const float* addend, int32_t add_cstride, int32_t add_coff, float* out,
                                           int32_t out_cstride, int32_t out_coff, hrv_stream_t stream) {
  HRV_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, "resize_nearest_nhwc: bad args");
  HRV_REQUIRE(in_cstride >= in_coff + C && out_cstride >= out_coff + C && (!addend || add_cstride >= add_coff + C),
              "resize_nearest_nhwc: channel slices out of range");
  const size_t total = (size_t)N * Ho * Wo * C;
  hipLaunchKernelGGL(resize_nearest_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, total, H, W, C, in_cstride,
                     in_coff, Ho, Wo, (float)H / (float)Ho, (float)W / (float)Wo, addend, add_cstride, add_coff, out, out_cstride, out_coff);
  return check_launch("resize_nearest_nhwc_kernel");
}

extern "C" int hrv_resize_nearest_bwd_nhwc_f32(const float* dout, int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t d_cstride,
                                               int32_t d_coff, float* dx, int32_t H, int32_t W, int32_t dx_cstride, int32_t dx_coff,
                                               int32_t accumulate, hrv_stream_t stream) {
  HRV_REQUIRE(dout && dx && N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, "resize_nearest_bwd_nhwc: bad args");
  HRV_REQUIRE(d_cstride >= d_coff + C && dx_cstride >= dx_coff + C, "resize_nearest_bwd_nhwc: channel slices out of range");
  const size_t total = (size_t)N * H * W * C;
  hipLaunchKernelGGL(resize_nearest_nhwc_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dout, total, H, W, C, d_cstride,
                     d_coff, Ho, Wo, (float)H / (float)Ho, (float)W / (float)Wo, dx, dx_cstride, dx_coff, accumulate);
  return check_launch("resize_nearest_nhwc_bwd_kernel");
}

extern "C" int hrv_resize_nearest_nchw_f32(const float* in, int32_t planes, int32_t H, int32_t W, int32_t Ho, int32_t Wo, float* out,
                                           hrv_stream_t stream) {
  HRV_REQUIRE(in && out && planes > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "resize_nearest: bad args");
  const size_t total = (size_t)planes * Ho * Wo;
  hipLaunchKernelGGL(resize_nearest_planes_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, total, H, W, Ho, Wo,
                     (float)H / (float)Ho, (float)W / (float)Wo, out);
  return check_launch("resize_nearest_planes_kernel");
}

extern "C" int hrv_resize_nearest_nchw_bwd_f32(const float* dout, int32_t planes, int32_t Ho, int32_t Wo, int32_t H, int32_t W, float* dx,
                                               hrv_stream_t stream) {
  HRV_REQUIRE(dout && dx && planes > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "resize_nearest_bwd: bad args");
  const size_t total = (size_t)planes * H * W;
  hipLaunchKernelGGL(resize_nearest_planes_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dout, total, H, W, Ho,
                     Wo, (float)H / (float)Ho, (float)W / (float)Wo, dx);
  return check_launch("resize_nearest_planes_bwd_kernel");
}

extern "C" int hrv_flow_warp_bwd_nhwc_f32(const hrv_flow_warp_bwd_t* d, hrv_stream_t stream) {
  HRV_REQUIRE(d && d->flow_up && d->dout && (d->dsrc || d->dflow), "flow_warp_bwd: null pointer");
  HRV_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->C > 0 && d->C % 4 == 0,
              "flow_warp_bwd: bad extent");
  HRV_REQUIRE(!d->dflow || d->src, "flow_warp_bwd: the flow gradient needs the sampled tensor");
  HRV_REQUIRE(((d->src_cstride | d->src_coff | d->dout_cstride | d->dout_coff | d->dsrc_cstride | d->dsrc_coff) & 3) == 0,
              "flow_warp_bwd: strides/offsets must be multiples of 4");
  HRV_REQUIRE(d->norm_x != 0.f && d->norm_y != 0.f, "flow_warp_bwd: zero normaliser");
  WarpBwdParams p;
  p.src = d->src; p.N = d->N; p.H = d->H; p.W = d->W; p.C4 = d->C / 4; p.scs = d->src_cstride; p.sco = d->src_coff;
  p.flow_up = d->flow_up; p.Ho = d->Ho; p.Wo = d->Wo; p.norm_x = d->norm_x; p.norm_y = d->norm_y;
  p.step_x = d->Wo > 1 ? 2.f / (float)(d->Wo - 1) : 0.f;
  p.step_y = d->Ho > 1 ? 2.f / (float)(d->Ho - 1) : 0.f;
  p.dout = d->dout; p.dcs = d->dout_cstride; p.dco = d->dout_coff;
  p.dsrc = d->dsrc; p.gcs = d->dsrc_cstride; p.gco = d->dsrc_coff;
  p.dflow = d->dflow; p.dflow_accumulate = d->dflow_accumulate;
  const size_t npix = (size_t)d->N * d->Ho * d->Wo;
  // wide tensors: d(src) by the LDS-privatised tile kernel, d(flow) (a gather, no atomics) by the per-pixel kernel
  const char* ev = hrv::env("HRV_WARP_BWD_TILED");
  const bool tiled = p.dsrc && d->C >= 16 && (!ev || atoi(ev) != 0);
  if (tiled) {
    const int tx = (d->Wo + WT - 1) / WT, ty = (d->Ho + WT - 1) / WT;
    hipLaunchKernelGGL(flow_warp_dsrc_tiled_kernel, dim3((unsigned)(tx * ty * d->N), (unsigned)((d->C + WCH - 1) / WCH)),
                       dim3(256), 0, (hipStream_t)stream, p, tx, ty);
    const int rc = check_launch("flow_warp_dsrc_tiled_kernel");
    if (rc || !p.dflow) return rc;
    p.dsrc = nullptr;
  }
  hipLaunchKernelGGL(flow_warp_bwd_kernel, dim3(grid_for(npix * 16)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("flow_warp_bwd_kernel");
}

extern "C" int hrv_grid_sample_nchw_f32(const float* in, int32_t N, int32_t C, int32_t H, int32_t W, const float* grid,
                                        int32_t Ho, int32_t Wo, float* out, hrv_stream_t stream) {
  HRV_REQUIRE(in && grid && out && N > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "grid_sample: bad args");
  hipLaunchKernelGGL(grid_sample_nchw_kernel, dim3(grid_for((size_t)N * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, in, N,
                     C, H, W, grid, Ho, Wo, out);
  return check_launch("grid_sample_nchw_kernel");
}

extern "C" int hrv_grid_sample_nchw_bwd_f32(const float* in, int32_t N, int32_t C, int32_t H, int32_t W, const float* grid,
                                            int32_t Ho, int32_t Wo, const float* dout, float* din, float* dgrid,
                                            hrv_stream_t stream) {
  HRV_REQUIRE(in && grid && dout && (din || dgrid) && N > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0,
              "grid_sample_bwd: bad args");
  hipLaunchKernelGGL(grid_sample_nchw_bwd_kernel, dim3(grid_for((size_t)N * Ho * Wo)), dim3(256), 0, (hipStream_t)stream,
                     in, N, C, H, W, grid, Ho, Wo, dout, din, dgrid);
  return check_launch("grid_sample_nchw_bwd_kernel");
}

extern "C" int hrv_softmax_nchw_f32(const float* x, int32_t N, int32_t C, int64_t HW, float* y, hrv_stream_t stream) {
  HRV_REQUIRE(x && y && N > 0 && C > 0 && C <= kMaxSoftmaxC && HW > 0, "softmax: bad args (C <= 64)");
  hipLaunchKernelGGL(softmax_nchw_kernel, dim3(grid_for((size_t)N * HW)), dim3(256), 0, (hipStream_t)stream, x, N, C,
                     (size_t)HW, y);
  return check_launch("softmax_nchw_kernel");
}

extern "C" int hrv_softmax_nchw_bwd_f32(const float* y, const float* dy, int32_t N, int32_t C, int64_t HW, float* dx,
                                        hrv_stream_t stream) {
  HRV_REQUIRE(y && dy && dx && N > 0 && C > 0 && HW > 0, "softmax_bwd: bad args");
  hipLaunchKernelGGL(softmax_nchw_bwd_kernel, dim3(grid_for((size_t)N * HW)), dim3(256), 0, (hipStream_t)stream, y, dy, N,
                     C, (size_t)HW, dx);
  return check_launch("softmax_nchw_bwd_kernel");
}

extern "C" int hrv_cross_entropy_nchw_f32(const float* x, const int64_t* target, int32_t N, int32_t C, int64_t HW,
                                          float gscale, float* grad, float* workspace, float* loss_out,
                                          hrv_stream_t stream) {
  HRV_REQUIRE(x && target && workspace && loss_out && N > 0 && C > 0 && C <= kMaxSoftmaxC && HW > 0,
              "cross_entropy: bad args (C <= 64)");
  int nb = grid_for((size_t)N * HW);
  nb = nb > 512 ? 512 : nb;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_nchw_kernel, dim3(nb), dim3(256), 0, st, x, (const long long*)target, N, C, (size_t)HW, gscale,
                     grad, workspace);
  int rc = check_launch("ce_nchw_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(64), 0, st, workspace, nb, loss_out);
  return check_launch("ce_final_kernel");
}

extern "C" int hrv_tv_loss_f32(const float* flow, int32_t N, int32_t H, int32_t W, float* grad, float* workspace,
                               float* loss_out, hrv_stream_t stream) {
  HRV_REQUIRE(flow && workspace && loss_out && N > 0 && H > 1 && W > 1, "tv_loss: bad args");
  const float sy = 1.f / ((float)N * (float)(H - 1) * (float)W * 2.f);
  const float sx = 1.f / ((float)N * (float)H * (float)(W - 1) * 2.f);
  int nb = grid_for((size_t)N * H * W * 2);
  nb = nb > 1024 ? 1024 : nb;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(tv_kernel, dim3(nb), dim3(256), 0, st, flow, N, H, W, sy, sx, grad, workspace);
  int rc = check_launch("tv_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(64), 0, st, workspace, nb, loss_out);
  return check_launch("sum_final_kernel");
}

extern "C" int hrv_tapsum_bwd_nhwc_f32(const float* dout, int32_t N, int32_t H, int32_t W, int32_t KH, int32_t KW,
                                       int32_t pad, int32_t Cout, int32_t dout_cstride, float* dy, int32_t dy_cstride,
                                       hrv_stream_t stream) {
  HRV_REQUIRE(dout && dy && N > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && pad >= 0 && Cout > 0, "tapsum_bwd: bad args");
  HRV_REQUIRE(KH == 2 * pad + 1 && KW == 2 * pad + 1, "tapsum_bwd: only 'same' stride-1 geometry (k = 2*pad+1)");
  const int ycp = (KH * KW * Cout + 3) / 4 * 4;
  HRV_REQUIRE(dout_cstride >= Cout && dy_cstride >= ycp, "tapsum_bwd: channel strides too small");
  const size_t total = (size_t)N * H * W * ycp;
  hipLaunchKernelGGL(tapsum_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dout, N, H, W, KH, KW,
                     pad, Cout, dout_cstride, dy, dy_cstride, ycp);
  return check_launch("tapsum_bwd_kernel");
}

extern "C" int hrv_mul_f32(const float* x, const float* m, int64_t n, float* out, hrv_stream_t stream) {
  HRV_REQUIRE(x && m && out && n > 0 && n % 4 == 0 && (((uintptr_t)x | (uintptr_t)m | (uintptr_t)out) & 15) == 0,
              "mul: n must be a multiple of 4 and the pointers 16-byte aligned");
  hipLaunchKernelGGL(mul_kernel, dim3(grid_for((size_t)n / 4)), dim3(256), 0, (hipStream_t)stream, x, m, (size_t)n / 4, out);
  return check_launch("mul_kernel");
}
