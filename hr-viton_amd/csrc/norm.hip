// Normalisation-side kernels of the generator / discriminator path (NHWC fp32):
//   * InstanceNorm statistics (per sample, per channel; biased variance) with the
//     SPADE noise term folded in:  v = x + z[n,w,h] * noise_scale[c]
//     (network_generator.py:104-110).  Two deterministic stages: shifted partial
//     sums per pixel-slab, then a double-precision fixed-order finalise.
//   * InstanceNorm apply + LeakyReLU (PatchGAN: network_generator.py:263-272,427).
//   * 3x3 stride-2 average pool, count_include_pad=False (:301-302).
// HBM-bound: one float4 per lane, wavefront-contiguous channel groups.
#include "hrv_common.h"

namespace hrv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct StatsParams {
  const float* x;
  int N, H, W, C4, cs, co;
  const float* z;   // [N][W][H] or null
  const float* ns;  // [C] or null
  int NB;           // pixel slabs per sample
  float* part;      // [N][NB][C4*4][2]
  // x = cat(nearest_up2(lo), hi) along channels, never materialised (up_g > 0): channel groups [0, up_g) come from `x` =
  // lo [N][H/2][W/2][cs] at (h >> 1, w >> 1), the others from `x2` = hi [N][H][W][cs2]
  const float* x2;
  int cs2, co2, up_g;
};

// element index of channel group g of pixel (n, pix = h * W + w) in its source tensor; *base receives that tensor
__device__ __forceinline__ const float* stats_src(const StatsParams& p, int n, int pix, int g) {
  if (p.up_g > 0) {
    if (g < p.up_g) {
      const int h = pix / p.W, w = pix - h * p.W;
      return p.x + (((size_t)n * (p.H >> 1) + (h >> 1)) * (p.W >> 1) + (w >> 1)) * p.cs + p.co + g * 4;
    }
    return p.x2 + ((size_t)n * p.H * p.W + pix) * p.cs2 + p.co2 + (g - p.up_g) * 4;
  }
  return p.x + ((size_t)n * p.H * p.W + pix) * p.cs + p.co + g * 4;
}

typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

template <bool BF>
__device__ __forceinline__ f32x4 stats_value(const StatsParams& p, int n, int pix, int g) {
  f32x4 v;
  const size_t idx = ((size_t)n * p.H * p.W + pix) * p.cs + p.co + g * 4;
  if constexpr (BF) {
    const u16x4 h = *reinterpret_cast<const u16x4*>(reinterpret_cast<const unsigned short*>(p.x) + idx);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_bit_cast(float, (unsigned)h[e] << 16);
  } else {
    v = *reinterpret_cast<const f32x4*>(p.x + idx);
  }
  if (p.z) {
    const int h = pix / p.W, w = pix - h * p.W;
    const float zz = p.z[((size_t)n * p.W + w) * p.H + h];
    v += zz * *reinterpret_cast<const f32x4*>(p.ns + g * 4);
  }
  return v;
}

template <bool BF>
__global__ __launch_bounds__(256) void instnorm_partial_kernel(const StatsParams p) {
  __shared__ f32x4 red[2][256];
  const int n = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
  const int HW = p.H * p.W;
  const int PB = (HW + p.NB - 1) / p.NB;
  const int p0 = b * PB, p1 = min(p0 + PB, HW);
  const int GB = p.C4 < NORM_GCAP ? p.C4 : NORM_GCAP;   // channel groups of this block (blockIdx.z picks the chunk)
  const int R = 256 / GB;                  // pixel rows in flight
  const int r = t / GB, gl = t - r * GB;
  {
    const int g = blockIdx.z * GB + gl;
    const bool active = r < R && g < p.C4;
    f32x4 s1 = (f32x4)(0.f), s2 = (f32x4)(0.f);
    if (active) {
      const f32x4 K = stats_value<BF>(p, n, 0, g);  // shift: kills the cancellation in E[v^2]-E[v]^2
      // four 16-byte loads in flight per thread (one load per iteration is latency-bound: 1.3 TB/s); the sums keep
      // ONE chain in pixel order, so the result is bit-identical to the plain loop
      int px = p0 + r;
      for (; px + 3 * R < p1; px += 4 * R) {
        f32x4 d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = stats_value<BF>(p, n, px + u * R, g) - K;
#pragma unroll
        for (int u = 0; u < 4; ++u) { s1 += d[u]; s2 += d[u] * d[u]; }
      }
      for (; px < p1; px += R) {
        const f32x4 d = stats_value<BF>(p, n, px, g) - K;
        s1 += d;
        s2 += d * d;
      }
    }
    red[0][t] = s1;
    red[1][t] = s2;
    __syncthreads();
    if (r == 0 && g < p.C4) {
      for (int rr = 1; rr < R; ++rr) {
        s1 += red[0][rr * GB + gl];
        s2 += red[1][rr * GB + gl];
      }
      float* dst = p.part + (((size_t)n * p.NB + b) * p.C4 * 4 + g * 4) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dst[2 * e] = s1[e];
        dst[2 * e + 1] = s2[e];
      }
    }
  }
}

// one WAVE per (n, c): the lanes sum the slab partials in parallel (fp64), then a fixed butterfly -- a thread per
// (n, c) walking up to 256 partials serially was 23 us of pure latency per normalisation
template <bool BF>
__global__ __launch_bounds__(256) void instnorm_finalize_kernel(const StatsParams p, float eps, float* __restrict__ mean,
                                                                float* __restrict__ rstd) {
  const int C = p.C4 * 4;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= p.N * C) return;
  const int n = i / C, c = i - n * C;
  double s1 = 0.0, s2 = 0.0;
  for (int b = lane; b < p.NB; b += 64) {
    const float* src = p.part + (((size_t)n * p.NB + b) * C + c) * 2;
    s1 += (double)src[0];
    s2 += (double)src[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  if (lane != 0) return;
  const size_t k0 = (size_t)n * p.H * p.W * p.cs + p.co + c;
  float K = BF ? __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(p.x)[k0] << 16)
               : (p.up_g > 0 ? stats_src(p, n, 0, c >> 2)[c & 3] : p.x[k0]);
  if (p.z) K += p.z[(size_t)n * p.W * p.H] * p.ns[c];
  const double cnt = (double)p.H * p.W;
  const double m = s1 / cnt;
  double var = s2 / cnt - m * m;
  var = var < 0.0 ? 0.0 : var;
  mean[i] = (float)((double)K + m);
  rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

// Two normalisations over the SAME x with different noise terms (a learned-shortcut SPADEResBlock: norm_0 and norm_s both
// normalise the block input, network_generator.py:158-166): x is read ONCE, both shifted partial sums keep the single
// kernel's summation order (same loop, one chain per statistic) -- bit-identical to two separate passes.
struct Stats2Params {
  StatsParams a;          // x, geometry, z / ns / part of the first norm
  const float* z2;
  const float* ns2;
  float* part2;
};

__global__ __launch_bounds__(256) void instnorm_partial2_kernel(const Stats2Params q) {
  __shared__ f32x4 red[4][256];
  const StatsParams& p = q.a;
  const int n = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
  const int HW = p.H * p.W;
  const int PB = (HW + p.NB - 1) / p.NB;
  const int p0 = b * PB, p1 = min(p0 + PB, HW);
  const int GB = p.C4 < NORM_GCAP ? p.C4 : NORM_GCAP;
  const int R = 256 / GB;
  const int r = t / GB, gl = t - r * GB;
  const int g = blockIdx.z * GB + gl;
  const bool active = r < R && g < p.C4;
  f32x4 s1 = (f32x4)(0.f), s2 = (f32x4)(0.f), u1 = (f32x4)(0.f), u2 = (f32x4)(0.f);
  if (active) {
    const f32x4 na = *reinterpret_cast<const f32x4*>(p.ns + g * 4), nb = *reinterpret_cast<const f32x4*>(q.ns2 + g * 4);
    // (the thread's channel group is fixed: its source tensor / stride are resolved once)
    const bool lo = p.up_g > 0 && g < p.up_g;
    const bool hi2 = p.up_g > 0 && !lo;
    const int xcs = hi2 ? p.cs2 : p.cs, Wl = p.W >> 1;
    const float* const xb = hi2 ? p.x2 + (size_t)n * HW * p.cs2 + p.co2 + (g - p.up_g) * 4
                                : p.x + (size_t)n * (lo ? (p.H >> 1) * Wl : HW) * p.cs + p.co + g * 4;
    auto val = [&](int pix, f32x4& va, f32x4& vb) {
      const int h = pix / p.W, w = pix - h * p.W;
      const f32x4 x = *reinterpret_cast<const f32x4*>(xb + (size_t)(lo ? (h >> 1) * Wl + (w >> 1) : pix) * xcs);
      const size_t zi = ((size_t)n * p.W + w) * p.H + h;
      va = x + p.z[zi] * na;
      vb = x + q.z2[zi] * nb;
    };
    f32x4 Ka, Kb;
    val(0, Ka, Kb);
    int px = p0 + r;
    for (; px + 3 * R < p1; px += 4 * R) {
      f32x4 da[4], db[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        val(px + u * R, da[u], db[u]);
        da[u] -= Ka;
        db[u] -= Kb;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s1 += da[u]; s2 += da[u] * da[u]; u1 += db[u]; u2 += db[u] * db[u]; }
    }
    for (; px < p1; px += R) {
      f32x4 da, db;
      val(px, da, db);
      da -= Ka; db -= Kb;
      s1 += da; s2 += da * da; u1 += db; u2 += db * db;
    }
  }
  red[0][t] = s1; red[1][t] = s2; red[2][t] = u1; red[3][t] = u2;
  __syncthreads();
  if (r == 0 && g < p.C4) {
    for (int rr = 1; rr < R; ++rr) {
      s1 += red[0][rr * GB + gl]; s2 += red[1][rr * GB + gl];
      u1 += red[2][rr * GB + gl]; u2 += red[3][rr * GB + gl];
    }
    const size_t o = (((size_t)n * p.NB + b) * p.C4 * 4 + g * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p.part[o + 2 * e] = s1[e]; p.part[o + 2 * e + 1] = s2[e];
      q.part2[o + 2 * e] = u1[e]; q.part2[o + 2 * e + 1] = u2[e];
    }
  }
}

// out = lrelu((x - mean) * rstd)   (InstanceNorm2d(affine=False) + LeakyReLU(0.2), in place allowed)
__global__ void instnorm_apply_kernel(const float* __restrict__ x, int N, int HW, int C4, int cs, int co,
                                      const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                      float slope, float* __restrict__ out, int ocs, int oco) {
  const size_t total = (size_t)N * HW * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % C4);
    const size_t pix = i / C4;
    const int n = (int)(pix / HW);
    f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * cs + co + g * 4);
    const f32x4 m = *reinterpret_cast<const f32x4*>(mean + (size_t)n * C4 * 4 + g * 4);
    const f32x4 r = *reinterpret_cast<const f32x4*>(rstd + (size_t)n * C4 * 4 + g * 4);
    v = (v - m) * r;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act, slope);
    *reinterpret_cast<f32x4*>(out + pix * ocs + oco + g * 4) = v;
  }
}

// F.avg_pool2d(k=3, s=2, p=1, count_include_pad=False)
__global__ void avgpool3s2_kernel(const float* __restrict__ x, int N, int H, int W, int C4, int cs, int co, int Ho,
                                  int Wo, float* __restrict__ out, int ocs, int oco) {
  const size_t total = (size_t)N * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % C4);
    const size_t pix = i / C4;
    const int wo = (int)(pix % Wo);
    const size_t t = pix / Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    f32x4 s = (f32x4)(0.f);
    int cnt = 0;
    for (int dy = -1; dy <= 1; ++dy) {
      const int h = 2 * ho + dy;
      if (h < 0 || h >= H) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int w = 2 * wo + dx;
        if (w < 0 || w >= W) continue;
        s += *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + h) * W + w) * cs + co + g * 4);
        ++cnt;
      }
    }
    *reinterpret_cast<f32x4*>(out + pix * ocs + oco + g * 4) = s / (float)cnt;
  }
}

static inline int grid_for(size_t work, int block = 256) {
  size_t g = (work + block - 1) / block;
  const size_t cap = 256 * 16;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace hrv

using namespace hrv;

extern "C" int64_t hrv_instnorm_workspace_elems(int32_t N, int32_t H, int32_t W, int32_t C) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return -1;
  return (int64_t)N * norm_slabs(H * W) * ((C + 3) / 4 * 4) * 2;
}

template <bool BF>
static int instnorm_stats_impl(const void* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t cstride, int32_t coff,
                               const float* noise_z, const float* noise_scale, float eps, float* workspace, float* mean,
                               float* rstd, hrv_stream_t stream) {
  HRV_REQUIRE(x && workspace && mean && rstd && N > 0 && H > 0 && W > 0, "instnorm_stats: bad args");
  HRV_REQUIRE(C > 0 && C % 4 == 0 && cstride % 4 == 0 && coff % 4 == 0 && coff + C <= cstride,
              "instnorm_stats: channels must be multiples of 4 and in range");
  HRV_REQUIRE((noise_z == nullptr) == (noise_scale == nullptr), "instnorm_stats: noise_z and noise_scale go together");
  HRV_REQUIRE((((uintptr_t)x) & (BF ? 7 : 15)) == 0 && (((uintptr_t)noise_scale) & 15) == 0, "instnorm_stats: alignment");
  StatsParams p;
  p.x = (const float*)x; p.N = N; p.H = H; p.W = W; p.C4 = C / 4; p.cs = cstride; p.co = coff;
  p.z = noise_z; p.ns = noise_scale;
  p.NB = norm_slabs(H * W);
  p.part = workspace;
  p.x2 = nullptr; p.cs2 = p.co2 = p.up_g = 0;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(instnorm_partial_kernel<BF>, dim3(p.NB, N, norm_chunks(p.C4)), dim3(256), 0, st, p);
  int rc = check_launch("instnorm_partial_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(instnorm_finalize_kernel<BF>, dim3((N * C + 3) / 4), dim3(256), 0, st, p, eps, mean, rstd);
  return check_launch("instnorm_finalize_kernel");
}

static int instnorm_stats2_impl(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t cstride, int32_t coff, const float* x2,
                                int32_t cs2, int32_t co2, int32_t up_c, const float* z_a, const float* ns_a, const float* z_b,
                                const float* ns_b, float eps, float* workspace, float* mean_a, float* rstd_a, float* mean_b, float* rstd_b,
                                hrv_stream_t stream) {
  HRV_REQUIRE(x && workspace && mean_a && rstd_a && mean_b && rstd_b && z_a && ns_a && z_b && ns_b && N > 0 && H > 0 && W > 0,
              "instnorm_stats2: bad args");
  HRV_REQUIRE(C > 0 && C % 4 == 0 && cstride % 4 == 0 && coff % 4 == 0, "instnorm_stats2: channels must be multiples of 4");
  if (up_c > 0) {
    HRV_REQUIRE(x2 && up_c % 4 == 0 && up_c < C && H % 2 == 0 && W % 2 == 0 && coff + up_c <= cstride && cs2 % 4 == 0 && co2 % 4 == 0 &&
                    co2 + (C - up_c) <= cs2 && ((uintptr_t)x2 & 15) == 0,
                "instnorm_stats2: upsampled source (up_c %d of C %d, %d x %d)", up_c, C, H, W);
  } else {
    HRV_REQUIRE(coff + C <= cstride, "instnorm_stats2: slice out of range");
  }
  HRV_REQUIRE(((((uintptr_t)x) | (uintptr_t)ns_a | (uintptr_t)ns_b) & 15) == 0, "instnorm_stats2: alignment");
  Stats2Params q;
  StatsParams& p = q.a;
  p.x = x; p.N = N; p.H = H; p.W = W; p.C4 = C / 4; p.cs = cstride; p.co = coff;
  p.x2 = x2; p.cs2 = cs2; p.co2 = co2; p.up_g = up_c / 4;
  p.z = z_a; p.ns = ns_a;
  p.NB = norm_slabs(H * W);
  p.part = workspace;
  q.z2 = z_b; q.ns2 = ns_b;
  q.part2 = workspace + (size_t)N * p.NB * C * 2;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(instnorm_partial2_kernel, dim3(p.NB, N, norm_chunks(p.C4)), dim3(256), 0, st, q);
  int rc = check_launch("instnorm_partial2_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(instnorm_finalize_kernel<false>, dim3((N * C + 3) / 4), dim3(256), 0, st, p, eps, mean_a, rstd_a);
  StatsParams pb = p;
  pb.z = z_b; pb.ns = ns_b; pb.part = q.part2;
  hipLaunchKernelGGL(instnorm_finalize_kernel<false>, dim3((N * C + 3) / 4), dim3(256), 0, st, pb, eps, mean_b, rstd_b);
  return check_launch("instnorm_finalize_kernel");
}

extern "C" int hrv_instnorm_stats2_nhwc_f32(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t cstride, int32_t coff,
                                            const float* z_a, const float* ns_a, const float* z_b, const float* ns_b, float eps,
                                            float* workspace, float* mean_a, float* rstd_a, float* mean_b, float* rstd_b,
                                            hrv_stream_t stream) {
  return instnorm_stats2_impl(x, N, H, W, C, cstride, coff, nullptr, 0, 0, 0, z_a, ns_a, z_b, ns_b, eps, workspace, mean_a, rstd_a, mean_b,
                              rstd_b, stream);
}

extern "C" int hrv_instnorm_stats2_up_nhwc_f32(const float* lo, int32_t lo_cstride, int32_t lo_coff, int32_t up_channels, const float* hi,
                                               int32_t hi_cstride, int32_t hi_coff, int32_t N, int32_t H, int32_t W, int32_t C, const float* z_a,
                                               const float* ns_a, const float* z_b, const float* ns_b, float eps, float* workspace,
                                               float* mean_a, float* rstd_a, float* mean_b, float* rstd_b, hrv_stream_t stream) {
  HRV_REQUIRE(up_channels > 0, "instnorm_stats2_up: up_channels");
  return instnorm_stats2_impl(lo, N, H, W, C, lo_cstride, lo_coff, hi, hi_cstride, hi_coff, up_channels, z_a, ns_a, z_b, ns_b, eps, workspace,
                              mean_a, rstd_a, mean_b, rstd_b, stream);
}

extern "C" int hrv_instnorm_stats_nhwc_f32(const float* x, int32_t N, int32_t H, int32_t W, int32_t C,
                                           int32_t cstride, int32_t coff, const float* noise_z,
                                           const float* noise_scale, float eps, float* workspace, float* mean,
                                           float* rstd, hrv_stream_t stream) {
  return instnorm_stats_impl<false>(x, N, H, W, C, cstride, coff, noise_z, noise_scale, eps, workspace, mean, rstd, stream);
}

extern "C" int hrv_instnorm_stats_nhwc_bf16(const uint16_t* x, int32_t N, int32_t H, int32_t W, int32_t C,
                                            int32_t cstride, int32_t coff, const float* noise_z,
                                            const float* noise_scale, float eps, float* workspace, float* mean,
                                            float* rstd, hrv_stream_t stream) {
  return instnorm_stats_impl<true>(x, N, H, W, C, cstride, coff, noise_z, noise_scale, eps, workspace, mean, rstd, stream);
}

extern "C" int hrv_instnorm_apply_nhwc_f32(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t cstride,
                                           int32_t coff, const float* mean, const float* rstd, int32_t act,
                                           float act_slope, float* out, int32_t out_cstride, int32_t out_coff,
                                           hrv_stream_t stream) {
  HRV_REQUIRE(x && out && mean && rstd && N > 0 && H > 0 && W > 0, "instnorm_apply: bad args");
  HRV_REQUIRE(C > 0 && C % 4 == 0 && cstride % 4 == 0 && coff % 4 == 0 && out_cstride % 4 == 0 && out_coff % 4 == 0 &&
                  coff + C <= cstride && out_coff + C <= out_cstride,
              "instnorm_apply: channels must be multiples of 4 and in range");
  const size_t total = (size_t)N * H * W * (C / 4);
  hipLaunchKernelGGL(instnorm_apply_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, N, H * W,
                     C / 4, cstride, coff, mean, rstd, act, act_slope, out, out_cstride, out_coff);
  return check_launch("instnorm_apply_kernel");
}

extern "C" int hrv_avgpool3x3s2_nhwc_f32(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t cstride,
                                         int32_t coff, float* out, int32_t out_cstride, int32_t out_coff,
                                         hrv_stream_t stream) {
  HRV_REQUIRE(x && out && N > 0 && H > 0 && W > 0, "avgpool: bad args");
  HRV_REQUIRE(C > 0 && C % 4 == 0 && cstride % 4 == 0 && coff % 4 == 0 && out_cstride % 4 == 0 && out_coff % 4 == 0 &&
                  coff + C <= cstride && out_coff + C <= out_cstride,
              "avgpool: channels must be multiples of 4 and in range");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(avgpool3s2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C / 4,
                     cstride, coff, Ho, Wo, out, out_cstride, out_coff);
  return check_launch("avgpool3s2_kernel");
}
