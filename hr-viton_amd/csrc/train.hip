// Training-side HBM-bound kernels (NHWC fp32): InstanceNorm / SPADE backward, loss
// reductions with their gradients, 2x2 gradient down-sum (nearest-upsample backward),
// avg-pool backward, 2x2 max-pool forward/backward (VGG19), fused Adam, and the
// GEMVs of the spectral-norm power iteration.  All reductions are two-stage with a
// fixed summation order (deterministic).
#include <string.h>

#include "hrv_common.h"

namespace hrv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int grid_for(size_t work, int block = 256) {
  size_t g = (work + block - 1) / block;
  const size_t cap = 256 * 16;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__device__ __forceinline__ float dact(float y, int act, float slope) {
  // derivative of ReLU / LeakyReLU expressed through the activation's OUTPUT y
  if (act == HRV_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == HRV_ACT_LRELU) return y > 0.f ? 1.f : slope;
  return 1.f;
}

// ---------------------------------------------------------------------------
// SPADE / InstanceNorm backward, stage 1 (elementwise + per-(n,c) partial sums).
//   forward:  v = x + z*ns;  nh = (v - mean)*rstd;  out = act(nh*g1p + beta)   (g1p = 1+gamma)
//   given dout:  dpre = dout * act'(out);  dnh = dpre*g1p;  dgamma = dpre*nh;  dbeta = dpre
// plain InstanceNorm (+act) is the same with g1p == NULL (=1) and no dgamma/dbeta outputs.
// Writes dnh (needed again by stage 2), optionally dgb = [dgamma | dbeta] (2C channels) and
// the partial sums S1 = sum dnh, S2 = sum dnh*nh per (n, slab, c).
// ---------------------------------------------------------------------------
// x = cat(nearest_up2(lo), hi) along channels, never materialised (up_g > 0): channel groups [0, up_g) come from `x` = lo
// [N][H/2][W/2][x_cs] at (h >> 1, w >> 1), the others from `x2` = hi [N][H][W][x2_cs]
struct XSrc {
  const float* x; int x_cs, x_co;
  const float* x2; int x2_cs, x2_co, up_g;
  int H, W;
};
// a thread's channel group g is fixed: its source (tensor, stride, low-resolution or not) is resolved once, per pixel only the
// pixel index differs
struct XThread {
  const float* base;      // channel group g of pixel 0 of sample n
  int cs, lo, W, Wl;
};
__device__ __forceinline__ XThread xsrc_thread(const XSrc& s, int n, int g) {
  XThread t;
  t.W = s.W; t.Wl = s.W >> 1;
  t.lo = (s.up_g > 0 && g < s.up_g) ? 1 : 0;
  if (s.up_g > 0 && !t.lo) {
    t.cs = s.x2_cs;
    t.base = s.x2 + (size_t)n * s.H * s.W * s.x2_cs + s.x2_co + (g - s.up_g) * 4;
  } else {
    t.cs = s.x_cs;
    t.base = s.x + (size_t)n * (t.lo ? (s.H >> 1) * (s.W >> 1) : s.H * s.W) * s.x_cs + s.x_co + g * 4;
  }
  return t;
}
__device__ __forceinline__ const float* xsrc_ptr(const XThread& t, int px) {
  int q = px;
  if (t.lo) {
    const int h = px / t.W, w = px - h * t.W;
    q = (h >> 1) * t.Wl + (w >> 1);
  }
  return t.base + (size_t)q * t.cs;
}

struct NormBwdParams {
  XSrc xs;
  const float* z; const float* ns;           // noise (nullable)
  const float* mean; const float* rstd;      // [N][C]
  const float* out; int out_cs, out_co;      // activation output (mask), nullable when act == NONE
  const float* g1p; int g_cs, g_co;          // 1+gamma, nullable
  const float* dout; int do_cs, do_co;
  float* dnh; int dn_cs, dn_co;
  float* dgb; int dgb_cs, dgb_co;            // nullable; [.., 2C]: dgamma at [0,C), dbeta at [C,2C)
  int N, H, W, C4, act; float slope;
  int NB; float* part;                       // [N][NB][C][2]
  int dgb_bf16, out_bf16;                    // storage of dgb / out: bf16 when only matrix cores (and this mask) read them
  int g1p_bf16;                              // (1 + gamma) stored as bf16 (the dedicated gamma|beta kernel writes it so)
  int dnh_bf16;                              // dnh (stage 1 -> stage 2) stored as bf16
  int dout_bf16;                             // dout stored as bf16 (the data gradient of a bf16-stored SPADE output)
  int dbeta_in_place;                        // dout IS the dbeta half of dgb (its producer wrote it there, activation derivative applied): not stored again
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4_bf16(const void* base, size_t elem) {
  const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem);
  f32x4 v;
  v[0] = __builtin_bit_cast(float, u.x << 16); v[1] = __builtin_bit_cast(float, u.x & 0xFFFF0000u);
  v[2] = __builtin_bit_cast(float, u.y << 16); v[3] = __builtin_bit_cast(float, u.y & 0xFFFF0000u);
  return v;
}
typedef __bf16 bf16x4t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_bf16(void* base, size_t elem, f32x4 v) {
  *reinterpret_cast<bf16x4t*>(reinterpret_cast<unsigned short*>(base) + elem) = __builtin_convertvector(v, bf16x4t);
}

__global__ __launch_bounds__(256) void norm_bwd_stage1_kernel(const NormBwdParams p) {
  __shared__ f32x4 red[2][256];
  const int n = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
  const int HW = p.H * p.W, C = p.C4 * 4;
  const int PB = (HW + p.NB - 1) / p.NB;
  const int p0 = b * PB, p1 = min(p0 + PB, HW);
  const int GB = p.C4 < NORM_GCAP ? p.C4 : NORM_GCAP;   // channel groups of this block (blockIdx.z picks the chunk)
  const int R = 256 / GB;
  const int r = t / GB, gl = t - r * GB;
  {
    const int g = blockIdx.z * GB + gl;
    f32x4 s1 = (f32x4)(0.f), s2 = (f32x4)(0.f);
    if (r < R && g < p.C4) {
      const f32x4 mu = ld4(p.mean + (size_t)n * C + g * 4), rs = ld4(p.rstd + (size_t)n * C + g * 4);
      const f32x4 ns4 = p.z ? ld4(p.ns + g * 4) : (f32x4)(0.f);
      const XThread xt = xsrc_thread(p.xs, n, g);
      // two pixels per iteration: all eight loads of both are requested before the first result is stored (the stores may
      // alias the loads as far as the compiler knows, so a plain loop keeps one pixel's four loads in flight per thread);
      // every value and the order of the two sums are those of the plain loop
      struct In { f32x4 v, d, o, g1; float zz; };
      auto load = [&](int px) {
        In L;
        const size_t pix = (size_t)n * HW + px;
        L.v = ld4(xsrc_ptr(xt, px));
        L.zz = 0.f;
        if (p.z) {
          const int h = px / p.W, w = px - h * p.W;
          L.zz = p.z[((size_t)n * p.W + w) * p.H + h];
        }
        L.d = p.dout_bf16 ? ld4_bf16(p.dout, pix * p.do_cs + p.do_co + g * 4) : ld4(p.dout + pix * p.do_cs + p.do_co + g * 4);
        L.o = (f32x4)(0.f);
        if (p.act != HRV_ACT_NONE) {
          const size_t oe = pix * p.out_cs + p.out_co + g * 4;
          L.o = p.out_bf16 ? ld4_bf16(p.out, oe) : ld4(p.out + oe);
        }
        L.g1 = (f32x4)(1.f);
        if (p.g1p) {
          const size_t ge1 = pix * p.g_cs + p.g_co + g * 4;
          L.g1 = p.g1p_bf16 ? ld4_bf16(p.g1p, ge1) : ld4(p.g1p + ge1);
        }
        return L;
      };
      auto finish = [&](int px, const In& L) {
        const size_t pix = (size_t)n * HW + px;
        f32x4 v = L.v;
        if (p.z) v += L.zz * ns4;
        const f32x4 nh = (v - mu) * rs;
        f32x4 dpre = L.d;
        if (p.act != HRV_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dpre[e] *= dact(L.o[e], p.act, p.slope);
        }
        f32x4 dnh = dpre;
        if (p.g1p) dnh *= L.g1;
        if (p.dnh_bf16) st4_bf16(p.dnh, pix * p.dn_cs + p.dn_co + g * 4, dnh);
        else *reinterpret_cast<f32x4*>(p.dnh + pix * p.dn_cs + p.dn_co + g * 4) = dnh;
        if (p.dgb) {
          const size_t ge = pix * p.dgb_cs + p.dgb_co + g * 4;
          if (p.dgb_bf16) {
            st4_bf16(p.dgb, ge, dpre * nh);
            if (!p.dbeta_in_place) st4_bf16(p.dgb, ge + C, dpre);
          } else {
            *reinterpret_cast<f32x4*>(p.dgb + ge) = dpre * nh;
            if (!p.dbeta_in_place) *reinterpret_cast<f32x4*>(p.dgb + ge + C) = dpre;
          }
        }
        s1 += dnh;
        s2 += dnh * nh;
      };
      int px = p0 + r;
      for (; px + R < p1; px += 2 * R) {
        const In A = load(px), B = load(px + R);
        finish(px, A);
        finish(px + R, B);
      }
      if (px < p1) finish(px, load(px));
    }
    red[0][t] = s1;
    red[1][t] = s2;
    __syncthreads();
    if (r == 0 && g < p.C4) {
      for (int rr = 1; rr < R; ++rr) { s1 += red[0][rr * GB + gl]; s2 += red[1][rr * GB + gl]; }
      float* dst = p.part + (((size_t)n * p.NB + b) * C + g * 4) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { dst[2 * e] = s1[e]; dst[2 * e + 1] = s2[e]; }
    }
  }
}

// fixed-order reduction of the slab partials: m1[n][c] = S1/HW, m2[n][c] = S2/HW
// 16 lanes per (sample, channel): lane l sums slabs l, l + 16, ... in double, then a fixed butterfly inside the 16-lane group
// (deterministic).  (One thread per (n, c) walking up to 256 slabs took 35 us; 31 of these per training step.)
__global__ void norm_bwd_finalize_kernel(const float* __restrict__ part, int N, int NB, int C, int HW,
                                         float* __restrict__ m1, float* __restrict__ m2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 4, l = t & 15;
  const bool live = i < N * C;
  const int ii = live ? i : 0;
  const int n = ii / C, c = ii - n * C;
  double s1 = 0.0, s2 = 0.0;
  for (int b = l; b < NB; b += 16) {
    const float* src = part + (((size_t)n * NB + b) * C + c) * 2;
    s1 += (double)src[0];
    s2 += (double)src[1];
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 16);
    s2 += __shfl_xor(s2, o, 16);
  }
  if (live && l == 0) {
    m1[i] = (float)(s1 / HW);
    m2[i] = (float)(s2 / HW);
  }
}

// stage 2: dx = rstd * (dnh - m1 - nh*m2)  (+ optional accumulate into dx), and partial sums of
// dx*z per (n, slab, c) for the noise_scale gradient.
struct NormBwd2Params {
  XSrc xs;
  const float* z; const float* ns;
  const float* mean; const float* rstd; const float* m1; const float* m2;
  const float* dnh; int dn_cs, dn_co; int dnh_bf16;
  float* dx; int dx_cs, dx_co; int accumulate;
  int dx_bf16;
  int N, H, W, C4;
  int NB; float* part;  // [N][NB][C] (only when z != NULL)
};

__global__ __launch_bounds__(256) void norm_bwd_stage2_kernel(const NormBwd2Params p) {
  __shared__ f32x4 red[256];
  const int n = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
  const int HW = p.H * p.W, C = p.C4 * 4;
  const int PB = (HW + p.NB - 1) / p.NB;
  const int p0 = b * PB, p1 = min(p0 + PB, HW);
  const int GB = p.C4 < NORM_GCAP ? p.C4 : NORM_GCAP;
  const int R = 256 / GB;
  const int r = t / GB, gl = t - r * GB;
  {
    const int g = blockIdx.z * GB + gl;
    f32x4 sz = (f32x4)(0.f);
    if (r < R && g < p.C4) {
      const size_t sc = (size_t)n * C + g * 4;
      const f32x4 mu = ld4(p.mean + sc), rs = ld4(p.rstd + sc), a1 = ld4(p.m1 + sc), a2 = ld4(p.m2 + sc);
      const f32x4 ns4 = p.z ? ld4(p.ns + g * 4) : (f32x4)(0.f);
      const XThread xt = xsrc_thread(p.xs, n, g);
      // two pixels per iteration, loads of both first (see stage 1); values and the order of the sum are unchanged
      struct In { f32x4 v, dn, acc; float zz; };
      auto load = [&](int px) {
        In L;
        const size_t pix = (size_t)n * HW + px;
        L.v = ld4(xsrc_ptr(xt, px));
        L.zz = 0.f;
        if (p.z) {
          const int h = px / p.W, w = px - h * p.W;
          L.zz = p.z[((size_t)n * p.W + w) * p.H + h];
        }
        const size_t de = pix * p.dn_cs + p.dn_co + g * 4;
        L.dn = p.dnh_bf16 ? ld4_bf16(p.dnh, de) : ld4(p.dnh + de);
        L.acc = (f32x4)(0.f);
        if (!p.dx_bf16 && p.accumulate) L.acc = ld4(p.dx + pix * p.dx_cs + p.dx_co + g * 4);
        return L;
      };
      auto finish = [&](int px, const In& L) {
#pragma clang fp contract(off)      // (norm_bwd2_stage2_kernel computes the same values in one pass: keep the roundings identical)
        const size_t pix = (size_t)n * HW + px;
        f32x4 v = L.v;
        if (p.z) v += L.zz * ns4;
        const f32x4 nh = (v - mu) * rs;
        f32x4 d = rs * (L.dn - a1 - nh * a2);
        sz += d * L.zz;
        if (p.dx_bf16) {
          st4_bf16(p.dx, pix * p.dx_cs + p.dx_co + g * 4, d);
        } else {
          float* o = p.dx + pix * p.dx_cs + p.dx_co + g * 4;
          if (p.accumulate) d += L.acc;
          *reinterpret_cast<f32x4*>(o) = d;
        }
      };
      int px = p0 + r;
      for (; px + R < p1; px += 2 * R) {
        const In A = load(px), B = load(px + R);
        finish(px, A);
        finish(px + R, B);
      }
      if (px < p1) finish(px, load(px));
    }
    if (p.z) {
      red[t] = sz;
      __syncthreads();
      if (r == 0 && g < p.C4) {
        for (int rr = 1; rr < R; ++rr) sz += red[rr * GB + gl];
        *reinterpret_cast<f32x4*>(p.part + ((size_t)n * p.NB + b) * C + g * 4) = sz;
      }
    }
  }
}

// ---- two normalisations over the SAME x (norm_0 and norm_s of a learned-shortcut SPADEResBlock both normalise the block
// input, network_generator.py:158-166): x is read once per stage, dx = dx_a + dx_b is written once (no read-modify-write of
// the first norm's result).  Every value is computed as in the single kernels (dx: one fp32 add, commutative), so the results
// are bit-identical to two sequential calls with dx_accumulate on the second.
__global__ __launch_bounds__(256) void norm_bwd2_stage1_kernel(const NormBwdParams pa, const NormBwdParams pb) {
  __shared__ f32x4 red[4][256];
  const NormBwdParams& p = pa;                       // geometry and x are shared
  const int n = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
  const int HW = p.H * p.W, C = p.C4 * 4;
  const int PB = (HW + p.NB - 1) / p.NB;
  const int p0 = b * PB, p1 = min(p0 + PB, HW);
  const int GB = p.C4 < NORM_GCAP ? p.C4 : NORM_GCAP;
  const int R = 256 / GB;
  const int r = t / GB, gl = t - r * GB;
  const int g = blockIdx.z * GB + gl;
  f32x4 s1a = (f32x4)(0.f), s2a = (f32x4)(0.f), s1b = (f32x4)(0.f), s2b = (f32x4)(0.f);
  if (r < R && g < p.C4) {
    const f32x4 mua = ld4(pa.mean + (size_t)n * C + g * 4), rsa = ld4(pa.rstd + (size_t)n * C + g * 4);
    const f32x4 mub = ld4(pb.mean + (size_t)n * C + g * 4), rsb = ld4(pb.rstd + (size_t)n * C + g * 4);
    const f32x4 nsa = pa.z ? ld4(pa.ns + g * 4) : (f32x4)(0.f), nsb = pb.z ? ld4(pb.ns + g * 4) : (f32x4)(0.f);
    const XThread xt = xsrc_thread(p.xs, n, g);
    struct In1 { f32x4 d, o, g1; float zz; };
    struct In { f32x4 v; In1 a, b; };
    // (the per-norm pieces take their parameter block by reference to the kernel argument itself: no pointer tables, which
    //  would force the arguments into scratch memory)
    auto load1 = [&](const NormBwdParams& q, size_t pix, int h, int w) {
      In1 L;
      L.zz = q.z ? q.z[((size_t)n * p.W + w) * p.H + h] : 0.f;
      L.d = q.dout_bf16 ? ld4_bf16(q.dout, pix * q.do_cs + q.do_co + g * 4) : ld4(q.dout + pix * q.do_cs + q.do_co + g * 4);
      L.o = (f32x4)(0.f);
      if (q.act != HRV_ACT_NONE) {
        const size_t oe = pix * q.out_cs + q.out_co + g * 4;
        L.o = q.out_bf16 ? ld4_bf16(q.out, oe) : ld4(q.out + oe);
      }
      L.g1 = (f32x4)(1.f);
      if (q.g1p) {
        const size_t ge1 = pix * q.g_cs + q.g_co + g * 4;
        L.g1 = q.g1p_bf16 ? ld4_bf16(q.g1p, ge1) : ld4(q.g1p + ge1);
      }
      return L;
    };
    auto load = [&](int px) {
      In L;
      const size_t pix = (size_t)n * HW + px;
      L.v = ld4(xsrc_ptr(xt, px));
      const int h = px / p.W, w = px - h * p.W;
      L.a = load1(pa, pix, h, w);
      L.b = load1(pb, pix, h, w);
      return L;
    };
    auto finish1 = [&](const NormBwdParams& q, size_t pix, f32x4 v, const In1& L, f32x4 mu, f32x4 rs, f32x4 ns4, f32x4& s1, f32x4& s2) {
      if (q.z) v += L.zz * ns4;
      const f32x4 nh = (v - mu) * rs;
      f32x4 dpre = L.d;
      if (q.act != HRV_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dpre[e] *= dact(L.o[e], q.act, q.slope);
      }
      f32x4 dnh = dpre;
      if (q.g1p) dnh *= L.g1;
      if (q.dnh_bf16) st4_bf16(q.dnh, pix * q.dn_cs + q.dn_co + g * 4, dnh);
      else *reinterpret_cast<f32x4*>(q.dnh + pix * q.dn_cs + q.dn_co + g * 4) = dnh;
      if (q.dgb) {
        const size_t ge = pix * q.dgb_cs + q.dgb_co + g * 4;
        if (q.dgb_bf16) {
          st4_bf16(q.dgb, ge, dpre * nh);
          if (!q.dbeta_in_place) st4_bf16(q.dgb, ge + C, dpre);
        } else {
          *reinterpret_cast<f32x4*>(q.dgb + ge) = dpre * nh;
          if (!q.dbeta_in_place) *reinterpret_cast<f32x4*>(q.dgb + ge + C) = dpre;
        }
      }
      s1 += dnh;
      s2 += dnh * nh;
    };
    auto finish = [&](int px, const In& L) {
      const size_t pix = (size_t)n * HW + px;
      finish1(pa, pix, L.v, L.a, mua, rsa, nsa, s1a, s2a);
      finish1(pb, pix, L.v, L.b, mub, rsb, nsb, s1b, s2b);
    };
    // one pixel per iteration: its seven 16-byte loads (x + three per norm) are as many as the single kernel keeps in flight
    // with two pixels, at half the registers of a two-pixel body (231 -> two waves per SIMD)
    for (int px = p0 + r; px < p1; px += R) finish(px, load(px));
  }
  red[0][t] = s1a; red[1][t] = s2a; red[2][t] = s1b; red[3][t] = s2b;
  __syncthreads();
  if (r == 0 && g < p.C4) {
    for (int rr = 1; rr < R; ++rr) {
      s1a += red[0][rr * GB + gl]; s2a += red[1][rr * GB + gl];
      s1b += red[2][rr * GB + gl]; s2b += red[3][rr * GB + gl];
    }
    const size_t o = (((size_t)n * p.NB + b) * C + g * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pa.part[o + 2 * e] = s1a[e]; pa.part[o + 2 * e + 1] = s2a[e];
      pb.part[o + 2 * e] = s1b[e]; pb.part[o + 2 * e + 1] = s2b[e];
    }
  }
}

__global__ __launch_bounds__(256) void norm_bwd2_stage2_kernel(const NormBwd2Params pa, const NormBwd2Params pb) {
  __shared__ f32x4 red[2][256];
  const NormBwd2Params& p = pa;
  const int n = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
  const int HW = p.H * p.W, C = p.C4 * 4;
  const int PB = (HW + p.NB - 1) / p.NB;
  const int p0 = b * PB, p1 = min(p0 + PB, HW);
  const int GB = p.C4 < NORM_GCAP ? p.C4 : NORM_GCAP;
  const int R = 256 / GB;
  const int r = t / GB, gl = t - r * GB;
  const int g = blockIdx.z * GB + gl;
  f32x4 sza = (f32x4)(0.f), szb = (f32x4)(0.f);
  if (r < R && g < p.C4) {
    const size_t sc = (size_t)n * C + g * 4;
    const f32x4 mua = ld4(pa.mean + sc), rsa = ld4(pa.rstd + sc), a1a = ld4(pa.m1 + sc), a2a = ld4(pa.m2 + sc);
    const f32x4 mub = ld4(pb.mean + sc), rsb = ld4(pb.rstd + sc), a1b = ld4(pb.m1 + sc), a2b = ld4(pb.m2 + sc);
    const f32x4 nsa = pa.z ? ld4(pa.ns + g * 4) : (f32x4)(0.f), nsb = pb.z ? ld4(pb.ns + g * 4) : (f32x4)(0.f);
    const XThread xt = xsrc_thread(p.xs, n, g);
    struct In { f32x4 v, dna, dnb; float za, zb; };
    auto load = [&](int px) {
      In L;
      const size_t pix = (size_t)n * HW + px;
      L.v = ld4(xsrc_ptr(xt, px));
      const int h = px / p.W, w = px - h * p.W;
      const size_t zi = ((size_t)n * p.W + w) * p.H + h;
      L.za = pa.z ? pa.z[zi] : 0.f;
      L.zb = pb.z ? pb.z[zi] : 0.f;
      const size_t dea = pix * pa.dn_cs + pa.dn_co + g * 4, deb = pix * pb.dn_cs + pb.dn_co + g * 4;
      L.dna = pa.dnh_bf16 ? ld4_bf16(pa.dnh, dea) : ld4(pa.dnh + dea);
      L.dnb = pb.dnh_bf16 ? ld4_bf16(pb.dnh, deb) : ld4(pb.dnh + deb);
      return L;
    };
    auto finish = [&](int px, const In& L) {
#pragma clang fp contract(off)      // (d_a and d_b are rounded products, their sum one add: as the two sequential calls compute them)
      const size_t pix = (size_t)n * HW + px;
      f32x4 va = L.v, vb = L.v;
      if (pa.z) va += L.za * nsa;
      if (pb.z) vb += L.zb * nsb;
      const f32x4 nha = (va - mua) * rsa, nhb = (vb - mub) * rsb;
      const f32x4 da = rsa * (L.dna - a1a - nha * a2a);
      const f32x4 db = rsb * (L.dnb - a1b - nhb * a2b);
      sza += da * L.za;
      szb += db * L.zb;
      *reinterpret_cast<f32x4*>(p.dx + pix * p.dx_cs + p.dx_co + g * 4) = db + da;
    };
    int px = p0 + r;
    for (; px + R < p1; px += 2 * R) {
      const In A = load(px), B = load(px + R);
      finish(px, A);
      finish(px + R, B);
    }
    if (px < p1) finish(px, load(px));
  }
  red[0][t] = sza; red[1][t] = szb;
  __syncthreads();
  if (r == 0 && g < p.C4) {
    for (int rr = 1; rr < R; ++rr) { sza += red[0][rr * GB + gl]; szb += red[1][rr * GB + gl]; }
    if (pa.z) *reinterpret_cast<f32x4*>(pa.part + ((size_t)n * p.NB + b) * C + g * 4) = sza;
    if (pb.z) *reinterpret_cast<f32x4*>(pb.part + ((size_t)n * p.NB + b) * C + g * 4) = szb;
  }
}

// ---------------------------------------------------------------------------
// losses: value + gradient in one pass.  mode: 0 L1 |a-b| ; 1 hinge-D fake max(1+a,0) ;
// 2 hinge-D real max(1-a,0) ; 3 -a (generator hinge / wgan) ; 4 (a-b)^2 (LSGAN/MSE)
// grad[i] = gscale * dl/da ; loss = lscale * sum(l)   (callers pass 1/numel etc.)
// ---------------------------------------------------------------------------
template <bool AB>
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                   int mode_flags, float gscale, float* __restrict__ grad,
                                                   float* __restrict__ part, int vec) {
  __shared__ float red[256];
  float s = 0.f;
  const int mode = mode_flags & 15;
  const bool relu_mask = (mode_flags & 16) != 0;   // a = ReLU(pre): the gradient is handed back w.r.t. pre (x (a > 0))
  const bool grad_bf16 = (mode_flags & 32) != 0;   // the gradient is stored as bf16 (its readers: matrix cores, pool backward)
  auto term = [&](float x, float y, float& g) -> float {
    float l;
    if (mode == 0) { const float d = x - y; l = fabsf(d); g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
    else if (mode == 1) { l = fmaxf(1.f + x, 0.f); g = x > -1.f ? 1.f : 0.f; }
    else if (mode == 2) { l = fmaxf(1.f - x, 0.f); g = x < 1.f ? -1.f : 0.f; }
    else if (mode == 3) { l = -x; g = -1.f; }
    else { const float d = x - y; l = d * d; g = 2.f * d; }
    g = (relu_mask && !(x > 0.f)) ? 0.f : gscale * g;
    return l;
  };
  typedef unsigned short u16x4v __attribute__((ext_vector_type(4)));
  auto ld4v = [](const float* p, size_t k) -> f32x4 {      // 4 consecutive elements starting at element k
    if constexpr (AB) {
      const u16x4v h = *reinterpret_cast<const u16x4v*>(reinterpret_cast<const unsigned short*>(p) + k);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __builtin_bit_cast(float, (unsigned)h[e] << 16);
      return v;
    } else {
      return *reinterpret_cast<const f32x4*>(p + k);
    }
  };
  // 16-byte (fp32) / 8-byte (bf16) accesses: four consecutive elements per thread and iteration
  const size_t n4 = vec ? n / 4 : 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 x4 = ld4v(a, 4 * i);
    const f32x4 y4 = b ? ld4v(b, 4 * i) : (f32x4)(0.f);
    f32x4 g4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float g;
      s += term(x4[e], y4[e], g);
      g4[e] = g;
    }
    if (grad) {
      if (grad_bf16) st4_bf16(grad, 4 * i, g4);
      else *reinterpret_cast<f32x4*>(grad + 4 * i) = g4;
    }
  }
  for (size_t i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float x, y = 0.f, g;
    if constexpr (AB) {
      x = __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(a)[i] << 16);
      if (b) y = __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(b)[i] << 16);
    } else {
      x = a[i];
      if (b) y = b[i];
    }
    s += term(x, y, g);
    if (grad) {
      if (grad_bf16) reinterpret_cast<unsigned short*>(grad)[i] = (unsigned short)(__builtin_bit_cast(unsigned, __builtin_bit_cast(unsigned, g) + 0x7fffu + ((__builtin_bit_cast(unsigned, g) >> 16) & 1u)) >> 16);
      else grad[i] = g;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// one wave: lane l sums part[l], part[l + 64], ... in double, then a fixed butterfly (deterministic).  (One thread walking
// up to 1024 partials took 25 us; 44 of these per training step.)
__global__ void loss_final_kernel(const float* __restrict__ part, int nb, float lscale, float* __restrict__ out,
                                  int accumulate) {
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) s += (double)part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (threadIdx.x == 0) {
    const float v = (float)(s * (double)lscale);
    out[0] = accumulate ? out[0] + v : v;
  }
}

// x *= s_host * (s_dev ? s_dev[0] : 1)   (upstream loss-gradient scalar without a host sync)
__global__ void scale_kernel(float* __restrict__ x, size_t n, float s_host, const float* __restrict__ s_dev) {
  const float s = s_dev ? s_host * s_dev[0] : s_host;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= s;
}

// out = dy * (1 - y*y)   (tanh backward through its output)   /   out (+)= a  (gradient accumulation)
__global__ void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, size_t n, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float t = y[i];
    out[i] = dy[i] * (1.f - t * t);
  }
}

// d <- d * act'(y)  in place (activation derivative through the activation's output)
__global__ void act_bwd_kernel(float* __restrict__ d, int dcs, int dco, const float* __restrict__ y, int ycs, int yco, int C4,
                               size_t npix, int act, float slope) {
  const size_t total = npix * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % C4);
    const size_t pix = i / C4;
    float* o = d + pix * dcs + dco + g * 4;
    f32x4 v = ld4(o);
    const f32x4 yy = ld4(y + pix * ycs + yco + g * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= dact(yy[e], act, slope);
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

__global__ void add_slice_kernel(const float* __restrict__ a, int acs, int aco, float* __restrict__ out, int ocs, int oco,
                                 int C4, size_t npix, int accumulate) {
  const size_t total = npix * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % C4);
    const size_t pix = i / C4;
    f32x4 v = ld4(a + pix * acs + aco + g * 4);
    float* o = out + pix * ocs + oco + g * 4;
    if (accumulate) v += ld4(o);
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

// ---------------------------------------------------------------------------
// nearest-x2 upsample backward: dlo[n,h,w,c] = sum of the 2x2 block of dhi
// ---------------------------------------------------------------------------
__global__ void downsum2x2_kernel(const float* __restrict__ dhi, int N, int Hl, int Wl, int C4, int hcs, int hco,
                                  float* __restrict__ dlo, int lcs, int lco, int accumulate) {
  const size_t total = (size_t)N * Hl * Wl * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % C4);
    const size_t pix = i / C4;
    const int w = (int)(pix % Wl);
    const size_t t = pix / Wl;
    const int h = (int)(t % Hl);
    const int n = (int)(t / Hl);
    const float* s = dhi + (((size_t)n * 2 * Hl + 2 * h) * 2 * Wl + 2 * w) * hcs + hco + g * 4;
    f32x4 v = ld4(s) + ld4(s + hcs) + ld4(s + (size_t)2 * Wl * hcs) + ld4(s + (size_t)2 * Wl * hcs + hcs);
    if (accumulate == 2) {      // bf16 result (element strides): a gradient only matrix cores read
      st4_bf16(dlo, pix * lcs + lco + g * 4, v);
      continue;
    }
    float* o = dlo + pix * lcs + lco + g * 4;
    if (accumulate) v += ld4(o);
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

// avg_pool2d(3, s2, p1, count_include_pad=False) backward: dx[h,w] = sum over the (<= 4) windows that
// contain (h,w) of dy[ho,wo] / count(ho,wo)
__global__ void avgpool3s2_bwd_kernel(const float* __restrict__ dy, int N, int H, int W, int C4, int Ho, int Wo,
                                      int ycs, int yco, float* __restrict__ dx, int xcs, int xco, int accumulate) {
  const size_t total = (size_t)N * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % C4);
    const size_t pix = i / C4;
    const int w = (int)(pix % W);
    const size_t t = pix / W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    f32x4 s = (f32x4)(0.f);
    for (int ho = h / 2; ho <= (h + 1) / 2; ++ho) {
      if (ho >= Ho) continue;
      const int ch = min(2 * ho + 1, H - 1) - max(2 * ho - 1, 0) + 1;
      for (int wo = w / 2; wo <= (w + 1) / 2; ++wo) {
        if (wo >= Wo) continue;
        const int cw = min(2 * wo + 1, W - 1) - max(2 * wo - 1, 0) + 1;
        s += ld4(dy + ((size_t)(n * Ho + ho) * Wo + wo) * ycs + yco + g * 4) / (float)(ch * cw);
      }
    }
    float* o = dx + pix * xcs + xco + g * 4;
    if (accumulate) s += ld4(o);
    *reinterpret_cast<f32x4*>(o) = s;
  }
}

// ---------------------------------------------------------------------------
// VGG19: 2x2 stride-2 max pool, forward and backward (first maximum in (0,0),(0,1),(1,0),(1,1)
// scan order takes the gradient, like torch)
// ---------------------------------------------------------------------------
// XB: x (and y) are bf16-stored (mixed-precision VGG activations: matrix cores, pools and the L1 taps are the only readers)
template <bool XB>
__device__ __forceinline__ f32x4 ldg4(const float* base, size_t idx) {
  if constexpr (XB) {
    // loaded as a vector of the STORED element type: reading the bf16 data through uint2 / f32x2 lvalues here came out of the
    // compiler as one dword load with every lane a copy of element 0 (type-punned access)
    typedef unsigned short u16x4v __attribute__((ext_vector_type(4)));
    const u16x4v h = *reinterpret_cast<const u16x4v*>(reinterpret_cast<const unsigned short*>(base) + idx);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_bit_cast(float, (unsigned)h[e] << 16);
    return v;
  } else {
    return ld4(base + idx);
  }
}

template <bool XB>
__global__ void maxpool2_kernel(const float* __restrict__ x, int N, int Ho, int Wo, int C4, int xcs, float* __restrict__ y,
                                int ycs) {
  const size_t total = (size_t)N * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % C4);
    const size_t pix = i / C4;
    const int wo = (int)(pix % Wo);
    const size_t t = pix / Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const size_t s = (((size_t)n * 2 * Ho + 2 * ho) * 2 * Wo + 2 * wo) * xcs + g * 4;
    if constexpr (XB) {
      // bf16 storage: the inputs are ReLU outputs (>= 0, never NaN), for which the order of the 16-bit patterns read as SIGNED
      // integers IS the order of the values (-0 = 0x8000 = -32768 sorts below every +x: a ReLU written as max(v, v * 0) stores
      // -0 for v < 0) -- an integer max on the stored elements, exact, no conversion
      typedef short u16x4v __attribute__((ext_vector_type(4)));
      const unsigned short* xs = reinterpret_cast<const unsigned short*>(x);
      const u16x4v a = *reinterpret_cast<const u16x4v*>(xs + s), b = *reinterpret_cast<const u16x4v*>(xs + s + xcs),
                   c = *reinterpret_cast<const u16x4v*>(xs + s + (size_t)2 * Wo * xcs),
                   d = *reinterpret_cast<const u16x4v*>(xs + s + (size_t)2 * Wo * xcs + xcs);
      u16x4v o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const short p = a[e] > b[e] ? a[e] : b[e], q = c[e] > d[e] ? c[e] : d[e];
        o[e] = p > q ? p : q;
      }
      *reinterpret_cast<u16x4v*>(reinterpret_cast<unsigned short*>(y) + pix * ycs + g * 4) = o;
    } else {
      const f32x4 a = ld4(x + s), b = ld4(x + s + xcs), c = ld4(x + s + (size_t)2 * Wo * xcs), d = ld4(x + s + (size_t)2 * Wo * xcs + xcs);
      f32x4 m;
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
      *reinterpret_cast<f32x4*>(y + pix * ycs + g * 4) = m;
    }
  }
}

// XB: x is bf16-stored; GB: dy and dx are bf16-stored too (mixed-precision VGG backward: gradients that only matrix cores
// and this routing read)
template <bool XB, bool GB = false>
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int N, int Ho, int Wo,
                                    int C4, int xcs, int ycs, float* __restrict__ dx, int relu) {
  const size_t total = (size_t)N * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % C4);
    const size_t pix = i / C4;
    const int wo = (int)(pix % Wo);
    const size_t t = pix / Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const size_t base = (((size_t)n * 2 * Ho + 2 * ho) * 2 * Wo + 2 * wo) * xcs + g * 4;
    const size_t o01 = xcs, o10 = (size_t)2 * Wo * xcs, o11 = o10 + xcs;
    const f32x4 a = ldg4<XB>(x, base), b = ldg4<XB>(x, base + o01), c = ldg4<XB>(x, base + o10), d = ldg4<XB>(x, base + o11);
    const f32x4 gy = GB ? ld4_bf16(dy, pix * ycs + g * 4) : ld4(dy + pix * ycs + g * 4);
    f32x4 ga = (f32x4)(0.f), gb = ga, gc = ga, gd = ga;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float m = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
      const float gv = (relu && !(m > 0.f)) ? 0.f : gy[e];      // relu: x = ReLU(pre), the gradient goes on to pre
      if (a[e] == m) ga[e] = gv;
      else if (b[e] == m) gb[e] = gv;
      else if (c[e] == m) gc[e] = gv;
      else gd[e] = gv;
    }
    if constexpr (GB) {
      st4_bf16(dx, base, ga); st4_bf16(dx, base + o01, gb); st4_bf16(dx, base + o10, gc); st4_bf16(dx, base + o11, gd);
    } else {
      *reinterpret_cast<f32x4*>(dx + base) = ga;
      *reinterpret_cast<f32x4*>(dx + base + o01) = gb;
      *reinterpret_cast<f32x4*>(dx + base + o10) = gc;
      *reinterpret_cast<f32x4*>(dx + base + o11) = gd;
    }
  }
}

// out (+)= a over bf16-stored tensors (fp32 sum, one rounding)
__global__ void add_slice_bf16_kernel(const unsigned short* __restrict__ a, int acs, int aco, unsigned short* __restrict__ out,
                                      int ocs, int oco, int C4, size_t npix, int accumulate) {
  const size_t total = npix * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % C4);
    const size_t pix = i / C4;
    f32x4 v = ld4_bf16(a, pix * acs + aco + g * 4);
    const size_t oe = pix * ocs + oco + g * 4;
    if (accumulate) v += ld4_bf16(out, oe);
    st4_bf16(out, oe, v);
  }
}

// ---------------------------------------------------------------------------
// Adam (torch.optim.Adam semantics: no amsgrad, optional L2 weight decay), fused over one
// flat buffer.  step_size = lr / (1 - b1^t);  denom = sqrt(v)/sqrt(1 - b2^t) + eps
// ---------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                            float bc1, float bc2_sqrt, float gscale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float wi = w[i];
    if (wd != 0.f) gi += wd * wi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    w[i] = wi - (lr / bc1) * (mi / denom);
  }
}

// The same update with its step-dependent scalars on the DEVICE (hipGraph-capturable training iteration: a replay must not
// freeze the step count or the learning rate into the launch arguments).  adam_hyper_kernel: ++step; hyper = {lr, 1 - b1^t,
// sqrt(1 - b2^t)} (the same expressions the host evaluates for hrv_adam_f32, here with the device's powf / sqrtf).
__global__ void adam_hyper_kernel(int* __restrict__ step, const float* __restrict__ lr, float b1, float b2,
                                  float* __restrict__ hyper) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int t = step[0] + 1;
    step[0] = t;
    hyper[0] = lr[0];
    hyper[1] = 1.f - powf(b1, (float)t);
    hyper[2] = sqrtf(1.f - powf(b2, (float)t));
  }
}

__global__ void adam_dev_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, size_t n, const float* __restrict__ hyper, float b1, float b2, float eps,
                                float wd, float gscale) {
  const float lr = hyper[0], bc1 = hyper[1], bc2_sqrt = hyper[2];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float wi = w[i];
    if (wd != 0.f) gi += wd * wi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    w[i] = wi - (lr / bc1) * (mi / denom);
  }
}

// ---------------------------------------------------------------------------
// Spectral-norm power iteration GEMVs on W [R][K] (row-major):  y = W x  and  y = W^T x
// (one block per row / per 64-column strip; fixed order => deterministic)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void gemv_rows_body(const float* __restrict__ Wm, const float* __restrict__ x, int K,
                                               float* __restrict__ y, int r) {
  __shared__ float red[256];
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) s += Wm[(size_t)r * K + k] * x[k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) y[r] = red[0];
}

__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ Wm, const float* __restrict__ x, int R,
                                                        int K, float* __restrict__ y) {
  gemv_rows_body(Wm, x, K, y, blockIdx.x);
}

__device__ __forceinline__ void gemv_cols_body(const float* __restrict__ Wm, const float* __restrict__ x, int R, int K,
                                               float* __restrict__ y, int blk) {
  // block = 64 columns; 16 row-groups x 16 lanes of 4 columns (16-byte loads when K % 4 == 0), four rows in flight
  // per thread (a 4-row-group version with one scalar load per iteration was pure latency: 51 us per call, 2.8 ms per
  // training step); partial sums are combined in row-group order => deterministic
  __shared__ float red[16][64];
  const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c0 = blk * 64 + cq * 4;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if ((K & 3) == 0 && c0 + 3 < K) {
    int r = rg;
    for (; r + 48 < R; r += 64) {
      f32x4 v[4];
      float xr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = *reinterpret_cast<const f32x4*>(Wm + (size_t)(r + 16 * u) * K + c0);
        xr[u] = x[r + 16 * u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += v[u][e] * xr[u];
    }
    for (; r < R; r += 16) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(Wm + (size_t)r * K + c0);
      const float xr = x[r];
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += v[e] * xr;
    }
  } else {
    for (int r = rg; r < R; r += 16) {
      const float xr = x[r];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + e < K) s[e] += Wm[(size_t)r * K + c0 + e] * xr;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rg][cq * 4 + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = blk * 64 + threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][threadIdx.x];
    if (c < K) y[c] = t;
  }
}

__global__ __launch_bounds__(256) void gemv_cols_kernel(const float* __restrict__ Wm, const float* __restrict__ x, int R,
                                                        int K, float* __restrict__ y) {
  gemv_cols_body(Wm, x, R, K, y, blockIdx.x);
}

// ---------------------------------------------------------------------------
// Batched spectral norm: the power iteration of EVERY spectral-normalised convolution of a network in four launches
// (the per-layer sequence above is eight launches of a few microseconds each; the generator has 27 such layers and runs
// twice per training step).  The job table travels in the kernel arguments; a block finds its job by a scan of the
// block-offset prefix (<= SN_MAX entries, wave-uniform).
// ---------------------------------------------------------------------------
constexpr int SN_MAX = 40;
struct SnJob {
  const float* w;
  float *u, *v, *wv, *sigma, *u_keep, *v_keep;
  int R, K;
};
struct SnBatch {
  SnJob job[SN_MAX];
  int col_blk[SN_MAX + 1], row_blk[SN_MAX + 1];
  int n;
  float eps;
};

__device__ __forceinline__ int sn_find(const int* __restrict__ prefix, int n, int b) {
  int j = 0;
  while (j + 1 < n && prefix[j + 1] <= b) ++j;
  return j;
}

__global__ __launch_bounds__(256) void sn_gemv_cols_batched_kernel(const SnBatch B) {   // v' = W^T u
  const int j = sn_find(B.col_blk, B.n, blockIdx.x);
  const SnJob& J = B.job[j];
  gemv_cols_body(J.w, J.u, J.R, J.K, J.v, blockIdx.x - B.col_blk[j]);
}

__global__ __launch_bounds__(256) void sn_gemv_rows_batched_kernel(const SnBatch B) {   // wv = W v
  const int j = sn_find(B.row_blk, B.n, blockIdx.x);
  const SnJob& J = B.job[j];
  gemv_rows_body(J.w, J.v, J.K, J.wv, blockIdx.x - B.row_blk[j]);
}

__device__ __forceinline__ float block_sum256(float s) {
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float t = red[0];
  __syncthreads();
  return t;
}

// one block per job: v <- v / max(||v||, eps), copy kept for the backward
__global__ __launch_bounds__(256) void sn_normalize_v_batched_kernel(const SnBatch B) {
  const SnJob& J = B.job[blockIdx.x];
  float s = 0.f;
  for (int i = threadIdx.x; i < J.K; i += 256) s += J.v[i] * J.v[i];
  const float inv = 1.f / fmaxf(sqrtf(block_sum256(s)), B.eps);
  for (int i = threadIdx.x; i < J.K; i += 256) {
    const float t = J.v[i] * inv;
    J.v[i] = t;
    if (J.v_keep) J.v_keep[i] = t;
  }
}

// one block per job: (iterate) u <- wv / max(||wv||, eps); sigma = u . wv; copies kept for the backward
__global__ __launch_bounds__(256) void sn_finish_batched_kernel(const SnBatch B, int iterate) {
  const SnJob& J = B.job[blockIdx.x];
  if (iterate) {
    float s = 0.f;
    for (int i = threadIdx.x; i < J.R; i += 256) s += J.wv[i] * J.wv[i];
    const float inv = 1.f / fmaxf(sqrtf(block_sum256(s)), B.eps);
    for (int i = threadIdx.x; i < J.R; i += 256) J.u[i] = J.wv[i] * inv;
  }
  float d = 0.f;
  for (int i = threadIdx.x; i < J.R; i += 256) {
    const float ui = J.u[i];            // each thread re-reads the elements it wrote
    d += ui * J.wv[i];
    if (J.u_keep) J.u_keep[i] = ui;
  }
  if (!iterate && J.v_keep)
    for (int i = threadIdx.x; i < J.K; i += 256) J.v_keep[i] = J.v[i];
  d = block_sum256(d);
  if (threadIdx.x == 0) J.sigma[0] = d;
}

// x <- x / max(||x||, eps); also returns ||x|| in out_norm (single block)
__global__ __launch_bounds__(256) void normalize_kernel(float* __restrict__ x, int n, float eps, float* __restrict__ out_norm) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i] * x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float nrm = sqrtf(red[0]);
  const float inv = 1.f / fmaxf(nrm, eps);
  for (int i = threadIdx.x; i < n; i += 256) x[i] *= inv;
  if (threadIdx.x == 0 && out_norm) out_norm[0] = nrm;
}

// dot(a, b) over n elements -> out[0] (two-stage)
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          size_t n, float* __restrict__ part) {
  __shared__ float red[256];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i] * b[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// spectral-norm weight-gradient transform (torch SpectralNorm backward with u, v constants):
//   dW_orig[r][k] = (G[r][k] - c * u[r] * v[k]) / sigma,   c = <G, W_orig> / sigma = <G, W_sn>
__global__ void sn_grad_kernel(const float* __restrict__ G, const float* __restrict__ u, const float* __restrict__ v,
                               const float* __restrict__ dotGW, const float* __restrict__ sigma, int R, int K,
                               float* __restrict__ out, int accumulate) {
  const size_t total = (size_t)R * K;
  const float sg = sigma[0];
  const float c = dotGW[0] / sg;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / K), k = (int)(i - (size_t)r * K);
    const float d = (G[i] - c * u[r] * v[k]) / sg;
    out[i] = accumulate ? out[i] + d : d;
  }
}

// ---------------------------------------------------------------------------
// Per-step parameter staging of a SPADE norm / a SPADE block in ONE launch each (they were 5-9 torch fill / index_copy /
// cat launches per norm and forward: ~750 launches of a few microseconds per training step).
// ---------------------------------------------------------------------------
// bias of the fused gamma|beta convolution in its interleaved column order (64g + l: gamma[32g + l] | beta[32g + l - 32])
// and the noise scale padded to the channel stride
__global__ __launch_bounds__(256) void spade_vec_prep_kernel(const float* __restrict__ gb, const float* __restrict__ bb,
                                                             const float* __restrict__ ns, int C, int ncols, int Cp,
                                                             float* __restrict__ bc, float* __restrict__ ns_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < ncols) {
    const int l = i & 63, c = (i >> 6) * 32 + (l & 31);
    bc[i] = c < C ? (l < 32 ? gb[c] : bb[c]) : 0.f;
  }
  if (i < Cp) ns_out[i] = i < C ? ns[i] : 0.f;
}

// ... for every SPADE norm of a network in ONE launch (blockIdx.y = the norm): 58 launches of 4 us per training iteration otherwise
constexpr int VEC_MAXJOBS = 32;
struct VecJobs {
  const float* gb[VEC_MAXJOBS]; const float* bb[VEC_MAXJOBS]; const float* ns[VEC_MAXJOBS];
  int C[VEC_MAXJOBS], off_bc[VEC_MAXJOBS], off_ns[VEC_MAXJOBS];
};
__global__ __launch_bounds__(256) void spade_vec_prep_multi_kernel(const VecJobs j, float* __restrict__ bc_all, float* __restrict__ ns_all) {
  const int q = blockIdx.y, C = j.C[q];
  const int ncols = (C + 31) / 32 * 64, Cp = (C + 3) / 4 * 4;
  float* const bc = bc_all + j.off_bc[q];
  float* const nso = ns_all + j.off_ns[q];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < ncols; i += gridDim.x * 256) {
    const int l = i & 63, c = (i >> 6) * 32 + (l & 31);
    bc[i] = c < C ? (l < 32 ? j.gb[q][c] : j.bb[q][c]) : 0.f;
    if (i < Cp) nso[i] = i < C ? j.ns[q][i] : 0.f;
  }
}

// conv_shared (label_nc -> hid, 3x3) of the n norms of a block as ONE 1x1 weight over the tap-expanded label map
// (ops.tap_expand: tap-major, cp channels per tap): wt[(i*hid + o)][tap*cp + ch] = w_i[o][ch][tap]; bt = concat of biases
struct SharedTaps {
  const float* w[4];
  const float* b[4];
  float* gw[4];
  float* gb[4];
  int n, hid, c, cp;
};
__global__ __launch_bounds__(256) void shared_taps_prep_kernel(const SharedTaps p, float* __restrict__ wt, float* __restrict__ bt) {
  const int K = 9 * p.cp, total = p.n * p.hid * K;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int k = i % K, row = i / K;
    const int tap = k / p.cp, ch = k - tap * p.cp;
    const int j = row / p.hid, o = row - j * p.hid;
    wt[i] = ch < p.c ? p.w[j][((size_t)o * p.c + ch) * 9 + tap] : 0.f;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < p.n * p.hid; i += gridDim.x * 256) bt[i] = p.b[i / p.hid][i % p.hid];
}
// the inverse map for the gradients: gw_i[o][ch][tap] = dw[(i*hid + o)][tap*cp + ch], gb_i = db[i*hid ..]
__global__ __launch_bounds__(256) void shared_taps_grad_kernel(const SharedTaps p, const float* __restrict__ dw,
                                                               const float* __restrict__ db) {
  const int per = p.hid * p.c * 9, total = p.n * per, K = 9 * p.cp;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int j = i / per, r = i - j * per;
    const int tap = r % 9, ch = (r / 9) % p.c, o = r / (9 * p.c);
    p.gw[j][r] = dw[(size_t)(j * p.hid + o) * K + tap * p.cp + ch];
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < p.n * p.hid; i += gridDim.x * 256) p.gb[i / p.hid][i % p.hid] = db[i];
}

}  // namespace hrv

using namespace hrv;

extern "C" int hrv_spade_vec_prep_f32(const float* gamma_bias, const float* beta_bias, const float* noise_scale, int32_t C,
                                      float* bias_interleaved, float* noise_scale_padded, hrv_stream_t stream) {
  HRV_REQUIRE(gamma_bias && beta_bias && noise_scale && bias_interleaved && noise_scale_padded && C > 0, "spade_vec_prep: bad args");
  const int ncols = (C + 31) / 32 * 64, Cp = (C + 3) / 4 * 4;
  hipLaunchKernelGGL(spade_vec_prep_kernel, dim3((ncols + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma_bias, beta_bias,
                     noise_scale, C, ncols, Cp, bias_interleaved, noise_scale_padded);
  return check_launch("spade_vec_prep_kernel");
}

extern "C" int hrv_spade_vec_prep_multi_f32(int32_t n, const float* const* gamma_bias, const float* const* beta_bias,
                                            const float* const* noise_scale, const int32_t* C, const int32_t* off_bias, const int32_t* off_noise,
                                            float* bias_interleaved_all, float* noise_scale_padded_all, hrv_stream_t stream) {
  HRV_REQUIRE(n > 0 && n <= VEC_MAXJOBS && gamma_bias && beta_bias && noise_scale && C && off_bias && off_noise && bias_interleaved_all &&
                  noise_scale_padded_all, "spade_vec_prep_multi: 1 .. %d norms", VEC_MAXJOBS);
  VecJobs j;
  memset(&j, 0, sizeof(j));
  int cmax = 0;
  for (int i = 0; i < n; ++i) {
    HRV_REQUIRE(gamma_bias[i] && beta_bias[i] && noise_scale[i] && C[i] > 0 && off_bias[i] >= 0 && off_noise[i] >= 0 && off_bias[i] % 4 == 0 &&
                    off_noise[i] % 4 == 0, "spade_vec_prep_multi: job %d", i);
    j.gb[i] = gamma_bias[i]; j.bb[i] = beta_bias[i]; j.ns[i] = noise_scale[i];
    j.C[i] = C[i]; j.off_bc[i] = off_bias[i]; j.off_ns[i] = off_noise[i];
    if (C[i] > cmax) cmax = C[i];
  }
  const int ncols = (cmax + 31) / 32 * 64;
  hipLaunchKernelGGL(spade_vec_prep_multi_kernel, dim3((ncols + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, j, bias_interleaved_all,
                     noise_scale_padded_all);
  return check_launch("spade_vec_prep_multi_kernel");
}

static int shared_taps_fill(SharedTaps& p, const float* const* w, const float* const* b, float* const* gw, float* const* gb,
                            int32_t n, int32_t hid, int32_t c, int32_t cp) {
  HRV_REQUIRE(n >= 1 && n <= 4 && hid > 0 && c > 0 && cp >= c, "shared_taps: n in 1..4, cp >= c");
  memset(&p, 0, sizeof(p));
  for (int i = 0; i < n; ++i) {
    if (w) { HRV_REQUIRE(w[i] && b[i], "shared_taps: null weight"); p.w[i] = w[i]; p.b[i] = b[i]; }
    if (gw) { HRV_REQUIRE(gw[i] && gb[i], "shared_taps: null gradient"); p.gw[i] = gw[i]; p.gb[i] = gb[i]; }
  }
  p.n = n; p.hid = hid; p.c = c; p.cp = cp;
  return HRV_OK;
}

extern "C" int hrv_shared_taps_prep_f32(const float* const* w, const float* const* b, int32_t n, int32_t hid, int32_t c,
                                        int32_t cp, float* wt, float* bt, hrv_stream_t stream) {
  HRV_REQUIRE(w && b && wt && bt, "shared_taps_prep: null pointer");
  SharedTaps p;
  const int rc = shared_taps_fill(p, w, b, nullptr, nullptr, n, hid, c, cp);
  if (rc) return rc;
  hipLaunchKernelGGL(shared_taps_prep_kernel, dim3(grid_for((size_t)n * hid * 9 * cp)), dim3(256), 0, (hipStream_t)stream, p, wt, bt);
  return check_launch("shared_taps_prep_kernel");
}

extern "C" int hrv_shared_taps_grad_f32(const float* dw, const float* db, int32_t n, int32_t hid, int32_t c, int32_t cp,
                                        float* const* gw, float* const* gb, hrv_stream_t stream) {
  HRV_REQUIRE(dw && db && gw && gb, "shared_taps_grad: null pointer");
  SharedTaps p;
  const int rc = shared_taps_fill(p, nullptr, nullptr, gw, gb, n, hid, c, cp);
  if (rc) return rc;
  hipLaunchKernelGGL(shared_taps_grad_kernel, dim3(grid_for((size_t)n * hid * c * 9)), dim3(256), 0, (hipStream_t)stream, p, dw, db);
  return check_launch("shared_taps_grad_kernel");
}

extern "C" int64_t hrv_norm_bwd_workspace_elems(int32_t N, int32_t H, int32_t W, int32_t C) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return -1;
  const int nb = norm_slabs(H * W);
  return (int64_t)N * nb * ((C + 3) / 4 * 4) * 2 + (int64_t)N * ((C + 3) / 4 * 4) * 2;
}

static int norm_bwd_check(const hrv_norm_bwd_t* d) {
  HRV_REQUIRE(d && d->x && d->mean && d->rstd && d->dout && d->dnh && d->dx && d->workspace, "norm_bwd: null pointer");
  HRV_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->C % 4 == 0, "norm_bwd: extent (C %% 4 == 0)");
  HRV_REQUIRE((d->noise_z == nullptr) == (d->noise_scale == nullptr), "norm_bwd: noise_z/noise_scale go together");
  HRV_REQUIRE(d->act == HRV_ACT_NONE || d->out, "norm_bwd: activation output needed for its derivative");
  HRV_REQUIRE(((d->x_cstride | d->x_coff | d->out_cstride | d->out_coff | d->g1p_cstride | d->g1p_coff | d->dout_cstride |
                d->dout_coff | d->dnh_cstride | d->dnh_coff | d->dgb_cstride | d->dgb_coff | d->dx_cstride | d->dx_coff) & 3) == 0,
              "norm_bwd: strides/offsets must be multiples of 4");
  const int C = d->C;
  if (d->x_up_channels > 0) {
    HRV_REQUIRE(d->x2 && d->x_up_channels % 4 == 0 && d->x_up_channels < C && d->H % 2 == 0 && d->W % 2 == 0 &&
                    d->x_coff + d->x_up_channels <= d->x_cstride && d->x2_cstride % 4 == 0 && d->x2_coff % 4 == 0 &&
                    d->x2_coff + (C - d->x_up_channels) <= d->x2_cstride && ((uintptr_t)d->x2 & 15) == 0,
                "norm_bwd: upsampled source (%d of %d channels, %d x %d)", d->x_up_channels, C, d->H, d->W);
  }
  HRV_REQUIRE(!(d->dx_bf16 && d->dx_accumulate), "norm_bwd: a bf16 dx cannot be accumulated into");
  return HRV_OK;
}

// workspace layout: [N][nb][C][2] slab partials (stage 1; reused as [N][nb][C] by stage 2) | m1 [N][C] | m2 [N][C]
static void norm_bwd_fill(const hrv_norm_bwd_t* d, NormBwdParams& p, NormBwd2Params& q, float*& m1, float*& m2) {
  const int C = d->C, nb = norm_slabs(d->H * d->W);
  float* part = d->workspace;
  m1 = part + (size_t)d->N * nb * C * 2;
  m2 = m1 + (size_t)d->N * C;
  XSrc xs;
  xs.x = d->x; xs.x_cs = d->x_cstride; xs.x_co = d->x_coff; xs.H = d->H; xs.W = d->W;
  xs.x2 = d->x2; xs.x2_cs = d->x2_cstride; xs.x2_co = d->x2_coff; xs.up_g = d->x_up_channels / 4;
  p.xs = xs; p.z = d->noise_z; p.ns = d->noise_scale;
  p.mean = d->mean; p.rstd = d->rstd; p.out = d->out; p.out_cs = d->out_cstride; p.out_co = d->out_coff;
  p.g1p = d->g1p; p.g_cs = d->g1p_cstride; p.g_co = d->g1p_coff; p.g1p_bf16 = d->g1p_bf16;
  p.dout = d->dout; p.do_cs = d->dout_cstride; p.do_co = d->dout_coff; p.dout_bf16 = d->dout_bf16;
  p.dnh = d->dnh; p.dn_cs = d->dnh_cstride; p.dn_co = d->dnh_coff; p.dnh_bf16 = d->dnh_bf16;
  p.dgb = d->dgb; p.dgb_cs = d->dgb_cstride; p.dgb_co = d->dgb_coff;
  p.N = d->N; p.H = d->H; p.W = d->W; p.C4 = C / 4; p.act = d->act; p.slope = d->act_slope; p.NB = nb; p.part = part;
  p.dgb_bf16 = d->dgb_bf16; p.out_bf16 = d->out_bf16;
  // dout handed over as the dbeta half of dgb (same buffer, same pixel stride, C channels up, same storage, no activation left
  // to differentiate): dbeta = dout is already where it belongs
  p.dbeta_in_place = (d->dgb != nullptr && (const void*)d->dout == (const void*)d->dgb && d->dout_cstride == d->dgb_cstride &&
                      d->dout_coff == d->dgb_coff + C && d->dout_bf16 == d->dgb_bf16 && d->act == HRV_ACT_NONE) ? 1 : 0;
  q.xs = xs; q.z = d->noise_z; q.ns = d->noise_scale;
  q.mean = d->mean; q.rstd = d->rstd; q.m1 = m1; q.m2 = m2;
  q.dnh = d->dnh; q.dn_cs = d->dnh_cstride; q.dn_co = d->dnh_coff; q.dnh_bf16 = d->dnh_bf16;
  q.dx = d->dx; q.dx_cs = d->dx_cstride; q.dx_co = d->dx_coff; q.accumulate = d->dx_accumulate;
  q.dx_bf16 = d->dx_bf16;
  q.N = d->N; q.H = d->H; q.W = d->W; q.C4 = C / 4; q.NB = nb; q.part = part;  // partials are free again in stage 2
}

extern "C" int hrv_spade_norm_bwd_nhwc_f32(const hrv_norm_bwd_t* d, hrv_stream_t stream) {
  int rc = norm_bwd_check(d);
  if (rc) return rc;
  const int HW = d->H * d->W, C = d->C;
  const int nb = norm_slabs(HW);
  hipStream_t st = (hipStream_t)stream;
  NormBwdParams p;
  NormBwd2Params q;
  float *m1, *m2;
  norm_bwd_fill(d, p, q, m1, m2);
  hipLaunchKernelGGL(norm_bwd_stage1_kernel, dim3(nb, d->N, norm_chunks(C / 4)), dim3(256), 0, st, p);
  rc = check_launch("norm_bwd_stage1_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((d->N * C * 16 + 255) / 256), dim3(256), 0, st, p.part, d->N, nb, C, HW, m1, m2);
  rc = check_launch("norm_bwd_finalize_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(norm_bwd_stage2_kernel, dim3(nb, d->N, norm_chunks(C / 4)), dim3(256), 0, st, q);
  rc = check_launch("norm_bwd_stage2_kernel");
  if (rc) return rc;
  if (d->noise_z && d->dnoise_scale) {
    hipLaunchKernelGGL(sum_rows_kernel<>, dim3((C + 15) / 16), dim3(256), 0, st, p.part, d->N * nb, C, d->dnoise_scale,
                       d->dns_accumulate);
    rc = check_launch("sum_rows_kernel");
  }
  return rc;
}

extern "C" int hrv_spade_norm_bwd2_nhwc_f32(const hrv_norm_bwd_t* a, const hrv_norm_bwd_t* b, hrv_stream_t stream) {
  int rc = norm_bwd_check(a);
  if (rc) return rc;
  rc = norm_bwd_check(b);
  if (rc) return rc;
  HRV_REQUIRE(a->x == b->x && a->x2 == b->x2 && a->x_cstride == b->x_cstride && a->x_coff == b->x_coff && a->x_up_channels == b->x_up_channels &&
                  a->N == b->N && a->H == b->H && a->W == b->W && a->C == b->C,
              "norm_bwd2: both norms must normalise the same x");
  HRV_REQUIRE(!a->dx_bf16 && !a->dx_accumulate && a->workspace != b->workspace && a->dnh != b->dnh, "norm_bwd2: dx fp32 (written, = dx_a + dx_b); separate scratch");
  const int HW = a->H * a->W, C = a->C;
  const int nb = norm_slabs(HW);
  hipStream_t st = (hipStream_t)stream;
  NormBwdParams pa, pb;
  NormBwd2Params qa, qb;
  float *m1a, *m2a, *m1b, *m2b;
  norm_bwd_fill(a, pa, qa, m1a, m2a);
  norm_bwd_fill(b, pb, qb, m1b, m2b);
  hipLaunchKernelGGL(norm_bwd2_stage1_kernel, dim3(nb, a->N, norm_chunks(C / 4)), dim3(256), 0, st, pa, pb);
  rc = check_launch("norm_bwd2_stage1_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((a->N * C * 16 + 255) / 256), dim3(256), 0, st, pa.part, a->N, nb, C, HW, m1a, m2a);
  hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((a->N * C * 16 + 255) / 256), dim3(256), 0, st, pb.part, a->N, nb, C, HW, m1b, m2b);
  rc = check_launch("norm_bwd_finalize_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(norm_bwd2_stage2_kernel, dim3(nb, a->N, norm_chunks(C / 4)), dim3(256), 0, st, qa, qb);
  rc = check_launch("norm_bwd2_stage2_kernel");
  if (rc) return rc;
  const hrv_norm_bwd_t* ds[2] = {a, b};
  const NormBwdParams* ps[2] = {&pa, &pb};
  for (int k = 0; k < 2; ++k)
    if (ds[k]->noise_z && ds[k]->dnoise_scale) {
      hipLaunchKernelGGL(sum_rows_kernel<>, dim3((C + 15) / 16), dim3(256), 0, st, ps[k]->part, a->N * nb, C, ds[k]->dnoise_scale,
                         ds[k]->dns_accumulate);
      rc = check_launch("sum_rows_kernel");
    }
  return rc;
}

extern "C" int hrv_loss_f32(const float* a, const float* b, int64_t n, int32_t mode, float lscale, float gscale,
                            float* grad, float* workspace, float* loss_out, int32_t accumulate, hrv_stream_t stream) {
  HRV_REQUIRE(a && workspace && loss_out && n > 0 && mode >= 0 && (mode & 15) <= 4 && (mode & ~31) == 0, "loss: bad args");
  HRV_REQUIRE(((mode & 15) != 0 && (mode & 15) != 4) || b, "loss: mode needs a target tensor");
  const int nb = grid_for((size_t)n) > 1024 ? 1024 : grid_for((size_t)n);
  hipStream_t st = (hipStream_t)stream;
  const int vec = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)grad) & 15) == 0;
  hipLaunchKernelGGL(loss_kernel<false>, dim3(nb), dim3(256), 0, st, a, b, (size_t)n, mode, gscale, grad, workspace, vec);
  int rc = check_launch("loss_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, workspace, nb, lscale, loss_out, accumulate);
  return check_launch("loss_final_kernel");
}

// the same over bf16-stored a and b (mixed-precision VGG taps); loss and gradient stay fp32
extern "C" int hrv_loss_bf16in_f32(const uint16_t* a, const uint16_t* b, int64_t n, int32_t mode, float lscale, float gscale,
                                   float* grad, float* workspace, float* loss_out, int32_t accumulate, hrv_stream_t stream) {
  HRV_REQUIRE(a && workspace && loss_out && n > 0 && mode >= 0 && (mode & 15) <= 4 && (mode & ~63) == 0, "loss_bf16in: bad args");
  HRV_REQUIRE(((mode & 15) != 0 && (mode & 15) != 4) || b, "loss_bf16in: mode needs a target tensor");
  const int nb = grid_for((size_t)n) > 1024 ? 1024 : grid_for((size_t)n);
  hipStream_t st = (hipStream_t)stream;
  const int vec = ((((uintptr_t)a | (uintptr_t)b) & 7) | ((uintptr_t)grad & 15)) == 0;
  hipLaunchKernelGGL(loss_kernel<true>, dim3(nb), dim3(256), 0, st, (const float*)a, (const float*)b, (size_t)n, mode, gscale, grad,
                     workspace, vec);
  int rc = check_launch("loss_kernel[bf16 in]");
  if (rc) return rc;
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, workspace, nb, lscale, loss_out, accumulate);
  return check_launch("loss_final_kernel");
}

extern "C" int hrv_downsum2x2_nhwc_f32(const float* dhi, int32_t N, int32_t Hl, int32_t Wl, int32_t C, int32_t hi_cstride,
                                       int32_t hi_coff, float* dlo, int32_t lo_cstride, int32_t lo_coff,
                                       int32_t accumulate, hrv_stream_t stream) {
  HRV_REQUIRE(dhi && dlo && N > 0 && Hl > 0 && Wl > 0 && C > 0 && C % 4 == 0 &&
                  ((hi_cstride | hi_coff | lo_cstride | lo_coff) & 3) == 0, "downsum2x2: bad args");
  const size_t total = (size_t)N * Hl * Wl * (C / 4);
  hipLaunchKernelGGL(downsum2x2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dhi, N, Hl, Wl, C / 4,
                     hi_cstride, hi_coff, dlo, lo_cstride, lo_coff, accumulate);
  return check_launch("downsum2x2_kernel");
}

extern "C" int hrv_avgpool3x3s2_bwd_nhwc_f32(const float* dy, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dy_cstride,
                                             int32_t dy_coff, float* dx, int32_t dx_cstride, int32_t dx_coff,
                                             int32_t accumulate, hrv_stream_t stream) {
  HRV_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 &&
                  ((dy_cstride | dy_coff | dx_cstride | dx_coff) & 3) == 0, "avgpool_bwd: bad args");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)N * H * W * (C / 4);
  hipLaunchKernelGGL(avgpool3s2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, N, H, W, C / 4,
                     Ho, Wo, dy_cstride, dy_coff, dx, dx_cstride, dx_coff, accumulate);
  return check_launch("avgpool3s2_bwd_kernel");
}

extern "C" int hrv_maxpool2x2_nhwc_f32(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, float* y,
                                       hrv_stream_t stream) {
  HRV_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0, "maxpool: bad args");
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, N, H / 2, W / 2, C / 4, C,
                     y, C);
  return check_launch("maxpool2_kernel");
}

// bf16-stored x and y (mixed-precision VGG activations)
extern "C" int hrv_maxpool2x2_nhwc_bf16(const uint16_t* x, int32_t N, int32_t H, int32_t W, int32_t C, uint16_t* y,
                                        hrv_stream_t stream) {
  HRV_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0, "maxpool_bf16: bad args");
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, N, H / 2,
                     W / 2, C / 4, C, (float*)y, C);
  return check_launch("maxpool2_kernel[bf16]");
}

extern "C" int hrv_maxpool2x2_bwd_nhwc_f32(const float* x, const float* dy, int32_t N, int32_t H, int32_t W, int32_t C,
                                           float* dx, hrv_stream_t stream) {
  HRV_REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0, "maxpool_bwd: bad args");
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2_bwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, dy, N, H / 2, W / 2,
                     C / 4, C, C, dx, 0);
  return check_launch("maxpool2_bwd_kernel");
}

// the same with the ReLU derivative of the pooled tensor fused (x = ReLU(pre): dx is the gradient w.r.t. pre)
extern "C" int hrv_maxpool2x2_bwd_relu_nhwc_f32(const float* x, const float* dy, int32_t N, int32_t H, int32_t W, int32_t C,
                                                float* dx, hrv_stream_t stream) {
  HRV_REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0, "maxpool_bwd_relu: bad args");
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2_bwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, dy, N, H / 2, W / 2,
                     C / 4, C, C, dx, 1);
  return check_launch("maxpool2_bwd_kernel[relu]");
}

// ... with a bf16-stored x (dy, dx fp32)
extern "C" int hrv_maxpool2x2_bwd_relu_nhwc_xbf16(const uint16_t* x, const float* dy, int32_t N, int32_t H, int32_t W, int32_t C,
                                                  float* dx, hrv_stream_t stream) {
  HRV_REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0, "maxpool_bwd_xbf16: bad args");
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2_bwd_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, dy, N,
                     H / 2, W / 2, C / 4, C, C, dx, 1);
  return check_launch("maxpool2_bwd_kernel[relu, bf16 x]");
}

// ... with bf16-stored x, dy AND dx
extern "C" int hrv_maxpool2x2_bwd_relu_nhwc_bf16(const uint16_t* x, const uint16_t* dy, int32_t N, int32_t H, int32_t W, int32_t C,
                                                 uint16_t* dx, hrv_stream_t stream) {
  HRV_REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0, "maxpool_bwd_bf16: bad args");
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL((maxpool2_bwd_kernel<true, true>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                     (const float*)dy, N, H / 2, W / 2, C / 4, C, C, (float*)dx, 1);
  return check_launch("maxpool2_bwd_kernel[relu, bf16]");
}

extern "C" int hrv_add_slice_nhwc_bf16(const uint16_t* a, int32_t a_cstride, int32_t a_coff, uint16_t* out, int32_t out_cstride,
                                       int32_t out_coff, int32_t C, int64_t npix, int32_t accumulate, hrv_stream_t stream) {
  HRV_REQUIRE(a && out && npix > 0 && C > 0 && C % 4 == 0 && ((a_cstride | a_coff | out_cstride | out_coff) & 3) == 0,
              "add_slice_bf16: bad args");
  hipLaunchKernelGGL(add_slice_bf16_kernel, dim3(grid_for((size_t)npix * (C / 4))), dim3(256), 0, (hipStream_t)stream, a, a_cstride,
                     a_coff, out, out_cstride, out_coff, C / 4, (size_t)npix, accumulate);
  return check_launch("add_slice_bf16_kernel");
}

extern "C" int hrv_adam_f32(float* w, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int32_t step, float grad_scale, hrv_stream_t stream) {
  HRV_REQUIRE(w && g && m && v && n > 0 && step >= 1, "adam: bad args");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, (size_t)n, lr,
                     beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
  return check_launch("adam_kernel");
}

extern "C" int hrv_adam_hyper_f32(int32_t* step_dev, const float* lr_dev, float beta1, float beta2, float* hyper_dev,
                                  hrv_stream_t stream) {
  HRV_REQUIRE(step_dev && lr_dev && hyper_dev, "adam_hyper: null pointer");
  hipLaunchKernelGGL(adam_hyper_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev, lr_dev, beta1, beta2, hyper_dev);
  return check_launch("adam_hyper_kernel");
}

extern "C" int hrv_adam_dev_f32(float* w, const float* g, float* m, float* v, int64_t n, const float* hyper_dev, float beta1,
                                float beta2, float eps, float weight_decay, float grad_scale, hrv_stream_t stream) {
  HRV_REQUIRE(w && g && m && v && hyper_dev && n > 0, "adam_dev: bad args");
  hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, (size_t)n, hyper_dev,
                     beta1, beta2, eps, weight_decay, grad_scale);
  return check_launch("adam_dev_kernel");
}

extern "C" int hrv_spectral_norm_f32(const float* w, int32_t R, int32_t K, float* u, float* v, int32_t power_iterations,
                                     float eps, float* wv_scratch, float* sigma_out, hrv_stream_t stream) {
  HRV_REQUIRE(w && u && v && wv_scratch && sigma_out && R > 0 && K > 0 && power_iterations >= 0, "spectral_norm: bad args");
  hipStream_t st = (hipStream_t)stream;
  for (int it = 0; it < power_iterations; ++it) {
    // v <- normalize(W^T u) ; u <- normalize(W v)    (torch SpectralNorm.compute_weight)
    hipLaunchKernelGGL(gemv_cols_kernel, dim3((K + 63) / 64), dim3(256), 0, st, w, u, R, K, v);
    hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(256), 0, st, v, K, eps, (float*)nullptr);
    hipLaunchKernelGGL(gemv_rows_kernel, dim3(R), dim3(256), 0, st, w, v, R, K, u);
    hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(256), 0, st, u, R, eps, (float*)nullptr);
  }
  // sigma = u . (W v)
  hipLaunchKernelGGL(gemv_rows_kernel, dim3(R), dim3(256), 0, st, w, v, R, K, wv_scratch);
  hipLaunchKernelGGL(dot_partial_kernel, dim3(1), dim3(256), 0, st, u, wv_scratch, (size_t)R, sigma_out);
  return check_launch("spectral_norm kernels");
}

extern "C" int hrv_spectral_norm_batched_f32(const hrv_sn_job_t* jobs, int32_t n_jobs, int32_t power_iterations, float eps,
                                             float* wv_scratch, hrv_stream_t stream) {
  HRV_REQUIRE(jobs && n_jobs > 0 && wv_scratch && power_iterations >= 0, "spectral_norm_batched: bad args");
  hipStream_t st = (hipStream_t)stream;
  float* scratch = wv_scratch;
  for (int j0 = 0; j0 < n_jobs; j0 += SN_MAX) {
    SnBatch B;
    B.n = n_jobs - j0 < SN_MAX ? n_jobs - j0 : SN_MAX;
    B.eps = eps;
    B.col_blk[0] = B.row_blk[0] = 0;
    for (int j = 0; j < B.n; ++j) {
      const hrv_sn_job_t& s = jobs[j0 + j];
      HRV_REQUIRE(s.w && s.u && s.v && s.sigma && s.R > 0 && s.K > 0, "spectral_norm_batched: job %d", j0 + j);
      SnJob& J = B.job[j];
      J.w = s.w; J.u = s.u; J.v = s.v; J.sigma = s.sigma; J.u_keep = s.u_keep; J.v_keep = s.v_keep; J.R = s.R; J.K = s.K;
      J.wv = scratch;
      scratch += s.R;
      B.col_blk[j + 1] = B.col_blk[j] + (s.K + 63) / 64;
      B.row_blk[j + 1] = B.row_blk[j] + s.R;
    }
    for (int it = 0; it < power_iterations; ++it) {
      // v <- normalize(W^T u) ; u <- normalize(W v)    (torch SpectralNorm.compute_weight); the last pass also yields sigma
      hipLaunchKernelGGL(sn_gemv_cols_batched_kernel, dim3(B.col_blk[B.n]), dim3(256), 0, st, B);
      hipLaunchKernelGGL(sn_normalize_v_batched_kernel, dim3(B.n), dim3(256), 0, st, B);
      hipLaunchKernelGGL(sn_gemv_rows_batched_kernel, dim3(B.row_blk[B.n]), dim3(256), 0, st, B);
      // sigma = u . (W v) with the normalised u and the SAME W v (torch recomputes it: identical operands)
      hipLaunchKernelGGL(sn_finish_batched_kernel, dim3(B.n), dim3(256), 0, st, B, 1);
    }
    if (power_iterations == 0) {
      hipLaunchKernelGGL(sn_gemv_rows_batched_kernel, dim3(B.row_blk[B.n]), dim3(256), 0, st, B);
      hipLaunchKernelGGL(sn_finish_batched_kernel, dim3(B.n), dim3(256), 0, st, B, 0);
    }
  }
  return check_launch("spectral_norm_batched kernels");
}

extern "C" int hrv_spectral_norm_bwd_f32(const float* G, const float* w_orig, const float* u, const float* v,
                                         const float* sigma, int32_t R, int32_t K, float* workspace, float* dw_orig,
                                         int32_t accumulate, hrv_stream_t stream) {
  HRV_REQUIRE(G && w_orig && u && v && sigma && workspace && dw_orig && R > 0 && K > 0, "spectral_norm_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)R * K;
  const int nb = grid_for(n) > 512 ? 512 : grid_for(n);
  hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(256), 0, st, G, w_orig, n, workspace);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, workspace, nb, 1.0f, workspace + 512, 0);
  hipLaunchKernelGGL(sn_grad_kernel, dim3(grid_for(n)), dim3(256), 0, st, G, u, v, workspace + 512, sigma, R, K, dw_orig,
                     accumulate);
  return check_launch("spectral_norm_bwd kernels");
}

extern "C" int hrv_tanh_bwd_f32(const float* dy, const float* y, int64_t n, float* out, hrv_stream_t stream) {
  HRV_REQUIRE(dy && y && out && n > 0, "tanh_bwd: bad args");
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, dy, y, (size_t)n, out);
  return check_launch("tanh_bwd_kernel");
}

extern "C" int hrv_add_slice_nhwc_f32(const float* a, int32_t a_cstride, int32_t a_coff, float* out, int32_t out_cstride,
                                      int32_t out_coff, int32_t C, int64_t npix, int32_t accumulate, hrv_stream_t stream) {
  HRV_REQUIRE(a && out && npix > 0 && C > 0 && C % 4 == 0 && ((a_cstride | a_coff | out_cstride | out_coff) & 3) == 0,
              "add_slice: bad args");
  hipLaunchKernelGGL(add_slice_kernel, dim3(grid_for((size_t)npix * (C / 4))), dim3(256), 0, (hipStream_t)stream, a,
                     a_cstride, a_coff, out, out_cstride, out_coff, C / 4, (size_t)npix, accumulate);
  return check_launch("add_slice_kernel");
}

extern "C" int hrv_act_bwd_nhwc_f32(float* d, int32_t d_cstride, int32_t d_coff, const float* y, int32_t y_cstride,
                                    int32_t y_coff, int32_t C, int64_t npix, int32_t act, float act_slope,
                                    hrv_stream_t stream) {
  HRV_REQUIRE(d && y && npix > 0 && C > 0 && C % 4 == 0 && ((d_cstride | d_coff | y_cstride | y_coff) & 3) == 0,
              "act_bwd: bad args");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for((size_t)npix * (C / 4))), dim3(256), 0, (hipStream_t)stream, d, d_cstride,
                     d_coff, y, y_cstride, y_coff, C / 4, (size_t)npix, act, act_slope);
  return check_launch("act_bwd_kernel");
}

extern "C" int hrv_scale_f32(float* x, int64_t n, float s_host, const float* s_dev, hrv_stream_t stream) {
  HRV_REQUIRE(x && n > 0, "scale: bad args");
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, s_host, s_dev);
  return check_launch("scale_kernel");
}
