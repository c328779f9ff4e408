// Parse-map glue between the two networks (test_generator.py:161-217,
// train_generator.py:217-275): cloth-mask composition, 15x15 Gaussian, argmax ->
// one-hot(13) -> 7-class merge, occlusion handling.  NHWC fp32, HBM-bound.
#include "hrv_common.h"

namespace hrv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int grid_for(size_t work, int block = 256) {
  size_t g = (work + block - 1) / block;
  const size_t cap = 256 * 16;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// x[p][ch] *= m[p][mch]   (fake_segmap[:,3] *= warped_cm, test_generator.py:167-176)
__global__ void mul_channel_kernel(float* __restrict__ x, int xcs, int ch, const float* __restrict__ m, int mcs,
                                   int mch, int binarize, size_t npix) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    float v = m[i * mcs + mch];
    if (binarize) v = v > 0.5f ? 1.f : 0.f;
    x[i * xcs + ch] *= v;
  }
}

struct GaussParams {
  float g[32];
  int k, r;
};

// One separable pass of the depthwise Gaussian with zero padding (tgm GaussianBlur:
// conv2d(padding=(k-1)//2, groups=C)).  dir 0: along W, dir 1: along H.
__global__ void gauss_pass_kernel(const float* __restrict__ in, int N, int H, int W, int C4, int cs,
                                  float* __restrict__ out, const GaussParams gp, int dir) {
  const size_t total = (size_t)N * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % C4);
    const size_t pix = i / C4;
    const int w = (int)(pix % W);
    const size_t t = pix / W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    f32x4 acc = (f32x4)(0.f);
    for (int j = 0; j < gp.k; ++j) {
      const int hh = dir ? h + j - gp.r : h;
      const int ww = dir ? w : w + j - gp.r;
      if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
      acc += gp.g[j] * *reinterpret_cast<const f32x4*>(in + ((size_t)(n * H + hh) * W + ww) * cs + g * 4);
    }
    *reinterpret_cast<f32x4*>(out + pix * cs + g * 4) = acc;
  }
}

// The same pass with EIGHT outputs per thread along the filter axis: a sliding window of 8 + k - 1 loads instead of 8 k (the one-
// output form re-reads every input k = 15 times out of L2: 0.43 ms for the 4 x 1024 x 768 x 16 map of make_parse, 1.8 TB/s).  Every
// output is the same expression as above -- taps in ascending order, out-of-range taps skipped -- so the values are bit-identical.
constexpr int GAUSS_T = 8;
__global__ void gauss_pass8_kernel(const float* __restrict__ in, int N, int H, int W, int C4, int cs, float* __restrict__ out,
                                   const GaussParams gp, int dir) {
  const int L = dir ? H : W, other = dir ? W : H;
  const int segs = (L + GAUSS_T - 1) / GAUSS_T;
  // thread order: channel group fastest, then the axis that is contiguous in memory (dir 1: the pixel along W; dir 0: the segment)
  const size_t total = (size_t)N * other * segs * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % C4);
    size_t t = i / C4;
    int seg, line;
    if (dir) { line = (int)(t % W); t /= W; seg = (int)(t % segs); t /= segs; }
    else { seg = (int)(t % segs); t /= segs; line = (int)(t % H); t /= H; }
    const int n = (int)t;
    const int p0 = seg * GAUSS_T;
    const size_t step = dir ? (size_t)W * cs : (size_t)cs;                       // elements between neighbours along the filter axis
    const float* base = in + ((size_t)n * H * W + (dir ? (size_t)line : (size_t)line * W)) * cs + g * 4;
    f32x4 acc[GAUSS_T];
#pragma unroll
    for (int o = 0; o < GAUSS_T; ++o) acc[o] = (f32x4)(0.f);
    // input position q = p0 - r + m, m = 0 .. T + k - 2, feeds output o = m - j with tap j; per output the taps arrive in ascending j
    // only if m ascends -- which it does
    for (int m = 0; m < GAUSS_T + gp.k - 1; ++m) {
      const int q = p0 - gp.r + m;
      if (q < 0 || q >= L) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)q * step);
#pragma unroll
      for (int o = 0; o < GAUSS_T; ++o) {
        const int j = m - o;
        if (j >= 0 && j < gp.k) acc[o] += gp.g[j] * v;
      }
    }
    float* ob = out + ((size_t)n * H * W + (dir ? (size_t)line : (size_t)line * W)) * cs + g * 4;
#pragma unroll
    for (int o = 0; o < GAUSS_T; ++o)
      if (p0 + o < L) *reinterpret_cast<f32x4*>(ob + (size_t)(p0 + o) * step) = acc[o];
  }
}

__constant__ int kMerge13to7[13] = {0, 3, 1, 2, 1, 4, 5, 1, 1, 1, 1, 1, 6};  // test_generator.py:188-196

// argmax over the first `nclass` channels (first maximum wins, like torch.argmax), one-hot
// scatter and the 13->7 label merge, in one pass.  labels: int64 [N,H,W]; parse7: NHWC [.,8].
__global__ void parse_argmax_kernel(const float* __restrict__ g, int cs, int nclass, size_t npix,
                                    long long* __restrict__ labels, float* __restrict__ parse7, int pcs) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    const float* p = g + i * cs;
    int best = 0;
    float bv = p[0];
    for (int c = 1; c < nclass; ++c) {
      const float v = p[c];
      if (v > bv) { bv = v; best = c; }
    }
    if (labels) labels[i] = best;
    const int m = kMerge13to7[best];
    f32x4 lo = (f32x4)(0.f), hi = (f32x4)(0.f);
    if (m < 4) lo[m] = 1.f; else hi[m - 4] = 1.f;
    *reinterpret_cast<f32x4*>(parse7 + i * pcs) = lo;
    *reinterpret_cast<f32x4*>(parse7 + i * pcs + 4) = hi;
  }
}

// --occlusion (test_generator.py:214-216): cm' = cm - (sum_{c in {1,2,5..12}} softmax(g)[c]) * cm;
// cloth' = cloth*cm' + (1 - cm').  cloth: NHWC [.,4] (3 real), cm: [npix] in/out.
__global__ void occlusion_kernel(const float* __restrict__ g, int gcs, int nclass, float* cloth, int ccs,
                                 float* cm, int mcs, int mch, size_t npix) {  // cloth / cm may alias
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    const float* p = g + i * gcs;
    float mx = p[0];
    for (int c = 1; c < nclass; ++c) mx = fmaxf(mx, p[c]);
    float den = 0.f, sel = 0.f;
    for (int c = 0; c < nclass; ++c) {
      const float e = expf(p[c] - mx);
      den += e;
      if (c == 1 || c == 2 || c >= 5) sel += e;
    }
    const float m0 = cm[i * mcs + mch];
    const float m1 = m0 - (sel / den) * m0;
    cm[i * mcs + mch] = m1;
    for (int c = 0; c < 3; ++c) cloth[i * ccs + c] = cloth[i * ccs + c] * m1 + (1.f - m1);
  }
}

// Boundary pre-processing resize on NCHW planes (test_generator.py:144-150: the inputs
// are brought to 256x192 with F.interpolate bilinear / nearest before the tocg).
__global__ void resize_planes_kernel(const float* __restrict__ in, int planes, int H, int W, int Ho, int Wo,
                                     float rh, float rw, int nearest, float* __restrict__ out) {
  const size_t total = (size_t)planes * Ho * Wo;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int wo = (int)(i % Wo);
    const size_t t = i / Wo;
    const int ho = (int)(t % Ho);
    const size_t pl = t / Ho;
    const float* src = in + pl * (size_t)H * W;
    if (nearest) {
      // torch nearest: src = min(floor(dst * scale), in - 1), scale = in/out as float
      int hs = (int)floorf((float)ho * rh), ws = (int)floorf((float)wo * rw);
      hs = hs < H - 1 ? hs : H - 1;
      ws = ws < W - 1 ? ws : W - 1;
      out[i] = src[(size_t)hs * W + ws];
    } else {
      float sy = rh * ((float)ho + 0.5f) - 0.5f, sx = rw * ((float)wo + 0.5f) - 0.5f;
      sy = sy < 0.f ? 0.f : sy;
      sx = sx < 0.f ? 0.f : sx;
      int y0 = (int)sy, x0 = (int)sx;
      y0 = y0 < H - 1 ? y0 : H - 1;
      x0 = x0 < W - 1 ? x0 : W - 1;
      const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
      float ly = sy - (float)y0, lx = sx - (float)x0;
      ly = ly < 0.f ? 0.f : (ly > 1.f ? 1.f : ly);
      lx = lx < 0.f ? 0.f : (lx > 1.f ? 1.f : lx);
      const float v00 = src[(size_t)y0 * W + x0], v01 = src[(size_t)y0 * W + x1];
      const float v10 = src[(size_t)y1 * W + x0], v11 = src[(size_t)y1 * W + x1];
      out[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
  }
}


// ---------------------------------------------------------------------------
// Tap expansion (im2col of a narrow tensor): out[n,h,w, t*G + g] = 16-byte group g of the source pixel under
// tap t = kh*KW + kw of a stride-1 'same' KHxKW window (zero outside), with the conv engine's nearest
// down-sampling of the source folded in.  Turns the 7(8)-channel conv_shared 3x3 of every SPADENorm
// (network_generator.py:97,113) into a 1x1 convolution over 72 dense channels: the label map is expanded
// once per resolution and shared by the block's two or three norms.  Element-size agnostic (16-byte groups).
// ---------------------------------------------------------------------------
__global__ void tap_expand_kernel(const uint4* __restrict__ src, int N, int H, int W, int G, int scs16, int sco16,
                                  int down, int KH, int KW, int pad, uint4* __restrict__ out) {
  const int taps = KH * KW;
  const size_t total = (size_t)N * H * W * taps * G;
  const int Hs = H << down, Ws = W << down;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    size_t t = i / G;
    const int tap = (int)(t % taps);
    const size_t pix = t / taps;
    const int w = (int)(pix % W);
    const size_t t2 = pix / W;
    const int h = (int)(t2 % H);
    const int n = (int)(t2 / H);
    const int hh = h + tap / KW - pad, ww = w + tap % KW - pad;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (hh >= 0 && hh < H && ww >= 0 && ww < W)
      v = src[((size_t)(n * Hs + (hh << down)) * Ws + (ww << down)) * scs16 + sco16 + g];
    out[i] = v;
  }
}
}  // namespace hrv

using namespace hrv;

extern "C" int hrv_mul_channel_nhwc_f32(float* x, int32_t x_cstride, int32_t ch, const float* m, int32_t m_cstride,
                                        int32_t m_ch, int32_t binarize, int64_t npix, hrv_stream_t stream) {
  HRV_REQUIRE(x && m && npix > 0 && ch >= 0 && ch < x_cstride && m_ch >= 0 && m_ch < m_cstride, "mul_channel: bad args");
  hipLaunchKernelGGL(mul_channel_kernel, dim3(grid_for((size_t)npix)), dim3(256), 0, (hipStream_t)stream, x, x_cstride,
                     ch, m, m_cstride, m_ch, binarize, (size_t)npix);
  return check_launch("mul_channel_kernel");
}

extern "C" int hrv_gauss_blur_nhwc_f32(const float* in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t cstride,
                                       const float* taps_host, int32_t ksize, float* tmp, float* out,
                                       hrv_stream_t stream) {
  HRV_REQUIRE(in && tmp && out && taps_host && N > 0 && H > 0 && W > 0, "gauss_blur: bad args");
  HRV_REQUIRE(ksize > 0 && ksize <= 31 && (ksize & 1), "gauss_blur: ksize must be odd and <= 31");
  HRV_REQUIRE(C > 0 && C % 4 == 0 && cstride == C, "gauss_blur: dense NHWC with C %% 4 == 0 required");
  GaussParams gp;
  for (int i = 0; i < 32; ++i) gp.g[i] = i < ksize ? taps_host[i] : 0.f;
  gp.k = ksize;
  gp.r = (ksize - 1) / 2;
  const size_t total = (size_t)N * H * W * (C / 4);
  hipStream_t st = (hipStream_t)stream;
  const char* e8 = hrv::env("HRV_GAUSS8");
  if (e8 && e8[0] == '0') {          // the one-output-per-thread form (A/B)
    hipLaunchKernelGGL(gauss_pass_kernel, dim3(grid_for(total)), dim3(256), 0, st, in, N, H, W, C / 4, cstride, tmp, gp, 0);
    int rc0 = check_launch("gauss_pass_kernel(W)");
    if (rc0) return rc0;
    hipLaunchKernelGGL(gauss_pass_kernel, dim3(grid_for(total)), dim3(256), 0, st, tmp, N, H, W, C / 4, cstride, out, gp, 1);
    return check_launch("gauss_pass_kernel(H)");
  }
  const size_t tw = (size_t)N * H * ((W + GAUSS_T - 1) / GAUSS_T) * (C / 4), th = (size_t)N * W * ((H + GAUSS_T - 1) / GAUSS_T) * (C / 4);
  hipLaunchKernelGGL(gauss_pass8_kernel, dim3(grid_for(tw)), dim3(256), 0, st, in, N, H, W, C / 4, cstride, tmp, gp, 0);
  int rc = check_launch("gauss_pass8_kernel(W)");
  if (rc) return rc;
  hipLaunchKernelGGL(gauss_pass8_kernel, dim3(grid_for(th)), dim3(256), 0, st, tmp, N, H, W, C / 4, cstride, out, gp, 1);
  return check_launch("gauss_pass8_kernel(H)");
}

extern "C" int hrv_parse_argmax_nhwc_f32(const float* g, int32_t cstride, int32_t nclass, int64_t npix,
                                         int64_t* labels, float* parse7, int32_t parse_cstride, hrv_stream_t stream) {
  HRV_REQUIRE(g && parse7 && npix > 0 && nclass == 13 && cstride >= 13, "parse_argmax: bad args (13-class map expected)");
  HRV_REQUIRE(parse_cstride >= 8 && parse_cstride % 4 == 0 && ((uintptr_t)parse7 & 15) == 0, "parse_argmax: parse7 layout");
  hipLaunchKernelGGL(parse_argmax_kernel, dim3(grid_for((size_t)npix)), dim3(256), 0, (hipStream_t)stream, g, cstride,
                     nclass, (size_t)npix, (long long*)labels, parse7, parse_cstride);
  return check_launch("parse_argmax_kernel");
}

extern "C" int hrv_occlusion_nhwc_f32(const float* g, int32_t g_cstride, int32_t nclass, float* cloth,
                                      int32_t cloth_cstride, float* cm, int32_t cm_cstride, int32_t cm_ch,
                                      int64_t npix, hrv_stream_t stream) {
  HRV_REQUIRE(g && cloth && cm && npix > 0 && nclass > 0 && nclass <= g_cstride && cloth_cstride >= 3 &&
                  cm_ch >= 0 && cm_ch < cm_cstride, "occlusion: bad args");
  hipLaunchKernelGGL(occlusion_kernel, dim3(grid_for((size_t)npix)), dim3(256), 0, (hipStream_t)stream, g, g_cstride,
                     nclass, cloth, cloth_cstride, cm, cm_cstride, cm_ch, (size_t)npix);
  return check_launch("occlusion_kernel");
}

extern "C" int hrv_resize_nchw_f32(const float* in, int32_t planes, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                   int32_t nearest, float* out, hrv_stream_t stream) {
  HRV_REQUIRE(in && out && planes > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "resize_nchw: bad args");
  const size_t total = (size_t)planes * Ho * Wo;
  hipLaunchKernelGGL(resize_planes_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, planes, H, W,
                     Ho, Wo, (float)H / (float)Ho, (float)W / (float)Wo, nearest, out);
  return check_launch("resize_planes_kernel");
}

extern "C" int hrv_tap_expand_nhwc(const void* src, int32_t N, int32_t H, int32_t W, int32_t group16_per_pixel,
                                   int32_t src_stride16, int32_t src_off16, int32_t down_shift, int32_t KH, int32_t KW,
                                   int32_t pad, void* out, hrv_stream_t stream) {
  HRV_REQUIRE(src && out && N > 0 && H > 0 && W > 0 && group16_per_pixel > 0 && src_stride16 >= src_off16 + group16_per_pixel,
              "tap_expand: bad args");
  HRV_REQUIRE(KH > 0 && KW > 0 && KH == 2 * pad + 1 && KW == 2 * pad + 1 && down_shift >= 0 && down_shift <= 7,
              "tap_expand: 'same' stride-1 window, down_shift in [0,7]");
  HRV_REQUIRE((((uintptr_t)src | (uintptr_t)out) & 15) == 0, "tap_expand: 16-byte alignment");
  const size_t total = (size_t)N * H * W * KH * KW * group16_per_pixel;
  hipLaunchKernelGGL(tap_expand_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, N, H, W,
                     group16_per_pixel, src_stride16, src_off16, down_shift, KH, KW, pad, (uint4*)out);
  return check_launch("tap_expand_kernel");
}
