// HBM-bound sampling kernels of the HR-VITON path (NHWC, fp32, float4 per lane):
// layout converters, bilinear resize (+fused addend) and the appearance-flow
// warp (flow upsample + normalise + base grid + bilinear/border grid_sample
// in one pass).  Reference ops replaced: see include/hrviton_hip.h.
#include <stdarg.h>
#include <string.h>

#include "hrv_common.h"
#include <mutex>
#include <string>
#include <unordered_map>

namespace hrv {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static inline int grid_for(size_t work, int block = 256) {
  size_t g = (work + block - 1) / block;
  const size_t cap = 256 * 16;  // 256 CUs x 16 blocks, grid-stride the rest
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---------------------------------------------------------------- layout
// Layout converters at the module boundary: LDS-tiled transposes.  A block moves 64 pixels x up to 64 channels:
// the NCHW side is accessed plane by plane with 64 consecutive pixels per wave (coalesced), the NHWC side as the
// contiguous run of the tile's pixels (coalesced); the [64][65] LDS tile absorbs the stride change.
constexpr int LT_P = 64, LT_C = 64;

// ``zpad`` > 0: the destination is a whole channel-padded tensor (co == 0, cs == C + zpad): the pad channels are
// written as zeros here (the conv engine requires them to read as zero) instead of by a fill of the entire tensor.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, int N, int C, int HW,
                                                           float* __restrict__ out, int cs, int co, int zpad) {
  __shared__ float tile[LT_P][LT_C + 1];
  const int tiles_per_img = (HW + LT_P - 1) / LT_P;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = blockIdx.x; t < N * tiles_per_img; t += gridDim.x) {
    const int n = t / tiles_per_img, p0 = (t - n * tiles_per_img) * LT_P;
    const int np = min(LT_P, HW - p0);
    for (int c0 = 0; c0 < C; c0 += LT_C) {
      const int nc = min(LT_C, C - c0);
      for (int c = wave; c < nc; c += 4)
        if (lane < np) tile[lane][c] = in[((size_t)n * C + c0 + c) * HW + p0 + lane];
      __syncthreads();
      for (int i = threadIdx.x; i < np * nc; i += 256) {
        const int p = i / nc, c = i - p * nc;
        out[((size_t)n * HW + p0 + p) * cs + co + c0 + c] = tile[p][c];
      }
      if (zpad > 0 && c0 + nc == C)
        for (int i = threadIdx.x; i < np * zpad; i += 256) out[((size_t)n * HW + p0 + i / zpad) * cs + co + C + i % zpad] = 0.f;
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, int cs, int co, int N, int C,
                                                           int HW, float* __restrict__ out) {
  __shared__ float tile[LT_P][LT_C + 1];
  const int tiles_per_img = (HW + LT_P - 1) / LT_P;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = blockIdx.x; t < N * tiles_per_img; t += gridDim.x) {
    const int n = t / tiles_per_img, p0 = (t - n * tiles_per_img) * LT_P;
    const int np = min(LT_P, HW - p0);
    for (int c0 = 0; c0 < C; c0 += LT_C) {
      const int nc = min(LT_C, C - c0);
      for (int i = threadIdx.x; i < np * nc; i += 256) {
        const int p = i / nc, c = i - p * nc;
        tile[p][c] = in[((size_t)n * HW + p0 + p) * cs + co + c0 + c];
      }
      __syncthreads();
      for (int c = wave; c < nc; c += 4)
        if (lane < np) out[((size_t)n * C + c0 + c) * HW + p0 + lane] = tile[lane][c];
      __syncthreads();
    }
  }
}

__global__ void nchw_f32_to_nhwc_bf16_kernel(const float* __restrict__ in, int N, int C, int HW,
                                            unsigned short* __restrict__ out, int cs, int co, int zpad) {
  const size_t total = (size_t)N * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / HW, hw = i - n * HW;
    const float* src = in + n * (size_t)C * HW + hw;
    unsigned short* dst = out + i * cs + co;
    for (int c = 0; c < C; ++c) {
      unsigned u = __builtin_bit_cast(unsigned, src[(size_t)c * HW]);
      u += 0x7fffu + ((u >> 16) & 1u);
      dst[c] = (unsigned short)(u >> 16);
    }
    for (int c = 0; c < zpad; ++c) dst[C + c] = 0;
  }
}

__global__ void nhwc_bf16_to_nchw_f32_kernel(const unsigned short* __restrict__ in, int cs, int co, int N, int C, int HW,
                                             float* __restrict__ out) {
  const size_t total = (size_t)N * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / HW, hw = i - n * HW;
    const unsigned short* src = in + i * cs + co;
    float* dst = out + n * (size_t)C * HW + hw;
    for (int c = 0; c < C; ++c) dst[(size_t)c * HW] = __builtin_bit_cast(float, (unsigned)src[c] << 16);
  }
}

// ---------------------------------------------------------------- bilinear
// torch area_pixel_compute_source_index (align_corners=False, not cubic):
//   src = max(ratio*(dst+0.5) - 0.5, 0); i0 = (int)src; i1 = i0 + (i0 < in-1); l1 = src - i0
struct Lin {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lin lin_src(int dst, int in_size, float ratio) {
  float src = ratio * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  int i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  Lin r;
  r.i0 = i0;
  r.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  float l1 = src - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
  r.l1 = l1;
  r.l0 = 1.f - l1;
  return r;
}

__device__ __forceinline__ float4 f4_bilerp(float4 v00, float4 v01, float4 v10, float4 v11, float wl0, float wl1,
                                            float hl0, float hl1) {
  float4 o;
  o.x = hl0 * (wl0 * v00.x + wl1 * v01.x) + hl1 * (wl0 * v10.x + wl1 * v11.x);
  o.y = hl0 * (wl0 * v00.y + wl1 * v01.y) + hl1 * (wl0 * v10.y + wl1 * v11.y);
  o.z = hl0 * (wl0 * v00.z + wl1 * v01.z) + hl1 * (wl0 * v10.z + wl1 * v11.z);
  o.w = hl0 * (wl0 * v00.w + wl1 * v01.w) + hl1 * (wl0 * v10.w + wl1 * v11.w);
  return o;
}

__global__ void resize_bilinear_kernel(const float* __restrict__ in, int N, int H, int W, int C4, int ics, int ico,
                                       int Ho, int Wo, float rh, float rw, const float* __restrict__ add, int acs,
                                       int aco, float* __restrict__ out, int ocs, int oco) {
  const size_t total = (size_t)N * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const size_t pix = i / C4;
    const int wo = (int)(pix % Wo);
    const size_t t = pix / Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const Lin ly = lin_src(ho, H, rh), lx = lin_src(wo, W, rw);
    const float* base = in + (size_t)n * H * W * ics + ico + c4 * 4;
    const float4 v00 = *reinterpret_cast<const float4*>(base + ((size_t)ly.i0 * W + lx.i0) * ics);
    const float4 v01 = *reinterpret_cast<const float4*>(base + ((size_t)ly.i0 * W + lx.i1) * ics);
    const float4 v10 = *reinterpret_cast<const float4*>(base + ((size_t)ly.i1 * W + lx.i0) * ics);
    const float4 v11 = *reinterpret_cast<const float4*>(base + ((size_t)ly.i1 * W + lx.i1) * ics);
    float4 o = f4_bilerp(v00, v01, v10, v11, lx.l0, lx.l1, ly.l0, ly.l1);
    if (add) {
      const float4 a = *reinterpret_cast<const float4*>(add + pix * acs + aco + c4 * 4);
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    *reinterpret_cast<float4*>(out + pix * ocs + oco + c4 * 4) = o;
  }
}

// ---------------------------------------------------------------- flow warp
struct WarpParams {
  const float* src;
  int N, H, W, C4, scs, sco;
  const float* flow;
  int fh, fw, Ho, Wo;
  float rh, rw, norm_x, norm_y;
  float step_x, step_y;  // fp32 linspace step 2/(Wo-1), 2/(Ho-1)
  float* out;
  int ocs, oco;
  float* flow_up;
};

// torch.linspace(-1, 1, n)[i] in fp32 (networks.py:162-163): first half from the
// start, second half from the end.
__device__ __forceinline__ float lin_m1_1(int i, int n, float step) {
  return i < n / 2 ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}

__global__ void flow_warp_kernel(const WarpParams p) {
  const size_t total = (size_t)p.N * p.Ho * p.Wo * p.C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % p.C4);
    const size_t pix = i / p.C4;
    const int wo = (int)(pix % p.Wo);
    const size_t t = pix / p.Wo;
    const int ho = (int)(t % p.Ho);
    const int n = (int)(t / p.Ho);
    // 1) bilinear upsample of the previous flow (2 channels, NHWC)
    const Lin ly = lin_src(ho, p.fh, p.rh), lx = lin_src(wo, p.fw, p.rw);
    const float2* fb = reinterpret_cast<const float2*>(p.flow) + (size_t)n * p.fh * p.fw;
    const float2 f00 = fb[(size_t)ly.i0 * p.fw + lx.i0], f01 = fb[(size_t)ly.i0 * p.fw + lx.i1];
    const float2 f10 = fb[(size_t)ly.i1 * p.fw + lx.i0], f11 = fb[(size_t)ly.i1 * p.fw + lx.i1];
    const float fx = ly.l0 * (lx.l0 * f00.x + lx.l1 * f01.x) + ly.l1 * (lx.l0 * f10.x + lx.l1 * f11.x);
    const float fy = ly.l0 * (lx.l0 * f00.y + lx.l1 * f01.y) + ly.l1 * (lx.l0 * f10.y + lx.l1 * f11.y);
    if (p.flow_up && c4 == 0) reinterpret_cast<float2*>(p.flow_up)[pix] = make_float2(fx, fy);
    // 2) normalise, add the linspace(-1,1) base grid
    const float gx = fx / p.norm_x + lin_m1_1(wo, p.Wo, p.step_x);
    const float gy = fy / p.norm_y + lin_m1_1(ho, p.Ho, p.step_y);
    // 3) grid_sample: un-normalise (align_corners=False), clamp to the border, 4 taps
    float ix = ((gx + 1.f) * (float)p.W - 1.f) / 2.f;
    float iy = ((gy + 1.f) * (float)p.H - 1.f) / 2.f;
    ix = fminf(fmaxf(ix, 0.f), (float)(p.W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(p.H - 1));
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - x0f, wx0 = (x0f + 1.f) - ix;
    const float wy1 = iy - y0f, wy0 = (y0f + 1.f) - iy;
    const float* sb = p.src + (size_t)n * p.H * p.W * p.scs + p.sco + c4 * 4;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    auto tap = [&](int x, int y, float w) {
      if (x >= 0 && x < p.W && y >= 0 && y < p.H) {
        const float4 v = *reinterpret_cast<const float4*>(sb + ((size_t)y * p.W + x) * p.scs);
        o.x += v.x * w; o.y += v.y * w; o.z += v.z * w; o.w += v.w * w;
      }
    };
    tap(x0, y0, wx0 * wy0);
    tap(x1, y0, wx1 * wy0);
    tap(x0, y1, wx0 * wy1);
    tap(x1, y1, wx1 * wy1);
    *reinterpret_cast<float4*>(p.out + pix * p.ocs + p.oco + c4 * 4) = o;
  }
}


// ---------------------------------------------------------------- tap-sum
// Second half of a KxK convolution with a tiny Cout (the 768->2 flow_conv,
// networks.py:85-92): the conv engine first runs it as a 1x1 convolution with
// Cout' = KH*KW*Cout "tap channels" (y[q][tap][co] = sum_c x[q][c] w[co][c][tap]),
// reading the wide input exactly once; this kernel then gathers
//   out[p][co] = bias[co] + sum_tap y[p + off(tap)][tap][co] (+ residual)
// with zero padding (out-of-range q contributes nothing).  stride 1 only.
__global__ void tapsum_kernel(const float* __restrict__ y, int N, int H, int W, int KH, int KW, int pad, int Cout,
                              int ycs, const float* __restrict__ bias, const float* __restrict__ res, int rcs,
                              float* __restrict__ out, int ocs) {
  const size_t total = (size_t)N * H * W * Cout;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout);
    const size_t pix = i / Cout;
    const int w = (int)(pix % W);
    const size_t t = pix / W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float acc = 0.f;
    for (int kh = 0; kh < KH; ++kh) {
      const int hh = h + kh - pad;
      if (hh < 0 || hh >= H) continue;
      for (int kw = 0; kw < KW; ++kw) {
        const int ww = w + kw - pad;
        if (ww < 0 || ww >= W) continue;
        acc += y[((size_t)(n * H + hh) * W + ww) * ycs + (kh * KW + kw) * Cout + co];
      }
    }
    if (bias) acc += bias[co];
    if (res) acc += res[pix * rcs + co];
    out[pix * ocs + co] = acc;
  }
}

// ------------------------------------------------------------------ space <-> depth (factor 2)
// out[n][y/2][x/2][((y&1)*2 + (x&1))*C + c] = in[n][y][x][c]: a 4x4 stride-2 pad-2 convolution over C channels becomes a
// 2x2 stride-1 pad-1 convolution over 4C (PatchGAN model0, 10 channels: one 64-k row per tap wastes 5/6 of the matrix
// cores and of the gather otherwise -- gen_train.S2DConv).  dir 0: in -> out, dir 1: the inverse.
__global__ void space_depth2_kernel(const float* __restrict__ a, int N, int H, int W, int C4, int cs, int co,
                                    float* __restrict__ b, int dir) {
  const size_t total = (size_t)N * H * W * C4;
  const int Hc = H >> 1, Wc = W >> 1, C = C4 * 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t t = i / C4;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const size_t full = (((size_t)n * H + y) * W + x) * cs + co + 4 * c4;
    const size_t cell = ((((size_t)n * Hc + (y >> 1)) * Wc + (x >> 1)) * 4 + (y & 1) * 2 + (x & 1)) * C + 4 * c4;
    if (dir == 0) *reinterpret_cast<float4*>(b + cell) = *reinterpret_cast<const float4*>(a + full);
    else *reinterpret_cast<float4*>(b + full) = *reinterpret_cast<const float4*>(a + cell);
  }
}

// ------------------------------------------------------------------ channel concat of an NHWC and an NCHW tensor into NHWC
// out[n][p][0:Ca] = a[n][p][co : co+Ca],  out[n][p][Ca : Ca+Cb] = b[n][:, p],  pad channels up to ocs = 0  (one thread a pixel:
// the plane reads of b are coalesced across the wave, a pixel's output row is contiguous)
__global__ void concat_nhwc_nchw_kernel(const float* __restrict__ a, int Ca, int acs, int aco, const float* __restrict__ b,
                                        int Cb, int N, int HW, float* __restrict__ out, int ocs) {
  const size_t total = (size_t)N * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / HW), p = (int)(i - (size_t)n * HW);
    float* o = out + i * ocs;
    const float* ap = a + i * acs + aco;
    for (int c = 0; c < Ca; ++c) o[c] = ap[c];
    const float* bp = b + (size_t)n * Cb * HW + p;
    for (int c = 0; c < Cb; ++c) o[Ca + c] = bp[(size_t)c * HW];
    for (int c = Ca + Cb; c < ocs; ++c) o[c] = 0.f;
  }
}

// the PatchGAN shape (8-channel label rows, at most 12 output channels): 16-byte loads of the label row, 16-byte stores
template <int CA, int CB>
__global__ void concat_nhwc_nchw_v4_kernel(const float* __restrict__ a, const float* __restrict__ b, int N, int HW,
                                           float* __restrict__ out) {
  static_assert(CA <= 8 && CA + CB <= 12, "one 8-float label row, three float4 of output");
  const size_t total = (size_t)N * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / HW), p = (int)(i - (size_t)n * HW);
    const float4 a0 = *reinterpret_cast<const float4*>(a + i * 8), a1 = *reinterpret_cast<const float4*>(a + i * 8 + 4);
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float* bp = b + (size_t)n * CB * HW + p;
    float o[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) o[c] = c < CA ? av[c] : (c < CA + CB ? bp[(size_t)(c - CA) * HW] : 0.f);
    float4* op = reinterpret_cast<float4*>(out + i * 12);
    op[0] = make_float4(o[0], o[1], o[2], o[3]);
    op[1] = make_float4(o[4], o[5], o[6], o[7]);
    op[2] = make_float4(o[8], o[9], o[10], o[11]);
  }
}

}  // namespace hrv

using namespace hrv;

extern "C" const char* hrv_version(void) { return "hrviton-hip 0.1 (gfx950)"; }
extern "C" const char* hrv_last_error(void) { return g_err; }

namespace hrv {
int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev < HRV_MAX_DEVICES ? dev : HRV_MAX_DEVICES - 1;
}
int device_cus() {
  static int cus[HRV_MAX_DEVICES] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  if (dev < HRV_MAX_DEVICES && cus[dev] > 0) return cus[dev];
  hipDeviceProp_t prop;
  int n = 0;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
  if (n <= 0) n = 256;
  if (dev < HRV_MAX_DEVICES) cus[dev] = n;
  return n;
}
// HRV_* switches: the environment is read once per name and cached.  The cache hands out pointers that stay valid for the life of
// the process: a reload (hrv_diag_reload_env) re-reads every cached name and, where the value changed, points the entry at a NEW
// copy -- the old one is never freed, so a launch on another thread (the autograd backward thread) that still parses the pointer it
// got before the reload reads valid memory (clear() here was a use-after-free under that race).
static std::mutex g_env_mu;
static std::unordered_map<std::string, const char*>& env_cache() {
  static std::unordered_map<std::string, const char*> c;
  return c;
}
static const char* env_copy(const char* e) {
  if (!e) return nullptr;
  char* c = (char*)malloc(strlen(e) + 1);      // (deliberately leaked: see above)
  if (c) strcpy(c, e);
  return c;
}
const char* env(const char* name) {
  std::lock_guard<std::mutex> lk(g_env_mu);
  auto& c = env_cache();
  auto it = c.find(name);
  if (it == c.end()) it = c.emplace(std::string(name), env_copy(::getenv(name))).first;
  return it->second;
}
static bool g_reserved_explicit = false;      // hrv_set_reserved_cus was called: a reload keeps that value
static int g_reserved_cus = -1;      // -1: not set yet (HRV_RESERVE_CUS is read once)
int persistent_cus() {
  if (g_reserved_cus < 0) {
    const char* e = hrv::env("HRV_RESERVE_CUS");
    int k = e ? atoi(e) : 0;
    g_reserved_cus = k < 0 ? 0 : k;
  }
  const int n = device_cus() - g_reserved_cus;
  return n < 8 ? 8 : n;
}
static unsigned long long* g_tlog = nullptr;
static long long g_tlog_tiles = 0;
unsigned long long* diag_tlog(long long tiles) { return (g_tlog != nullptr && tiles <= g_tlog_tiles) ? g_tlog : nullptr; }
}  // namespace hrv

extern "C" int hrv_diag_reload_env(void) {
  std::lock_guard<std::mutex> lk(hrv::g_env_mu);
  for (auto& kv : hrv::env_cache()) {
    const char* e = ::getenv(kv.first.c_str());
    const bool same = (e == nullptr && kv.second == nullptr) || (e != nullptr && kv.second != nullptr && strcmp(e, kv.second) == 0);
    if (!same) kv.second = hrv::env_copy(e);
  }
  if (!hrv::g_reserved_explicit) hrv::g_reserved_cus = -1;      // HRV_RESERVE_CUS is read again
  return HRV_OK;
}

extern "C" int hrv_set_reserved_cus(int32_t k) {
  HRV_REQUIRE(k >= 0 && k < 4096, "set_reserved_cus: %d", k);
  hrv::g_reserved_cus = k;
  hrv::g_reserved_explicit = true;
  return HRV_OK;
}

extern "C" int hrv_persistent_cus(void) { return hrv::persistent_cus(); }

extern "C" int hrv_diag_set_tlog(void* buf, int64_t tiles) {
  HRV_REQUIRE((buf == nullptr) == (tiles == 0) && tiles >= 0, "diag_set_tlog: buf and tiles go together");
  hrv::g_tlog = (unsigned long long*)buf;
  hrv::g_tlog_tiles = tiles;
  return HRV_OK;
}

extern "C" int hrv_device_check(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
    set_error("no HIP device visible");
    return HRV_ERR_NODEV;
  }
  hipDeviceProp_t prop;
  int dev = 0;
  hipGetDevice(&dev);
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    set_error("hipGetDeviceProperties failed");
    return HRV_ERR_NODEV;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device is %s, this library is built for gfx950 only", prop.gcnArchName);
    return HRV_ERR_NODEV;
  }
  return HRV_OK;
}

extern "C" int hrv_nchw_to_nhwc_f32(const float* in, int32_t N, int32_t C, int32_t H, int32_t W, float* out,
                                    int32_t out_cstride, int32_t out_coff, int32_t zero_tail, hrv_stream_t stream) {
  HRV_REQUIRE(in && out && N > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad args");
  HRV_REQUIRE(out_coff >= 0 && out_coff + C <= out_cstride, "nchw_to_nhwc: slice out of range");
  HRV_REQUIRE(zero_tail >= 0 && out_coff + C + zero_tail <= out_cstride, "nchw_to_nhwc: zero_tail %d leaves the pixel (coff %d, C %d, cstride %d)",
              zero_tail, out_coff, C, out_cstride);
  const size_t total = (size_t)N * H * W;
  (void)total;
  const size_t tiles = (size_t)N * (((size_t)H * W + LT_P - 1) / LT_P);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)(tiles > 16384 ? 16384 : tiles)), dim3(256), 0,
                     (hipStream_t)stream, in, N, C, H * W, out, out_cstride, out_coff, zero_tail);
  return check_launch("nchw_to_nhwc_kernel");
}

extern "C" int hrv_nhwc_to_nchw_f32(const float* in, int32_t in_cstride, int32_t in_coff, int32_t N, int32_t C,
                                    int32_t H, int32_t W, float* out, hrv_stream_t stream) {
  HRV_REQUIRE(in && out && N > 0 && C > 0 && H > 0 && W > 0, "nhwc_to_nchw: bad args");
  HRV_REQUIRE(in_coff >= 0 && in_coff + C <= in_cstride, "nhwc_to_nchw: slice out of range");
  const size_t total = (size_t)N * H * W;
  (void)total;
  const size_t tiles = (size_t)N * (((size_t)H * W + LT_P - 1) / LT_P);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)(tiles > 16384 ? 16384 : tiles)), dim3(256), 0,
                     (hipStream_t)stream, in, in_cstride, in_coff, N, C, H * W, out);
  return check_launch("nhwc_to_nchw_kernel");
}

extern "C" int hrv_nchw_f32_to_nhwc_bf16(const float* in, int32_t N, int32_t C, int32_t H, int32_t W, uint16_t* out,
                                         int32_t out_cstride, int32_t out_coff, int32_t zero_tail, hrv_stream_t stream) {
  HRV_REQUIRE(in && out && N > 0 && C > 0 && H > 0 && W > 0, "nchw_f32_to_nhwc_bf16: bad args");
  HRV_REQUIRE(out_coff >= 0 && out_coff + C <= out_cstride, "nchw_f32_to_nhwc_bf16: slice out of range");
  HRV_REQUIRE(zero_tail >= 0 && out_coff + C + zero_tail <= out_cstride, "nchw_f32_to_nhwc_bf16: zero_tail %d leaves the pixel", zero_tail);
  const size_t total = (size_t)N * H * W;
  hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, N, C,
                     H * W, out, out_cstride, out_coff, zero_tail);
  return check_launch("nchw_f32_to_nhwc_bf16_kernel");
}

extern "C" int hrv_nhwc_bf16_to_nchw_f32(const uint16_t* in, int32_t in_cstride, int32_t in_coff, int32_t N, int32_t C,
                                         int32_t H, int32_t W, float* out, hrv_stream_t stream) {
  HRV_REQUIRE(in && out && N > 0 && C > 0 && H > 0 && W > 0, "nhwc_bf16_to_nchw_f32: bad args");
  HRV_REQUIRE(in_coff >= 0 && in_coff + C <= in_cstride, "nhwc_bf16_to_nchw_f32: slice out of range");
  const size_t total = (size_t)N * H * W;
  hipLaunchKernelGGL(nhwc_bf16_to_nchw_f32_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in,
                     in_cstride, in_coff, N, C, H * W, out);
  return check_launch("nhwc_bf16_to_nchw_f32_kernel");
}

extern "C" int hrv_resize_bilinear_nhwc_f32(const float* in, int32_t N, int32_t H, int32_t W, int32_t C,
                                            int32_t in_cstride, int32_t in_coff, int32_t Ho, int32_t Wo, float rh,
                                            float rw, const float* addend, int32_t add_cstride, int32_t add_coff,
                                            float* out, int32_t out_cstride, int32_t out_coff,
                                            hrv_stream_t stream) {
  HRV_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "resize_bilinear: bad args");
  HRV_REQUIRE(C > 0 && C % 4 == 0 && in_cstride % 4 == 0 && in_coff % 4 == 0 && out_cstride % 4 == 0 && out_coff % 4 == 0,
              "resize_bilinear: channel counts/offsets must be multiples of 4");
  HRV_REQUIRE(in_coff + C <= in_cstride && out_coff + C <= out_cstride, "resize_bilinear: slice out of range");
  HRV_REQUIRE(!addend || (add_cstride % 4 == 0 && add_coff % 4 == 0 && add_coff + C <= add_cstride),
              "resize_bilinear: addend slice");
  HRV_REQUIRE((((uintptr_t)in | (uintptr_t)out | (uintptr_t)addend) & 15) == 0, "resize_bilinear: 16-byte alignment");
  const size_t total = (size_t)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, N, H, W,
                     C / 4, in_cstride, in_coff, Ho, Wo, rh, rw, addend, add_cstride, add_coff, out, out_cstride,
                     out_coff);
  return check_launch("resize_bilinear_kernel");
}

extern "C" int hrv_flow_warp_nhwc_f32(const hrv_flow_warp_t* d, hrv_stream_t stream) {
  HRV_REQUIRE(d && d->src && d->flow && d->out, "flow_warp: null pointer");
  HRV_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->fh > 0 && d->fw > 0 && d->Ho > 0 && d->Wo > 0, "flow_warp: extent");
  HRV_REQUIRE(d->C > 0 && d->C % 4 == 0 && d->src_cstride % 4 == 0 && d->src_coff % 4 == 0 &&
                  d->out_cstride % 4 == 0 && d->out_coff % 4 == 0,
              "flow_warp: channel counts/offsets must be multiples of 4");
  HRV_REQUIRE(d->src_coff + d->C <= d->src_cstride && d->out_coff + d->C <= d->out_cstride, "flow_warp: slice range");
  HRV_REQUIRE((((uintptr_t)d->src | (uintptr_t)d->out) & 15) == 0 && ((uintptr_t)d->flow & 7) == 0 &&
                  ((uintptr_t)d->flow_up & 7) == 0,
              "flow_warp: alignment");
  HRV_REQUIRE(d->norm_x != 0.f && d->norm_y != 0.f, "flow_warp: zero normaliser");
  WarpParams p;
  p.src = d->src; p.N = d->N; p.H = d->H; p.W = d->W; p.C4 = d->C / 4; p.scs = d->src_cstride; p.sco = d->src_coff;
  p.flow = d->flow; p.fh = d->fh; p.fw = d->fw; p.Ho = d->Ho; p.Wo = d->Wo;
  p.rh = d->rh; p.rw = d->rw; p.norm_x = d->norm_x; p.norm_y = d->norm_y;
  p.step_x = d->Wo > 1 ? 2.0f / (float)(d->Wo - 1) : 0.f;
  p.step_y = d->Ho > 1 ? 2.0f / (float)(d->Ho - 1) : 0.f;
  p.out = d->out; p.ocs = d->out_cstride; p.oco = d->out_coff; p.flow_up = d->flow_up;
  const size_t total = (size_t)d->N * d->Ho * d->Wo * p.C4;
  hipLaunchKernelGGL(flow_warp_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("flow_warp_kernel");
}

extern "C" int hrv_tapsum_nhwc_f32(const float* y, int32_t N, int32_t H, int32_t W, int32_t KH, int32_t KW,
                                   int32_t pad, int32_t Cout, int32_t y_cstride, const float* bias,
                                   const float* residual, int32_t res_cstride, float* out, int32_t out_cstride,
                                   hrv_stream_t stream) {
  HRV_REQUIRE(y && out && N > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && pad >= 0 && Cout > 0, "tapsum: bad args");
  HRV_REQUIRE(KH == 2 * pad + 1 && KW == 2 * pad + 1, "tapsum: only 'same' stride-1 geometry (k = 2*pad+1)");
  HRV_REQUIRE(y_cstride >= KH * KW * Cout && out_cstride >= Cout && (!residual || res_cstride >= Cout),
              "tapsum: channel strides too small");
  const size_t total = (size_t)N * H * W * Cout;
  hipLaunchKernelGGL(tapsum_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, y, N, H, W, KH, KW,
                     pad, Cout, y_cstride, bias, residual, res_cstride, out, out_cstride);
  return check_launch("tapsum_kernel");
}

// [N,H,W,C] (a channel slice of a wider fp32 tensor) -> dense [N,H/2,W/2,4C], channel = ((y&1)*2 + (x&1))*C + c.
extern "C" int hrv_space_to_depth2_nhwc_f32(const float* in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_cstride,
                                            int32_t in_coff, float* out, hrv_stream_t stream) {
  HRV_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && C > 0, "space_to_depth2: bad args");
  HRV_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && in_cstride % 4 == 0 && in_coff % 4 == 0 && in_coff + C <= in_cstride &&
                  (((uintptr_t)in | (uintptr_t)out) & 15) == 0,
              "space_to_depth2: even extents, 4-channel granules");
  const size_t total = (size_t)N * H * W * (C / 4);
  hipLaunchKernelGGL(space_depth2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, N, H, W, C / 4, in_cstride,
                     in_coff, out, 0);
  return check_launch("space_depth2_kernel");
}

// the inverse: dense [N,H/2,W/2,4C] -> dense [N,H,W,C]  (H, W: the FULL extents)
extern "C" int hrv_depth_to_space2_nhwc_f32(const float* in, int32_t N, int32_t H, int32_t W, int32_t C, float* out,
                                            hrv_stream_t stream) {
  HRV_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && C > 0, "depth_to_space2: bad args");
  HRV_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0,
              "depth_to_space2: even extents, 4-channel granules");
  const size_t total = (size_t)N * H * W * (C / 4);
  hipLaunchKernelGGL(space_depth2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, N, H, W, C / 4, C, 0,
                     out, 1);
  return check_launch("space_depth2_kernel");
}

// cat((a, b), dim=channel) with a NHWC (a channel slice) and b NCHW, written NHWC with zeroed pad channels -- the PatchGAN
// input cat((parse, image), 1) of train_generator.py:283-284 without materialising NCHW copies.
extern "C" int hrv_concat_nhwc_nchw_f32(const float* a, int32_t Ca, int32_t a_cstride, int32_t a_coff, const float* b, int32_t Cb,
                                        int32_t N, int32_t H, int32_t W, float* out, int32_t out_cstride, hrv_stream_t stream) {
  HRV_REQUIRE(a && b && out && Ca > 0 && Cb > 0 && N > 0 && H > 0 && W > 0, "concat_nhwc_nchw: bad args");
  HRV_REQUIRE(a_coff >= 0 && a_coff + Ca <= a_cstride && Ca + Cb <= out_cstride, "concat_nhwc_nchw: channel ranges");
  const size_t total = (size_t)N * H * W;
  if (Ca == 7 && Cb == 3 && a_cstride == 8 && a_coff == 0 && out_cstride == 12 && (((uintptr_t)a | (uintptr_t)out) & 15) == 0) {
    hipLaunchKernelGGL((concat_nhwc_nchw_v4_kernel<7, 3>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a, b, N, H * W, out);
    return check_launch("concat_nhwc_nchw_v4_kernel");
  }
  hipLaunchKernelGGL(concat_nhwc_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a, Ca, a_cstride, a_coff, b,
                     Cb, N, H * W, out, out_cstride);
  return check_launch("concat_nhwc_nchw_kernel");
}
