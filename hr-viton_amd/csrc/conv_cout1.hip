// Convolutions with ONE output channel -- the last layer of every PatchGAN scale (network_generator.py NLayerDiscriminator:
// Conv2d(nf, 1, kernel_size=4, stride=1, padding=2); networks.py:389-393 for the condition generator's discriminator):
// forward, data gradient and weight gradient.  The implicit-GEMM engine pads the single column to a 64-column tile: at
// 2 x 4 x 131 x 99 pixels x 256 channels the forward took 0.23 ms (3.6 TFLOP/s, 0.44 TB/s), the data gradient 0.09-0.13 ms,
// the weight gradient 0.19 ms.  None of them is a matrix product worth the matrix cores: the forward is a 4096-long dot
// product per pixel (HBM floor: one read of x), the data gradient an outer product (one write of dx), the weight gradient
// a reduction over pixels.  fp32 FMAs, fp32 accumulation; ``round_bf16`` rounds both operands to bf16 first -- the
// arithmetic of the bf16 matrix-core engine these kernels replace in mixed-precision training.
#include "conv_params.h"

namespace hrv {

__device__ __forceinline__ float c1_rb(float v, int round_bf16) {
  if (!round_bf16) return v;
  unsigned u = __builtin_bit_cast(unsigned, v);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  return __builtin_bit_cast(float, u);
}

struct Cout1Params {
  const float* x; int N, H, W, C, xcs, xco;       // fp32 NHWC source (C % 4 == 0, 16-byte aligned rows)
  const float* w;                                   // [1][C][K][K] fp32 (OIHW)
  const float* sigma; float wscale;                 // weights are multiplied by wscale / sigma[0]
  const float* bias;                                // [1] or null
  int K, pad, Ho, Wo;
  float* y; int ycs, yco;                           // forward output / data-gradient input: [N][Ho][Wo][ycs], channel yco
  float* dx; int dcs, dco;                          // data gradient output [N][H][W][dcs]
  const float* add; int acs, aco;                   // optional tensor added to dx (feature-matching gradient)
  float* part; int S;                               // weight gradient: partials [S][C*K*K + 1] (last: bias)
  int round_bf16;
};

constexpr int C1_STRIP = 8;
constexpr int C1_ROWS = 4;

// forward: one wave per block of C1_ROWS x C1_STRIP output pixels; lane = 4 channels (x channel groups of 256).  Round 6: the first
// build walked one output row per wave with its weights in LDS -- 0.116 ms for the 102 MB of f2 at 8 x 129 x 97 x 256, and neither the
// 5.5-fold re-read of the input out of L2 nor the one-load-at-a-time column loop was the bound (a row-sliding window and eleven loads
// in flight moved it to 0.100): every (column, kernel row, tap, output) product re-read its weight from LDS, 1.26 MB per wave, the
// CU's LDS bandwidth.  The 16 taps x 4 channels of a lane now sit in 64 registers (zeros beyond K), the C1_ROWS + K - 1 input rows of
// the block are read once each with all their columns requested up front.
__global__ __launch_bounds__(256) void cout1_fwd_kernel(const Cout1Params p) {
  const int KK = p.K * p.K;
  const float mul = p.sigma ? p.wscale / p.sigma[0] : p.wscale;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strips_per_row = (p.Wo + C1_STRIP - 1) / C1_STRIP, row_groups = (p.Ho + C1_ROWS - 1) / C1_ROWS;
  const int total = p.N * row_groups * strips_per_row;
  const float b = p.bias ? p.bias[0] : 0.f;
  for (int s = blockIdx.x * 4 + wave; s < total; s += gridDim.x * 4) {
    const int sx = s % strips_per_row;
    const int t = s / strips_per_row;
    const int rg = t % row_groups, n = t / row_groups;
    const int wo0 = sx * C1_STRIP, ho0 = rg * C1_ROWS;
    float acc[C1_ROWS][C1_STRIP];
#pragma unroll
    for (int a = 0; a < C1_ROWS; ++a)
#pragma unroll
      for (int o = 0; o < C1_STRIP; ++o) acc[a][o] = 0.f;
    for (int c0 = lane * 4; c0 < p.C; c0 += 256) {
      float4 wq[4][4];                                             // [kernel row][tap]: the lane's four channels
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
          const bool in = r < p.K && kw < p.K;
          const float* wp_ = p.w + (size_t)c0 * KK + (in ? r * p.K + kw : 0);
          wq[r][kw].x = in ? c1_rb(wp_[0] * mul, p.round_bf16) : 0.f;
          wq[r][kw].y = in ? c1_rb(wp_[KK] * mul, p.round_bf16) : 0.f;
          wq[r][kw].z = in ? c1_rb(wp_[2 * KK] * mul, p.round_bf16) : 0.f;
          wq[r][kw].w = in ? c1_rb(wp_[3 * KK] * mul, p.round_bf16) : 0.f;
        }
#pragma unroll
      for (int ri = 0; ri < C1_ROWS + 3; ++ri) {                   // input row ho0 + ri - pad (K <= 4)
        const int h = ho0 + ri - p.pad;
        if (ri >= C1_ROWS + p.K - 1 || h < 0 || h >= p.H) continue;
        const float* xrow = p.x + ((size_t)n * p.H + h) * p.W * p.xcs + p.xco + c0;
        // input columns wo0 - pad .. wo0 + STRIP - 1 - pad + K - 1, all requested before the first is used; column q feeds output
        // o = q - kw (kw = 0..K-1) of the output rows ri - r (kernel row r = 0..K-1) that lie inside the block
        float4 xq[C1_STRIP + 3];
#pragma unroll
        for (int q = 0; q < C1_STRIP + 3; ++q) {
          const int wi = wo0 + q - p.pad;
          const bool ok = q < C1_STRIP + p.K - 1 && wi >= 0 && wi < p.W;
          xq[q] = ok ? *reinterpret_cast<const float4*>(xrow + (size_t)wi * p.xcs) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < C1_STRIP + 3; ++q) {
          float4 xv = xq[q];
          xv.x = c1_rb(xv.x, p.round_bf16); xv.y = c1_rb(xv.y, p.round_bf16);
          xv.z = c1_rb(xv.z, p.round_bf16); xv.w = c1_rb(xv.w, p.round_bf16);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int orow = ri - r;                               // compile-time: ri and r are unrolled
            if (orow < 0 || orow >= C1_ROWS) continue;
#pragma unroll
            for (int o = 0; o < C1_STRIP; ++o) {
              const int kw = q - o;                                // compile-time; taps beyond K hold zero weights
              if (kw >= 0 && kw < 4) {
                const float4 wv = wq[r][kw];
                acc[orow][o] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int a = 0; a < C1_ROWS; ++a)
#pragma unroll
      for (int o = 0; o < C1_STRIP; ++o) {
        float v = acc[a][o];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0 && wo0 + o < p.Wo && ho0 + a < p.Ho) p.y[(((size_t)n * p.Ho + ho0 + a) * p.Wo + wo0 + o) * p.ycs + p.yco] = v + b;
      }
  }
}

// The K rows of dY one INPUT row (n, h) sees, zero-padded, in LDS:  dyl[kh][K + wo] = dY[n][h - kh + pad][wo]  (0 outside).
// A tap (kh, kw) of input pixel w then reads dyl[kh][K + w - kw + pad] -- no bounds checks in the inner loops.
__device__ __forceinline__ void c1_stage_dy(const Cout1Params& p, int n, int h, float* dyl, int LW) {
  for (int idx = threadIdx.x; idx < p.K * LW; idx += 256) {
    const int kh = idx / LW, wo = idx - kh * LW - p.K;
    const int ho = h - kh + p.pad;
    float v = 0.f;
    if (wo >= 0 && wo < p.Wo && ho >= 0 && ho < p.Ho)
      v = c1_rb(p.y[(((size_t)n * p.Ho + ho) * p.Wo + wo) * p.ycs + p.yco], p.round_bf16);
    dyl[idx] = v;
  }
}

// data gradient: dx[n][h][w][c] = sum_{kh,kw} dy[n][h - kh + pad][w - kw + pad] * w[c][kh][kw]  (+ add).  One block per
// input row; weights [K*K][C] and the dY rows in LDS; a wave per pixel, a lane per 4 channels (x channel groups of 256).
// (First build: one wave per pixel with its 16 dY values fetched from global memory one after the other -- 0.099 ms at
// 2 x 4 x 130 x 98 x 256, no faster than the padded matrix-core path.)
__global__ __launch_bounds__(256) void cout1_dgrad_kernel(const Cout1Params p) {
  extern __shared__ float smem[];
  const int KK = p.K * p.K, LW = p.Wo + 2 * p.K;
  float* const wl = smem;                           // [K*K][C]
  float* const dyl = smem + KK * p.C;               // [K][LW]
  const float mul = p.sigma ? p.wscale / p.sigma[0] : p.wscale;
  for (int i = threadIdx.x; i < KK * p.C; i += 256) {
    const int tap = i / p.C, c = i - tap * p.C;
    wl[i] = c1_rb(p.w[(size_t)c * KK + tap] * mul, p.round_bf16);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x; row < p.N * p.H; row += gridDim.x) {
    const int n = row / p.H, h = row - n * p.H;
    __syncthreads();                                // (weights staged / previous row's dyl fully read)
    c1_stage_dy(p, n, h, dyl, LW);
    __syncthreads();
    for (int w_ = wave; w_ < p.W; w_ += 4) {
      const size_t px = (size_t)row * p.W + w_;
      float g[4][4];
#pragma unroll
      for (int kh = 0; kh < 4; ++kh)
#pragma unroll
        for (int kw = 0; kw < 4; ++kw)
          g[kh][kw] = (kh < p.K && kw < p.K) ? dyl[kh * LW + p.K + w_ - kw + p.pad] : 0.f;
      for (int c0 = lane * 4; c0 < p.C; c0 += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < 4; ++kh)
#pragma unroll
          for (int kw = 0; kw < 4; ++kw)
            if (kh < p.K && kw < p.K) {
              const float4 wv = *reinterpret_cast<const float4*>(wl + (size_t)(kh * p.K + kw) * p.C + c0);
              acc.x += g[kh][kw] * wv.x; acc.y += g[kh][kw] * wv.y; acc.z += g[kh][kw] * wv.z; acc.w += g[kh][kw] * wv.w;
            }
        if (p.add) {
          const float4 a = *reinterpret_cast<const float4*>(p.add + px * p.acs + p.aco + c0);
          acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
        }
        *reinterpret_cast<float4*>(p.dx + px * p.dcs + p.dco + c0) = acc;
      }
    }
  }
}

// weight gradient partials: one block per input row (n, h), thread = channel; the dY rows in LDS (broadcast reads).
//   part[row][c][kh][kw] = sum_w x[n][h][w][c] * dY[n][h - kh + pad][w - kw + pad],   part[row][C*K*K] = sum of dY row h
//   (+ the rows below the image, h = H-1 only): the bias gradient.
// (First build: a slab walk with 16 bounds-checked global dY loads per pixel and thread: 0.72 ms against the engine's 0.19.)
__global__ __launch_bounds__(256) void cout1_wgrad_kernel(const Cout1Params p) {
  extern __shared__ float smem[];
  __shared__ float red[4];
  const int KK = p.K * p.K, LW = p.Wo + 2 * p.K;
  float* const dyl = smem;                          // [K][LW]
  const int row = blockIdx.x, n = row / p.H, h = row - n * p.H;
  const int c = blockIdx.y * 256 + threadIdx.x;
  c1_stage_dy(p, n, h, dyl, LW);
  __syncthreads();
  float acc[4][4];
#pragma unroll
  for (int kh = 0; kh < 4; ++kh)
#pragma unroll
    for (int kw = 0; kw < 4; ++kw) acc[kh][kw] = 0.f;
  if (c < p.C) {
    const float* xrow = p.x + ((size_t)row * p.W) * p.xcs + p.xco + c;
#pragma unroll 8
    for (int w_ = 0; w_ < p.W; ++w_) {      // (eight pixels in flight per thread: 4-byte loads, one block per row -- latency-bound at four)
      const float xv = c1_rb(xrow[(size_t)w_ * p.xcs], p.round_bf16);
      const float* dq = dyl + p.K + w_ + p.pad;
#pragma unroll
      for (int kh = 0; kh < 4; ++kh)
#pragma unroll
        for (int kw = 0; kw < 4; ++kw)
          if (kh < p.K && kw < p.K) acc[kh][kw] += xv * dq[kh * LW - kw];
    }
  }
  float* dst = p.part + (size_t)row * (p.C * KK + 1);
  if (c < p.C) {
#pragma unroll
    for (int kh = 0; kh < 4; ++kh)
#pragma unroll
      for (int kw = 0; kw < 4; ++kw)
        if (kh < p.K && kw < p.K) dst[c * KK + kh * p.K + kw] = acc[kh][kw];
  }
  if (blockIdx.y == 0) {                            // bias partial: the UNROUNDED dY of output row h (and of the rows >= H)
    float bs = 0.f;
    const int ho_end = h == p.H - 1 ? p.Ho : h + 1;
    for (int ho = h; ho < ho_end; ++ho)
      for (int wo = threadIdx.x; wo < p.Wo; wo += 256) bs += p.y[(((size_t)n * p.Ho + ho) * p.Wo + wo) * p.ycs + p.yco];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) bs += __shfl_xor(bs, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bs;
    __syncthreads();
    if (threadIdx.x == 0) dst[p.C * KK] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// dw[col] (+)= sum_s part[s][col] (col < cols), dbias[0] (+)= sum_s part[s][cols]: 16 columns x 16 row groups per block,
// double accumulation, fixed combination order (the shape of hrv_common.h's sum_rows_block, rows cols + 1 apart)
__global__ __launch_bounds__(256) void cout1_reduce_kernel(const float* __restrict__ part, int S, int cols, float* __restrict__ dw,
                                                           int accumulate, float* __restrict__ dbias, int dbias_accumulate) {
  __shared__ double red[16][16];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s = 0.0;
  if (c <= cols)
    for (int r = rg; r < S; r += 16) s += (double)part[(size_t)r * (cols + 1) + c];
  red[rg][cl] = s;
  __syncthreads();
  if (threadIdx.x < 16 && c <= cols) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][cl];
    if (c < cols) dw[c] = accumulate ? dw[c] + (float)t : (float)t;
    else if (dbias) dbias[0] = dbias_accumulate ? dbias[0] + (float)t : (float)t;
  }
}

}  // namespace hrv

using namespace hrv;

static int c1_fill(const hrv_conv_cout1_t* d, Cout1Params& p, const char* who) {
  HRV_REQUIRE(d && d->x && d->w_oihw && d->y, "%s: null pointer", who);
  HRV_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->C % 4 == 0 && d->C <= 2048, "%s: extents (C %% 4 == 0, C <= 2048)", who);
  HRV_REQUIRE(d->K >= 1 && d->K <= 4 && d->pad >= 0 && d->pad < d->K, "%s: kernel size 1..4, stride 1", who);
  HRV_REQUIRE(d->x_cstride % 4 == 0 && d->x_coff % 4 == 0 && d->x_coff + d->C <= d->x_cstride && ((uintptr_t)d->x & 15) == 0,
              "%s: source slice (4-channel granules)", who);
  HRV_REQUIRE(d->y_coff >= 0 && d->y_coff < d->y_cstride, "%s: y channel", who);
  p.x = d->x; p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->C; p.xcs = d->x_cstride; p.xco = d->x_coff;
  p.w = d->w_oihw; p.sigma = d->sigma; p.wscale = d->wscale; p.bias = d->bias;
  p.K = d->K; p.pad = d->pad; p.Ho = d->H + 2 * d->pad - d->K + 1; p.Wo = d->W + 2 * d->pad - d->K + 1;
  p.y = d->y; p.ycs = d->y_cstride; p.yco = d->y_coff;
  p.dx = d->dx; p.dcs = d->dx_cstride; p.dco = d->dx_coff;
  p.add = d->add; p.acs = d->add_cstride; p.aco = d->add_coff;
  p.part = d->workspace; p.S = 0;
  p.round_bf16 = d->round_bf16 ? 1 : 0;
  HRV_REQUIRE(p.Ho > 0 && p.Wo > 0, "%s: empty output", who);
  return HRV_OK;
}

static int c1_lds(const void* fn, int bytes, const char* who) {
  if (bytes > 48 * 1024 &&
      hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) failed", who, bytes);
    return HRV_ERR_LAUNCH;
  }
  return HRV_OK;
}

extern "C" int hrv_conv_cout1_fwd_f32(const hrv_conv_cout1_t* d, hrv_stream_t stream) {
  Cout1Params p;
  int rc = c1_fill(d, p, "conv_cout1_fwd");
  if (rc) return rc;
  const int lds = 0;                 // (the weights sit in registers since round 6)
  const int strips = p.N * ((p.Ho + C1_ROWS - 1) / C1_ROWS) * ((p.Wo + C1_STRIP - 1) / C1_STRIP);
  int grid = (strips + 3) / 4;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(cout1_fwd_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
  return check_launch("cout1_fwd_kernel");
}

extern "C" int hrv_conv_cout1_dgrad_f32(const hrv_conv_cout1_t* d, hrv_stream_t stream) {
  Cout1Params p;
  int rc = c1_fill(d, p, "conv_cout1_dgrad");
  if (rc) return rc;
  HRV_REQUIRE(d->dx && d->dx_cstride % 4 == 0 && d->dx_coff % 4 == 0 && d->dx_coff + d->C <= d->dx_cstride &&
                  ((uintptr_t)d->dx & 15) == 0, "conv_cout1_dgrad: dx slice");
  HRV_REQUIRE(!d->add || (d->add_cstride % 4 == 0 && d->add_coff % 4 == 0 && ((uintptr_t)d->add & 15) == 0), "conv_cout1_dgrad: add slice");
  const int lds = (p.K * p.K * p.C + p.K * (p.Wo + 2 * p.K)) * 4;
  HRV_REQUIRE(lds <= 160 * 1024, "conv_cout1_dgrad: weights + dY rows exceed the LDS (%d bytes)", lds);
  rc = c1_lds(reinterpret_cast<const void*>(&cout1_dgrad_kernel), lds, "conv_cout1_dgrad");
  if (rc) return rc;
  int grid = p.N * p.H;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(cout1_dgrad_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, p);
  return check_launch("cout1_dgrad_kernel");
}

extern "C" int32_t hrv_conv_cout1_wgrad_slabs(int32_t N, int32_t Ho, int32_t Wo) {
  (void)Wo;
  return N * Ho;            // one partial row per INPUT row (N * H <= N * Ho of them are used)
}

// workspace: hrv_conv_cout1_wgrad_slabs() x (C*K*K + 1) floats; dw [1][C][K][K] and dbias [1] (+)= the slab sums
extern "C" int hrv_conv_cout1_wgrad_f32(const hrv_conv_cout1_t* d, float* dw, int32_t accumulate, float* dbias,
                                        int32_t dbias_accumulate, hrv_stream_t stream) {
  Cout1Params p;
  int rc = c1_fill(d, p, "conv_cout1_wgrad");
  if (rc) return rc;
  HRV_REQUIRE(d->workspace && dw, "conv_cout1_wgrad: null pointer");
  HRV_REQUIRE(p.H <= p.Ho, "conv_cout1_wgrad: pad < (K - 1) / 2 is not built (the bias rows are walked from row h on)");
  p.S = p.N * p.H;
  hipStream_t st = (hipStream_t)stream;
  const int lds = p.K * (p.Wo + 2 * p.K) * 4;
  HRV_REQUIRE(lds <= 64 * 1024, "conv_cout1_wgrad: dY rows exceed 64 KB of LDS");
  rc = c1_lds(reinterpret_cast<const void*>(&cout1_wgrad_kernel), lds, "conv_cout1_wgrad");
  if (rc) return rc;
  hipLaunchKernelGGL(cout1_wgrad_kernel, dim3(p.S, (p.C + 255) / 256), dim3(256), lds, st, p);
  rc = check_launch("cout1_wgrad_kernel");
  if (rc) return rc;
  const int cols = p.C * p.K * p.K;
  hipLaunchKernelGGL(cout1_reduce_kernel, dim3((cols + 1 + 15) / 16), dim3(256), 0, st, p.part, p.S, cols, dw, accumulate, dbias,
                     dbias_accumulate);
  return check_launch("cout1_reduce_kernel");
}
