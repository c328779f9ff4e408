// Training-side convolution kernels (NHWC fp32, gfx950 MFMA):
//   * device-side weight packing for the forward / data-gradient engine (weights change
//     every optimiser step, so the host packer is only for frozen/inference weights);
//   * weight gradient  dW[co][ci][kh][kw] = sum_p dY[p][co] * Xgather[p][kh,kw][ci]
//     as an MFMA GEMM whose reduction dimension is the pixel index, split over pixel
//     slabs with a fixed-order second stage (deterministic);
//   * bias gradient (column sums of dY).
// The data gradient itself is the forward engine (conv_f32.hip) run on dY with
// transposed / flipped (and, for stride 2, per-phase) weights packed here.
#include <stdlib.h>
#include <string.h>

#include "hrv_common.h"

namespace hrv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;

// ------------------------------------------------------------------ device pack
struct PackParams {
  const float* w;  // [Cout][CinTot][KH][KW]
  int Cout, CinTot, KH, KW;
  int transposed;  // 0: rows = cout, k runs over the sources' input channels (forward)
                   // 1: rows = cin,  k runs over cout (data gradient: one "source" of Cout channels)
  int nsrc;
  int src_cpad[HRV_MAX_SRC], src_creal[HRV_MAX_SRC], src_cbase[HRV_MAX_SRC], src_chunk0[HRV_MAX_SRC];
  int chunks_total;
  int KHp, KWp;          // packed tap grid
  int kh_of[8], kw_of[8];  // packed tap -> weight tap (or -1: absent)
  int rows, rows_pad;    // output rows (Cout or CinTot) and their padded count
  float wscale;
  const float* sigma;    // optional device scalar: weights are divided by sigma[0] (spectral norm)
  float* out;            // [KHp*KWp*chunks_total][rows_pad][bke]  (fp32, or bf16 when `bf16`)
  int bke;               // k-values per packed row: 16 (fp32 engine), 32 / 64 (bf16 engine, 64- / 128-byte rows)
  int bf16;
  // Weight PAIRS (the SPADE gamma / beta convolutions, network_generator.py:93-118, packed as ONE matrix without a
  // concatenated copy): the Cout axis of the virtual weight is built from w (gamma) and w2 (beta), rows_each couts each.
  //   pair_mode 1: interleaved 32|32 -- virtual cout 64g + l = gamma[32g + l] (l < 32) / beta[32g + l - 32]
  //                (the fused modulate epilogue wants gamma_c and beta_c in the same lane)
  //   pair_mode 2: concatenated  -- virtual cout c = gamma[c] (c < pair_split) / beta[c - pair_split]
  const float* w2;
  int pair_mode, rows_each, pair_split;
};

// virtual cout -> (weight pointer, real cout) of a pair; false: a padding row
__device__ __forceinline__ bool pack_pair_src(const PackParams& p, int co, const float*& wp, int& cr) {
  wp = p.w; cr = co;
  if (p.pair_mode == 1) {
    const int l = co & 63;
    cr = (co >> 6) * 32 + (l & 31);
    if (l >= 32) wp = p.w2;
    return cr < p.rows_each;
  }
  if (p.pair_mode == 2) {
    if (co >= p.pair_split) { wp = p.w2; cr = co - p.pair_split; }
    return cr < p.rows_each;
  }
  return true;
}

// one COLUMN of the packed matrix: the (row, k) position ``i`` (over [chunk][row][k]) for every tap.  A thread reads the
// KH*KW taps of its (cout, cin) pair -- contiguous in the OIHW parameter -- once, and writes one element per tap plane
// (consecutive lanes = consecutive k: coalesced).  (One element per thread re-fetched the same 32-byte sectors once per
// tap: the batched pack of the generator, 140 M elements, took 0.3 ms per launch = gather-bound.)
template <typename IDX>
__device__ __forceinline__ void pack_col(const PackParams& p, const IDX i, const float mul) {
  const int BKp = p.bke;                 // 16 / 32 / 64
  const int k = (int)(i & (IDX)(BKp - 1));
  const IDX t = i >> (31 - __clz(BKp));
  const int row = (int)(t % (IDX)p.rows_pad);
  const int chunk = (int)(t / (IDX)p.rows_pad);
  const float* src = nullptr;            // &w[co][ci][0][0], or null: a padding position
  if (row < p.rows) {
    int s = 0;
#pragma unroll
    for (int q = 1; q < HRV_MAX_SRC; ++q)
      if (q < p.nsrc && chunk >= p.src_chunk0[q]) s = q;
    const int c = (chunk - p.src_chunk0[s]) * BKp + k;  // channel within source s
    if (c < p.src_creal[s]) {
      const int cc = p.src_cbase[s] + c;
      const int co = p.transposed ? cc : row;
      const int ci = p.transposed ? row : cc;
      const float* wp;
      int cr;
      if (pack_pair_src(p, co, wp, cr)) src = wp + ((size_t)cr * p.CinTot + ci) * p.KH * p.KW;
    }
  }
  const size_t plane = (size_t)p.chunks_total * p.rows_pad * BKp;      // elements per tap
  size_t o = (size_t)i;
  for (int jh = 0; jh < p.KHp; ++jh) {
    const int kh = p.kh_of[jh];
    for (int jw = 0; jw < p.KWp; ++jw, o += plane) {
      const int kw = p.kw_of[jw];
      const float v = (src && kh >= 0 && kw >= 0) ? src[kh * p.KW + kw] * mul : 0.f;
      if (p.bf16) {  // round to nearest even
        unsigned u = __builtin_bit_cast(unsigned, v);
        u += 0x7fffu + ((u >> 16) & 1u);
        reinterpret_cast<unsigned short*>(p.out)[o] = (unsigned short)(u >> 16);
      } else {
        p.out[o] = v;
      }
    }
  }
}

// store 4 consecutive packed values (o % 4 == 0)
__device__ __forceinline__ void pack_store4(const PackParams& p, const size_t o, const float v0, const float v1, const float v2,
                                            const float v3) {
  if (p.bf16) {
    auto rn = [](float v) -> unsigned {  // round to nearest even
      unsigned u = __builtin_bit_cast(unsigned, v);
      u += 0x7fffu + ((u >> 16) & 1u);
      return u >> 16;
    };
    uint2 w;
    w.x = rn(v0) | (rn(v1) << 16);
    w.y = rn(v2) | (rn(v3) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.out) + o) = w;
  } else {
    *reinterpret_cast<float4*>(p.out + o) = make_float4(v0, v1, v2, v3);
  }
}

// pack_col for FOUR consecutive k (i % 4 == 0): a quarter of the memory instructions (2-byte stores and 4-byte loads made
// the batched pack issue-bound: 1.5 ms per iteration for 1.3 GB).  Forward 3x3 packs read their 4 x 9 source floats as nine
// 16-byte loads (consecutive cin of one cout are contiguous in OIHW).
template <typename IDX>
__device__ __forceinline__ void pack_col4(const PackParams& p, const IDX i, const float mul) {
  const int BKp = p.bke;
  const int k = (int)(i & (IDX)(BKp - 1));
  const IDX t = i >> (31 - __clz(BKp));
  const int row = (int)(t % (IDX)p.rows_pad);
  const int chunk = (int)(t / (IDX)p.rows_pad);
  const int T = p.KH * p.KW;
  const float* src[4] = {nullptr, nullptr, nullptr, nullptr};
  if (row < p.rows) {
    int s = 0;
#pragma unroll
    for (int q = 1; q < HRV_MAX_SRC; ++q)
      if (q < p.nsrc && chunk >= p.src_chunk0[q]) s = q;
    const int c0 = (chunk - p.src_chunk0[s]) * BKp + k;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + j;
      if (c < p.src_creal[s]) {
        const int cc = p.src_cbase[s] + c;
        const int co = p.transposed ? cc : row;
        const int ci = p.transposed ? row : cc;
        const float* wp;
        int cr;
        if (pack_pair_src(p, co, wp, cr)) src[j] = wp + ((size_t)cr * p.CinTot + ci) * T;
      }
    }
  }
  const size_t plane = (size_t)p.chunks_total * p.rows_pad * BKp;
  size_t o = (size_t)i;
  if (!p.transposed && T == 9 && p.KHp == 3 && p.KWp == 3 && src[0] && src[3] == src[0] + 27 &&
      ((uintptr_t)src[0] & 15) == 0) {
    // (forward packs keep the tap order: packed tap q = source tap q)
    float f[36];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const float4 v = reinterpret_cast<const float4*>(src[0])[q];
      f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q, o += plane) pack_store4(p, o, f[q] * mul, f[9 + q] * mul, f[18 + q] * mul, f[27 + q] * mul);
    return;
  }
  for (int jh = 0; jh < p.KHp; ++jh) {
    const int kh = p.kh_of[jh];
    for (int jw = 0; jw < p.KWp; ++jw, o += plane) {
      const int kw = p.kw_of[jw];
      const bool ok = kh >= 0 && kw >= 0;
      const int tap = kh * p.KW + kw;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (ok && src[j]) ? src[j][tap] * mul : 0.f;
      pack_store4(p, o, v[0], v[1], v[2], v[3]);
    }
  }
}

__global__ void pack_weight_kernel(const PackParams p) {
  const size_t cols = (size_t)p.chunks_total * p.rows_pad * p.bke;
  const float mul = p.sigma ? p.wscale / p.sigma[0] : p.wscale;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cols; i += (size_t)gridDim.x * blockDim.x)
    pack_col(p, i, mul);
}

// Every weight pack of a network in ONE launch (a training step re-packs ~180 weights after each optimizer step: one
// 5-11 us launch each otherwise).  ``tbl``: the records hrv_conv2d_pack_weight_record filled, on the device; ``first``:
// [n + 1] first block of each record (PACK_MULTI_ELEMS columns per block).  Same arithmetic per element as above.
constexpr int PACK_MULTI_ELEMS = 1024;
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const PackParams* __restrict__ tbl,
                                                                const int* __restrict__ first, const int n) {
  const int b = blockIdx.x;
  int lo = 0, hi = n;                 // first[lo] <= b < first[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (first[mid] <= b) lo = mid; else hi = mid;
  }
  // the record moves to LDS: read through the table pointer, every field would be re-loaded from global memory after each
  // store (the compiler cannot rule out that the packed output aliases the table) -- first build: 0.94 ms per launch
  __shared__ PackParams sp;
  static_assert(sizeof(PackParams) % 4 == 0, "record is copied in dwords");
  for (int k = threadIdx.x; k < (int)(sizeof(PackParams) / 4); k += 256)
    reinterpret_cast<unsigned*>(&sp)[k] = reinterpret_cast<const unsigned*>(tbl + lo)[k];
  __syncthreads();
  const unsigned cols = (unsigned)((size_t)sp.chunks_total * sp.rows_pad * sp.bke);   // < 2^30 (host check)
  const float mul = sp.sigma ? sp.wscale / sp.sigma[0] : sp.wscale;
  const int T = sp.KH * sp.KW;
  if (sp.transposed && T <= 9 && (sp.rows_pad & (PACK_MULTI_ELEMS / sp.bke - 1)) == 0) {
    // Data-gradient packs (rows = cin, k = cout): a column-per-thread walk reads 36 bytes every CinTot*36 bytes -- gather-
    // bound.  Here a block owns RB = 1024 / bke consecutive rows x the bke couts of one chunk: for each cout the RB*T source
    // floats are CONTIGUOUS (w[co][row0 .. row0+RB)[taps]) -> coalesced reads into LDS, then one coalesced 2*bke-byte run
    // of the packed matrix per (tap, row).
    __shared__ float tile[1024 * 9 + 64];
    const int bke = sp.bke, RB = PACK_MULTI_ELEMS / bke, seg = RB * T, ld = seg + 1;
    const unsigned t0 = (unsigned)(b - first[lo]) * (unsigned)RB;          // first (chunk, row) of the block
    const int chunk = (int)(t0 / (unsigned)sp.rows_pad), row0 = (int)(t0 - (unsigned)chunk * sp.rows_pad);
    for (int idx = threadIdx.x; idx < bke * seg; idx += 256) {
      const int k = idx / seg, j = idx - k * seg;
      const int cc = chunk * bke + k;                                      // cout (virtual, for a pair)
      const int rl = j / T;
      float v = 0.f;
      const float* wp;
      int cr;
      if (cc < sp.src_creal[0] && row0 + rl < sp.rows && pack_pair_src(sp, cc, wp, cr))
        v = wp[((size_t)cr * sp.CinTot + row0) * T + j];
      tile[k * ld + j] = v;
    }
    __syncthreads();
    const size_t plane = (size_t)sp.chunks_total * sp.rows_pad * bke;
    const int kq = bke / 4;                                   // 4 consecutive k per thread: 8- / 16-byte stores
    const int k = (threadIdx.x % kq) * 4, rstep = 256 / kq;
    for (int jh = 0; jh < sp.KHp; ++jh) {
      const int kh = sp.kh_of[jh];
      for (int jw = 0; jw < sp.KWp; ++jw) {
        const int kw = sp.kw_of[jw];
        const int tap = (kh >= 0 && kw >= 0) ? kh * sp.KW + kw : -1;
        const size_t o0 = (size_t)(jh * sp.KWp + jw) * plane + ((size_t)chunk * sp.rows_pad + row0) * bke;
        for (int rl = threadIdx.x / kq; rl < RB; rl += rstep) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = tap >= 0 ? tile[(k + j) * ld + rl * T + tap] * mul : 0.f;
          pack_store4(sp, o0 + (size_t)rl * bke + k, v[0], v[1], v[2], v[3]);
        }
      }
    }
    return;
  }
  const unsigned i0 = ((unsigned)(b - first[lo]) * PACK_MULTI_ELEMS + threadIdx.x * 4);
  if (i0 < cols) pack_col4(sp, i0, mul);                      // (bke % 4 == 0: the four k share chunk and row)
}

// ------------------------------------------------------------------ wgrad
struct WgradParams {
  const float* dy;
  int dy_cs, dy_co, Cout;
  const float* x;
  int x_C, x_cs, x_co, x_up;  // source channels (padded to 4), stride, offset, nearest shift
  int N, H, W, Ho, Wo, KH, KW, stride, pad;
  int P;            // N*Ho*Wo
  int ci_base;      // position of this source on the weight's Cin axis
  int ci_real;      // real channels of this source (<= x_C)
  int CinTot;
  int co_tiles, ci_tiles, taps, S;
  float* ws;        // [S][taps][Cout][CinTot]  (only this source's ci range is written)
  int x_bf16;       // bf16-MMA kernel only: X is stored as bf16 (a tensor only matrix cores read: same rounding, half the bytes)
  float* bias_ws;   // [S][Cout] partial column sums of dY (bias gradient), or null: computed as one extra
                    // "ones" column right after the (tap, ci) columns -- same MFMAs, same fixed reduction order
};

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_wgrad_mfma_kernel(const WgradParams p) {
  constexpr int BMc = 32 * TM * WM;  // couts per block
  constexpr int BNc = 32 * TN * WN;  // cins per block
  constexpr int LY = BMc + 4, LX = BNc + 4;
  constexpr int YR = (BK * BMc / 4 + 255) / 256;  // float4 loads of dY per thread per pixel tile
  constexpr int XR = (BK * BNc / 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LY + LX)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  // Logical block id: slab-major (all tiles of one pixel slab are neighbours) and XCD-contiguous, so the
  // blocks that stream the same dY / X pixels run at the same time behind the same L2.
  int b = xcd_remap(blockIdx.x, p.co_tiles * p.ci_tiles * p.S);
  const int it = b % p.ci_tiles; b /= p.ci_tiles;
  const int ct = b % p.co_tiles;
  const int s = b / p.co_tiles;
  const int co0 = ct * BMc, col0 = it * BNc;   // columns = flattened (tap, ci): col = tap * x_C + ci

  const int ptiles = (p.P + BK - 1) / BK;
  const int t_begin = (int)(((long long)ptiles * s) / p.S);
  const int t_end = (int)(((long long)ptiles * (s + 1)) / p.S);

  const int sr = p.x_up > 0 ? p.x_up : 0, sl = p.x_up < 0 ? -p.x_up : 0;
  const int Hs = (p.H >> sr) << sl, Ws = (p.W >> sr) << sl;

  // every X load of a thread has the same column group (256 % (BNc/4) == 0) -> one (tap, ci) per thread
  static_assert(256 % (BNc / 4) == 0, "column group of a thread must not depend on the load index");
  const int my_col = col0 + (tid % (BNc / 4)) * 4;
  const int NT = p.taps * p.x_C;
  const bool ones_col = p.bias_ws != nullptr && my_col == NT;   // this thread's 4 columns are (1, 0, 0, 0)
  const bool col_ok = my_col < NT;
  const int my_tap = col_ok ? my_col / p.x_C : 0;
  const int my_ci = col_ok ? my_col - my_tap * p.x_C : 0;
  const int kh = my_tap / p.KW, kw = my_tap - kh * p.KW;

  f32x4 yreg[YR], xreg[XR];
  // per-thread gather rows: pixel (n, ho, wo) of X-row r at the current tile, advanced by BK
  // pixels per tile without integer division
  int xr_n[XR], xr_ho[XR], xr_wo[XR];
#pragma unroll
  for (int r = 0; r < XR; ++r) {
    const int idx = tid + 256 * r;
    const int row = idx / (BNc / 4);
    const int pix = t_begin * BK + row;
    const int pp = pix < p.P ? pix : 0;
    xr_n[r] = pp / (p.Ho * p.Wo);
    const int rem = pp - xr_n[r] * (p.Ho * p.Wo);
    xr_ho[r] = rem / p.Wo;
    xr_wo[r] = rem - xr_ho[r] * p.Wo;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#define WG_LOAD(T)                                                                                        \
  {                                                                                                       \
    _Pragma("unroll") for (int r = 0; r < YR; ++r) {                                                      \
      const int idx = tid + 256 * r;                                                                      \
      const int row = idx / (BMc / 4), c4 = idx - row * (BMc / 4);                                        \
      const int pix = (T)*BK + row;                                                                       \
      const int c = co0 + c4 * 4;                                                                         \
      const bool ok = idx < BK * BMc / 4 && pix < p.P && c < p.Cout;                                      \
      const size_t off = ok ? (size_t)pix * p.dy_cs + p.dy_co + c : (size_t)p.dy_co;                      \
      f32x4 v = *reinterpret_cast<const f32x4*>(p.dy + off);                                              \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = (ok && c + e < p.Cout) ? v[e] : 0.f;           \
      yreg[r] = v;                                                                                        \
    }                                                                                                     \
    _Pragma("unroll") for (int r = 0; r < XR; ++r) {                                                      \
      const int idx = tid + 256 * r;                                                                      \
      const int row = idx / (BNc / 4);                                                                    \
      const int pix = (T)*BK + row;                                                                       \
      const int n = xr_n[r], ho = xr_ho[r], wo = xr_wo[r];                                                \
      const int hi = ho * p.stride - p.pad + kh, wi = wo * p.stride - p.pad + kw;                         \
      const bool ok = idx < BK * BNc / 4 && pix < p.P && col_ok && (unsigned)hi < (unsigned)p.H &&        \
                      (unsigned)wi < (unsigned)p.W;                                                       \
      const int hic = min(max(hi, 0), p.H - 1), wic = min(max(wi, 0), p.W - 1);                           \
      const size_t off = ((size_t)(n * Hs + ((hic >> sr) << sl)) * Ws + ((wic >> sr) << sl)) * p.x_cs +   \
                         p.x_co + my_ci;                                                                  \
      f32x4 v = *reinterpret_cast<const f32x4*>(p.x + off);                                               \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;                               \
      if (ones_col) v[0] = (idx < BK * BNc / 4 && pix < p.P) ? 1.f : 0.f;                                 \
      xreg[r] = v;                                                                                        \
      /* advance this row by BK pixels for the next tile */                                               \
      int w2 = wo + BK, h2 = ho, n2 = n;                                                                  \
      while (w2 >= p.Wo) { w2 -= p.Wo; ++h2; }                                                            \
      while (h2 >= p.Ho) { h2 -= p.Ho; ++n2; }                                                            \
      xr_wo[r] = w2; xr_ho[r] = h2; xr_n[r] = n2 < p.N ? n2 : 0;                                          \
    }                                                                                                     \
  }

#define WG_STORE(BUF)                                                                       \
  {                                                                                         \
    float* Ys = smem + (BUF)*BK * (LY + LX);                                                \
    float* Xs = Ys + BK * LY;                                                               \
    _Pragma("unroll") for (int r = 0; r < YR; ++r) {                                        \
      const int idx = tid + 256 * r;                                                        \
      const int row = idx / (BMc / 4), c4 = idx - row * (BMc / 4);                          \
      if (idx < BK * BMc / 4) *reinterpret_cast<f32x4*>(Ys + row * LY + c4 * 4) = yreg[r];  \
    }                                                                                       \
    _Pragma("unroll") for (int r = 0; r < XR; ++r) {                                        \
      const int idx = tid + 256 * r;                                                        \
      const int row = idx / (BNc / 4), c4 = idx - row * (BNc / 4);                          \
      if (idx < BK * BNc / 4) *reinterpret_cast<f32x4*>(Xs + row * LX + c4 * 4) = xreg[r];  \
    }                                                                                       \
  }

#define WG_MMA(BUF)                                                                                   \
  {                                                                                                   \
    const float* Ys = smem + (BUF)*BK * (LY + LX) + wm * TM * 32 + l31;                               \
    const float* Xs = smem + (BUF)*BK * (LY + LX) + BK * LY + wn * TN * 32 + l31;                     \
    _Pragma("unroll") for (int kk = 0; kk < BK / 2; ++kk) {                                           \
      float a[TM], bb[TN];                                                                            \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) a[i] = Ys[(2 * kk + lh) * LY + i * 32];          \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) bb[j] = Xs[(2 * kk + lh) * LX + j * 32];         \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                              \
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);      \
    }                                                                                                 \
  }

  if (t_begin < t_end) {
    WG_LOAD(t_begin)
    WG_STORE(t_begin & 1)
    __syncthreads();
    for (int t = t_begin; t < t_end - 1; ++t) {
      WG_LOAD(t + 1)
      WG_MMA(t & 1)
      WG_STORE((t + 1) & 1)
      __syncthreads();
    }
    WG_MMA((t_end - 1) & 1)
  }
#undef WG_LOAD
#undef WG_STORE
#undef WG_MMA

  // D[i = cout][j = cin]: col = lane&31 (cin), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (cout)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + (wn * TN + j) * 32 + l31;
    if (col == p.taps * p.x_C && p.bias_ws) {   // the ones column: sum over this slab's pixels of dY[:, co]
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int co = co0 + (wm * TM + i) * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
          if (co < p.Cout) p.bias_ws[(size_t)s * p.Cout + co] = acc[i][j][e];
        }
      continue;
    }
    if (col >= p.taps * p.x_C) continue;
    const int tap = col / p.x_C, ci = col - tap * p.x_C;
    if (ci >= p.ci_real) continue;
    float* wsp = p.ws + ((size_t)s * p.taps + tap) * p.Cout * p.CinTot;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + (wm * TM + i) * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
        if (co < p.Cout) wsp[(size_t)co * p.CinTot + p.ci_base + ci] = acc[i][j][e];
      }
  }
}

// ------------------------------------------------------------------ wgrad, bf16 matrix cores over fp32 tensors
// Mixed-precision training: dY and X stay fp32 in HBM; they are rounded to bf16 (v_cvt_pk_bf16_f32) while being
// staged and multiplied by v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  That MFMA wants 8 consecutive k-values
// (= pixels here) per lane, but NHWC is pixel-major, so each load task covers a QUAD of 4 consecutive pixels x 4
// channels: the 4x4 block is transposed in registers for free and every channel's 4 pixels go to LDS as one
// 8-byte store into a [channel][pixel] image (row = 32 pixels = 64 B + 16 B pad -> conflict-free b128 fragment
// reads).  Needs Wo % 4 == 0 (a quad never straddles an image row); the caller falls back to the fp32 kernel
// otherwise (the odd-sized PatchGAN maps).
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2w __attribute__((ext_vector_type(2)));
typedef float f32x2w __attribute__((ext_vector_type(2)));
constexpr int BKP = 32;      // pixels per K-tile
constexpr int LKP = 40;      // LDS row stride in bf16 elements (32 + 8 pad)

__device__ __forceinline__ f32x4 widen_bf16x4(uint2 u) {   // 4 stored bf16 -> the same values as fp32
  f32x4 v;
  v[0] = __builtin_bit_cast(float, u.x << 16); v[1] = __builtin_bit_cast(float, u.x & 0xFFFF0000u);
  v[2] = __builtin_bit_cast(float, u.y << 16); v[3] = __builtin_bit_cast(float, u.y & 0xFFFF0000u);
  return v;
}

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const f32x2w v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2w));
}

template <int TM, int TN, int WM, int WN, bool XB = false, bool YB = false>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(const WgradParams p) {
  constexpr int BMc = 32 * TM * WM;  // couts per block
  constexpr int BNc = 32 * TN * WN;  // (tap, cin) columns per block
  // 4 channels per load task: 16 bytes of fp32, or 8 bytes of a bf16-STORED operand (XB / YB) widened in
  // registers.  (Measured: 8-channel / 16-byte packed tasks with a v_perm_b32 transpose leave half the block's
  // threads without staging work and run 25 % SLOWER; see DESIGN.md 6c for the PMC picture of this kernel.)
  constexpr int YC = 4, XC = 4;
  constexpr int YG = BMc / YC, XG = BNc / XC;           // channel groups per pixel
  constexpr int YT = (8 * YG + 255) / 256;              // quad tasks per thread per K-tile
  constexpr int XT = (8 * XG + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * (BMc + BNc) * LKP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  int b = xcd_remap(blockIdx.x, p.co_tiles * p.ci_tiles * p.S);
  const int it = b % p.ci_tiles; b /= p.ci_tiles;
  const int ct = b % p.co_tiles;
  const int s = b / p.co_tiles;
  const int co0 = ct * BMc, col0 = it * BNc;

  const int ptiles = (p.P + BKP - 1) / BKP;
  const int t_begin = (int)(((long long)ptiles * s) / p.S);
  const int t_end = (int)(((long long)ptiles * (s + 1)) / p.S);
  const int sr = p.x_up > 0 ? p.x_up : 0, sl = p.x_up < 0 ? -p.x_up : 0;
  const int Hs = (p.H >> sr) << sl, Ws = (p.W >> sr) << sl;

  // X tasks: task = (quad q in 0..7, column group g); every task of a thread has its own (tap, ci)
  int xq[XT], xtap_kh[XT], xtap_kw[XT], xci[XT], xn[XT], xho[XT], xwo[XT];
  bool xcol_ok[XT], xones[XT];
#pragma unroll
  for (int r = 0; r < XT; ++r) {
    const int task = tid + 256 * r; 
    const int g = task % XG;
    xq[r] = task / XG;
    const int col = col0 + g * XC;
    xones[r] = p.bias_ws != nullptr && task < 8 * XG && col == p.taps * p.x_C;
    xcol_ok[r] = task < 8 * XG && col < p.taps * p.x_C;
    const int tap = xcol_ok[r] ? col / p.x_C : 0;
    xci[r] = xcol_ok[r] ? col - tap * p.x_C : 0;
    xtap_kh[r] = tap / p.KW;
    xtap_kw[r] = tap - xtap_kh[r] * p.KW;
    const int pix = t_begin * BKP + xq[r] * 4;
    const int pp = pix < p.P ? pix : 0;
    xn[r] = pp / (p.Ho * p.Wo);
    const int rem = pp - xn[r] * (p.Ho * p.Wo);
    xho[r] = rem / p.Wo;
    xwo[r] = rem - xho[r] * p.Wo;
  }
  // staging registers: [task][pixel of the quad] = one 16-byte load (4 fp32 or 8 packed bf16 channels)
  f32x4 yreg[YT][4], xreg[XT][4];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#define WB_LOAD(T)                                                                                        \
  {                                                                                                       \
    _Pragma("unroll") for (int r = 0; r < YT; ++r) {                                                      \
      const int task = tid + 256 * r;                                                                     \
      const int g = task % YG, q = task / YG;                                                             \
      const int c = co0 + g * YC;                                                                         \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
        const int pix = (T)*BKP + q * 4 + j;                                                              \
        const bool ok = task < 8 * YG && pix < p.P && c < p.Cout;                                         \
        const size_t off = ok ? (size_t)pix * p.dy_cs + p.dy_co + c : (size_t)p.dy_co;                    \
        f32x4 v;                                                                                          \
        if constexpr (YB) {                                                                               \
          v = widen_bf16x4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.dy) + off)); \
        } else {                                                                                          \
          v = *reinterpret_cast<const f32x4*>(p.dy + off);                                                \
        }                                                                                                 \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = (ok && c + e < p.Cout) ? v[e] : 0.f;         \
        yreg[r][j] = v;                                                                                   \
      }                                                                                                   \
    }                                                                                                     \
    _Pragma("unroll") for (int r = 0; r < XT; ++r) {                                                      \
      const int n = xn[r], ho = xho[r], wo = xwo[r];                                                      \
      const int hi = ho * p.stride - p.pad + xtap_kh[r];                                                  \
      const bool row_ok = xcol_ok[r] && (unsigned)hi < (unsigned)p.H;                                     \
      const int hic = min(max(hi, 0), p.H - 1);                                                           \
      const size_t rowoff = (size_t)(n * Hs + ((hic >> sr) << sl)) * Ws;                                  \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
        const int pix = (T)*BKP + xq[r] * 4 + j;                                                          \
        const int wi = (wo + j) * p.stride - p.pad + xtap_kw[r];                                          \
        const bool ok = row_ok && pix < p.P && (unsigned)wi < (unsigned)p.W;                              \
        const int wic = min(max(wi, 0), p.W - 1);                                                         \
        const size_t off = (rowoff + ((wic >> sr) << sl)) * p.x_cs + p.x_co + xci[r];                     \
        f32x4 v;                                                                                          \
        if constexpr (XB) {                                                                               \
          v = widen_bf16x4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.x) + off)); \
        } else {                                                                                          \
          v = *reinterpret_cast<const f32x4*>(p.x + off);                                                 \
        }                                                                                                 \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;                             \
        if (xones[r]) v[0] = pix < p.P ? 1.f : 0.f;                                                       \
        xreg[r][j] = v;                                                                                   \
      }                                                                                                   \
      /* advance the quad by BKP pixels for the next tile (Wo % 4 == 0: it stays inside one image row) */ \
      int w2 = wo + BKP, h2 = ho, n2 = n;                                                                 \
      while (w2 >= p.Wo) { w2 -= p.Wo; ++h2; }                                                            \
      while (h2 >= p.Ho) { h2 -= p.Ho; ++n2; }                                                            \
      xwo[r] = w2; xho[r] = h2; xn[r] = n2 < p.N ? n2 : 0;                                                \
    }                                                                                                     \
  }

// one operand's quad (4 pixels x CG channels, REG[j] = pixel j) -> LDS: channel-major rows, the quad's 4 pixels
// of one channel as one 8-byte store
// LDS image: row = channel, 32 pixels (64 B) + 16 B pad.  The quad slot inside a row is XOR-swizzled with
// 2*((row >> 4) & 3): without it the 32 channel groups of a wave hit 4 bank groups with their 8-byte stores
// (8-way conflict: SQ_LDS_BANK_CONFLICT was 76 % of SQ_LDS_IDX_ACTIVE, profiles/r01_pmc_wgrad_bf16.txt); with it
// 2-way, the floor for a 512-byte wave store (270 -> 317 TFLOP/s on the gamma|beta shape).  The swizzle is even,
// so the two quads of one b128 fragment read stay adjacent and in order.
#define WB_STORE_QUAD(REG, DST, CH0, Q)                                                                   \
  {                                                                                                       \
    const int qs = ((Q) ^ ((((CH0) >> 4) & 3) * 2)) * 4;   /* CH0 % 4 == 0: all 4 rows share the swizzle */ \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                       \
      uint2 w2;                                                                                           \
      w2.x = pk_bf16(REG[0][e], REG[1][e]);                                                               \
      w2.y = pk_bf16(REG[2][e], REG[3][e]);                                                               \
      *reinterpret_cast<uint2*>(DST + ((CH0) + e) * LKP + qs) = w2;                                       \
    }                                                                                                     \
  }
#define WB_STORE(BUF)                                                                                     \
  {                                                                                                       \
    unsigned short* Ys = smem + (BUF) * (BMc + BNc) * LKP;                                                \
    unsigned short* Xs = Ys + BMc * LKP;                                                                  \
    _Pragma("unroll") for (int r = 0; r < YT; ++r) {                                                      \
      const int task = tid + 256 * r;                                                                     \
      const int g = task % YG, q = task / YG;                                                             \
      if (task < 8 * YG) WB_STORE_QUAD(yreg[r], Ys, g * YC, q)                                        \
    }                                                                                                     \
    _Pragma("unroll") for (int r = 0; r < XT; ++r) {                                                      \
      const int task = tid + 256 * r;                                                                     \
      const int g = task % XG;                                                                            \
      if (task < 8 * XG) WB_STORE_QUAD(xreg[r], Xs, g * XC, xq[r])                                    \
    }                                                                                                     \
  }

#define WB_MMA(BUF)                                                                                       \
  {                                                                                                       \
    const int yrow = wm * TM * 32 + l31, xrow = wn * TN * 32 + l31;                                       \
    const unsigned short* Ys = smem + (BUF) * (BMc + BNc) * LKP + yrow * LKP;                             \
    const unsigned short* Xs = smem + (BUF) * (BMc + BNc) * LKP + BMc * LKP + xrow * LKP;                 \
    _Pragma("unroll") for (int ks = 0; ks < BKP / 16; ++ks) {                                             \
      bf16x8w a[TM], bb[TN];                                                                              \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                    \
        const int sw = (((yrow + i * 32) >> 4) & 3) * 2;                                                  \
        a[i] = *reinterpret_cast<const bf16x8w*>(Ys + i * 32 * LKP + ((lh * 2 + ks * 4) ^ sw) * 4);       \
      }                                                                                                   \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                    \
        const int sw = (((xrow + j * 32) >> 4) & 3) * 2;                                                  \
        bb[j] = *reinterpret_cast<const bf16x8w*>(Xs + j * 32 * LKP + ((lh * 2 + ks * 4) ^ sw) * 4);      \
      }                                                                                                   \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                      \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                  \
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bb[j], acc[i][j], 0, 0, 0);       \
    }                                                                                                     \
  }

  if (t_begin < t_end) {
    WB_LOAD(t_begin)
    WB_STORE(t_begin & 1)
    __syncthreads();
    for (int t = t_begin; t < t_end - 1; ++t) {
      WB_LOAD(t + 1)
      WB_MMA(t & 1)
      WB_STORE((t + 1) & 1)
      __syncthreads();
    }
    WB_MMA((t_end - 1) & 1)
  }
#undef WB_LOAD
#undef WB_STORE
#undef WB_STORE_QUAD
#undef WB_MMA

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + (wn * TN + j) * 32 + l31;
    if (col == p.taps * p.x_C && p.bias_ws) {   // the ones column: bias-gradient partial of this slab
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int co = co0 + (wm * TM + i) * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
          if (co < p.Cout) p.bias_ws[(size_t)s * p.Cout + co] = acc[i][j][e];
        }
      continue;
    }
    if (col >= p.taps * p.x_C) continue;
    const int tap = col / p.x_C, ci = col - tap * p.x_C;
    if (ci >= p.ci_real) continue;
    float* wsp = p.ws + ((size_t)s * p.taps + tap) * p.Cout * p.CinTot;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + (wm * TM + i) * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
        if (co < p.Cout) wsp[(size_t)co * p.CinTot + p.ci_base + ci] = acc[i][j][e];
      }
  }
}

// dW[co][ci][tap] (torch OIHW) (+)= sum_s ws[s][tap][co][ci], for ci in [ci_base, ci_base+ci_real)
// Blocks [nb_main, gridDim.x) (when bias_ws is given) sum the bias partials bias_ws[s][co] instead -- one launch for both
// reductions of a weight gradient (71 + 83 launches per training iteration as two).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int S, int taps, int Cout, int CinTot, int ci_base,
                                    int ci_real, float* __restrict__ dw, int accumulate, int nb_main,
                                    const float* __restrict__ bias_ws, float* __restrict__ dbias, int dbias_accumulate) {
  if ((int)blockIdx.x >= nb_main) {
    sum_rows_block(blockIdx.x - nb_main, bias_ws, S, Cout, dbias, dbias_accumulate);
    return;
  }
  const size_t total = (size_t)Cout * ci_real * taps;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)nb_main * blockDim.x) {
    const int ci = (int)(i % ci_real);
    const size_t t = i / ci_real;
    const int tap = (int)(t % taps);
    const int co = (int)(t / taps);
    // eight independent loads in flight (the slab partials are megabytes apart: one load per iteration is pure
    // latency), summed in slab order -- the result does not depend on the unrolling
    const float* src = ws + ((size_t)tap * Cout + co) * CinTot + ci_base + ci;
    const size_t sstride = (size_t)taps * Cout * CinTot;
    float sum = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s + u) * sstride];
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; s < S; ++s) sum += src[(size_t)s * sstride];
    float* dst = dw + ((size_t)co * CinTot + ci_base + ci) * taps + tap;
    *dst = accumulate ? *dst + sum : sum;
  }
}

// (the bias partials bias_ws[s][co] are summed by the tail blocks of wgrad_reduce_kernel: sum_rows_block, hrv_common.h)

// ------------------------------------------------------------------ column sums (bias gradient)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int P, int C4, int cs, int co,
                                                             int NB, float* __restrict__ part) {
  __shared__ f32x4 red[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const int PB = (P + NB - 1) / NB;
  const int p0 = b * PB, p1 = min(p0 + PB, P);
  const int GB = C4 < 256 ? C4 : 256;
  const int R = 256 / GB;
  const int r = t / GB, gl = t - r * GB;
  for (int g0 = 0; g0 < C4; g0 += GB) {
    const int g = g0 + gl;
    f32x4 s1 = (f32x4)(0.f);
    if (r < R && g < C4)
      for (int px = p0 + r; px < p1; px += R) s1 += *reinterpret_cast<const f32x4*>(x + (size_t)px * cs + co + g * 4);
    red[t] = s1;
    __syncthreads();
    if (r == 0 && g < C4) {
      for (int rr = 1; rr < R; ++rr) s1 += red[rr * GB + gl];
      *reinterpret_cast<f32x4*>(part + ((size_t)b * C4 + g) * 4) = s1;
    }
    __syncthreads();
  }
}

__global__ void colsum_final_kernel(const float* __restrict__ part, int NB, int C, int Cpad, float* __restrict__ out,
                                    int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int b = 0; b < NB; ++b) s += (double)part[(size_t)b * Cpad + c];
  out[c] = accumulate ? out[c] + (float)s : (float)s;
}

static inline int grid_for(size_t work, int block = 256) {
  size_t g = (work + block - 1) / block;
  const size_t cap = 256 * 16;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

struct WTile { int TM, TN, WM, WN; };
// wgrad tiles (couts x cins per block).  Family A: all 4 waves share the dY fragments, each owns 32
// cins (BNc = 128, BMc = 32*TM exactly fits Cout up to 192).  Family B: cin tile 32 for tiny Cin.
static const WTile kWTiles[] = {
    {1, 1, 1, 4}, {2, 1, 1, 4}, {3, 1, 1, 4}, {4, 1, 1, 4}, {5, 1, 1, 4}, {6, 1, 1, 4},  // 0-5: 32..192 x 128
    {2, 2, 2, 2},                                                                          // 6: 128 x 128
    {1, 1, 4, 1}, {2, 1, 4, 1},                                                            // 7-8: 128/256 x 32
    {1, 1, 2, 2},                                                                          // 9: 64 x 64
};
static int wt_bm(int i) { return 32 * kWTiles[i].TM * kWTiles[i].WM; }
static int wt_bn(int i) { return 32 * kWTiles[i].TN * kWTiles[i].WN; }

static int pick_wtile(int Cout, int Cin) {
  if (Cin <= 32) return Cout > 128 ? 8 : 7;
  if (Cout <= 64 && Cin <= 64) return 9;
  const int tm = (Cout + 31) / 32;
  if (tm <= 6) return tm - 1;           // exact cout fit, no padded rows
  if (Cout % 128 == 0) return 6;
  // large Cout: the family-A tile with the least padding
  int best = 3, best_pad = 1 << 30;
  for (int t = 3; t <= 6; ++t) {
    const int bm = 32 * t, pad = (Cout + bm - 1) / bm * bm - Cout;
    if (pad <= best_pad) { best_pad = pad; best = t - 1; }
  }
  return best;
}

}  // namespace hrv

using namespace hrv;

static int pack_weight_dev_impl(const float* w_oihw_dev, int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc,
                                const int32_t* srcC, const int32_t* srcC_real, int32_t tile_cfg, int32_t mode,
                                int32_t stride, int32_t pad, int32_t phase_a, int32_t phase_b, float wscale,
                                const float* sigma_dev, void* out_dev, int32_t* out_geom, hrv_stream_t stream,
                                const int BK, const int as_bf16, const float* w2_dev = nullptr, int pair_mode = 0,
                                int rows_each = 0, PackParams* record = nullptr) {
  HRV_REQUIRE(w_oihw_dev && out_dev && srcC && srcC_real && nsrc >= 1 && nsrc <= HRV_MAX_SRC, "pack_dev: bad args");
  const int bn = hrv_conv2d_tile_bn(tile_cfg);
  HRV_REQUIRE(bn > 0, "pack_dev: bad tile_cfg %d", tile_cfg);
  HRV_REQUIRE(mode >= 0 && mode <= 2, "pack_dev: mode must be 0 (forward), 1 (dgrad stride 1), 2 (dgrad stride-2 phase)");
  HRV_REQUIRE(KH <= 8 && KW <= 8, "pack_dev: kernel too large");
  PackParams p;
  memset(&p, 0, sizeof(p));
  p.w = w_oihw_dev; p.Cout = Cout; p.KH = KH; p.KW = KW; p.wscale = wscale; p.sigma = sigma_dev; p.out = (float*)out_dev;
  p.bke = BK; p.bf16 = as_bf16;
  p.w2 = w2_dev; p.pair_mode = pair_mode; p.rows_each = rows_each;
  HRV_REQUIRE(pair_mode == 0 || (w2_dev && rows_each > 0 && ((pair_mode == 1 && mode == 0 && Cout == (rows_each + 31) / 32 * 64) ||
                                                              (pair_mode == 2 && mode == 1 && Cout % 2 == 0 && rows_each <= Cout / 2))),
              "pack_dev: weight pair (mode 1: forward rows interleaved 32|32, Cout = 64*ceil(rows_each/32); mode 2: data "
              "gradient over [gamma | beta] halves of Cout)");
  p.pair_split = Cout / 2;
  int cin = 0;
  for (int i = 0; i < nsrc; ++i) cin += srcC_real[i];
  p.CinTot = cin;
  int geom_pad_h = pad, geom_pad_w = pad;
  if (mode == 0) {
    p.transposed = 0; p.nsrc = nsrc; p.rows = Cout;
    int chunk0 = 0, cbase = 0;
    for (int i = 0; i < nsrc; ++i) {
      HRV_REQUIRE(srcC[i] % 4 == 0 && srcC_real[i] > 0 && srcC_real[i] <= srcC[i], "pack_dev: channel counts");
      p.src_cpad[i] = srcC[i]; p.src_creal[i] = srcC_real[i]; p.src_cbase[i] = cbase; p.src_chunk0[i] = chunk0;
      chunk0 += (srcC[i] + BK - 1) / BK; cbase += srcC_real[i];
    }
    p.chunks_total = chunk0; p.KHp = KH; p.KWp = KW;
    for (int j = 0; j < KH; ++j) p.kh_of[j] = j;
    for (int j = 0; j < KW; ++j) p.kw_of[j] = j;
  } else {
    // data gradient: the conv runs over dY (one source of Cout channels), rows are the Cin axis
    p.transposed = 1; p.nsrc = 1; p.rows = cin;
    p.src_cpad[0] = (Cout + 3) / 4 * 4; p.src_creal[0] = Cout; p.src_cbase[0] = 0; p.src_chunk0[0] = 0;
    p.chunks_total = (p.src_cpad[0] + BK - 1) / BK;
    if (mode == 1) {
      HRV_REQUIRE(stride == 1, "pack_dev: mode 1 is the stride-1 data gradient");
      p.KHp = KH; p.KWp = KW;
      for (int j = 0; j < KH; ++j) p.kh_of[j] = KH - 1 - j;
      for (int j = 0; j < KW; ++j) p.kw_of[j] = KW - 1 - j;
      geom_pad_h = KH - 1 - pad; geom_pad_w = KW - 1 - pad;
    } else {
      HRV_REQUIRE(stride == 2 && phase_a >= 0 && phase_a < 2 && phase_b >= 0 && phase_b < 2, "pack_dev: mode 2 phase");
      // dX[2h'+a] = sum_j dY[h' + j] * W[a + pad - 2j]  for the j with a valid tap
      auto build = [&](int a, int K, int* of, int& Kp, int& padp) {
        int jmin = 0, jmax = -1;
        bool any = false;
        for (int j = -8; j <= 8; ++j) {
          const int k = a + pad - 2 * j;
          if (k >= 0 && k < K) { if (!any) { jmin = j; any = true; } jmax = j; }
        }
        Kp = any ? jmax - jmin + 1 : 1;
        padp = any ? -jmin : 0;
        for (int jj = 0; jj < Kp; ++jj) {
          const int k = a + pad - 2 * (jj + jmin);
          of[jj] = (any && k >= 0 && k < K) ? k : -1;
        }
      };
      build(phase_a, KH, p.kh_of, p.KHp, geom_pad_h);
      build(phase_b, KW, p.kw_of, p.KWp, geom_pad_w);
    }
  }
  p.rows_pad = (p.rows + bn - 1) / bn * bn;
  if (out_geom) {  // {KHp, KWp, pad_h, pad_w, rows, rows_pad, chunks_total, packed elems}
    out_geom[0] = p.KHp; out_geom[1] = p.KWp; out_geom[2] = geom_pad_h; out_geom[3] = geom_pad_w;
    out_geom[4] = p.rows; out_geom[5] = p.rows_pad; out_geom[6] = p.chunks_total;
    out_geom[7] = p.KHp * p.KWp * p.chunks_total * p.rows_pad * BK;
  }
  const size_t total = (size_t)p.chunks_total * p.rows_pad * BK;       // columns: one thread packs every tap of a column
  if (record) {      // hrv_conv2d_pack_weight_record: the launch is the caller's hrv_conv2d_pack_weight_multi
    *record = p;
    return HRV_OK;
  }
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("pack_weight_kernel");
}

extern "C" int32_t hrv_conv2d_pack_record_bytes(void) { return (int32_t)sizeof(PackParams); }

// Fills ``record_host`` (hrv_conv2d_pack_record_bytes() bytes) with what hrv_conv2d_pack_weight_dev_{f32,bf16} /
// hrv_conv2d_pack_weight_pair_dev would launch for these arguments (as_bf16; w2_dev / pair_mode / rows_each: the pair
// form, else null / 0 / 0) and returns the packed geometry; *blocks = blocks of the record in the batched launch.
extern "C" int hrv_conv2d_pack_weight_record(const float* w_oihw_dev, int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc,
                                             const int32_t* srcC, const int32_t* srcC_real, int32_t tile_cfg, int32_t mode,
                                             int32_t stride, int32_t pad, int32_t phase_a, int32_t phase_b, float wscale,
                                             const float* sigma_dev, int32_t as_bf16, const float* w2_dev, int32_t pair_mode,
                                             int32_t rows_each, void* out_dev, int32_t* out_geom, void* record_host,
                                             int32_t* blocks) {
  HRV_REQUIRE(record_host && out_geom && blocks, "pack_record: null pointer");
  int bke = 16;
  if (as_bf16) {
    const int rb = hrv_conv2d_tile_row_bytes(tile_cfg);
    HRV_REQUIRE(rb == 64 || rb == 128, "pack_record: bad tile_cfg %d", tile_cfg);
    bke = rb / 2;
  }
  const int rc = pack_weight_dev_impl(w_oihw_dev, Cout, KH, KW, nsrc, srcC, srcC_real, tile_cfg, mode, stride, pad, phase_a,
                                      phase_b, wscale, sigma_dev, out_dev, out_geom, nullptr, bke, as_bf16 ? 1 : 0, w2_dev,
                                      pair_mode, rows_each, (PackParams*)record_host);
  if (rc) return rc;
  HRV_REQUIRE(out_geom[7] > 0 && (int64_t)out_geom[7] < (int64_t)1 << 30, "pack_record: packed size");
  const int64_t cols = (int64_t)out_geom[7] / ((int64_t)out_geom[0] * out_geom[1]);      // elements per tap plane
  *blocks = (int32_t)((cols + PACK_MULTI_ELEMS - 1) / PACK_MULTI_ELEMS);
  return HRV_OK;
}

// One launch over ``n`` records (device copies of the host records, in order) -- ``first_block_dev``: [n + 1] prefix sums
// of the records' block counts, ``blocks`` = first_block[n].
extern "C" int hrv_conv2d_pack_weight_multi(const void* records_dev, const int32_t* first_block_dev, int32_t n,
                                            int32_t blocks, hrv_stream_t stream) {
  HRV_REQUIRE(records_dev && first_block_dev && n > 0 && blocks > 0, "pack_multi: bad args");
  hipLaunchKernelGGL(pack_weight_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const PackParams*)records_dev, (const int*)first_block_dev, n);
  return check_launch("pack_weight_multi_kernel");
}

extern "C" int hrv_conv2d_pack_weight_dev_f32(const float* w_oihw_dev, int32_t Cout, int32_t KH, int32_t KW,
                                              int32_t nsrc, const int32_t* srcC, const int32_t* srcC_real,
                                              int32_t tile_cfg, int32_t mode, int32_t stride, int32_t pad,
                                              int32_t phase_a, int32_t phase_b, float wscale,
                                              const float* sigma_dev, float* out_dev, int32_t* out_geom,
                                              hrv_stream_t stream) {
  return pack_weight_dev_impl(w_oihw_dev, Cout, KH, KW, nsrc, srcC, srcC_real, tile_cfg, mode, stride, pad, phase_a,
                              phase_b, wscale, sigma_dev, out_dev, out_geom, stream, 16, 0);
}

// bf16 packing for the bf16 matrix-core engine ([kt][rows_pad][64] bf16 for the 128-byte-row tiles cfg 8/9,
// [..][32] for the 64-byte-row tiles); same modes / geometry output (out_geom[7] counts bf16 elements).
extern "C" int hrv_conv2d_pack_weight_dev_bf16(const float* w_oihw_dev, int32_t Cout, int32_t KH, int32_t KW,
                                               int32_t nsrc, const int32_t* srcC, const int32_t* srcC_real,
                                               int32_t tile_cfg, int32_t mode, int32_t stride, int32_t pad,
                                               int32_t phase_a, int32_t phase_b, float wscale,
                                               const float* sigma_dev, uint16_t* out_dev, int32_t* out_geom,
                                               hrv_stream_t stream) {
  // k-values per packed row follow the TILE's row size (128-byte rows: the LDS-DMA gather tiles 8-11 and every patch-mode
  // tile 16-19).  (A range test on 8..11 here packed 32-value rows for the patch tiles: wrong weights whenever a
  // training-path convolution picked tile 16-18 -- found by tools/diag/patch_train_check.py, pinned by
  // test_training_convs_on_patch_tiles_match_torch.)
  const int rb = hrv_conv2d_tile_row_bytes(tile_cfg);
  HRV_REQUIRE(rb == 64 || rb == 128, "pack_dev: bad tile_cfg %d", tile_cfg);
  const int bke = rb / 2;
  return pack_weight_dev_impl(w_oihw_dev, Cout, KH, KW, nsrc, srcC, srcC_real, tile_cfg, mode, stride, pad, phase_a,
                              phase_b, wscale, sigma_dev, out_dev, out_geom, stream, bke, 1);
}

// The same packing for a weight PAIR (SPADE conv_gamma / conv_beta, both [rows_each][Cin][KH][KW]) without materialising
// the combined matrix: pair_mode 1 = forward, virtual Cout = 64*ceil(rows_each/32) rows interleaved (gamma32 | beta32);
// pair_mode 2 = stride-1 data gradient (mode 1) over dY = [dgamma | dbeta], each half Cout/2 channels wide.
extern "C" int hrv_conv2d_pack_weight_pair_dev(const float* w_a_dev, const float* w_b_dev, int32_t rows_each, int32_t pair_mode,
                                               int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc, const int32_t* srcC,
                                               const int32_t* srcC_real, int32_t tile_cfg, int32_t mode, int32_t pad,
                                               int32_t as_bf16, void* out_dev, int32_t* out_geom, hrv_stream_t stream) {
  int bke = 16;
  if (as_bf16) {
    const int rb = hrv_conv2d_tile_row_bytes(tile_cfg);
    HRV_REQUIRE(rb == 64 || rb == 128, "pack_pair_dev: bad tile_cfg %d", tile_cfg);
    bke = rb / 2;
  }
  return pack_weight_dev_impl(w_a_dev, Cout, KH, KW, nsrc, srcC, srcC_real, tile_cfg, mode, 1, pad, 0, 0, 1.0f, nullptr, out_dev,
                              out_geom, stream, bke, as_bf16 ? 1 : 0, w_b_dev, pair_mode, rows_each);
}

extern "C" int64_t hrv_conv2d_wgrad_workspace_bytes(int32_t Cout, int32_t CinTot, int32_t KH, int32_t KW, int64_t P) {
  // upper bound on S is 256 slabs
  const int64_t ptiles = (P + BK - 1) / BK;
  int64_t S = ptiles / 4 < 1 ? 1 : ptiles / 4;
  if (S > 256) S = 256;
  return (S * KH * KW * (int64_t)Cout * CinTot + 256 * (int64_t)Cout) * (int64_t)sizeof(float);   // + bias partials
}

namespace hrv {
// wgrad_tr.hip: LDS-DMA + transposing-read weight gradient for bf16-stored operands (1: launched, 0: shape not served)
int wgrad_tr_try(const void* dy, int dy_cs, int dy_co, int Cout, const void* x, int x_C, int x_cs, int x_co, int x_C_real,
                 int ci_base, int CinTot, int N, int H, int W, int KH, int KW, int pad, float* workspace,
                 long long workspace_bytes, float* dbias, int dbias_accumulate, hipStream_t st, int* S_out);
// wgrad_s2.hip: the same for PatchGAN's 4x4 stride-2 pad-2 layers (1: launched, 0: shape not served)
int wgrad_s2_try(const void* dy, int dy_cs, int dy_co, int Cout, const void* x, int x_C, int x_cs, int x_co, int x_C_real, int ci_base,
                 int CinTot, int N, int H, int W, int Ho, int Wo, float* workspace, long long workspace_bytes, float* dbias, hipStream_t st,
                 int* S_out);
int wgrad_s2_serves(int Cout, int x_C, int x_cs, int x_co, int dy_cs, int dy_co, int N, int H, int W);
}

static int wgrad_impl(const float* dy, int32_t dy_cstride, int32_t dy_coff, int32_t Cout, const float* x, int32_t x_C,
                      int32_t x_cstride, int32_t x_coff, int32_t x_up_shift, int32_t x_C_real, int32_t ci_base,
                      int32_t CinTot, int32_t N, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t KH, int32_t KW,
                      int32_t stride, int32_t pad, float* workspace, int64_t workspace_bytes, float* dw_oihw,
                      int32_t accumulate, float* dbias, int32_t dbias_accumulate, hrv_stream_t stream,
                      const bool mma_bf16, const bool x_bf16 = false, const bool dy_bf16 = false) {
  HRV_REQUIRE(dy && x && workspace && dw_oihw, "wgrad: null pointer");
  HRV_REQUIRE(Cout > 0 && x_C > 0 && x_C % 4 == 0 && x_cstride % 4 == 0 && x_coff % 4 == 0 && dy_cstride % 4 == 0 &&
                  dy_coff % 4 == 0 && x_C_real > 0 && x_C_real <= x_C && ci_base >= 0 && ci_base + x_C_real <= CinTot,
              "wgrad: channel layout");
  HRV_REQUIRE(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "wgrad: geometry");
  HRV_REQUIRE((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)workspace) & 15) == 0, "wgrad: 16-byte alignment");
  HRV_REQUIRE(dy_coff + Cout <= dy_cstride + 3, "wgrad: dy slice");
  WgradParams p;
  p.dy = dy; p.dy_cs = dy_cstride; p.dy_co = dy_coff; p.Cout = Cout;
  p.x = x; p.x_C = x_C; p.x_cs = x_cstride; p.x_co = x_coff; p.x_up = x_up_shift;
  p.N = N; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
  p.P = N * Ho * Wo; p.ci_base = ci_base; p.ci_real = x_C_real; p.CinTot = CinTot;
  p.x_bf16 = x_bf16 ? 1 : 0;
  HRV_REQUIRE(!(x_bf16 || dy_bf16) || mma_bf16, "wgrad: bf16-stored operands exist for the bf16 matrix-core kernel only");
  HRV_REQUIRE(!dy_bf16 || x_bf16, "wgrad: storage_flags 1 (bf16 dY with fp32 X) is not built");
  if (mma_bf16 && x_bf16 && dy_bf16 && stride == 1 && Ho == H && Wo == W && x_up_shift == 0) {
    int S2 = 0;
    const int r = wgrad_tr_try(dy, dy_cstride, dy_coff, Cout, x, x_C, x_cstride, x_coff, x_C_real, ci_base, CinTot, N, H, W,
                               KH, KW, pad, workspace, workspace_bytes, dbias, dbias_accumulate, (hipStream_t)stream, &S2);
    if (r < 0) return r;
    if (r == 1) {      // (the kernel left its bias partials right behind the S2 weight slabs, as the kernels below do)
      const size_t total = (size_t)Cout * x_C_real * KH * KW;
      const int nb = grid_for(total);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb + (dbias ? (Cout + 15) / 16 : 0)), dim3(256), 0, (hipStream_t)stream, workspace,
                         S2, KH * KW, Cout, CinTot, ci_base, x_C_real, dw_oihw, accumulate, nb,
                         workspace + (size_t)S2 * KH * KW * Cout * CinTot, dbias, dbias_accumulate);
      return check_launch("wgrad_reduce_kernel");
    }
  }
  if (mma_bf16 && x_bf16 && dy_bf16 && stride == 2 && KH == 4 && KW == 4 && pad == 2 && x_up_shift == 0) {
    int S2 = 0;
    const int r = wgrad_s2_try(dy, dy_cstride, dy_coff, Cout, x, x_C, x_cstride, x_coff, x_C_real, ci_base, CinTot, N, H, W, Ho, Wo, workspace,
                               workspace_bytes, dbias, (hipStream_t)stream, &S2);
    if (r < 0) return r;
    if (r == 1) {
      const size_t total = (size_t)Cout * x_C_real * KH * KW;
      const int nb = grid_for(total);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb + (dbias ? (Cout + 15) / 16 : 0)), dim3(256), 0, (hipStream_t)stream, workspace,
                         S2, KH * KW, Cout, CinTot, ci_base, x_C_real, dw_oihw, accumulate, nb,
                         workspace + (size_t)S2 * KH * KW * Cout * CinTot, dbias, dbias_accumulate);
      return check_launch("wgrad_reduce_kernel");
    }
  }
  const int wt = pick_wtile(Cout, x_C);
  const int bm = wt_bm(wt), bn = wt_bn(wt);
  p.taps = KH * KW;
  // column tiles over (tap, ci) (+ the "ones" column group of the fused bias gradient)
  p.co_tiles = (Cout + bm - 1) / bm; p.ci_tiles = (p.taps * x_C + (dbias ? 4 : 0) + bn - 1) / bn;
  const int tiles = p.co_tiles * p.ci_tiles;
  HRV_REQUIRE(!mma_bf16 || Wo % 4 == 0, "wgrad (bf16 matrix cores): Wo must be a multiple of 4 (got %d)", Wo);
  const int ptiles = mma_bf16 ? (p.P + BKP - 1) / BKP : (p.P + BK - 1) / BK;
  int S = (1024 + tiles - 1) / tiles;
  if (S > ptiles / 4) S = ptiles / 4;
  if (S > 256) S = 256;
  if (S < 1) S = 1;
  const int64_t need = ((int64_t)S * p.taps * Cout * CinTot + (dbias ? (int64_t)S * Cout : 0)) * (int64_t)sizeof(float);
  HRV_REQUIRE(workspace_bytes >= need, "wgrad: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)need);
  p.S = S; p.ws = workspace;
  p.bias_ws = dbias ? workspace + (size_t)S * p.taps * Cout * CinTot : nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = tiles * S;
#define WG_CASE(I, A, B, Cc, D)                                                                                   \
  case I:                                                                                                         \
    if (mma_bf16 && dy_bf16) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<A, B, Cc, D, true, true>), dim3(nblk), dim3(256), 0, st, p); \
    else if (mma_bf16 && x_bf16) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<A, B, Cc, D, true>), dim3(nblk), dim3(256), 0, st, p); \
    else if (mma_bf16) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<A, B, Cc, D>), dim3(nblk), dim3(256), 0, st, p); \
    else hipLaunchKernelGGL((conv_wgrad_mfma_kernel<A, B, Cc, D>), dim3(nblk), dim3(256), 0, st, p);              \
    break;
  switch (wt) {
    WG_CASE(0, 1, 1, 1, 4) WG_CASE(1, 2, 1, 1, 4) WG_CASE(2, 3, 1, 1, 4) WG_CASE(3, 4, 1, 1, 4) WG_CASE(4, 5, 1, 1, 4)
    WG_CASE(5, 6, 1, 1, 4) WG_CASE(6, 2, 2, 2, 2) WG_CASE(7, 1, 1, 4, 1) WG_CASE(8, 2, 1, 4, 1)
    WG_CASE(9, 1, 1, 2, 2)
    default:
      set_error("wgrad: unknown tile %d", wt);
      return HRV_ERR_ARG;
  }
#undef WG_CASE
  int rc = check_launch("conv_wgrad_mfma_kernel");
  if (rc) return rc;
  const size_t total = (size_t)Cout * x_C_real * p.taps;
  const int nb = grid_for(total);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb + (dbias ? (Cout + 15) / 16 : 0)), dim3(256), 0, st, workspace, S, p.taps, Cout,
                     CinTot, ci_base, x_C_real, dw_oihw, accumulate, nb, p.bias_ws, dbias, dbias_accumulate);
  return check_launch("wgrad_reduce_kernel");
}

extern "C" int hrv_conv2d_wgrad_nhwc_f32(const float* dy, int32_t dy_cstride, int32_t dy_coff, int32_t Cout,
                                         const float* x, int32_t x_C, int32_t x_cstride, int32_t x_coff,
                                         int32_t x_up_shift, int32_t x_C_real, int32_t ci_base, int32_t CinTot,
                                         int32_t N, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t KH, int32_t KW,
                                         int32_t stride, int32_t pad, float* workspace, int64_t workspace_bytes,
                                         float* dw_oihw, int32_t accumulate, float* dbias, int32_t dbias_accumulate,
                                         hrv_stream_t stream) {
  return wgrad_impl(dy, dy_cstride, dy_coff, Cout, x, x_C, x_cstride, x_coff, x_up_shift, x_C_real, ci_base, CinTot, N, H,
                    W, Ho, Wo, KH, KW, stride, pad, workspace, workspace_bytes, dw_oihw, accumulate, dbias,
                    dbias_accumulate, stream, false);
}

// Same contract on the bf16 matrix cores (operands rounded to bf16 while staged, fp32 accumulate): the weight
// gradient of mixed-precision training.  Requires Wo % 4 == 0.
extern "C" int hrv_conv2d_wgrad_bf16mma_nhwc_f32(const float* dy, int32_t dy_cstride, int32_t dy_coff, int32_t Cout,
                                                 const float* x, int32_t x_C, int32_t x_cstride, int32_t x_coff,
                                                 int32_t x_up_shift, int32_t x_C_real, int32_t ci_base, int32_t CinTot,
                                                 int32_t N, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t KH,
                                                 int32_t KW, int32_t stride, int32_t pad, float* workspace,
                                                 int64_t workspace_bytes, float* dw_oihw, int32_t accumulate,
                                                 float* dbias, int32_t dbias_accumulate, hrv_stream_t stream) {
  return wgrad_impl(dy, dy_cstride, dy_coff, Cout, x, x_C, x_cstride, x_coff, x_up_shift, x_C_real, ci_base, CinTot, N, H,
                    W, Ho, Wo, KH, KW, stride, pad, workspace, workspace_bytes, dw_oihw, accumulate, dbias,
                    dbias_accumulate, stream, true);
}

// Same again with bf16-STORED operands (storage_flags bit0: dY, bit1: X; element counts): activations and
// gradients that only matrix cores read (ReLU(conv_shared(seg)), the expanded label map, [dgamma|dbeta]) are kept
// in bf16 by the mixed-precision training plan -- the MMA operand is the same bf16 value either way.
extern "C" int hrv_conv2d_wgrad_bf16mma_st_nhwc_f32(const void* dy, int32_t dy_cstride, int32_t dy_coff, int32_t Cout,
                                                    const void* x, int32_t x_C, int32_t x_cstride, int32_t x_coff,
                                                    int32_t x_up_shift, int32_t x_C_real, int32_t ci_base,
                                                    int32_t CinTot, int32_t N, int32_t H, int32_t W, int32_t Ho,
                                                    int32_t Wo, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                                                    float* workspace, int64_t workspace_bytes, float* dw_oihw,
                                                    int32_t accumulate, float* dbias, int32_t dbias_accumulate,
                                                    int32_t storage_flags, hrv_stream_t stream) {
  HRV_REQUIRE(storage_flags >= 0 && storage_flags <= 3, "wgrad: storage_flags");
  return wgrad_impl((const float*)dy, dy_cstride, dy_coff, Cout, (const float*)x, x_C, x_cstride, x_coff, x_up_shift,
                    x_C_real, ci_base, CinTot, N, H, W, Ho, Wo, KH, KW, stride, pad, workspace, workspace_bytes, dw_oihw,
                    accumulate, dbias, dbias_accumulate, stream, true, (storage_flags & 2) != 0, (storage_flags & 1) != 0);
}

extern "C" int hrv_colsum_nhwc_f32(const float* x, int64_t P, int32_t C, int32_t cstride, int32_t coff, float* workspace,
                                   int64_t workspace_bytes, float* out, int32_t accumulate, hrv_stream_t stream) {
  HRV_REQUIRE(x && workspace && out && P > 0 && C > 0, "colsum: bad args");
  HRV_REQUIRE(cstride % 4 == 0 && coff % 4 == 0 && coff + C <= cstride + 3 && (((uintptr_t)x | (uintptr_t)workspace) & 15) == 0,
              "colsum: layout");
  const int Cpad = (C + 3) / 4 * 4;
  HRV_REQUIRE(coff + Cpad <= cstride, "colsum: padded channel group must lie inside the pixel row");
  int nb = (int)((P + 1023) / 1024);
  nb = nb < 1 ? 1 : (nb > 256 ? 256 : nb);
  HRV_REQUIRE(workspace_bytes >= (int64_t)nb * Cpad * 4, "colsum: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, st, x, (int)P, Cpad / 4, cstride, coff, nb, workspace);
  int rc = check_launch("colsum_partial_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, st, workspace, nb, C, Cpad, out, accumulate);
  return check_launch("colsum_final_kernel");
}

extern "C" int hrv_conv2d_wgrad_s2_supported(int32_t Cout, int32_t x_C, int32_t x_cstride, int32_t x_coff, int32_t dy_cstride, int32_t dy_coff,
                                             int32_t N, int32_t H, int32_t W) {
  return hrv::wgrad_s2_serves(Cout, x_C, x_cstride, x_coff, dy_cstride, dy_coff, N, H, W);
}
