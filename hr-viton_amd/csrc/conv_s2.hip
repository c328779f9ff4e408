// PatchGAN's 4x4 stride-2 pad-2 convolution (NLayerDiscriminator, network_generator.py:263-272) over a bf16-stored NHWC
// feature map, and its data gradient, on the skeleton of conv_p2.hip -- the layers the generic tiles ran at 0.08 of the bf16
// MFMA peak (fp32 feature maps, strided gathers, four single-phase launches per data gradient).
//
// Both directions are a 2x2 stride-1 convolution over CELLS (a cell = the 2x2 pixels (2cy+dy, 2cx+dx)):
//   forward   Y[o]              = sum_{t,d} W[2t+d] * X[2 (o-1+t) + d]           per axis: cell o-1+t, sub-pixel d
//   gradient  dX[2c+d]          = sum_t     W[2t+d] * dY[c+1-t]
// mode 0 (forward): a K-chunk is 32 channels of ONE sub-pixel (dy, dx) -- the patch loader gathers the 17x17 cells of the tile's halo
//   at source pixels (2cy+dy, 2cx+dx), 64 contiguous bytes each; no space-to-depth copy of the feature map exists.
// mode 1 (data gradient): the source is dY at its own resolution, the COLUMNS are (phase (dy,dx), input channel): a column tile of
//   32 lies in one phase, its rows are stored at pixel (2cy+dy, 2cx+dx) of dX -- the four phases of the generic engine's four
//   launches are the column passes of one.  Optional in the epilogue: + a second gradient of the same tensor (the
//   feature-matching tap), * LeakyReLU'(mask) (the activation behind the forward layer that produced X).
// mode 2: a plain 2x2 stride-1 convolution (pad 1 on top / left) over a tensor that IS a space-to-depth image (PatchGAN's model0:
//   10 input channels -> 4 x 12 per cell).
//
// Block = 256 threads = 4 waves, a 16x16-cell tile x 128 (or 64) columns per pass, 2 x NTP accumulator tiles of 32x32 per wave
// (swapped operands: a lane ends up with 4 consecutive columns of one pixel).  The chunk's halo patch (pitch 20, 64 B per cell,
// 16-byte groups XOR-swizzled by (hx >> 2) & 3) is double-buffered by LDS-DMA; weights stream through a FOUR-stage ring in
// MFMA-fragment order ([pass][chunk][tap][column tile][k-step][lane][8 bf16]): a chunk is four k-tiles (taps), k-tile kt lives in
// stage kt % 4 = its tap, requested two k-tiles ahead, published by a counted s_waitcnt + one barrier per k-tile.
// LDS 80,384 B and <= 256 registers: two blocks per CU.
#include <string.h>

#include <type_traits>
#include <utility>

#include "conv_params.h"

namespace hrv {

#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned s2_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned s2_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void s2_store16(s2_u32x4 v, rsrc_t r, unsigned voff) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, 0, 0); }
__device__ __forceinline__ s2_u32x2 s2_load8(rsrc_t r, unsigned voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, soff, 0); }
__device__ __forceinline__ f32x4 s2_load16(rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
__device__ __forceinline__ s2_u32x2 s2_swap32(unsigned lo, unsigned hi) {
  const auto s = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
  s2_u32x2 r;
  r[0] = s[0]; r[1] = s[1];
  return r;
}
#else
typedef unsigned s2_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned s2_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void s2_store16(s2_u32x4, rsrc_t, unsigned) {}
__device__ inline s2_u32x2 s2_load8(rsrc_t, unsigned, int) { return s2_u32x2{0, 0}; }
__device__ inline f32x4 s2_load16(rsrc_t, unsigned, int) { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ inline s2_u32x2 s2_swap32(unsigned a, unsigned b) { return s2_u32x2{a, b}; }
#endif
// (by value: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 whatever the index, conv_p2.hip)
__device__ __forceinline__ unsigned s2_bits(float f) { return __builtin_bit_cast(unsigned, f); }

constexpr int S2_MAXP = 16;
constexpr int S2_PW = 20;                                // patch pitch in cells (a multiple of 4: the swizzle keys on hx)
constexpr int S2_PBUF = 23 * 1024;                       // 17 rows x 20 cells x 64 B = 21,760, DMA'd as 22 (+1 empty) pieces of 1 KB
constexpr int s2_sb(int ntp) { return ntp * 2048; }                                      // ring stage
constexpr int s2_patch_off(int ntp) { return 4 * s2_sb(ntp); }
constexpr int s2_cb_off(int ntp) { return s2_patch_off(ntp) + 2 * S2_PBUF; }
constexpr int s2_lds(int ntp) { return s2_cb_off(ntp) + 512; }
static_assert(2 * ((s2_lds(4) + 1279) / 1280) * 1280 <= 160 * 1024, "two blocks per CU");

struct S2Params {
  const void* src; int src_cs, src_co, Cin; unsigned src_bytes;     // bf16 NHWC; src_bytes: ONE image
  int Hs, Ws;               // source extent
  int sstep;                // 2: cell (cy, cx) of chunk sub-pixel (dy, dx) = source pixel (2cy+dy, 2cx+dx); 1: cell = source pixel
  int cpc, nchunk;          // 32-channel chunks per sub-pixel; chunks of a pass (sstep 2: 4 cpc)
  int org;                  // patch origin relative to the tile: -1 (forward) / 0 (data gradient)
  int N, Ht, Wt;            // tile grid extent in cells
  const void* wp; unsigned w_bytes;
  int npass;
  int ntp[S2_MAXP], tile0[S2_MAXP];
  unsigned woff[S2_MAXP];
  int m_tiles;
  int Cout;                 // columns
  const float* bias;
  int ostep, tpp;           // ostep 2: column tile ct lies in phase ct / tpp = (dy, dx): its rows go to pixel (2cy+dy, 2cx+dx), channel
                            // (ct % tpp) * 32 ...; ostep 1: pixel = cell, channel = column
  int Ho, Wo;               // output image extent
  void* out; int out_cs, out_co, out_f32;
  const void* res; int res_cs, res_co, res_f32;      // added before the activation (same pixel / channel mapping as out)
  int act; float slope;
  const void* mask; int mask_cs, mask_co; float mask_slope;          // bf16: out *= (mask > 0 ? 1 : mask_slope)
  int pp;                   // one (tile, pass) per unit of work
};

struct S2Plan {
  int npass, ntp[S2_MAXP], tile0[S2_MAXP];
  unsigned woff[S2_MAXP];
  int nchunk;
  long long bytes;
};

static int s2_chunks(int mode, int K) { return (mode == 0 ? 4 : 1) * ((K + 31) / 32); }

static bool s2_plan(int mode, int K, int cols, S2Plan& pl) {
  memset(&pl, 0, sizeof(pl));
  // columns in passes of 4 column tiles (128) and at most one of 2 (64): the PatchGAN's 64 / 128 / 256 / 512-column layers
  if (mode < 0 || mode > 2 || K < 1 || cols < 64 || cols % 64 != 0 || K % 8 != 0) return false;
  const int NT = cols / 32, n4 = NT / 4, rem = NT % 4;
  if (n4 + (rem ? 1 : 0) > S2_MAXP) return false;
  pl.npass = n4 + (rem ? 1 : 0);
  pl.nchunk = s2_chunks(mode, K);
  long long off = 0;
  int t0 = 0;
  for (int i = 0; i < pl.npass; ++i) {
    pl.ntp[i] = i < n4 ? 4 : rem;
    pl.tile0[i] = t0;
    t0 += pl.ntp[i];
    pl.woff[i] = (unsigned)off;
    off += (long long)pl.nchunk * 4 * pl.ntp[i] * 2048;
  }
  pl.bytes = off;
  return off < (long long)0xFFFFFFF0;
}

// ------------------------------------------------------------------------------------------------ weight packer
// mode 0: w = the layer's OIHW weight [cols][K][4][4]; chunk = (sub-pixel, 32 channels), tap (a, b): kh = 2a+dy, kw = 2b+dx
// mode 1: w = the FORWARD layer's OIHW weight [K][Cph][4][4] (its output channels are this call's K), columns (phase, ci):
//         patch tap (a, b) <-> t = 1 - a: kh = 2 (1-a) + dy, kw = 2 (1-b) + dx
// mode 2: w = [cols][K][2][2]
struct S2PackParams {
  S2Plan pl;
  int mode, K, cols, Cph;
  int split3;             // modes 0 / 2: K = 3 K0 over a source [hi | lo | hi]; weight thirds [hi(w) | hi(w) | w - hi(w)] of the K0-channel w
  const float* w;
  const float* sigma;     // optional: weights are divided by sigma[0] (spectral norm)
  float wscale;
  unsigned short* out;
};

__device__ __forceinline__ void s2_pack_group(const S2PackParams& p, const long long G) {
  if (G >= p.pl.bytes / 16) return;
  int pass = 0;
  for (int i = 1; i < p.pl.npass; ++i)
    if (G * 16 >= (long long)p.pl.woff[i]) pass = i;
  const int ntp = p.pl.ntp[pass];
  long long r = G - (long long)p.pl.woff[pass] / 16;
  const int lane = (int)(r & 63);
  r >>= 6;
  const int piece = (int)(r % (ntp * 2));
  const int kt = (int)(r / (ntp * 2));
  const int j = piece >> 1, s = piece & 1;
  const int chunk = kt >> 2, tap = kt & 3;
  const int a = tap >> 1, b = tap & 1;
  const int col = (p.pl.tile0[pass] + j) * 32 + (lane & 31);
  const int kk0 = s * 16 + (lane >> 5) * 8;
  const float sc = p.wscale / (p.sigma ? p.sigma[0] : 1.f);
  const int cpc = (p.K + 31) / 32;
  unsigned short v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float w = 0.f;
    int third = 0;
    if (col < p.cols) {
      if (p.mode == 0) {
        const int sub = chunk / cpc, c = (chunk - sub * cpc) * 32 + kk0 + e;
        if (c < p.K) {
          const int K0 = p.split3 ? p.K / 3 : p.K, c0 = p.split3 ? c % K0 : c;
          w = p.w[(((size_t)col * K0 + c0) * 4 + 2 * a + (sub >> 1)) * 4 + 2 * b + (sub & 1)];
          third = p.split3 ? c / K0 : 0;
        }
      } else if (p.mode == 1) {
        const int k = chunk * 32 + kk0 + e, ph = col / p.Cph, ci = col - ph * p.Cph;
        if (k < p.K) w = p.w[(((size_t)k * p.Cph + ci) * 4 + 2 * (1 - a) + (ph >> 1)) * 4 + 2 * (1 - b) + (ph & 1)];
      } else {
        const int k = chunk * 32 + kk0 + e;
        if (k < p.K) {
          const int K0 = p.split3 ? p.K / 3 : p.K, k0 = p.split3 ? k % K0 : k;
          w = p.w[(((size_t)col * K0 + k0) * 2 + a) * 2 + b];
          third = p.split3 ? k / K0 : 0;
        }
      }
    }
    const float ws = w * sc;
    v[e] = third == 2 ? f2bf(ws - bf2f(f2bf(ws))) : f2bf(ws);
  }
  uint4 o;
  o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
  o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
  reinterpret_cast<uint4*>(p.out)[G] = o;
}

__global__ __launch_bounds__(256) void s2_pack_kernel(const S2PackParams p) {
  s2_pack_group(p, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

// several weights in one launch (a PatchGAN pass packs 3 forward / 2 data-gradient weights per scale): blockIdx.y = the job
constexpr int S2_MAXJOBS = 8;
struct S2PackMulti { S2PackParams j[S2_MAXJOBS]; };
__global__ __launch_bounds__(256) void s2_pack_multi_kernel(const S2PackMulti m) {
  const S2PackParams& p = m.j[blockIdx.y];
  for (long long G = (long long)blockIdx.x * blockDim.x + threadIdx.x; G < p.pl.bytes / 16; G += (long long)gridDim.x * blockDim.x) s2_pack_group(p, G);
}

// ------------------------------------------------------------------------------------------------ the kernel
constexpr int s2_wait(int vm) { return (vm & 15) | (7 << 4) | (0 << 8) | ((vm >> 4) << 14); }   // vmcnt(vm) lgkmcnt(0)

struct S2Tile { int n, y0, x0; };
__device__ __forceinline__ S2Tile s2_tile(const S2Params& p, int bid) {
  const int tx = (p.Wt + 15) >> 4, ty = (p.Ht + 15) >> 4;
  const int mt = xcd_remap(bid, p.m_tiles);
  S2Tile t;
  t.n = mt / (tx * ty);
  const int rr = mt - t.n * (tx * ty);
  t.y0 = (rr / tx) << 4;
  t.x0 = (rr % tx) << 4;
  return t;
}

typedef __bf16 s2_bf16x4 __attribute__((ext_vector_type(4)));

// piece `pp` (0..21; beyond: an empty piece) of chunk (sub-pixel `sub`, channels 32 cc ..) of the tile's halo patch -> patch buffer
// `buf`.  16 cells x 4 groups of 8 channels per piece (linear patch order, pitch 20); the 16-byte groups of a cell are XOR-swizzled
// by (hx >> 2) & 3 on the SOURCE side.  Outside the source / beyond its channels / row or column 17+: zeros.
template <int NTP>
__device__ __forceinline__ void s2_patch_piece(const S2Params& p, unsigned char* const smem, const rsrc_t a_rsrc, const S2Tile T, const int sub,
                                               const int cc, const int buf, int pp, const int lane) {
  pp = pp < 23 ? pp : 22;
  const int P = pp * 16 + (lane >> 2), g = lane & 3;
  const int hy = (P * 3277) >> 16, hx = P - S2_PW * hy;              // P / 20, P % 20 (P < 368)
  const int y = p.sstep * (T.y0 + p.org + hy) + (sub >> 1), x = p.sstep * (T.x0 + p.org + hx) + (sub & 1);
  const int gs = g ^ ((hx >> 2) & 3);
  const bool ok = hy < 17 && hx < 17 && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws && cc * 32 + gs * 8 < p.Cin;
  const unsigned off = ((unsigned)(y * p.Ws + x) * (unsigned)p.src_cs + (unsigned)(p.src_co + cc * 32 + gs * 8)) * 2u;
  dma16(a_rsrc, reinterpret_cast<float*>(smem + s2_patch_off(NTP) + buf * S2_PBUF + pp * 1024), ok ? off : 0xFFFFFFF0u, 0u);
}

// The head of a (tile, pass): chunk 0 of the patch -> buffer 0 (6 pieces per wave), k-tiles 0 / 1 -> ring stages 0 / 1.
template <int NTP>
__device__ __forceinline__ void s2_head(const S2Params& p, const int pass, unsigned char* const smem, const S2Tile T, const int wave,
                                        const int lane) {
  constexpr int NPW = NTP * 2, NBW = (NPW + 3) / 4;
  const rsrc_t a_rsrc = make_rsrc(reinterpret_cast<const char*>(p.src) + (size_t)T.n * p.src_bytes, p.src_bytes);
  const rsrc_t w_rsrc = make_rsrc(p.wp, p.w_bytes);
#pragma unroll
  for (int k = 0; k < 6; ++k) s2_patch_piece<NTP>(p, smem, a_rsrc, T, 0, 0, 0, k * 4 + wave, lane);
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
      int idx = wave + 4 * k;
      idx = idx < NPW ? idx : (NPW >= 4 ? idx - 4 : idx % NPW);
      dma16(w_rsrc, reinterpret_cast<float*>(smem + q * s2_sb(NTP) + idx * 1024), (unsigned)lane * 16u,
            p.woff[pass] + (unsigned)q * (unsigned)(NPW * 1024) + (unsigned)idx * 1024u);
    }
}

// EPI 0: bias + activation; 2: + residual and / or mask (their registers stay out of the lean instance's allocation)
template <int NTP, int EPI>
__device__ __forceinline__ void s2_pass(const S2Params& p, const int pass, unsigned char* const smem, const S2Tile T, const bool load_consts,
                                        const bool wait_all, const int nxt_pass, const S2Tile NT_) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  constexpr int NPW = NTP * 2;                               // 1-KB pieces (column tile, k-step) of a k-tile
  constexpr int NBW = (NPW + 3) / 4;                         // DMA instructions per wave per k-tile
  constexpr int NST = 4 * NTP;                               // global stores of one epilogue per wave (bf16 out; fp32: 8 NTP)
  constexpr int NPT = 3;                                     // patch pieces per wave under each of taps 0 and 1
  unsigned char* const ring = smem;
  float* const cbuf = reinterpret_cast<float*>(smem + s2_cb_off(NTP));
  const rsrc_t w_rsrc = make_rsrc(p.wp, p.w_bytes);
  const rsrc_t a_rsrc = make_rsrc(reinterpret_cast<const char*>(p.src) + (size_t)T.n * p.src_bytes, p.src_bytes);
  const unsigned wbase = p.woff[pass];
  const int tile0 = p.tile0[pass];
  const int pt_n = T.n, pt_y0 = T.y0, pt_x0 = T.x0;
  const int nchunk = p.nchunk;

  auto dma_w = [&](const int kt, const int st, const int k) {
    int idx = wave + 4 * k;                                  // (a wave beyond the last piece re-requests an earlier one: same bytes, same place)
    idx = idx < NPW ? idx : (NPW >= 4 ? idx - 4 : idx % NPW);
    dma16(w_rsrc, reinterpret_cast<float*>(ring + st * s2_sb(NTP) + idx * 1024), (unsigned)lane * 16u,
          wbase + (unsigned)kt * (unsigned)(NPW * 1024) + (unsigned)idx * 1024u);
  };

  // every wave is done with the previous (tile, pass): its bias sits in cbuf
  __builtin_amdgcn_s_waitcnt(s2_wait(63));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (load_consts && tid < NTP * 32) cbuf[tid] = (p.bias && tile0 * 32 + tid < p.Cout) ? p.bias[tile0 * 32 + tid] : 0.f;

  const int ty = 4 * wave + (l31 >> 4), tx = l31 & 15;
  // fragment addresses: cell (ty + a, tx + b) of the patch, 16-byte group (2 s + lh) ^ ((hx >> 2) & 3)
  const unsigned a_l = (unsigned)((ty * S2_PW + tx) * 64);
  unsigned axk[2];
#pragma unroll
  for (int kw = 0; kw < 2; ++kw) axk[kw] = (unsigned)((lh ^ (((tx + kw) >> 2) & 3)) << 4);
  const unsigned char* const b_lb = ring + lane * 16;

  // The head (chunk 0, k-tiles 0 and 1) has landed.  Behind it in this wave's queue sit only the previous pass's epilogue
  // stores (they need not drain) -- unless this pass loaded constants, is the block's first, or the epilogue requested the head late
  if (wait_all || load_consts || EPI != 0) __builtin_amdgcn_s_waitcnt(s2_wait(0));
  else if (p.out_f32) __builtin_amdgcn_s_waitcnt(s2_wait(2 * NST));
  else __builtin_amdgcn_s_waitcnt(s2_wait(NST));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // the accumulators start at the bias (cbuf: this pass's columns, zeros without a bias; published by the barrier above)
  f32x16 acc[2][NTP];
#pragma unroll
  for (int j = 0; j < NTP; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(cbuf + j * 32 + 8 * g + 4 * lh);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[0][j][4 * g + e] = b[e]; acc[1][j][4 * g + e] = b[e]; }
    }

  // ---- main loop: chunks x 4 taps.  A fragments double-buffered, B fragments refilled in place (see conv_p2.hip)
  f32x4 fa[2][2], fb[NTP];
  const unsigned char* pbuf = smem + s2_patch_off(NTP);          // patch buffer of the current chunk
  int chunk = 0;
  int n_sub = 0, n_cc = 0;                                       // (sub-pixel, channel chunk) of chunk + 1
  auto advance = [&]() {
    ++n_cc;
    if (n_cc == p.cpc) { n_cc = 0; ++n_sub; }
  };
  advance();
#define S2_READ_A(SET, TAP, S)                                                                             \
  {                                                                                                        \
    const unsigned char* const ap_ = pbuf + a_l + (((TAP) >> 1) * S2_PW + ((TAP) & 1)) * 64;               \
    const unsigned ax_ = axk[(TAP) & 1] ^ (unsigned)((S) << 5);                                            \
    fa[SET][0] = *reinterpret_cast<const f32x4*>(ap_ + ax_);                                               \
    fa[SET][1] = *reinterpret_cast<const f32x4*>(ap_ + 2 * S2_PW * 64 + ax_);                              \
  }
#define S2_READ_B(J, ST, S) fb[J] = *reinterpret_cast<const f32x4*>(b_lb + (ST) * s2_sb(NTP) + ((J) * 2 + (S)) * 1024);
#define S2_STEP(SET, REFILL, ST, SN, DMAW, DMAP)                                                           \
  {                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < NTP; ++j) {                                                      \
      acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[j]),              \
                                                          __builtin_bit_cast(bf16x8, fa[SET][0]), acc[0][j], 0, 0, 0); \
      acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[j]),              \
                                                          __builtin_bit_cast(bf16x8, fa[SET][1]), acc[1][j], 0, 0, 0); \
      if constexpr (REFILL) { S2_READ_B(j, ST, SN) }                                                       \
      if constexpr (DMAW) { if (j < NBW) dma_w(kt + 2, (TAP_ + 2) & 3, j); }                               \
      if constexpr (DMAP) {                                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < NPT; ++i_)                                                 \
          if (j == (NTP >= 4 ? i_ + 1 : NTP - 1))                                                          \
            s2_patch_piece<NTP>(p, smem, a_rsrc, T, n_sub, n_cc, (chunk + 1) & 1, (TAP_ * NPT + i_) * 4 + wave, lane); \
      }                                                                                                    \
    }                                                                                                      \
  }
#define S2_ORDER(NA, REFILL, NW, NP)                                                                       \
  {                                                                                                        \
    if constexpr ((NA) > 0) __builtin_amdgcn_sched_group_barrier(0x100, (NA), 0);                          \
    _Pragma("unroll") for (int j = 0; j < NTP; ++j) {                                                      \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                   \
      if constexpr (REFILL) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                             \
      if (j < (NW)) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                                     \
      if ((NP) > 0 && (NTP >= 4 ? (j >= 1) : (j == NTP - 1)))                                              \
        __builtin_amdgcn_sched_group_barrier(0x010, NTP >= 4 ? 1 : (NP), 0);                               \
    }                                                                                                      \
  }
  int kt = 0;                     // current k-tile of the pass (chunk * 4 + tap)
  // One k-tile = tap TAP of the current chunk, in ring stage TAP.  On entry fa[0] / fb hold its k-step 0.
  // LASTC: the chunk is the last of the pass (no further chunk to prefetch; its taps 2 / 3 request no weights, tap 3 has no successor).
  auto ktile = [&](auto tap_c, auto lastc_c) {
    constexpr int TAP_ = decltype(tap_c)::value;
    constexpr bool LASTC = decltype(lastc_c)::value;
    constexpr bool DMAW = !(LASTC && TAP_ >= 2);
    constexpr bool DMAP = !LASTC && TAP_ < 2;
    constexpr bool NEXT = !(LASTC && TAP_ == 3);
    S2_READ_A(1, TAP_, 1)
    S2_STEP(0, true, TAP_, 1, DMAW, DMAP)
    S2_ORDER(2, true, DMAW ? NBW : 0, DMAP ? NPT : 0)
    if constexpr (NEXT) {
      asm volatile("" ::: "memory");
      // k-tile kt + 1 must have landed.  In flight may stay what this wave requested BEHIND it: the previous tap's patch pieces,
      // this tap's k-tile kt + 2 and this tap's patch pieces.  At tap 3 nothing of the next chunk's patch is left in flight.
      constexpr int NP_PREV = (!LASTC && TAP_ >= 1 && TAP_ <= 2) ? NPT : 0;
      __builtin_amdgcn_s_waitcnt(s2_wait((DMAW ? NBW : 0) + (DMAP ? NPT : 0) + NP_PREV));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr (TAP_ == 3) {                              // next k-tile: tap 0 of the next chunk, the other patch buffer
        pbuf = smem + s2_patch_off(NTP) + ((chunk + 1) & 1) * S2_PBUF;
      }
      S2_READ_A(0, (TAP_ + 1) & 3, 0)
      S2_STEP(1, true, (TAP_ + 1) & 3, 0, false, false)
      S2_ORDER(2, true, 0, 0)
    } else {
      S2_STEP(1, false, 0, 0, false, false)
    }
    ++kt;
  };
  auto chunk_body = [&](auto lastc_c) {
    ktile(std::integral_constant<int, 0>{}, lastc_c);
    ktile(std::integral_constant<int, 1>{}, lastc_c);
    ktile(std::integral_constant<int, 2>{}, lastc_c);
    ktile(std::integral_constant<int, 3>{}, lastc_c);
    ++chunk;
    advance();
  };
  // k-step 0 of k-tile 0
  S2_READ_A(0, 0, 0)
#pragma unroll
  for (int j = 0; j < NTP; ++j) { S2_READ_B(j, 0, 0) }
#pragma unroll 1
  for (int c = 0; c < nchunk - 1; ++c) chunk_body(std::false_type{});
  chunk_body(std::true_type{});
#undef S2_READ_A
#undef S2_READ_B
#undef S2_STEP
#undef S2_ORDER

  // ---- epilogue.  D layout (swapped operands): lane -> cell l31; regs 4g..4g+3 -> columns 8g + 4 lh + (0..3) of the tile.  The bias
  // is already in the accumulators.  Rows leave straight from the registers (conv_p2.hip): the lane pair (l, l + 32) holds columns
  // 8g .. 8g+3 and 8g+4 .. 8g+7 of ONE cell, v_permlane32_swap hands each lane 16 contiguous bytes.  A column tile's rows go to the
  // pixel of ITS phase (ostep 2).
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(s2_wait(63));
  __builtin_amdgcn_s_barrier();                // every wave is done with the patch buffers and the weight ring
  asm volatile("" ::: "memory");
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int l31e = lane_e & 31, lhe = lane_e >> 5;
  const size_t img_px = (size_t)p.Ho * p.Wo;
  const int oes = p.out_f32 ? 4 : 2;
  const rsrc_t o_rsrc = make_rsrc(reinterpret_cast<const char*>(p.out) + (size_t)pt_n * img_px * p.out_cs * oes,
                                  (unsigned)(img_px * p.out_cs * oes));
  const bool has_mask = EPI == 2 && p.mask != nullptr, has_res = EPI == 2 && p.res != nullptr;
  const int res_es = p.res_f32 ? 4 : 2;
  const rsrc_t r_rsrc = make_rsrc(has_res ? reinterpret_cast<const char*>(p.res) + (size_t)pt_n * img_px * p.res_cs * res_es : nullptr,
                                  has_res ? (unsigned)(img_px * p.res_cs * res_es) : 0u);
  const rsrc_t m_rsrc = make_rsrc(has_mask ? reinterpret_cast<const char*>(p.mask) + (size_t)pt_n * img_px * p.mask_cs * 2 : nullptr,
                                  has_mask ? (unsigned)(img_px * p.mask_cs * 2) : 0u);
  constexpr unsigned S2_OOB = 0xF0000000u;
  // this lane's two cells
  const int cxe = pt_x0 + (l31e & 15);
  int cye[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) cye[i] = pt_y0 + 4 * wave + (l31e >> 4) + 2 * i;
  // column tile ct -> (phase offsets a, b; first channel ch0) and the lane's pixel index / validity
  auto col_map = [&](const int j, int& ch0, unsigned (&pix)[2], bool (&pok)[2]) {
    const int ct = tile0 + j;
    int a = 0, b = 0;
    ch0 = ct * 32;
    if (p.ostep == 2) {
      const int ph = (ct >= p.tpp ? 1 : 0) + (ct >= 2 * p.tpp ? 1 : 0) + (ct >= 3 * p.tpp ? 1 : 0);
      ch0 = (ct - ph * p.tpp) * 32;
      a = ph >> 1; b = ph & 1;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int py = p.ostep * cye[i] + a, px = p.ostep * cxe + b;
      pok[i] = cye[i] < p.Ht && cxe < p.Wt && py < p.Ho && px < p.Wo;
      pix[i] = (unsigned)(py * p.Wo + px);
    }
  };
  s2_u32x2 mv[2][4];
  f32x4 rv[2][4];
  auto load_extra = [&](const int j) {
    int ch0;
    unsigned pix[2];
    bool pok[2];
    col_map(j, ch0, pix, pok);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (has_res) {
        const unsigned roff = pok[i] ? (pix[i] * (unsigned)p.res_cs + (unsigned)(p.res_co + ch0 + 4 * lhe)) * (unsigned)res_es : S2_OOB;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (p.res_f32) {
            rv[i][g] = s2_load16(r_rsrc, roff, 8 * g * 4);
          } else {
            const s2_u32x2 h = s2_load8(r_rsrc, roff, 8 * g * 2);
            rv[i][g][0] = __builtin_bit_cast(float, h[0] << 16); rv[i][g][1] = __builtin_bit_cast(float, h[0] & 0xFFFF0000u);
            rv[i][g][2] = __builtin_bit_cast(float, h[1] << 16); rv[i][g][3] = __builtin_bit_cast(float, h[1] & 0xFFFF0000u);
          }
        }
      }
      if (has_mask) {
        const unsigned moff = pok[i] ? (pix[i] * (unsigned)p.mask_cs + (unsigned)(p.mask_co + ch0 + 4 * lhe)) * 2u : S2_OOB;
#pragma unroll
        for (int g = 0; g < 4; ++g) mv[i][g] = s2_load8(m_rsrc, moff, 8 * g * 2);
      }
    }
  };
  // the next (tile, pass) of this block: its head flies while this epilogue computes and stores.  With a residual / mask the
  // head is requested behind the LAST column tile's loads (vmcnt completes in order: a load behind the head waits for the head)
  if (EPI == 0) {
    if (nxt_pass >= 0) s2_head<NTP>(p, nxt_pass, smem, NT_, wave, lane);
  } else {
    load_extra(0);
    if (NTP == 1 && nxt_pass >= 0) s2_head<NTP>(p, nxt_pass, smem, NT_, wave, lane);
  }
  const bool relu = p.act == HRV_ACT_RELU, lrelu = p.act == HRV_ACT_LRELU;
  const float sl = p.slope, msl = p.mask_slope;
#pragma unroll
  for (int j = 0; j < NTP; ++j) {
    f32x4 vv[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[i][g][e] = acc[i][j][4 * g + e];
      if (has_res) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[i][g][e] += rv[i][g][e];
      }
      if (relu) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[i][g][e] = fmaxf(vv[i][g][e], 0.f);      // (+0, never v * 0 = -0)
      } else if (lrelu) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[i][g][e] = fmaxf(vv[i][g][e], vv[i][g][e] * sl);
      }
      if (has_mask) {
        // out *= (mask > 0 ? 1 : mask_slope): the sign / zero test runs on the stored 16-bit patterns
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const s2_u32x2 m = mv[i][g];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned w_ = m[e >> 1];
            const bool keep = (e & 1) ? ((int)w_ > 0xFFFF) : ((short)(w_ & 0xFFFFu) > 0);
            vv[i][g][e] = keep ? vv[i][g][e] : vv[i][g][e] * msl;
          }
        }
      }
    }
    if (EPI != 0 && j + 1 < NTP) {
      load_extra(j + 1);          // (into the registers just consumed; under this column tile's conversion and stores)
      if (j + 2 == NTP && nxt_pass >= 0) s2_head<NTP>(p, nxt_pass, smem, NT_, wave, lane);
    }
    int ch0;
    unsigned pix[2];
    bool pok[2];
    col_map(j, ch0, pix, pok);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned pbase = pix[i] * (unsigned)p.out_cs + (unsigned)(p.out_co + ch0);
      if (!p.out_f32) {
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const s2_u32x2 X = __builtin_bit_cast(s2_u32x2, __builtin_convertvector(vv[i][2 * gp], s2_bf16x4));
          const s2_u32x2 Y = __builtin_bit_cast(s2_u32x2, __builtin_convertvector(vv[i][2 * gp + 1], s2_bf16x4));
          const s2_u32x2 s0 = s2_swap32(X[0], Y[0]), s1 = s2_swap32(X[1], Y[1]);
          const s2_u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
          const int gcol = 8 * (2 * gp + lhe);
          s2_store16(o, o_rsrc, (!pok[i] || (tile0 + j) * 32 + gcol >= p.Cout) ? 0xFFFFFFF0u : (pbase + (unsigned)gcol) * 2u);
        }
      } else {
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const f32x4 xa = vv[i][2 * gp], xb = vv[i][2 * gp + 1];
          const s2_u32x2 a0 = s2_swap32(s2_bits(xa[0]), s2_bits(xb[0]));
          const s2_u32x2 a1 = s2_swap32(s2_bits(xa[1]), s2_bits(xb[1]));
          const s2_u32x2 a2 = s2_swap32(s2_bits(xa[2]), s2_bits(xb[2]));
          const s2_u32x2 a3 = s2_swap32(s2_bits(xa[3]), s2_bits(xb[3]));
          const s2_u32x4 lo_ = {a0[0], a1[0], a2[0], a3[0]}, hi_ = {a0[1], a1[1], a2[1], a3[1]};
          const int gcol = 8 * (2 * gp + lhe);
          const int colg = (tile0 + j) * 32 + gcol;
          s2_store16(lo_, o_rsrc, (!pok[i] || colg >= p.Cout) ? 0xFFFFFFF0u : (pbase + (unsigned)gcol) * 4u);
          s2_store16(hi_, o_rsrc, (!pok[i] || colg + 4 >= p.Cout) ? 0xFFFFFFF0u : (pbase + (unsigned)gcol + 4u) * 4u);
        }
      }
    }
  }
}

template <int NTP, int EPI>
__global__ __launch_bounds__(256, 2) void conv_s2_kernel(const S2Params p, const int pass0, const int pass1) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[s2_lds(NTP)];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  // A unit of work = one tile with the launch's passes pass0 .. pass1 one after the other (the source patch of the later passes
  // comes out of L2) -- or, p.pp (fewer tiles than resident blocks), ONE (tile, pass)
  const int npg = pass1 - pass0;
  const int units = p.pp ? p.m_tiles * npg : p.m_tiles;
  auto unit_tile = [&](const int u) { return p.pp ? u / npg : u; };
  auto unit_pass = [&](const int u) { return p.pp ? pass0 + u % npg : pass0; };
  if ((int)blockIdx.x < units) s2_head<NTP>(p, unit_pass(blockIdx.x), smem, s2_tile(p, unit_tile(blockIdx.x)), wave, lane);
  int c_pass = -1;
#pragma unroll 1
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const S2Tile T = s2_tile(p, unit_tile(u));
    const int nu = u + gridDim.x;
    const S2Tile TN = s2_tile(p, unit_tile(nu < units ? nu : u));
    const int pa = unit_pass(u), pb = p.pp ? pa + 1 : pass1;
#pragma unroll 1
    for (int pass = pa; pass < pb; ++pass) {
      const bool lastp = pass == pb - 1;
      const int nxt_pass = !lastp ? pass + 1 : (nu < units ? unit_pass(nu) : -1);
      const bool lc = c_pass != pass;
      c_pass = pass;
      s2_pass<NTP, EPI>(p, pass, smem, T, lc, u == (int)blockIdx.x && pass == pa, nxt_pass, lastp ? TN : T);
    }
  }
}

}  // namespace hrv

using namespace hrv;

extern "C" int64_t hrv_conv_s2_packed_bytes(int32_t mode, int32_t K, int32_t cols) {
  S2Plan pl;
  if (!s2_plan(mode, K, cols, pl)) return -1;
  return pl.bytes;
}

// mode 1: (Ho, Wo) = the extent of dX (the forward layer's input); modes 0 / 2: of the layer's output
static void s2_grid(int mode, int Ho, int Wo, int& Ht, int& Wt) {
  Ht = mode == 1 ? (Ho + 1) / 2 : Ho;
  Wt = mode == 1 ? (Wo + 1) / 2 : Wo;
}

extern "C" int hrv_conv_s2_supported(int32_t mode, int32_t K, int32_t cols, int32_t Cph, int32_t N, int32_t Ho, int32_t Wo) {
  S2Plan pl;
  if (!s2_plan(mode, K, cols, pl)) return 0;
  if (mode == 0 && K % 32 != 0) return 0;                                  // a chunk lies in one sub-pixel
  if (mode == 1 && (Cph < 32 || Cph % 32 != 0 || cols != 4 * Cph)) return 0;      // a column tile lies in one phase
  int Ht, Wt;
  s2_grid(mode, Ho, Wo, Ht, Wt);
  const int64_t tiles = (int64_t)N * ((Ht + 15) / 16) * ((Wt + 15) / 16);
  const char* e = hrv::env("HRV_CONV_S2_MIN_TILES_X4");
  int q4 = e ? atoi(e) : 3;
  if (q4 < 1) q4 = 3;
  const int64_t units = tiles < 2 * (int64_t)persistent_cus() ? tiles * pl.npass : tiles;
  return 4 * units >= q4 * (int64_t)persistent_cus() ? 1 : 0;
}

extern "C" int hrv_conv_s2_pack_dev(int32_t mode_flags, const float* w, int32_t K, int32_t cols, int32_t Cph, const float* sigma, float wscale,
                                    void* out, hrv_stream_t stream);

static int s2_pack_fill(S2PackParams& pp, int32_t mode_flags, const float* w, int32_t K, int32_t cols, int32_t Cph, const float* sigma, float wscale,
                        void* out) {
  HRV_REQUIRE(w && out, "conv_s2_pack: null pointer");
  const int mode = mode_flags & 3, split3 = (mode_flags & HRV_S2_SPLIT3) ? 1 : 0;
  HRV_REQUIRE(!split3 || (mode != 1 && K % 3 == 0 && (K / 3) % 8 == 0 && (mode != 0 || (K / 3) % 32 == 0)),
              "conv_s2_pack: split3 is for modes 0 / 2, K = 3 x (a multiple of 8; mode 0: of 32) (mode %d, K %d)", mode, K);
  HRV_REQUIRE(s2_plan(mode, K, cols, pp.pl), "conv_s2_pack: unsupported shape (mode %d, K %d, columns %d)", mode, K, cols);
  HRV_REQUIRE(mode != 0 || K % 32 == 0, "conv_s2_pack: forward K must be a multiple of 32 (got %d)", K);
  HRV_REQUIRE(mode != 1 || (Cph >= 32 && Cph % 32 == 0 && cols == 4 * Cph), "conv_s2_pack: data gradient columns = 4 x Cph (Cph %d, columns %d)", Cph, cols);
  HRV_REQUIRE(((uintptr_t)out & 15) == 0, "conv_s2_pack: out must be 16-byte aligned");
  pp.mode = mode; pp.K = K; pp.cols = cols; pp.Cph = Cph > 0 ? Cph : 1; pp.split3 = split3;
  pp.w = w; pp.sigma = sigma; pp.wscale = wscale; pp.out = (unsigned short*)out;
  return HRV_OK;
}

extern "C" int hrv_conv_s2_pack_multi_dev(int32_t n, const hrv_s2_pack_job_t* jobs, hrv_stream_t stream) {
  HRV_REQUIRE(jobs && n > 0 && n <= S2_MAXJOBS, "conv_s2_pack_multi: 1 .. %d jobs", S2_MAXJOBS);
  S2PackMulti m;
  memset(&m, 0, sizeof(m));
  long long most = 0;
  for (int i = 0; i < n; ++i) {
    const int rc = s2_pack_fill(m.j[i], jobs[i].mode_flags, jobs[i].w, jobs[i].K, jobs[i].cols, jobs[i].Cph, jobs[i].sigma, jobs[i].wscale, jobs[i].out);
    if (rc) return rc;
    if (m.j[i].pl.bytes / 16 > most) most = m.j[i].pl.bytes / 16;
  }
  long long gx = (most + 255) / 256;
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(s2_pack_multi_kernel, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, m);
  return check_launch("s2_pack_multi_kernel");
}

extern "C" int hrv_conv_s2_bf16(const hrv_conv_s2_t* d, hrv_stream_t stream) {
  HRV_REQUIRE(d != nullptr, "conv_s2: null descriptor");
  S2Plan pl;
  HRV_REQUIRE(s2_plan(d->mode, d->K, d->cols, pl), "conv_s2: unsupported shape (mode %d, K %d, columns %d)", d->mode, d->K, d->cols);
  HRV_REQUIRE(d->mode != 0 || d->K % 32 == 0, "conv_s2: forward K must be a multiple of 32 (got %d)", d->K);
  HRV_REQUIRE(d->mode != 1 || (d->Cph >= 32 && d->Cph % 32 == 0 && d->cols == 4 * d->Cph), "conv_s2: data gradient columns = 4 x Cph");
  HRV_REQUIRE(d->N > 0 && d->Hs > 0 && d->Ws > 0 && d->Ho > 0 && d->Wo > 0, "conv_s2: bad extent");
  if (d->mode == 0) HRV_REQUIRE(d->Ho == d->Hs / 2 + 1 && d->Wo == d->Ws / 2 + 1, "conv_s2: forward extent (%d x %d -> %d x %d)", d->Hs, d->Ws, d->Ho, d->Wo);
  if (d->mode == 1) HRV_REQUIRE(d->Hs == d->Ho / 2 + 1 && d->Ws == d->Wo / 2 + 1, "conv_s2: data-gradient extent (dY %d x %d -> dX %d x %d)", d->Hs, d->Ws, d->Ho, d->Wo);
  if (d->mode == 2) HRV_REQUIRE((d->Ho == d->Hs || d->Ho == d->Hs + 1) && (d->Wo == d->Ws || d->Wo == d->Ws + 1), "conv_s2: 2x2 extent");
  HRV_REQUIRE(d->src && d->w_packed && d->out, "conv_s2: null pointer");
  HRV_REQUIRE(d->src_cstride % 8 == 0 && d->src_coff % 8 == 0 && d->src_coff + d->K <= d->src_cstride, "conv_s2: source slice");
  const int64_t sbytes = (int64_t)d->Hs * d->Ws * d->src_cstride * 2;
  HRV_REQUIRE(sbytes < (int64_t)0xFFFFFFF0, "conv_s2: one image of the source exceeds the 32-bit buffer range");
  const int oes = d->out_f32 ? 4 : 2, oal = d->out_f32 ? 4 : 8;
  const int ocols = d->mode == 1 ? d->Cph : d->cols;      // channels of the output image
  HRV_REQUIRE(d->out_cstride % oal == 0 && d->out_coff % oal == 0 && d->out_coff + ocols <= d->out_cstride, "conv_s2: out slice");
  HRV_REQUIRE((int64_t)d->Ho * d->Wo * d->out_cstride * oes < (int64_t)0xFFFFFFF0, "conv_s2: one image of `out` exceeds 4 GB");
  HRV_REQUIRE((((uintptr_t)d->src | (uintptr_t)d->w_packed | (uintptr_t)d->out) & 15) == 0 && ((uintptr_t)d->bias & 3) == 0, "conv_s2: alignment");
  HRV_REQUIRE(d->residual == nullptr || (d->res_cstride % 4 == 0 && d->res_coff % 4 == 0 && d->res_coff + ocols <= d->res_cstride &&
                                         ((uintptr_t)d->residual & 15) == 0 &&
                                         (int64_t)d->Ho * d->Wo * d->res_cstride * (d->res_f32 ? 4 : 2) < (int64_t)0xF0000000),
              "conv_s2: residual slice");
  HRV_REQUIRE(d->mask == nullptr || (d->mask_cstride % 4 == 0 && d->mask_coff % 4 == 0 && ((uintptr_t)d->mask & 7) == 0 &&
                                     d->mask_coff + ocols <= d->mask_cstride && (int64_t)d->Ho * d->Wo * d->mask_cstride * 2 < (int64_t)0xF0000000),
              "conv_s2: mask slice");
  S2Params p;
  memset(&p, 0, sizeof(p));
  p.src = d->src; p.src_cs = d->src_cstride; p.src_co = d->src_coff; p.Cin = d->K; p.src_bytes = (unsigned)sbytes;
  p.Hs = d->Hs; p.Ws = d->Ws;
  p.sstep = d->mode == 0 ? 2 : 1;
  p.cpc = (d->K + 31) / 32;
  p.nchunk = pl.nchunk;
  p.org = d->mode == 1 ? 0 : -1;
  p.N = d->N;
  s2_grid(d->mode, d->Ho, d->Wo, p.Ht, p.Wt);
  p.wp = d->w_packed; p.w_bytes = (unsigned)pl.bytes;
  p.npass = pl.npass;
  for (int i = 0; i < pl.npass; ++i) { p.ntp[i] = pl.ntp[i]; p.tile0[i] = pl.tile0[i]; p.woff[i] = pl.woff[i]; }
  p.m_tiles = d->N * ((p.Ht + 15) / 16) * ((p.Wt + 15) / 16);
  p.Cout = d->cols;
  p.bias = d->bias;
  p.ostep = d->mode == 1 ? 2 : 1;
  p.tpp = d->mode == 1 ? d->Cph / 32 : 1;
  p.Ho = d->Ho; p.Wo = d->Wo;
  p.out = d->out; p.out_cs = d->out_cstride; p.out_co = d->out_coff; p.out_f32 = d->out_f32;
  p.res = d->residual; p.res_cs = d->res_cstride; p.res_co = d->res_coff; p.res_f32 = d->res_f32;
  p.act = d->act; p.slope = d->act_slope;
  p.mask = d->mask; p.mask_cs = d->mask_cstride; p.mask_co = d->mask_coff; p.mask_slope = d->mask_slope;
  p.pp = p.m_tiles < 2 * persistent_cus() ? 1 : 0;
  for (int a = 0; a < pl.npass;) {
    int b = a;
    while (b < pl.npass && pl.ntp[b] == pl.ntp[a]) ++b;
    const long long units = p.pp ? (long long)p.m_tiles * (b - a) : p.m_tiles;
    const int cap = 2 * persistent_cus();
    const int grid = units < cap ? (int)units : cap;
    const dim3 g3(grid), b3(256);
    const hipStream_t st = (hipStream_t)stream;
    const bool lean = !p.res && !p.mask;
    if (pl.ntp[a] == 4) {
      if (lean) hipLaunchKernelGGL((conv_s2_kernel<4, 0>), g3, b3, 0, st, p, a, b);
      else hipLaunchKernelGGL((conv_s2_kernel<4, 2>), g3, b3, 0, st, p, a, b);
    } else {
      if (lean) hipLaunchKernelGGL((conv_s2_kernel<2, 0>), g3, b3, 0, st, p, a, b);
      else hipLaunchKernelGGL((conv_s2_kernel<2, 2>), g3, b3, 0, st, p, a, b);
    }
    a = b;
  }
  return check_launch("conv_s2_kernel");
}

// ------------------------------------------------------------------------------------------------ bf16-storage companions
// What the PatchGAN needs around the kernel above when its feature maps are STORED in bf16 (mixed precision): the space-to-depth
// image of model0's input (over cells, optionally split), InstanceNorm2d + LeakyReLU written in bf16, the upstream scalar applied to a bf16 loss gradient, and a
// width-padded copy of a bf16 dY for the quad-staged weight-gradient kernel (conv_bwd.hip: Wo % 4 == 0).
namespace hrv {

// The space-to-depth image of model0's input as conv_s2's mode 2 reads it: out[n][cy][cx][(dy*2+dx)*C + c] = in[n][2cy+dy][2cx+dx][c],
// bf16, over Hp x Wp >= H/2 x W/2 cells (cells / sub-pixels outside the image: zeros -- a one-cell border makes the layer a 'same' 2x2
// convolution, which is what the LDS-DMA weight-gradient kernel serves); split3: [hi | lo | hi] of that image (HRV_S2_SPLIT3).
__global__ void s2d_cells_kernel(const float* __restrict__ a, int N, int H, int W, int C4, int cs, int co, unsigned short* __restrict__ b, int Hp,
                                 int Wp, int split3) {
  const size_t total = (size_t)N * Hp * Wp * 4 * C4;
  const int C = C4 * 4, K0 = 4 * C, ocs = split3 ? 3 * K0 : K0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t t = i / C4;
    const int sub = (int)(t & 3); t >>= 2;
    const int cx = (int)(t % Wp); t /= Wp;
    const int cy = (int)(t % Hp);
    const int n = (int)(t / Hp);
    const int y = 2 * cy + (sub >> 1), x = 2 * cx + (sub & 1);
    f32x4 v = (f32x4)(0.f);
    if (y < H && x < W) v = *reinterpret_cast<const f32x4*>(a + (((size_t)n * H + y) * W + x) * cs + co + 4 * c4);
    unsigned short h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = f2bf(v[e]);
      l[e] = f2bf(v[e] - bf2f(h[e]));
    }
    uint2 hh, ll;
    hh.x = h[0] | ((unsigned)h[1] << 16); hh.y = h[2] | ((unsigned)h[3] << 16);
    ll.x = l[0] | ((unsigned)l[1] << 16); ll.y = l[2] | ((unsigned)l[3] << 16);
    unsigned short* o = b + (((size_t)n * Hp + cy) * Wp + cx) * ocs + sub * C + 4 * c4;
    *reinterpret_cast<uint2*>(o) = hh;
    if (split3) {
      *reinterpret_cast<uint2*>(o + K0) = ll;
      *reinterpret_cast<uint2*>(o + 2 * K0) = hh;
    }
  }
}

__global__ void instnorm_apply_bf16out_kernel(const float* __restrict__ x, int N, int HW, int C4, int cs, int co, const float* __restrict__ mean,
                                              const float* __restrict__ rstd, int act, float slope, unsigned short* __restrict__ out, int ocs,
                                              int oco) {
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
    uint2 o;
    o.x = f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    o.y = f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(out + pix * ocs + oco + g * 4) = o;
  }
}

__global__ void scale_bf16_kernel(unsigned short* __restrict__ x, size_t n4, float s_host, const float* __restrict__ s_dev) {
  const float s = s_host * (s_dev ? s_dev[0] : 1.f);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    uint2 u = reinterpret_cast<uint2*>(x)[i];
    const float a = bf2f((unsigned short)(u.x & 0xFFFFu)) * s, b = bf2f((unsigned short)(u.x >> 16)) * s;
    const float c = bf2f((unsigned short)(u.y & 0xFFFFu)) * s, d = bf2f((unsigned short)(u.y >> 16)) * s;
    u.x = f2bf(a) | ((unsigned)f2bf(b) << 16);
    u.y = f2bf(c) | ((unsigned)f2bf(d) << 16);
    reinterpret_cast<uint2*>(x)[i] = u;
  }
}

// out[r][w][:] = w < W ? in[r][w][:] : 0   (rows = N * H; 16-byte groups)
__global__ void pad_width_bf16_kernel(const uint4* __restrict__ in, size_t rows, int W, int Wp, int G, uint4* __restrict__ out) {
  const size_t total = rows * (size_t)Wp * G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    const size_t t = i / G;
    const int w = (int)(t % Wp);
    const size_t r = t / Wp;
    out[i] = w < W ? in[(r * W + w) * G + g] : uint4{0u, 0u, 0u, 0u};
  }
}

// out[p][0:C] = hi(x), out[p][C:2C] = bf16(x - hi(x)), out[p][2C:3C] = hi(x): the operand layout of a convolution that multiplies the
// three products hi*hi + lo*hi + hi*lo on the bf16 matrix cores (~16 mantissa bits)
__global__ void split3_bf16_kernel(const float* __restrict__ x, size_t npix, int C4, int cs, int co, unsigned short* __restrict__ out) {
  const size_t total = npix * C4;
  const int C = C4 * 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % C4);
    const size_t pix = i / C4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * cs + co + g * 4);
    unsigned short h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = f2bf(v[e]);
      l[e] = f2bf(v[e] - bf2f(h[e]));
    }
    uint2 hh, ll;
    hh.x = h[0] | ((unsigned)h[1] << 16); hh.y = h[2] | ((unsigned)h[3] << 16);
    ll.x = l[0] | ((unsigned)l[1] << 16); ll.y = l[2] | ((unsigned)l[3] << 16);
    unsigned short* o = out + pix * (3 * (size_t)C) + g * 4;
    *reinterpret_cast<uint2*>(o) = hh;
    *reinterpret_cast<uint2*>(o + C) = ll;
    *reinterpret_cast<uint2*>(o + 2 * C) = hh;
  }
}

static inline int s2_grid_for(size_t work) {
  size_t g = (work + 255) / 256;
  const size_t cap = 256 * 16;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace hrv

extern "C" int hrv_space_to_depth2_cells_bf16(const float* in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_cstride, int32_t in_coff,
                                              int32_t Hp, int32_t Wp, int32_t split3, uint16_t* out, hrv_stream_t stream) {
  HRV_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && C > 0, "space_to_depth2_cells: bad args");
  HRV_REQUIRE(C % 4 == 0 && in_cstride % 4 == 0 && in_coff % 4 == 0 && in_coff + C <= in_cstride && Hp >= (H + 1) / 2 && Wp >= (W + 1) / 2 &&
                  ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 7) == 0,
              "space_to_depth2_cells: 4-channel granules, Hp x Wp cells covering the image");
  const size_t total = (size_t)N * Hp * Wp * 4 * (C / 4);
  hipLaunchKernelGGL(s2d_cells_kernel, dim3(s2_grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, N, H, W, C / 4, in_cstride, in_coff, out, Hp, Wp,
                     split3 ? 1 : 0);
  return check_launch("s2d_cells_kernel");
}

extern "C" int hrv_instnorm_apply_nhwc_bf16out(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t cstride, int32_t coff,
                                               const float* mean, const float* rstd, int32_t act, float act_slope, uint16_t* out,
                                               int32_t out_cstride, int32_t out_coff, hrv_stream_t stream) {
  HRV_REQUIRE(x && mean && rstd && out && N > 0 && H > 0 && W > 0, "instnorm_apply_bf16out: bad args");
  HRV_REQUIRE(C > 0 && C % 4 == 0 && cstride % 4 == 0 && coff % 4 == 0 && out_cstride % 4 == 0 && out_coff % 4 == 0 && coff + C <= cstride &&
                  out_coff + C <= out_cstride && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 7) == 0,
              "instnorm_apply_bf16out: channels must be multiples of 4 and in range");
  const size_t total = (size_t)N * H * W * (C / 4);
  hipLaunchKernelGGL(instnorm_apply_bf16out_kernel, dim3(s2_grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, N, H * W, C / 4, cstride, coff,
                     mean, rstd, act, act_slope, out, out_cstride, out_coff);
  return check_launch("instnorm_apply_bf16out_kernel");
}

extern "C" int hrv_scale_bf16(uint16_t* x, int64_t n, float s_host, const float* s_dev, hrv_stream_t stream) {
  HRV_REQUIRE(x && n > 0 && n % 4 == 0 && ((uintptr_t)x & 7) == 0, "scale_bf16: element count must be a multiple of 4");
  hipLaunchKernelGGL(scale_bf16_kernel, dim3(s2_grid_for((size_t)n / 4)), dim3(256), 0, (hipStream_t)stream, x, (size_t)n / 4, s_host, s_dev);
  return check_launch("scale_bf16_kernel");
}

extern "C" int hrv_pad_width_nhwc_bf16(const uint16_t* in, int64_t rows, int32_t W, int32_t C, int32_t Wp, uint16_t* out, hrv_stream_t stream) {
  HRV_REQUIRE(in && out && rows > 0 && W > 0 && Wp >= W && C > 0 && C % 8 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0,
              "pad_width_bf16: dense rows of 16-byte channel groups");
  const size_t total = (size_t)rows * Wp * (C / 8);
  hipLaunchKernelGGL(pad_width_bf16_kernel, dim3(s2_grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint4*)in, (size_t)rows, W, Wp, C / 8,
                     (uint4*)out);
  return check_launch("pad_width_bf16_kernel");
}

extern "C" int hrv_split3_nhwc_bf16(const float* x, int64_t npix, int32_t C, int32_t cstride, int32_t coff, uint16_t* out, hrv_stream_t stream) {
  HRV_REQUIRE(x && out && npix > 0 && C > 0 && C % 4 == 0 && cstride % 4 == 0 && coff % 4 == 0 && coff + C <= cstride &&
                  ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 7) == 0,
              "split3: 4-channel granules");
  hipLaunchKernelGGL(split3_bf16_kernel, dim3(s2_grid_for((size_t)npix * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, (size_t)npix, C / 4, cstride,
                     coff, out);
  return check_launch("split3_bf16_kernel");
}

extern "C" int hrv_conv_s2_pack_dev(int32_t mode_flags, const float* w, int32_t K, int32_t cols, int32_t Cph, const float* sigma, float wscale,
                                    void* out, hrv_stream_t stream) {
  S2PackParams pp;
  const int rc = s2_pack_fill(pp, mode_flags, w, K, cols, Cph, sigma, wscale, out);
  if (rc) return rc;
  const long long groups = pp.pl.bytes / 16;
  hipLaunchKernelGGL(s2_pack_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pp);
  return check_launch("s2_pack_kernel");
}
