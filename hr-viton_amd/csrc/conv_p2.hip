// 3x3 stride-1 'same' convolution over ONE bf16-stored NHWC source, two blocks per CU -- the plain-convolution sibling of
// spade_fused.hip, for the layers whose generic patch tiles ran at 0.24-0.31 of the bf16 MFMA peak:
//   * the data gradient of SPADENorm.conv_gamma / conv_beta, d(actv) = conv^T([dgamma | dbeta]) * relu'(actv)
//     (network_generator.py:117-118; K = 2 C = 160 / 288 / 544 ..., 128 columns),
//   * VGG19's convolutions and their data gradients in the perceptual loss (networks.py:201-233; 128..512 channels).
//
// Block = 256 threads = 4 waves, a 16x16-pixel tile x 128 (or 64) columns per pass; wave w owns tile rows 4w..4w+3 (2 x 32
// pixels), 2 x NTP accumulator tiles of 32x32 (swapped operands: a lane ends up with 4 consecutive columns of one pixel).
// K runs over 32-channel CHUNKS of the source: the chunk's 18x18 halo patch sits in LDS (64 B per pixel, 16-byte groups
// XOR-swizzled by (hx >> 2) & 3: the tap-shifted b128 fragment reads of a lane group are bank-disjoint) and is
// DOUBLE-BUFFERED -- chunk c+1 arrives by LDS-DMA (one 1-KB piece per wave under each of the first six k-tiles) while the
// nine k-tiles (taps) of chunk c are multiplied.  Weights stream through a 3-stage ring in MFMA-fragment order
// ([pass][chunk][tap][column tile][k-step][lane][8 bf16], 8 KB per k-tile): requested two k-tiles ahead, published by a
// counted s_waitcnt + ONE barrier per k-tile, B fragments refilled in place behind the MFMAs that consumed them.
// LDS 72,192 B and <= 256 registers: two blocks per CU -- one block's epilogue and chunk switches run under the other's MFMAs.
// The head of the next (tile, pass) -- chunk 0 of its patch, its first two k-tiles -- is requested before the epilogue.
#include <string.h>

#include <type_traits>
#include <utility>

#include "conv_params.h"

namespace hrv {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void p2_store16(f32x4 v, rsrc_t r, unsigned voff) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, (int)voff, 0, 0);
}
typedef unsigned p2_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ p2_u32x2 p2_load8(rsrc_t r, unsigned voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, soff, 0); }
__device__ __forceinline__ f32x4 p2_load16(rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
// v_permlane32_swap: lanes 0..31 of the result pair hold (lo, lo of lane + 32), lanes 32..63 hold (hi of lane - 32, hi)
__device__ __forceinline__ p2_u32x2 p2_swap32(unsigned lo, unsigned hi) {
  const auto s = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
  p2_u32x2 r;
  r[0] = s[0]; r[1] = s[1];
  return r;
}
// (by value: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 whatever the index)
__device__ __forceinline__ unsigned p2_bits(float f) { return __builtin_bit_cast(unsigned, f); }
#else
__device__ inline void p2_store16(f32x4, rsrc_t, unsigned) {}
typedef unsigned p2_u32x2 __attribute__((ext_vector_type(2)));
__device__ inline p2_u32x2 p2_load8(rsrc_t, unsigned, int) { return p2_u32x2{0, 0}; }
__device__ inline f32x4 p2_load16(rsrc_t, unsigned, int) { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ inline p2_u32x2 p2_swap32(unsigned a, unsigned b) { return p2_u32x2{a, b}; }
__device__ inline unsigned p2_bits(float) { return 0u; }
#endif

__device__ __forceinline__ f32x4 p2_acc4(const f32x16& a, int g) {
  f32x4 r;
  r[0] = a[4 * g]; r[1] = a[4 * g + 1]; r[2] = a[4 * g + 2]; r[3] = a[4 * g + 3];
  return r;
}

constexpr int P2_MAXP = 16;
constexpr int P2_PW = 20;                                // patch pitch in pixels (a multiple of 4: the swizzle keys on hx)
constexpr int P2_PBUF = 23 * 1024;                       // 18 x 20 pixels x 64 B = 23,040, DMA'd as 23 pieces of 1 KB
// LDS layout by pass width NTP (column tiles of 32): [ring: 3 stages of 32 k x 32 NTP columns][patch x 2][bias].  NTP 4: 72,192 B,
// two blocks per CU; NTP 1 (the 32-column layers of the 1024 x 768 level: HBM / latency-bound, 104-122 registers): 53,760 B,
// THREE blocks per CU
constexpr int p2_sb(int ntp) { return ntp * 2048; }                                      // ring stage
constexpr int p2_patch_off(int ntp) { return 3 * p2_sb(ntp); }
constexpr int p2_cb_off(int ntp) { return p2_patch_off(ntp) + 2 * P2_PBUF; }
constexpr int p2_lds(int ntp) { return p2_cb_off(ntp) + 512; }
constexpr int p2_blocks_per_cu(int ntp) { return ntp == 1 ? 3 : 2; }
static_assert(2 * ((p2_lds(4) + 1279) / 1280) * 1280 <= 160 * 1024, "two blocks per CU");
static_assert(3 * ((p2_lds(1) + 1279) / 1280) * 1280 <= 160 * 1024, "three blocks per CU for single-tile passes");

struct P2Params {
  const void* src; int src_cs, src_co, Cin; unsigned src_bytes;     // bf16 NHWC; src_bytes: ONE image
  int N, H, W;
  const void* wp; unsigned w_bytes;
  int npass;
  int ntp[P2_MAXP];         // column tiles of 32 per pass (4; the last pass 1..3)
  int tile0[P2_MAXP];
  unsigned woff[P2_MAXP];
  int nchunk;               // Cin / 32
  int m_tiles;
  int Cout;
  const float* bias;        // [Cout] or nullptr
  const void* res; int res_cs, res_co, res_f32, res_after;           // residual (fp32 or bf16 NHWC slice) added before the activation,
                                                                     // or -- res_after -- behind activation and mask
  int act; float slope;
  const void* mask; int mask_cs, mask_co; float mask_slope;          // bf16: out *= (mask > 0 ? 1 : mask_slope)
  void* out; int out_cs, out_co, out_f32;
  unsigned long long* tlog;
  int pp;                   // one (tile, pass) per unit of work (see conv_p2_kernel)
};

struct P2Plan {
  int npass, ntp[P2_MAXP], tile0[P2_MAXP];
  unsigned woff[P2_MAXP];
  long long bytes;
};

static bool p2_plan(int Cin, int Cout, P2Plan& pl) {
  memset(&pl, 0, sizeof(pl));
  // K runs over 32-channel chunks; a source whose width is 16 off a multiple of 32 (the generator's 144 / 272-channel block inputs)
  // ends with a half-empty chunk: its upper 16 channels arrive as zeros (out-of-range DMA offsets) against zero weights
  // columns: any count; the last column tile may be partly padding: its weights are packed as zeros, its bias / mask / residual reads
  // and its stores are bounded at 16-byte groups (8 bf16 / 4 fp32 channels: `out` keeps its channels padded to that, the pad lanes get
  // act(0 + 0) = 0)
  // (K: any count -- the patch loader bounds 8-channel groups, the packer zero-fills k >= Cin; a source stores its channels padded to 8)
  if (Cin < 1 || Cout < 1) return false;
  const int NT = (Cout + 31) / 32;
  const int n4 = NT / 4, rem = NT % 4;
  if (n4 + (rem ? 1 : 0) > P2_MAXP) return false;
  pl.npass = n4 + (rem ? 1 : 0);
  long long off = 0;
  int t0 = 0;
  for (int i = 0; i < pl.npass; ++i) {
    pl.ntp[i] = i < n4 ? 4 : rem;
    pl.tile0[i] = t0;
    t0 += pl.ntp[i];
    pl.woff[i] = (unsigned)off;
    off += (long long)((Cin + 31) / 32) * 9 * pl.ntp[i] * 2048;
  }
  pl.bytes = off;
  return off < (long long)0xFFFFFFF0;
}

// ------------------------------------------------------------------------------------------------ weight packer
// mode 0 (forward):        w [Cout][Cin][3][3]; column = output channel, k = input channel
// mode 1 (data gradient):  w [Ck][Ccol][3][3] (the forward's OIHW weight: Ck = its output channels = channels of dY, Ccol =
//                          its input channels = columns), taps flipped
// mode 2 (data gradient over a PAIR): w, w2 [Cp][Ccol][3][3]; k < Cp -> w, k >= Cp -> w2 ([dgamma | dbeta]), taps flipped
struct P2PackParams {
  P2Plan pl;
  int mode, Cin, Cout, Cp;
  const float* w;
  const float* w2;
  const float* sigma;     // optional: weights are divided by sigma[0] (spectral norm)
  float wscale;
  unsigned short* out;
};

__global__ __launch_bounds__(256) void p2_pack_kernel(const P2PackParams p) {
  const long long G = (long long)blockIdx.x * blockDim.x + threadIdx.x;
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
  const int chunk = kt / 9, tap = kt - 9 * chunk;
  const int kh = tap / 3, kw = tap - 3 * kh;
  const int col = (p.pl.tile0[pass] + j) * 32 + (lane & 31);
  const int k0 = chunk * 32 + s * 16 + (lane >> 5) * 8;
  const float sc = p.wscale / (p.sigma ? p.sigma[0] : 1.f);
  unsigned short v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    float w = 0.f;
    if (col < p.Cout && k < p.Cin) {
      if (p.mode == 0) w = p.w[(((size_t)col * p.Cin + k) * 3 + kh) * 3 + kw];
      else if (p.mode == 1) w = p.w[(((size_t)k * p.Cout + col) * 3 + (2 - kh)) * 3 + (2 - kw)];
      else w = (k < p.Cp ? p.w : p.w2)[(((size_t)(k < p.Cp ? k : k - p.Cp) * p.Cout + col) * 3 + (2 - kh)) * 3 + (2 - kw)];
    }
    v[e] = f2bf(w * sc);
  }
  uint4 o;
  o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
  o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
  reinterpret_cast<uint4*>(p.out)[G] = o;
}

// ------------------------------------------------------------------------------------------------ the kernel
constexpr int p2_wait(int vm) { return (vm & 15) | (7 << 4) | (0 << 8) | ((vm >> 4) << 14); }   // vmcnt(vm) lgkmcnt(0)

struct P2Tile { int n, y0, x0; };
__device__ __forceinline__ P2Tile p2_tile(const P2Params& p, int bid) {
  const int tx = (p.W + 15) >> 4, ty = (p.H + 15) >> 4;
  const int mt = xcd_remap(bid, p.m_tiles);
  P2Tile t;
  t.n = mt / (tx * ty);
  const int rr = mt - t.n * (tx * ty);
  t.y0 = (rr / tx) << 4;
  t.x0 = (rr % tx) << 4;
  return t;
}

typedef __bf16 p2_bf16x4 __attribute__((ext_vector_type(4)));

// piece `pp` (0..22; 23 folds back: same bytes to the same place) of the 32-channel chunk `chunk` of the tile's halo patch
// -> patch buffer `buf`.  16 halo pixels x 4 groups of 8 channels per piece (linear patch order, pitch 20); the 16-byte
// groups of a pixel are XOR-swizzled by (hx >> 2) & 3 on the SOURCE side.  Out of the image / beyond pixel 360: zeros.
template <int NTP>
__device__ __forceinline__ void p2_patch_piece(const P2Params& p, unsigned char* const smem, const rsrc_t a_rsrc, const P2Tile T, const int chunk,
                                               const int buf, int pp, const int lane) {
  pp = pp < 23 ? pp : 22;
  const int P = pp * 16 + (lane >> 2), g = lane & 3;
  const int hy = (P * 3277) >> 16, hx = P - P2_PW * hy;              // P / 20, P % 20 (P < 368)
  const int y = T.y0 - 1 + hy, x = T.x0 - 1 + hx;
  const int gs = g ^ ((hx >> 2) & 3);
  const bool ok = hy < 18 && hx < 18 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W && chunk * 32 + gs * 8 < p.Cin;
  const unsigned off = ((unsigned)(y * p.W + x) * (unsigned)p.src_cs + (unsigned)(p.src_co + chunk * 32 + gs * 8)) * 2u;
#ifdef P2_EXP_L2PATCH      // timing experiment only (wrong results): every patch piece reads the same 23 KB -- L2 hits, same instruction stream
  dma16(a_rsrc, reinterpret_cast<float*>(smem + p2_patch_off(NTP) + buf * P2_PBUF + pp * 1024), (unsigned)(pp * 1024 + lane * 16), 0u);
#else
  dma16(a_rsrc, reinterpret_cast<float*>(smem + p2_patch_off(NTP) + buf * P2_PBUF + pp * 1024), ok ? off : 0xFFFFFFF0u, 0u);
#endif
}

// The head of a (tile, pass): chunk 0 of the patch -> buffer 0 (6 pieces per wave), k-tiles 0 / 1 -> ring stages 0 / 1.
template <int NTP>
__device__ __forceinline__ void p2_head(const P2Params& p, const int pass, unsigned char* const smem, const P2Tile T, const int wave,
                                        const int lane) {
  constexpr int NPW = NTP * 2, NBW = (NPW + 3) / 4;
  const rsrc_t a_rsrc = make_rsrc(reinterpret_cast<const char*>(p.src) + (size_t)T.n * p.src_bytes, p.src_bytes);
  const rsrc_t w_rsrc = make_rsrc(p.wp, p.w_bytes);
#pragma unroll
  for (int k = 0; k < 6; ++k) p2_patch_piece<NTP>(p, smem, a_rsrc, T, 0, 0, k * 4 + wave, lane);
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
      int idx = wave + 4 * k;
      idx = idx < NPW ? idx : (NPW >= 4 ? idx - 4 : idx % NPW);
      dma16(w_rsrc, reinterpret_cast<float*>(smem + q * p2_sb(NTP) + idx * 1024), (unsigned)lane * 16u,
            p.woff[pass] + (unsigned)q * (unsigned)(NPW * 1024) + (unsigned)idx * 1024u);
    }
}

// EPI: what the epilogue can be asked for -- 0: bias + activation; 1: + mask; 2: mask and / or residual in any combination (the
// lean instances keep the registers of the others' operands out of the allocation)
template <int NTP, int EPI>
__device__ __forceinline__ void p2_pass(const P2Params& p, const int pass, unsigned char* const smem, const P2Tile T, const int bid,
                                        const bool load_consts, const bool wait_all, const bool first, const bool last, const int nxt_pass,
                                        const P2Tile NT_) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  constexpr int NPW = NTP * 2;                               // 1-KB pieces (column tile, k-step) of a k-tile
  constexpr int NBW = (NPW + 3) / 4;                         // DMA instructions per wave per k-tile (NTP 4, 3: 2; NTP 2, 1: 1)
  constexpr int NST = 4 * NTP;                               // global stores of one epilogue per wave (bf16 out; fp32: 8 NTP)
  unsigned char* const ring = smem;
  float* const cbuf = reinterpret_cast<float*>(smem + p2_cb_off(NTP));
  const rsrc_t w_rsrc = make_rsrc(p.wp, p.w_bytes);
  const rsrc_t a_rsrc = make_rsrc(reinterpret_cast<const char*>(p.src) + (size_t)T.n * p.src_bytes, p.src_bytes);
  const unsigned wbase = p.woff[pass];
  const int tile0 = p.tile0[pass];
  const int pt_n = T.n, pt_y0 = T.y0, pt_x0 = T.x0;
  const int nchunk = p.nchunk;

  auto dma_w = [&](const int kt, const int st, const int k) {
    int idx = wave + 4 * k;                                  // (a wave beyond the last piece re-requests an earlier one: same bytes, same place)
    idx = idx < NPW ? idx : (NPW >= 4 ? idx - 4 : idx % NPW);
    dma16(w_rsrc, reinterpret_cast<float*>(ring + st * p2_sb(NTP) + idx * 1024), (unsigned)lane * 16u,
          wbase + (unsigned)kt * (unsigned)(NPW * 1024) + (unsigned)idx * 1024u);
  };

  // every wave is done with the previous (tile, pass): its epilogue's staging scratch lives in patch buffer 1, its bias in cbuf
  __builtin_amdgcn_s_waitcnt(p2_wait(63));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (p.tlog && tid == 0 && first) p.tlog[(size_t)bid * 8 + 0] = wall_clock64();
  if (load_consts && tid < NTP * 32) cbuf[tid] = (p.bias && tile0 * 32 + tid < p.Cout) ? p.bias[tile0 * 32 + tid] : 0.f;

  const int ty = 4 * wave + (l31 >> 4), tx = l31 & 15;
  // fragment addresses: pixel (ty + kh, tx + kw) of the patch, 16-byte group (2 s + lh) ^ ((hx >> 2) & 3)
  const unsigned a_l = (unsigned)((ty * P2_PW + tx) * 64);
  unsigned axk[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) axk[kw] = (unsigned)((lh ^ (((tx + kw) >> 2) & 3)) << 4);
  const unsigned char* const b_lb = ring + lane * 16;

  // The head (chunk 0, k-tiles 0 and 1) has landed.  Behind it in this wave's queue sit only the previous pass's epilogue
  // stores (they need not drain) -- unless this pass loaded constants or is the block's first
  if (wait_all || load_consts) __builtin_amdgcn_s_waitcnt(p2_wait(0));
  else if (p.out_f32) __builtin_amdgcn_s_waitcnt(p2_wait(2 * NST));
  else __builtin_amdgcn_s_waitcnt(p2_wait(NST));
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
  if (p.tlog && tid == 0 && first) p.tlog[(size_t)bid * 8 + 1] = wall_clock64();

  // ---- main loop: chunks x 9 taps.  A fragments double-buffered, B fragments refilled in place (see spade_fused.hip)
  f32x4 fa[2][2], fb[NTP];
  const unsigned char* pbuf = smem + p2_patch_off(NTP);          // patch buffer of the current chunk
  int chunk = 0;
#define P2_READ_A(SET, TAP, S)                                                                             \
  {                                                                                                        \
    const unsigned char* const ap_ = pbuf + a_l + (((TAP) / 3) * P2_PW + ((TAP) % 3)) * 64;                \
    const unsigned ax_ = axk[(TAP) % 3] ^ (unsigned)((S) << 5);                                            \
    fa[SET][0] = *reinterpret_cast<const f32x4*>(ap_ + ax_);                                               \
    fa[SET][1] = *reinterpret_cast<const f32x4*>(ap_ + 2 * P2_PW * 64 + ax_);                              \
  }
#define P2_READ_B(J, ST, S) fb[J] = *reinterpret_cast<const f32x4*>(b_lb + (ST) * p2_sb(NTP) + ((J) * 2 + (S)) * 1024);
#define P2_STEP(SET, REFILL, ST, SN, DMAW, DMAP)                                                           \
  {                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < NTP; ++j) {                                                      \
      acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[j]),              \
                                                          __builtin_bit_cast(bf16x8, fa[SET][0]), acc[0][j], 0, 0, 0); \
      acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[j]),              \
                                                          __builtin_bit_cast(bf16x8, fa[SET][1]), acc[1][j], 0, 0, 0); \
      if constexpr (REFILL) { P2_READ_B(j, ST, SN) }                                                       \
      if constexpr (DMAW) { if (j < NBW) dma_w(kt + 2, (TAP_ + 2) % 3, j); }                               \
      if constexpr (DMAP) { if (j == NTP - 1) p2_patch_piece<NTP>(p, smem, a_rsrc, T, chunk + 1, (chunk + 1) & 1, TAP_ * 4 + wave, lane); } \
    }                                                                                                      \
  }
#define P2_ORDER(NA, REFILL, NW, NP)                                                                       \
  {                                                                                                        \
    if constexpr ((NA) > 0) __builtin_amdgcn_sched_group_barrier(0x100, (NA), 0);                          \
    _Pragma("unroll") for (int j = 0; j < NTP; ++j) {                                                      \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                   \
      if constexpr (REFILL) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                             \
      if (j < (NW)) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                                     \
      if ((NP) > 0 && j == NTP - 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                     \
    }                                                                                                      \
  }
  // ---- the mask of the epilogue (EPI >= 1): 4 bf16 per (pixel, group of the lane's own columns), one column tile at a time.  One
  // offset register per pixel (out of the image: an offset no constant below brings back into range -> zeros), the (j, g) part
  // rides in the instruction's scalar offset; columns beyond Cout read whatever lies there (at worst zeros past the end of the
  // tensor): those lanes store nothing.  EPI == 1 requests column tile 0 under the LAST chunk's tap 6 (the loads are younger
  // than every LDS-DMA piece of the pass: the counted waits behind them allow 8 more in flight) and double-buffers the rest.
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  constexpr unsigned P2_OOB = 0xF0000000u;
  constexpr bool MASK_EARLY = EPI == 1;
  u32x2_t mv[2][2][4];
  unsigned moff[2] = {P2_OOB, P2_OOB};
  rsrc_t m_rsrc = make_rsrc(nullptr, 0u);
  auto mask_setup = [&]() {
    int lane_m = lane;
    asm volatile("" : "+v"(lane_m));        // (lane-only index arithmetic: kept out of the persistent loop's live ranges)
    const int tym = 4 * wave + ((lane_m & 31) >> 4), pxm = pt_x0 + (lane_m & 15);
    m_rsrc = make_rsrc(reinterpret_cast<const char*>(p.mask) + (size_t)pt_n * p.H * p.W * p.mask_cs * 2, (unsigned)((size_t)p.H * p.W * p.mask_cs * 2));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int py = pt_y0 + tym + 2 * i;
      moff[i] = (py < p.H && pxm < p.W) ? ((unsigned)(py * p.W + pxm) * (unsigned)p.mask_cs + (unsigned)(p.mask_co + tile0 * 32 + 4 * (lane_m >> 5))) * 2u : P2_OOB;
    }
  };
  auto load_mask = [&](const int j, u32x2_t (&m)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) m[i][g] = p2_load8(m_rsrc, moff[i], (j * 32 + 8 * g) * 2);
  };
  int kt = 0;                     // current k-tile of the pass (chunk * 9 + tap)
  // One k-tile = tap TAP of the current chunk, in ring stage TAP % 3.  On entry fa[0] / fb hold its k-step 0.
  // LASTC: the chunk is the last of the pass (no further chunk to prefetch; its taps 7 / 8 request no weights, tap 8 has no
  // successor).
  auto ktile = [&](auto tap_c, auto lastc_c) {
    constexpr int TAP_ = decltype(tap_c)::value;
    constexpr bool LASTC = decltype(lastc_c)::value;
    constexpr bool DMAW = !(LASTC && TAP_ >= 7);
    constexpr bool DMAP = !LASTC && TAP_ < 6;
    constexpr bool NEXT = !(LASTC && TAP_ == 8);
    constexpr int ST = TAP_ % 3;
    P2_READ_A(1, TAP_, 1)
    P2_STEP(0, true, ST, 1, DMAW, DMAP)
    P2_ORDER(2, true, DMAW ? NBW : 0, DMAP ? 1 : 0)
    if constexpr (MASK_EARLY && LASTC && TAP_ == 6) {
      mask_setup();
      load_mask(0, mv[0]);
    }
    if constexpr (NEXT) {
      asm volatile("" ::: "memory");
      // k-tile kt + 1 must have landed.  In flight may stay what this wave requested BEHIND it: the previous tap's patch piece (requested
      // late in that tap, after k-tile kt + 1), this tap's k-tile kt + 2 and this tap's patch piece -- a patch piece (HBM, not the
      // L2-resident weight stream) gets two k-tiles to arrive instead of one
      constexpr bool DMAP_PREV = !LASTC && TAP_ >= 1 && TAP_ <= 6;
      constexpr int MASK_FLY = (MASK_EARLY && LASTC && TAP_ >= 6) ? 8 : 0;      // (the mask loads of column tile 0: youngest in the queue)
      __builtin_amdgcn_s_waitcnt(p2_wait((DMAW ? NBW : 0) + (DMAP ? 1 : 0) + (DMAP_PREV ? 1 : 0) + MASK_FLY));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr (TAP_ == 8) {                              // next k-tile: tap 0 of the next chunk, the other patch buffer
        pbuf = smem + p2_patch_off(NTP) + ((chunk + 1) & 1) * P2_PBUF;
      }
      P2_READ_A(0, (TAP_ + 1) % 9, 0)
      P2_STEP(1, true, (TAP_ + 1) % 3, 0, false, false)
      P2_ORDER(2, true, 0, 0)
    } else {
      P2_STEP(1, false, 0, 0, false, false)
    }
    ++kt;
  };
  auto chunk_body = [&](auto lastc_c) {
    ktile(std::integral_constant<int, 0>{}, lastc_c);
    ktile(std::integral_constant<int, 1>{}, lastc_c);
    ktile(std::integral_constant<int, 2>{}, lastc_c);
    ktile(std::integral_constant<int, 3>{}, lastc_c);
    ktile(std::integral_constant<int, 4>{}, lastc_c);
    ktile(std::integral_constant<int, 5>{}, lastc_c);
    ktile(std::integral_constant<int, 6>{}, lastc_c);
    ktile(std::integral_constant<int, 7>{}, lastc_c);
    ktile(std::integral_constant<int, 8>{}, lastc_c);
    ++chunk;
  };
  // k-step 0 of k-tile 0
  P2_READ_A(0, 0, 0)
#pragma unroll
  for (int j = 0; j < NTP; ++j) { P2_READ_B(j, 0, 0) }
#pragma unroll 1
  for (int c = 0; c < nchunk - 1; ++c) chunk_body(std::false_type{});
  chunk_body(std::true_type{});
#undef P2_READ_A
#undef P2_READ_B
#undef P2_STEP
#undef P2_ORDER
  if (p.tlog && tid == 0 && last) p.tlog[(size_t)bid * 8 + 2] = wall_clock64();

  // ---- epilogue.  D layout (swapped operands): lane -> pixel l31; regs 4g..4g+3 -> columns 8g + 4 lh + (0..3) of the tile.
  // The bias is already in the accumulators.  What is left per element is the activation, the mask and the residual -- each
  // behind a wave-uniform branch around a whole column tile, nothing per-element that a descriptor flag decides (measured: the
  // co-resident block's MFMA stream leaves this wave about one VALU issue per MFMA slot, so the epilogue's length is its VALU
  // count; 400 VALU + an LDS round trip per column tile took 4 us each, tools/p2_timeline.py) -- then rows leave straight from
  // the registers: the lane pair (l, l + 32) holds columns 8g .. 8g+3 and 8g+4 .. 8g+7 of ONE pixel, v_permlane32_swap hands the
  // lower lane both halves of an even group and the upper lane both halves of the odd group next to it, and every lane stores 16
  // contiguous bytes (bf16: 8 columns; fp32: 2 x 4 columns).  No LDS staging, no wait on the stores.
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(p2_wait(63));
  __builtin_amdgcn_s_barrier();                // every wave is done with the patch buffers and the weight ring
  asm volatile("" ::: "memory");
  unsigned long long tl_e0 = 0, tl_pack = 0;
  if (p.tlog && tid == 0 && last) tl_e0 = wall_clock64();
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int l31e = lane_e & 31, lhe = lane_e >> 5;
  const size_t img_px = (size_t)p.H * p.W;
  const int oes = p.out_f32 ? 4 : 2;
  const rsrc_t o_rsrc = make_rsrc(reinterpret_cast<const char*>(p.out) + (size_t)pt_n * img_px * p.out_cs * oes,
                                  (unsigned)(img_px * p.out_cs * oes));
  const bool has_mask = EPI == 1 || (EPI == 2 && p.mask != nullptr), has_res = EPI == 2 && p.res != nullptr;
  const int res_es = p.res_f32 ? 4 : 2;
  const rsrc_t r_rsrc = make_rsrc(has_res ? reinterpret_cast<const char*>(p.res) + (size_t)pt_n * img_px * p.res_cs * res_es : nullptr,
                                  has_res ? (unsigned)(img_px * p.res_cs * res_es) : 0u);
  // this lane's two pixels
  const int tye = 4 * wave + (l31e >> 4), pxe = pt_x0 + (l31e & 15);
  unsigned pix[2], roff[2];
  bool pok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int py = pt_y0 + tye + 2 * i;
    pok[i] = py < p.H && pxe < p.W;
    pix[i] = (unsigned)(py * p.W + pxe);
    roff[i] = pok[i] ? (pix[i] * (unsigned)p.res_cs + (unsigned)(p.res_co + tile0 * 32 + 4 * lhe)) * (unsigned)res_es : P2_OOB;
  }
  if (!MASK_EARLY && has_mask) {
    mask_setup();
    load_mask(0, mv[0]);
  }
  // the next (tile, pass) of this block: its head flies while this epilogue computes and stores
  if (nxt_pass >= 0) p2_head<NTP>(p, nxt_pass, smem, NT_, wave, lane);
  const bool relu = p.act == HRV_ACT_RELU, lrelu = p.act == HRV_ACT_LRELU;
  const float sl = p.slope, msl = p.mask_slope;
#pragma unroll
  for (int j = 0; j < NTP; ++j) {
    if (MASK_EARLY && j + 1 < NTP) load_mask(j + 1, mv[(j + 1) & 1]);       // (column tile j's mask has been in flight for a whole tile)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 vv[4], rv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) vv[g] = p2_acc4(acc[i][j], g);
      if (has_res) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (p.res_f32) {
            rv[g] = p2_load16(r_rsrc, roff[i], (j * 32 + 8 * g) * 4);
          } else {
            const u32x2_t h = p2_load8(r_rsrc, roff[i], (j * 32 + 8 * g) * 2);
            rv[g][0] = __builtin_bit_cast(float, h[0] << 16); rv[g][1] = __builtin_bit_cast(float, h[0] & 0xFFFF0000u);
            rv[g][2] = __builtin_bit_cast(float, h[1] << 16); rv[g][3] = __builtin_bit_cast(float, h[1] & 0xFFFF0000u);
          }
        }
        if (!p.res_after) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[g][e] += rv[g][e];
        }
      }
      if (relu) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[g][e] = fmaxf(vv[g][e], 0.f);      // (+0, never v * 0 = -0)
      } else if (lrelu) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[g][e] = fmaxf(vv[g][e], vv[g][e] * sl);
      }
      if (has_mask) {
        // out *= (mask > 0 ? 1 : mask_slope): the sign / zero test runs on the stored 16-bit patterns
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u32x2_t m = mv[MASK_EARLY ? (j & 1) : 0][i][g];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned w_ = m[e >> 1];
            const bool keep = (e & 1) ? ((int)w_ > 0xFFFF) : ((short)(w_ & 0xFFFFu) > 0);
            vv[g][e] = keep ? vv[g][e] : vv[g][e] * msl;
          }
        }
      }
      if (has_res && p.res_after) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[g][e] += rv[g][e];
      }
      const unsigned pbase = pix[i] * (unsigned)p.out_cs + (unsigned)(p.out_co + (tile0 + j) * 32);
      if (!p.out_f32) {
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const u32x2_t X = __builtin_bit_cast(u32x2_t, __builtin_convertvector(vv[2 * gp], p2_bf16x4));
          const u32x2_t Y = __builtin_bit_cast(u32x2_t, __builtin_convertvector(vv[2 * gp + 1], p2_bf16x4));
          const u32x2_t s0 = p2_swap32(X[0], Y[0]), s1 = p2_swap32(X[1], Y[1]);
          const u32x4_t o = {s0[0], s1[0], s0[1], s1[1]};
          const int gcol = 8 * (2 * gp + lhe);
          p2_store16(__builtin_bit_cast(f32x4, o), o_rsrc,
                     (!pok[i] || (tile0 + j) * 32 + gcol >= p.Cout) ? 0xFFFFFFF0u : (pbase + (unsigned)gcol) * 2u);
        }
      } else {
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const f32x4 xa = vv[2 * gp], xb = vv[2 * gp + 1];
          const u32x2_t a0 = p2_swap32(p2_bits(xa[0]), p2_bits(xb[0]));
          const u32x2_t a1 = p2_swap32(p2_bits(xa[1]), p2_bits(xb[1]));
          const u32x2_t a2 = p2_swap32(p2_bits(xa[2]), p2_bits(xb[2]));
          const u32x2_t a3 = p2_swap32(p2_bits(xa[3]), p2_bits(xb[3]));
          const u32x4_t lo_ = {a0[0], a1[0], a2[0], a3[0]}, hi_ = {a0[1], a1[1], a2[1], a3[1]};
          const f32x4 lo4 = __builtin_bit_cast(f32x4, lo_), hi4 = __builtin_bit_cast(f32x4, hi_);
          const int gcol = 8 * (2 * gp + lhe);
          const int colg = (tile0 + j) * 32 + gcol;
          p2_store16(lo4, o_rsrc, (!pok[i] || colg >= p.Cout) ? 0xFFFFFFF0u : (pbase + (unsigned)gcol) * 4u);
          p2_store16(hi4, o_rsrc, (!pok[i] || colg + 4 >= p.Cout) ? 0xFFFFFFF0u : (pbase + (unsigned)gcol + 4u) * 4u);
        }
      }
    }
    if (!MASK_EARLY && has_mask && j + 1 < NTP) load_mask(j + 1, mv[0]);      // (into the registers just consumed)
    if (p.tlog && tid == 0 && last) tl_pack |= ((wall_clock64() - tl_e0) & 0xFFFFull) << (16 * (j & 3));      // diag: ticks since the epilogue began
  }
  if (p.tlog && tid == 0 && last) {
    p.tlog[(size_t)bid * 8 + 7] = tl_pack;
    p.tlog[(size_t)bid * 8 + 6] = wall_clock64();
  }
}

template <int NTP, int EPI>
__global__ __launch_bounds__(256, p2_blocks_per_cu(NTP)) void conv_p2_kernel(const P2Params p, const int pass0, const int pass1) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[p2_lds(NTP)];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  // A unit of work = one tile with the launch's passes pass0 .. pass1 one after the other (the source patch of the later passes
  // comes out of L2) -- or, p.pp (fewer tiles than resident blocks: the 64 x 48 / 32 x 24 levels), ONE (tile, pass): the passes
  // of a tile run on different CUs at the same time
  const int npg = pass1 - pass0;
  const int units = p.pp ? p.m_tiles * npg : p.m_tiles;
  auto unit_tile = [&](const int u) { return p.pp ? u / npg : u; };
  auto unit_pass = [&](const int u) { return p.pp ? pass0 + u % npg : pass0; };
  if ((int)blockIdx.x < units) p2_head<NTP>(p, unit_pass(blockIdx.x), smem, p2_tile(p, unit_tile(blockIdx.x)), wave, lane);
  int c_pass = -1;
#pragma unroll 1
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int bid = unit_tile(u);
    if (p.tlog && threadIdx.x == 0) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      p.tlog[(size_t)bid * 8 + 4] = ((unsigned long long)xcc << 32) | hw;
      p.tlog[(size_t)bid * 8 + 5] = blockIdx.x;
    }
    const P2Tile T = p2_tile(p, bid);
    const int nu = u + gridDim.x;
    const P2Tile TN = p2_tile(p, unit_tile(nu < units ? nu : u));
    const int pa = unit_pass(u), pb = p.pp ? pa + 1 : pass1;
#pragma unroll 1
    for (int pass = pa; pass < pb; ++pass) {
      const bool lastp = pass == pb - 1;
      const int nxt_pass = !lastp ? pass + 1 : (nu < units ? unit_pass(nu) : -1);
      const bool lc = c_pass != pass;
      c_pass = pass;
      p2_pass<NTP, EPI>(p, pass, smem, T, bid, lc, u == (int)blockIdx.x && pass == pa, pass == pa, lastp, nxt_pass, lastp ? TN : T);
    }
    if (p.tlog) {
      __builtin_amdgcn_s_waitcnt(p2_wait(0));
      if (threadIdx.x == 0) p.tlog[(size_t)bid * 8 + 3] = wall_clock64();
    }
  }
}

}  // namespace hrv

using namespace hrv;

extern "C" int64_t hrv_conv_p2_packed_bytes(int32_t Cin, int32_t Cout) {
  P2Plan pl;
  if (!p2_plan(Cin, Cout, pl)) return -1;
  return pl.bytes;
}

extern "C" int hrv_conv_p2_supported(int32_t Cin, int32_t Cout, int32_t N, int32_t H, int32_t W) {
  P2Plan pl;
  if (!p2_plan(Cin, Cout, pl)) return 0;
  const int64_t tiles = (int64_t)N * ((H + 15) / 16) * ((W + 15) / 16);
  // two blocks per CU; measured down to 1.5 tiles per CU (VGG19's 128 x 96 level at 8 images: 384 tiles, 908 -> 1050+ TF/s) the kernel
  // beats the generic tiles clearly; round 5 took the threshold to 0.75 tiles per CU (the generator's 128 x 96 level at 4 images: 192
  // tiles) on a same-box alternating A/B of the whole iteration -- 72.28 -> 71.96 ms, three rounds, every pair in favour
  // (profiles/r05_ab_p2_threshold.txt).  HRV_CONV_P2_MIN_TILES_X4: the threshold in quarter-tiles per CU
  const char* e = hrv::env("HRV_CONV_P2_MIN_TILES_X4");      // (cached by hrv::env: no getenv here after the first call)
  int q4 = e ? atoi(e) : 3;
  if (q4 < 1) q4 = 3;
  // (round 6: counted in UNITS of work -- a level with fewer tiles than resident blocks spreads the column passes of a tile over
  //  the CUs, so what has to fill the chip is tiles x passes: the 64 x 48 level's 48 tiles x 4 passes of a 512-column layer)
  const int64_t units = tiles < 2 * (int64_t)persistent_cus() ? tiles * pl.npass : tiles;
  return 4 * units >= q4 * (int64_t)persistent_cus() ? 1 : 0;
}

extern "C" int hrv_conv_p2_pack_dev(int32_t mode, const float* w, const float* w2, int32_t Cin, int32_t Cout, const float* sigma, float wscale,
                                    void* out, hrv_stream_t stream) {
  HRV_REQUIRE(mode >= 0 && mode <= 2 && w && out && (mode != 2 || w2), "conv_p2_pack: bad arguments (mode %d)", mode);
  P2PackParams pp;
  HRV_REQUIRE(p2_plan(Cin, Cout, pp.pl), "conv_p2_pack: unsupported shape (K %d, columns %d)", Cin, Cout);
  HRV_REQUIRE(((uintptr_t)out & 15) == 0, "conv_p2_pack: out must be 16-byte aligned");
  HRV_REQUIRE(mode != 2 || Cin % 2 == 0, "conv_p2_pack: pair mode needs an even K");
  pp.mode = mode; pp.Cin = Cin; pp.Cout = Cout; pp.Cp = Cin / 2;
  pp.w = w; pp.w2 = w2; pp.sigma = sigma; pp.wscale = wscale; pp.out = (unsigned short*)out;
  const long long groups = pp.pl.bytes / 16;
  hipLaunchKernelGGL(p2_pack_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pp);
  return check_launch("p2_pack_kernel");
}

extern "C" int hrv_conv_p2_bf16(const hrv_conv_p2_t* d, hrv_stream_t stream) {
  HRV_REQUIRE(d != nullptr, "conv_p2: null descriptor");
  P2Plan pl;
  HRV_REQUIRE(p2_plan(d->Cin, d->Cout, pl), "conv_p2: unsupported shape (K %d, columns %d)", d->Cin, d->Cout);
  HRV_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && (int64_t)d->N * d->H * d->W < ((int64_t)1 << 31), "conv_p2: bad extent");
  HRV_REQUIRE(d->src && d->w_packed && d->out, "conv_p2: null pointer");
  HRV_REQUIRE(d->src_cstride % 8 == 0 && d->src_coff % 8 == 0 && d->src_coff + ((d->Cin + 7) & ~7) <= d->src_cstride, "conv_p2: source slice");
  const int64_t sbytes = (int64_t)d->H * d->W * d->src_cstride * 2;
  HRV_REQUIRE(sbytes < (int64_t)0xFFFFFFF0, "conv_p2: one image of the source exceeds the 32-bit buffer range");
  const int oes = d->out_f32 ? 4 : 2, oal = d->out_f32 ? 4 : 8;
  HRV_REQUIRE(d->out_cstride % oal == 0 && d->out_coff % oal == 0 && d->out_coff + ((d->Cout + oal - 1) / oal) * oal <= d->out_cstride,
              "conv_p2: out slice (channels padded to 16 bytes)");
  const int c4 = (d->Cout + 3) & ~3;
  HRV_REQUIRE(d->residual == nullptr || (d->res_cstride % 4 == 0 && d->res_coff % 4 == 0 && d->res_coff + c4 <= d->res_cstride &&
                                         ((uintptr_t)d->residual & 15) == 0),
              "conv_p2: residual slice");
  HRV_REQUIRE((int64_t)d->H * d->W * d->out_cstride * oes < (int64_t)0xFFFFFFF0, "conv_p2: one image of `out` exceeds 4 GB");
  HRV_REQUIRE((((uintptr_t)d->src | (uintptr_t)d->w_packed | (uintptr_t)d->out) & 15) == 0 && ((uintptr_t)d->bias & 3) == 0, "conv_p2: alignment");
  HRV_REQUIRE(d->mask == nullptr || (d->mask_cstride % 4 == 0 && d->mask_coff % 4 == 0 && ((uintptr_t)d->mask & 7) == 0 &&
                                     d->mask_coff + c4 <= d->mask_cstride),
              "conv_p2: mask slice");
  P2Params p;
  memset(&p, 0, sizeof(p));
  p.src = d->src; p.src_cs = d->src_cstride; p.src_co = d->src_coff; p.Cin = d->Cin; p.src_bytes = (unsigned)sbytes;
  p.N = d->N; p.H = d->H; p.W = d->W;
  p.wp = d->w_packed; p.w_bytes = (unsigned)pl.bytes;
  p.npass = pl.npass; p.nchunk = (d->Cin + 31) / 32;
  for (int i = 0; i < pl.npass; ++i) { p.ntp[i] = pl.ntp[i]; p.tile0[i] = pl.tile0[i]; p.woff[i] = pl.woff[i]; }
  p.m_tiles = d->N * ((d->H + 15) / 16) * ((d->W + 15) / 16);
  p.Cout = d->Cout;
  p.bias = d->bias; p.act = d->act; p.slope = d->act_slope;
  p.res = d->residual; p.res_cs = d->res_cstride; p.res_co = d->res_coff; p.res_f32 = d->res_f32; p.res_after = d->res_after_mask;
  p.mask = d->mask; p.mask_cs = d->mask_cstride; p.mask_co = d->mask_coff; p.mask_slope = d->mask_slope;
  p.out = d->out; p.out_cs = d->out_cstride; p.out_co = d->out_coff; p.out_f32 = d->out_f32;
  p.tlog = diag_tlog(p.m_tiles);
  p.pp = p.m_tiles < 2 * persistent_cus() ? 1 : 0;
  if (p.pp) p.tlog = nullptr;          // (the timeline's slots are per tile)
  for (int a = 0; a < pl.npass;) {
    int b = a;
    while (b < pl.npass && pl.ntp[b] == pl.ntp[a]) ++b;
    const long long units = p.pp ? (long long)p.m_tiles * (b - a) : p.m_tiles;
    const int cap = p2_blocks_per_cu(pl.ntp[a]) * persistent_cus();      // resident blocks: three per CU for single-tile passes
    const int grid = units < cap ? (int)units : cap;
    const dim3 g3(grid), b3(256);
    const hipStream_t st = (hipStream_t)stream;
    const int epi = p.res ? 2 : (p.mask ? 1 : 0);
#define P2_LAUNCH(NTP_)                                                                                                 \
  {                                                                                                                     \
    if (epi == 0) hipLaunchKernelGGL((conv_p2_kernel<NTP_, 0>), g3, b3, 0, st, p, a, b);                                \
    else if (epi == 1) hipLaunchKernelGGL((conv_p2_kernel<NTP_, 1>), g3, b3, 0, st, p, a, b);                           \
    else hipLaunchKernelGGL((conv_p2_kernel<NTP_, 2>), g3, b3, 0, st, p, a, b);                                         \
  }
    if (pl.ntp[a] == 4) P2_LAUNCH(4)
    else if (pl.ntp[a] == 3) P2_LAUNCH(3)
    else if (pl.ntp[a] == 2) P2_LAUNCH(2)
    else P2_LAUNCH(1)
#undef P2_LAUNCH
    a = b;
  }
  return check_launch("conv_p2_kernel");
}
