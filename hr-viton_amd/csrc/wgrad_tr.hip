// Weight gradient of a stride-1 'same' convolution whose operands are STORED in bf16 (mixed-precision training:
// dY = [dgamma|dbeta] / d(conv out), X = actv / a SPADE-modulated activation -- tensors only matrix cores read):
//
//     dW[co][tap][ci] = sum_p dY[p][co] * X[p + tap][ci]            (reference: autograd of nn.Conv2d,
//                                                                     network_generator.py:98-99,117-121,141-143)
//
// The bf16 MFMA wants 8 consecutive k (= PIXELS here) per lane, but NHWC is pixel-major.  conv_wgrad_bf16_kernel
// (conv_bwd.hip) transposes quads in registers: 26 VALU instructions per MFMA, MFMA-busy 13 %
// (profiles/r01_pmc_wgrad_bf16.txt).  This kernel has NO VALU in the staging path:
//   * both operands travel global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds) in their natural [pixel][channel]
//     order: a tile is one image-row segment of 64 pixels, so dY is ONE contiguous run and X (with its KW-1 halo
//     pixels) another; 3-4 stages in flight with counted vmcnt across a fence-less s_barrier;
//   * fragments are read with ds_read_b64_tr_b16: 16 lanes fetch a [4 pixels][16 channels] block and receive it
//     transposed -- lane j holds channel j's 4 pixels, i.e. half an MFMA operand.  Every lane passes its own address,
//     so a tap is just a pixel offset into the X patch (no im2col, no per-tap restaging);
//   * a block owns a (cout tile) x (one kernel row: KW taps x 32*XC input channels) tile of dW and keeps it in
//     registers (TM x TN x 16 accumulators per lane) while it streams its slab of the image; the kernel rows /
//     cout tiles of one slab are neighbouring blocks of one XCD (they re-read the same dY rows out of L2).
// LDS rows are padded to (4 mod 8) 16-byte slots so the four pixel rows of a transposing read fall into the four
// 64-byte bank quarters (pad slots are DMA'd as zeros by out-of-range offsets).
// Partial sums go to the same [S][tap][Cout][CinTot] workspace as the other weight-gradient kernels (fixed-order
// reduce => deterministic).
#include "hrv_common.h"

namespace hrv {

typedef float wt_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 wt_bf16x8 __attribute__((ext_vector_type(8)));
typedef short wt_s16x4 __attribute__((ext_vector_type(4)));
typedef short wt_s16x8 __attribute__((ext_vector_type(8)));

struct WgradTrParams {
  const void* dy; int dy_cs, dy_co, Cout;
  const void* x; int x_cs, x_co, x_C;       // x_C: channels of this source, multiple of 8
  int N, H, W, KH, KW, pad;
  int CinTot, ci_base, ci_real;
  int co_tiles, col_tiles, S;               // col tile = (kernel row kh, group range)
  int gpt;                                  // 32-channel groups per tap = ceil(x_C / 32)
  int row_mode;                             // 1: a block's column groups are the KW * gpt groups of ONE kernel row (the rest of its
                                            //    WN * TN group slots idle) -- sources whose width is not 128
  int tiles_per_row, n_tiles;               // 64-pixel row segments
  float* ws;
  float* bias_ws;                           // [S][Cout] column sums of dY (bias gradient), or null
};

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t wt_rsrc_t;
__device__ __forceinline__ wt_rsrc_t wt_make_rsrc(const void* base) {
  // every out-of-image lane is masked explicitly (voffset = 0xFFFFFFF0 >= num_records => the DMA writes zeros);
  // in-range voffsets are small, the tile position travels in the (unchecked) scalar offset
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0x7FFFFFF0, 0x00020000);
}
__device__ __forceinline__ void wt_dma16(wt_rsrc_t r, unsigned char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ wt_s16x4 wt_tr_read(const unsigned char* lds) {
  typedef __attribute__((address_space(3))) wt_s16x4 lds_v;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)lds);
}
#else
struct wt_rsrc_t { int unused; };
__device__ inline wt_rsrc_t wt_make_rsrc(const void*) { return wt_rsrc_t{0}; }
__device__ inline void wt_dma16(wt_rsrc_t, unsigned char*, unsigned, unsigned) {}
__device__ inline wt_s16x4 wt_tr_read(const unsigned char*) { return wt_s16x4{0, 0, 0, 0}; }
#endif

constexpr int wt_pad_slots(int s) { return s + ((4 - (s & 7)) & 7); }   // next count == 4 (mod 8)

// TM x 32 couts and TN groups (of 32 (tap, ci) columns) per wave; WM x WN waves; XC = 32-channel chunks of X a block
// stages; PR = image rows of the X patch: 1 (all groups of a block lie in ONE kernel row: the 128-channel sources) or
// KH (a block covers every tap: the thin convolutions of the 1024x768 level, 32..96 channels on either side).
template <int TM, int TN, int WM, int WN, int XC, int PR>
__global__ __launch_bounds__(256) void conv_wgrad_tr_kernel(const WgradTrParams p) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int TW = 64;                          // pixels per tile
  // LDS rows [pixel][channel], padded to (4 mod 8) 16-byte slots: a transposing read of a 32-lane group touches 4
  // pixel rows x 64 bytes, and banks are (address / 4) mod 64, so the four rows must start in four different 64-byte
  // quarters of the 256-byte bank line.  Measured on dense 256-byte X rows (profiles/r02_pmc_wgrad_tr.txt, first
  // build): SQ_LDS_BANK_CONFLICT = 53 % of SQ_LDS_IDX_ACTIVE = exactly the 4-way conflict of the 6 X reads per k-step
  // next to 10 conflict-free dY reads on 320-byte rows.
  constexpr int RDY = wt_pad_slots(4 * TM * WM);  // 16-byte slots per dY pixel row
  constexpr int RX = wt_pad_slots(4 * XC);        // 16-byte slots per X patch pixel
  constexpr int PXMAX = TW + 2;                   // patch pixels (KW <= 3)
  constexpr int NDY = RDY;                        // dY DMA instructions per tile (64 pixels x RDY slots / 64 lanes)
  static_assert(NDY % 4 == 0, "dY instructions split evenly over the waves");
  constexpr int NDYW = NDY / 4;
  constexpr int NX = (PR * PXMAX * RX + 63) / 64; // X DMA instructions per tile
  constexpr int NXW = (NX + 3) / 4;               // per wave (the last ones may repeat instruction NX-1: benign)
  constexpr int DYB = NDY * 1024, XB = NX * 1024, STAGE = DYB + XB;
  constexpr int NS = (163840 / STAGE) >= 4 ? 4 : (163840 / STAGE);
  static_assert(NS >= 2, "at least two stages must fit the 160 KB LDS");
  constexpr int NPW = NDYW + NXW;                 // DMA instructions per wave per stage
  static_assert(NPW * (NS - 2) < 64, "vmcnt is a 6-bit counter");
  constexpr int WAIT_RUN = ((NPW * (NS - 2)) & 15) | (7 << 4) | (0 << 8) | (((NPW * (NS - 2)) >> 4) << 14);
  constexpr int WAIT_ALL = 0 | (7 << 4) | (0 << 8);
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int g = lane >> 4, i16 = lane & 15, l31 = lane & 31, lh = lane >> 5;

  // logical block id: slab-major, the (cout tile, column tile) jobs of one slab are neighbours on one XCD
  int b = xcd_remap(blockIdx.x, p.co_tiles * p.col_tiles * p.S);
  const int ct = b % p.col_tiles; b /= p.col_tiles;
  const int cot = b % p.co_tiles;
  const int s = b / p.co_tiles;
  const int co0 = cot * (32 * TM * WM);
  constexpr int NGB = WN * TN;                    // column groups of this block
  const int gidx0 = p.row_mode ? ct * p.KW * p.gpt : ct * NGB;   // first global group (tap-major: tap * gpt + chunk)
  const int tap0 = gidx0 / p.gpt;
  const int kh = PR == 1 ? tap0 / p.KW : 0;       // first kernel row of the patch
  const int chunk_lo = (NGB >= p.gpt) ? 0 : gidx0 % p.gpt;   // first 32-channel chunk staged (whole taps: all of them)

  const int t_begin = (int)(((long long)p.n_tiles * s) / p.S);
  const int t_end = (int)(((long long)p.n_tiles * (s + 1)) / p.S);

  // ---- DMA lane constants.  The buffer resources are based at the first image row of this block's slab (64-bit
  // pointer arithmetic once per block), so the per-tile scalar offsets stay small whatever the tensor size (the
  // 384-channel actv tensor of up_4 is 2.4 GB).
  const int r_base = t_begin / p.tiles_per_row;                    // global row index n*H + y of the slab's first tile
  const wt_rsrc_t dy_rsrc = wt_make_rsrc((const char*)p.dy + ((long long)r_base * p.W * p.dy_cs + p.dy_co + co0) * 2);
  // X base shifted back by `pad` rows and `pad` pixels: patch pixel 0 of a tile is image column x0 - pad, its row may
  // be row - pad (both masked when outside the image; the address is never dereferenced then)
  const wt_rsrc_t x_rsrc = wt_make_rsrc((const char*)p.x + (((long long)(r_base - p.pad) * p.W - p.pad) * p.x_cs + p.x_co + chunk_lo * 32) * 2);
  unsigned dy_voff[NDYW];
  int dy_p[NDYW];
#pragma unroll
  for (int q = 0; q < NDYW; ++q) {
    const int slot = 64 * (wave + 4 * q) + lane;
    const int pp = slot / RDY, sl = slot - pp * RDY;
    const bool ok = sl < 4 * TM * WM && co0 + 8 * sl < p.Cout;
    dy_p[q] = ok ? pp : 1 << 20;                                   // pixel of the tile (>= any width: never valid)
    dy_voff[q] = (unsigned)((pp * p.dy_cs + 8 * sl) * 2);
  }
  unsigned x_voff[NXW];
  int x_p[NXW], x_row[NXW];
  const int px_used = TW + p.KW - 1;
#pragma unroll
  for (int q = 0; q < NXW; ++q) {
    int j = wave + 4 * q;
    j = j < NX ? j : NX - 1;
    const int slot = 64 * j + lane;
    const int pq = slot / RX, sl = slot - pq * RX;                  // patch pixel (row-major over PR rows), slot
    const int prow = pq / PXMAX, pp = pq - prow * PXMAX;
    const bool ok = prow < PR && pp < px_used && sl < 4 * XC && (chunk_lo * 32 + 8 * sl) < p.x_C;
    x_p[q] = ok ? pp : 1 << 20;
    x_row[q] = prow < PR ? prow : 0;
    x_voff[q] = (unsigned)(((prow * p.W + pp) * p.x_cs + 8 * sl) * 2);
  }

  // ---- fragment lane constants (bytes inside a stage)
  //  a (dY): pixel 8*(g>>1) + (i16>>2) (+4 for the second read, +16 per k-step), channels wm*TM*32 + 16*(g&1) + 4*(i16&3) (+32 per tm)
  const int a_base = (8 * (g >> 1) + (i16 >> 2)) * (RDY * 16) + (wm * TM * 32 + 16 * (g & 1) + 4 * (i16 & 3)) * 2;
  int b_base[TN], b_tap[TN], b_chunk[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int gi = gidx0 + wn * TN + j;
    int tap = gi / p.gpt, chunk = gi - tap * p.gpt;
    if (p.row_mode && wn * TN + j >= p.KW * p.gpt) tap = p.KH * p.KW;   // an idle slot of a row-aligned block
    // groups past the last tap (and idle slots) are computed on the first staged tap / chunk and dropped
    const bool live = tap < p.KH * p.KW;
    const int tc = live ? tap : kh * p.KW;
    if (!live) chunk = chunk_lo;
    const int khj = tc / p.KW, kw = tc - khj * p.KW;
    b_tap[j] = tap; b_chunk[j] = chunk;
    b_base[j] = DYB + ((khj - kh) * PXMAX + 8 * (g >> 1) + (i16 >> 2) + kw) * (RX * 16) +
                ((chunk - chunk_lo) * 32 + 16 * (g & 1) + 4 * (i16 & 3)) * 2;
  }

  wt_f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Bias gradient = column sums of dY: one extra MFMA per k-step against a constant B fragment whose column 0 is all
  // ones (D[co][0] = sum_k dY[k][co]).  The TM cout tiles of a slab are spread over the 4 waves of its kernel-row
  // blocks (wave w of the kh block takes tile kh*4 + w), so no wave carries more than one extra MFMA per 15.
  const int bias_i = (p.bias_ws != nullptr && ct < 3) ? ct * 4 + wave : -1;      // wave-uniform
  wt_f32x16 acc_b;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc_b[e] = 0.f;
  wt_bf16x8 ones;
  {
    const short one = l31 == 0 ? (short)0x3F80 : (short)0;
    const wt_s16x8 o8 = {one, one, one, one, one, one, one, one};
    ones = __builtin_bit_cast(wt_bf16x8, o8);
  }

  // tile t -> (image row r = n*H + y, segment xt)
  auto issue = [&](int t, int buf) {
    const int r = t / p.tiles_per_row, xt = t - r * p.tiles_per_row;
    const int x0 = xt * TW;
    const int n = r / p.H, y = r - n * p.H;
    unsigned char* sb = smem + buf * STAGE;
    {
      const unsigned soff = (unsigned)(((r - r_base) * p.W + x0) * p.dy_cs * 2);
      const int lim = p.W - x0;                                    // valid pixels of this segment
#pragma unroll
      for (int q = 0; q < NDYW; ++q)
        wt_dma16(dy_rsrc, sb + (wave + 4 * q) * 1024, dy_p[q] < lim ? dy_voff[q] : 0xFFFFFFF0u, soff);
    }
    {
      // patch row `prow` is image row y + kh + prow - pad of this sample; the scalar offset points at patch row 0
      // (never negative: the resource base sits `pad` rows before the slab), invalid rows / columns are masked per lane
      unsigned rowmask = 0;
#pragma unroll
      for (int pr = 0; pr < PR; ++pr) rowmask |= ((unsigned)(y + kh + pr - p.pad) < (unsigned)p.H) ? (1u << pr) : 0u;
      const unsigned soff = (unsigned)(((r + kh - r_base) * p.W + x0) * p.x_cs * 2);
      const int lo = p.pad - x0, hi = p.W - x0 + p.pad;            // patch pixel pp is image column x0 - pad + pp
#pragma unroll
      for (int q = 0; q < NXW; ++q) {
        int j = wave + 4 * q;
        j = j < NX ? j : NX - 1;
        const bool ok = ((rowmask >> x_row[q]) & 1u) && x_p[q] >= lo && x_p[q] < hi;
        wt_dma16(x_rsrc, sb + DYB + j * 1024, ok ? x_voff[q] : 0xFFFFFFF0u, soff);
      }
    }
  };

  // ---- fragment reads: inline asm (the ds_read_tr16 builtin makes hipcc wait vmcnt(0) for every pending LDS-DMA before
  // the first read of a k-step, which serialises the pipeline; plain asm reads are invisible to that pass), so the
  // LDS counter is managed by hand: reads of k-step k+1 are issued in two halves around the MFMAs of k-step k,
  // "lgkmcnt(half)" at the top of a step says the CURRENT step's fragments have all landed (LDS returns in order).
  constexpr int NR = 2 * (TM + TN);                 // tr reads per k-step
  constexpr int NH1 = NR / 2, NH2 = NR - NH1;
  static_assert(NH1 <= 15, "lgkmcnt is a 4-bit counter");
  constexpr int KSTEPS = TW / 16;
  static_assert(KSTEPS % 2 == 0, "fragment sets alternate by k-step parity");
  wt_s16x4 fr[2][NR];                               // [set][read]: reads 2i, 2i+1 = a[i] (lo, hi); 2TM + 2j, +1 = b[j]
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
#define WT_READ(DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
  // reads [R0, R1) of k-step KS of the stage at byte address SB (a VGPR base per operand) into set SET
#define WT_READS(SET, KS, R0, R1, ABASE, BBASE)                                                            \
  {                                                                                                        \
    _Pragma("unroll") for (int r = (R0); r < (R1); ++r) {                                                  \
      if (r < 2 * TM) {                                                                                    \
        const int i = r >> 1, hi = r & 1;                                                                  \
        WT_READ(fr[SET][r], ABASE, (KS) * (16 * RDY * 16) + i * 64 + hi * (4 * RDY * 16));                 \
      } else {                                                                                             \
        const int j = (r - 2 * TM) >> 1, hi = r & 1;                                                       \
        WT_READ(fr[SET][r], BBASE[j], (KS) * (16 * RX * 16) + hi * (4 * RX * 16));                         \
      }                                                                                                    \
    }                                                                                                      \
  }
#define WT_FRAG(SET, R) __builtin_bit_cast(wt_bf16x8, __builtin_shufflevector(fr[SET][2 * (R)], fr[SET][2 * (R) + 1], 0, 1, 2, 3, 4, 5, 6, 7))
  // MFMAs [M0, M1) of the TM x TN grid (row-major) on set SET
#define WT_MMAS(SET, M0, M1)                                                                               \
  {                                                                                                        \
    _Pragma("unroll") for (int m = (M0); m < (M1); ++m) {                                                  \
      const int i = m / TN, j = m - i * TN;                                                                \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WT_FRAG(SET, i), WT_FRAG(SET, TM + j), acc[i][j], 0, 0, 0); \
    }                                                                                                      \
    if ((M1) == TM * TN) {                                                                                 \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                       \
        if (bias_i == i) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WT_FRAG(SET, i), ones, acc_b, 0, 0, 0); \
    }                                                                                                      \
  }
  constexpr int WAIT_H1 = 0x3F | (7 << 4) | (NH1 << 8) | (3 << 14);    // lgkmcnt(NH1), vmcnt untouched
  constexpr int WAIT_L0 = 0x3F | (7 << 4) | (0 << 8) | (3 << 14);      // lgkmcnt(0)

  if (t_begin < t_end) {
    // prologue: NS-1 tiles in flight, the first one landed
#pragma unroll
    for (int q = 0; q < NS - 1; ++q)
      if (t_begin + q < t_end) issue(t_begin + q, q);
    if (t_begin + NS - 1 <= t_end) __builtin_amdgcn_s_waitcnt(WAIT_RUN);
    else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int rb = 0, wb = NS - 1;
    unsigned a_addr = lds0 + (unsigned)a_base;
    unsigned b_addr[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) b_addr[j] = lds0 + (unsigned)b_base[j];
    WT_READS(0, 0, 0, NR, a_addr, b_addr)                 // first k-step of the first tile
    for (int t = t_begin; t < t_end; ++t) {
      const bool more = t + NS - 1 < t_end;
      if (more) issue(t + NS - 1, wb);      // that buffer was read in tile t-1: every wave passed the barrier after its reads
      const int nb = rb == NS - 1 ? 0 : rb + 1;
      const unsigned a_next = lds0 + (unsigned)(a_base + nb * STAGE);
      unsigned b_next[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) b_next[j] = lds0 + (unsigned)(b_base[j] + nb * STAGE);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks + 1 < KSTEPS) {
          WT_READS(nxt, ks + 1, 0, NH1, a_addr, b_addr)
          __builtin_amdgcn_s_waitcnt(WAIT_H1);             // set `cur` has landed
        } else {
          // last k-step of the tile: every LDS read of this tile has been issued; once they are back the stage is
          // free, and tile t+1 must have landed before its first fragments are fetched
          if (t + 1 < t_end) {
            if (more) __builtin_amdgcn_s_waitcnt(WAIT_RUN);   // lgkmcnt(0) + this wave's DMA of tile t+1
            else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            WT_READS(nxt, 0, 0, NH1, a_next, b_next)
            __builtin_amdgcn_s_waitcnt(WAIT_H1);
          } else {
            __builtin_amdgcn_s_waitcnt(WAIT_L0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        WT_MMAS(cur, 0, (TM * TN) / 2)
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < KSTEPS) {
          WT_READS(nxt, ks + 1, NH1, NR, a_addr, b_addr)
        } else if (t + 1 < t_end) {
          WT_READS(nxt, 0, NH1, NR, a_next, b_next)
        }
        __builtin_amdgcn_sched_barrier(0);
        WT_MMAS(cur, (TM * TN) / 2, TM * TN)
        __builtin_amdgcn_sched_barrier(0);
      }
      a_addr = a_next;
#pragma unroll
      for (int j = 0; j < TN; ++j) b_addr[j] = b_next[j];
      rb = nb;
      wb = wb == NS - 1 ? 0 : wb + 1;
    }
  }
#undef WT_READ
#undef WT_READS
#undef WT_FRAG
#undef WT_MMAS

  // D[i = cout][j = ci]: col = lane&31 (ci), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (cout)
  const int taps = p.KH * p.KW;
  if (bias_i >= 0 && bias_i < TM && l31 == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = co0 + bias_i * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);      // WM == 1: tile i covers couts co0 + 32 i ..
      if (co < p.Cout) p.bias_ws[(size_t)s * p.Cout + co] = acc_b[e];
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int ci = b_chunk[j] * 32 + l31;
    if (b_tap[j] >= taps || ci >= p.ci_real) continue;
    float* wsp = p.ws + ((size_t)s * taps + b_tap[j]) * p.Cout * p.CinTot;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + (wm * TM + i) * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
        if (co < p.Cout) wsp[(size_t)co * p.CinTot + p.ci_base + ci] = acc[i][j][e];
      }
  }
}

// (bias gradient, second stage: the per-slab column sums are summed by the caller's wgrad_reduce_kernel launch)

// Host side.  Returns 1 when the kernel was launched (partials in `workspace`, *S_out slabs), 0 when the shape is
// not one it serves (the caller falls back to conv_wgrad_bf16_kernel), < 0 on error.
int wgrad_tr_try(const void* dy, int dy_cs, int dy_co, int Cout, const void* x, int x_C, int x_cs, int x_co, int x_C_real,
                 int ci_base, int CinTot, int N, int H, int W, int KH, int KW, int pad, float* workspace,
                 long long workspace_bytes, float* dbias, int dbias_accumulate, hipStream_t st, int* S_out) {
  const char* env = hrv::env("HRV_WGRAD_TR");
  if (env && env[0] == '0') return 0;
  if (KW < 1 || KW > 3 || KH != KW || pad != KH / 2) return 0;
  if ((dy_cs | dy_co | x_cs | x_co | x_C) & 7) return 0;                         // 16-byte DMA granules (Cout itself may be
                                                                                  // anything: rows >= Cout are never written)
  const long long P = (long long)N * H * W;
  // low-resolution levels: weight-bound, old kernel.  HRV_WGRAD_TR_MIN_PIX: the smallest N*H*W this kernel takes -- 8192 since round 5
  // (the 64 x 48 level at 4 images: three same-box pairs of the whole iteration, each 0.16-0.27 ms in favour,
  // profiles/r05_ab_wgrad_tr_min.txt); 32768 before
  const char* emin = hrv::env("HRV_WGRAD_TR_MIN_PIX");
  const long long pmin = emin ? atoll(emin) : 8192;
  if (P < (pmin > 0 ? pmin : 8192) || W < 32) return 0;
  const int gpt = (x_C + 31) / 32;
  const int taps = KH * KW;
  WgradTrParams p;
  p.dy = dy; p.dy_cs = dy_cs; p.dy_co = dy_co; p.Cout = Cout;
  p.x = x; p.x_cs = x_cs; p.x_co = x_co; p.x_C = x_C;
  p.N = N; p.H = H; p.W = W; p.KH = KH; p.KW = KW; p.pad = pad;
  p.CinTot = CinTot; p.ci_base = ci_base; p.ci_real = x_C_real;
  p.gpt = gpt;
  p.tiles_per_row = (W + 63) / 64;
  p.n_tiles = N * H * p.tiles_per_row;
  // shape classes:  0 = 128-channel source, a block = the KW taps of one kernel row x all 128 channels x <= 160 couts
  //                 1..3 = thin layers (<= 32 couts, <= 96 source channels): a block = every tap x every channel
  //                 4..7 = other source widths (round 4): 4: 144 / 160 channels (5 groups), 6: 272 / 288 (9), 7: 256 (8) -- a block =
  //                        the KW taps of one kernel row x all groups x 64 couts (row mode); 5: 64 channels x 64 couts, every tap
  int cls = -1, tm = 0;
  p.row_mode = 0;
  if (gpt == 4 && KW == 3) {
    cls = 0;
    p.co_tiles = (Cout + 159) / 160;
    tm = (((Cout + p.co_tiles - 1) / p.co_tiles) + 31) / 32;                     // 1..5
    p.col_tiles = KH * KW * gpt / 12;                                              // WN 4 x TN 3 groups per block
  } else if (KW == 3 && Cout % 64 == 0 && (gpt == 5 || gpt == 9 || gpt == 8)) {
    cls = gpt == 5 ? 4 : (gpt == 9 ? 6 : 7);
    p.row_mode = 1;
    p.co_tiles = Cout / 64; tm = 2;
    p.col_tiles = KH;
  } else if (KW == 3 && Cout % 64 == 0 && gpt == 2) {
    cls = 5;
    p.co_tiles = Cout / 64; tm = 2;
    p.col_tiles = 1;
  } else if (KW == 2 && Cout % 64 == 0 && gpt == 2) {
    cls = 8;      // 2x2 over <= 64 channels (round 6: PatchGAN's model0 over its space-to-depth image, conv_s2.hip mode 2)
    p.co_tiles = Cout / 64; tm = 2;
    p.col_tiles = 1;
  } else if (Cout <= 32 && taps * gpt <= 28 && gpt <= 3) {
    cls = gpt == 3 ? (taps == 1 ? 3 : 1) : (gpt == 1 && taps == 9 ? 2 : -1);
    p.co_tiles = 1; p.col_tiles = 1; tm = 1;
  }
  if (cls < 0) return 0;
  const int jobs = p.co_tiles * p.col_tiles;
  // one block per CU (a block owns 126-152 KB of LDS): the grid must NOT exceed the CU count, or the surplus blocks
  // run as a second round on an otherwise idle chip (first build: 258 blocks, kernel time 2x the wave lifetime)
  const int n_cu = persistent_cus();
  int S = n_cu / jobs;
  if (S > p.n_tiles / 8) S = p.n_tiles / 8;
  if (S > 256) S = 256;
  if (S < 1) S = 1;
  // per-tile scalar offsets are relative to the slab's first row: the slab's extent must fit 31 bits
  const long long slab_rows = p.n_tiles / S / p.tiles_per_row + 4 + KH;
  if (slab_rows * W * (long long)(dy_cs > x_cs ? dy_cs : x_cs) * 2 >= 0x7FF00000LL) return 0;
  const long long need = ((long long)S * KH * KW * Cout * CinTot + 256LL * Cout) * 4;
  if (workspace_bytes < need) {
    set_error("wgrad_tr: workspace too small (%lld < %lld)", workspace_bytes, need);
    return HRV_ERR_ARG;
  }
  p.S = S; p.ws = workspace;
  p.bias_ws = dbias ? workspace + (size_t)S * KH * KW * Cout * CinTot : nullptr;     // [S][Cout], written by the kh blocks
  const int nblk = jobs * S;
  if (cls == 0) {
    switch (tm) {
      case 1: hipLaunchKernelGGL((conv_wgrad_tr_kernel<1, 3, 1, 4, 4, 1>), dim3(nblk), dim3(256), 0, st, p); break;
      case 2: hipLaunchKernelGGL((conv_wgrad_tr_kernel<2, 3, 1, 4, 4, 1>), dim3(nblk), dim3(256), 0, st, p); break;
      case 3: hipLaunchKernelGGL((conv_wgrad_tr_kernel<3, 3, 1, 4, 4, 1>), dim3(nblk), dim3(256), 0, st, p); break;
      case 4: hipLaunchKernelGGL((conv_wgrad_tr_kernel<4, 3, 1, 4, 4, 1>), dim3(nblk), dim3(256), 0, st, p); break;
      case 5: hipLaunchKernelGGL((conv_wgrad_tr_kernel<5, 3, 1, 4, 4, 1>), dim3(nblk), dim3(256), 0, st, p); break;
      default: return 0;
    }
  } else if (cls == 4) {      // 3x3 over 144 / 160 channels: 15 groups per kernel row -> 4 waves x 4
    hipLaunchKernelGGL((conv_wgrad_tr_kernel<2, 4, 1, 4, 5, 1>), dim3(nblk), dim3(256), 0, st, p);
  } else if (cls == 5) {      // 3x3 over 64 channels: 18 groups -> 4 waves x 5, every tap in one block
    hipLaunchKernelGGL((conv_wgrad_tr_kernel<2, 5, 1, 4, 2, 3>), dim3(nblk), dim3(256), 0, st, p);
  } else if (cls == 6) {      // 3x3 over 272 / 288 channels: 27 groups per kernel row -> 4 waves x 7
    hipLaunchKernelGGL((conv_wgrad_tr_kernel<2, 7, 1, 4, 9, 1>), dim3(nblk), dim3(256), 0, st, p);
  } else if (cls == 7) {      // 3x3 over 256 channels: 24 groups per kernel row -> 4 waves x 6
    hipLaunchKernelGGL((conv_wgrad_tr_kernel<2, 6, 1, 4, 8, 1>), dim3(nblk), dim3(256), 0, st, p);
  } else if (cls == 8) {      // 2x2 over <= 64 channels: 8 groups -> 4 waves x 2, every tap in one block
    hipLaunchKernelGGL((conv_wgrad_tr_kernel<2, 2, 1, 4, 2, 2>), dim3(nblk), dim3(256), 0, st, p);
  } else if (cls == 1) {      // 3x3 over <= 96 channels: 27 groups -> 4 waves x 7
    hipLaunchKernelGGL((conv_wgrad_tr_kernel<1, 7, 1, 4, 3, 3>), dim3(nblk), dim3(256), 0, st, p);
  } else if (cls == 2) {      // 3x3 over 32 channels: 9 groups -> 4 waves x 3
    hipLaunchKernelGGL((conv_wgrad_tr_kernel<1, 3, 1, 4, 1, 3>), dim3(nblk), dim3(256), 0, st, p);
  } else {                    // 1x1 over <= 96 channels: 3 groups -> 4 waves x 1
    hipLaunchKernelGGL((conv_wgrad_tr_kernel<1, 1, 1, 4, 3, 1>), dim3(nblk), dim3(256), 0, st, p);
  }
  int rc = check_launch("conv_wgrad_tr_kernel");
  if (rc) return rc;
  (void)dbias_accumulate;     // the caller's reduce launch sums bias_ws ([S][Cout], right behind the S weight slabs) as well
  *S_out = S;
  return 1;
}

}  // namespace hrv
