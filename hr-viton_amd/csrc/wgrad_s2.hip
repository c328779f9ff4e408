// Weight gradient of PatchGAN's 4x4 stride-2 pad-2 convolution (NLayerDiscriminator, network_generator.py:263-272) over bf16-STORED
// operands, on the skeleton of wgrad_tr.hip (LDS-DMA staging in natural [pixel][channel] order, transposing fragment reads, a block
// keeps its tile of dW in registers while it streams a slab of the image):
//
//     dW[co][kh][kw][ci] = sum_{n,oy,ox} dY[n][oy][ox][co] * X[n][2 oy + kh - 2][2 ox + kw - 2][ci]
//
// A tile is one segment of 64 OUTPUT pixels of one dY row; a block owns ONE kernel row kh x (the 4 kw taps x all 32-channel chunks
// of X) x a cout tile.  Its X patch is the single input row 2 oy + kh - 2, pixels 2 x0 - 2 .. 2 x0 + 127 (130 pixels, one
// contiguous run by LDS-DMA); ds_read_b64_tr_b16 takes every lane's own address, so the four k (= output pixel) rows of a
// transposed 4x16 block are simply TWO patch pixels apart and a tap kw is a one-pixel offset: no im2col, no space-to-depth copy.
// X rows are padded to (2 mod 8) 16-byte slots: the four rows of a transposing read, two pixels apart, then start in the four
// 64-byte quarters of the 256-byte bank line (wgrad_tr.hip pads to 4 mod 8 for rows one pixel apart).
// The generic quad-transposing kernel ran these layers at 250-420 TFLOP/s behind a width-padding copy of dY.
// Partial sums go to the [S][tap][Cout][CinTot] workspace of the other weight-gradient kernels (fixed-order reduce).
#include "hrv_common.h"

namespace hrv {

typedef float ws_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ws_bf16x8 __attribute__((ext_vector_type(8)));
typedef short ws_s16x4 __attribute__((ext_vector_type(4)));
typedef short ws_s16x8 __attribute__((ext_vector_type(8)));

struct WgradS2Params {
  const void* dy; int dy_cs, dy_co, Cout;
  const void* x; int x_cs, x_co, x_C;       // x_C: channels of X, multiple of 8
  int N, H, W, Ho, Wo;
  int CinTot, ci_base, ci_real;
  int co_tiles, S;
  int gpt;                                  // 32-channel groups of X = ceil(x_C / 32) (== XC of the instance)
  int tiles_per_row, n_tiles;               // 64-pixel segments of dY rows
  float* ws;
  float* bias_ws;                           // [S][Cout] column sums of dY (bias gradient), or null
};

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t ws_rsrc_t;
__device__ __forceinline__ ws_rsrc_t ws_make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0x7FFFFFF0, 0x00020000);
}
__device__ __forceinline__ void ws_dma16(ws_rsrc_t r, unsigned char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
#else
struct ws_rsrc_t { int unused; };
__device__ inline ws_rsrc_t ws_make_rsrc(const void*) { return ws_rsrc_t{0}; }
__device__ inline void ws_dma16(ws_rsrc_t, unsigned char*, unsigned, unsigned) {}
#endif

constexpr int ws_pad4(int s) { return s + ((4 - (s & 7)) & 7); }   // next count == 4 (mod 8): dY rows, one pixel apart
constexpr int ws_pad2(int s) { return s + ((2 - (s & 7)) & 7); }   // next count == 2 (mod 8): X rows, read two pixels apart

// TM x 32 couts per block (WM == 1), TN groups per wave over WN = 4 waves: 4 TN = 4 kw x XC chunks
template <int TM, int TN, int XC>
__global__ __launch_bounds__(256) void conv_wgrad_s2_kernel(const WgradS2Params p) {
  static_assert(4 * TN == 4 * XC, "a block covers the 4 kw taps x XC chunks of one kernel row");
  constexpr int TW = 64;                          // output pixels per tile
  constexpr int RDY = ws_pad4(4 * TM);            // 16-byte slots per dY pixel row
  constexpr int RX = ws_pad2(4 * XC);             // 16-byte slots per X patch pixel
  constexpr int PXMAX = 2 * TW + 2;               // patch pixels: 2 x0 - 2 .. 2 x0 + 127
  constexpr int NDY = RDY;                        // dY DMA instructions per tile (64 pixels x RDY slots / 64 lanes)
  static_assert(NDY % 4 == 0, "dY instructions split evenly over the waves");
  constexpr int NDYW = NDY / 4;
  constexpr int NX = (PXMAX * RX + 63) / 64;      // X DMA instructions per tile
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
  const int wn = wave;
  const int g = lane >> 4, i16 = lane & 15, l31 = lane & 31, lh = lane >> 5;

  // logical block id: slab-major, the (cout tile, kernel row) jobs of one slab are neighbours on one XCD
  int b = xcd_remap(blockIdx.x, p.co_tiles * 4 * p.S);
  const int kh = b & 3; b >>= 2;
  const int cot = b % p.co_tiles;
  const int s = b / p.co_tiles;
  const int co0 = cot * (32 * TM);

  const int t_begin = (int)(((long long)p.n_tiles * s) / p.S);
  const int t_end = (int)(((long long)p.n_tiles * (s + 1)) / p.S);

  // ---- DMA lane constants.  The buffer resources are based at the first rows of this block's slab, so the per-tile scalar
  // offsets stay small whatever the tensor size.  X: based two rows and two pixels before input row 2 oy_b of image n_b (never
  // dereferenced outside the image: such lanes are masked)
  const int r_base = t_begin / p.tiles_per_row;                    // dY row index n*Ho + oy of the slab's first tile
  const int n_b = r_base / p.Ho, oy_b = r_base - n_b * p.Ho;
  const long long xrow_b = (long long)n_b * p.H + 2 * oy_b;         // X row index of (n_b, 2 oy_b)
  const ws_rsrc_t dy_rsrc = ws_make_rsrc((const char*)p.dy + ((long long)r_base * p.Wo * p.dy_cs + p.dy_co + co0) * 2);
  const ws_rsrc_t x_rsrc = ws_make_rsrc((const char*)p.x + (((xrow_b - 2) * p.W - 2) * p.x_cs + p.x_co) * 2);
  unsigned dy_voff[NDYW];
  int dy_p[NDYW];
#pragma unroll
  for (int q = 0; q < NDYW; ++q) {
    const int slot = 64 * (wave + 4 * q) + lane;
    const int pp = slot / RDY, sl = slot - pp * RDY;
    const bool ok = sl < 4 * TM && co0 + 8 * sl < p.Cout;
    dy_p[q] = ok ? pp : 1 << 20;                                   // pixel of the tile (>= any width: never valid)
    dy_voff[q] = (unsigned)((pp * p.dy_cs + 8 * sl) * 2);
  }
  unsigned x_voff[NXW];
  int x_p[NXW];
#pragma unroll
  for (int q = 0; q < NXW; ++q) {
    int j = wave + 4 * q;
    j = j < NX ? j : NX - 1;
    const int slot = 64 * j + lane;
    const int pp = slot / RX, sl = slot - pp * RX;                  // patch pixel, slot
    const bool ok = pp < PXMAX && sl < 4 * XC && 8 * sl < p.x_C;
    x_p[q] = ok ? pp : 1 << 20;
    x_voff[q] = (unsigned)((pp * p.x_cs + 8 * sl) * 2);
  }

  // ---- fragment lane constants (bytes inside a stage)
  //  a (dY): pixel 8*(g>>1) + (i16>>2) (+4 for the second read, +16 per k-step), channels 16*(g&1) + 4*(i16&3) (+32 per tm)
  const int a_base = (8 * (g >> 1) + (i16 >> 2)) * (RDY * 16) + (16 * (g & 1) + 4 * (i16 & 3)) * 2;
  //  b (X): group gi = wn*TN + j = kw * XC + chunk; output pixel q -> patch pixel 2 q + kw
  int b_base[TN], b_kw[TN], b_chunk[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int gi = wn * TN + j;
    const int kw = gi / XC, chunk = gi - kw * XC;
    b_kw[j] = kw; b_chunk[j] = chunk;
    b_base[j] = DYB + (2 * (8 * (g >> 1) + (i16 >> 2)) + kw) * (RX * 16) + (chunk * 32 + 16 * (g & 1) + 4 * (i16 & 3)) * 2;
  }

  ws_f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Bias gradient = column sums of dY: one extra MFMA per k-step against a constant B fragment whose column 0 is all ones.  Wave w
  // of the kh == 0 block takes cout tile w.
  const int bias_i = (p.bias_ws != nullptr && kh == 0 && wave < TM) ? wave : -1;      // wave-uniform
  ws_f32x16 acc_b;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc_b[e] = 0.f;
  ws_bf16x8 ones;
  {
    const short one = l31 == 0 ? (short)0x3F80 : (short)0;
    const ws_s16x8 o8 = {one, one, one, one, one, one, one, one};
    ones = __builtin_bit_cast(ws_bf16x8, o8);
  }

  // tile t -> (dY row r = n*Ho + oy, segment xt)
  auto issue = [&](int t, int buf) {
    const int r = t / p.tiles_per_row, xt = t - r * p.tiles_per_row;
    const int x0 = xt * TW;
    const int n = r / p.Ho, oy = r - n * p.Ho;
    unsigned char* sb = smem + buf * STAGE;
    {
      const unsigned soff = (unsigned)(((r - r_base) * p.Wo + x0) * p.dy_cs * 2);
      const int lim = p.Wo - x0;                                   // valid pixels of this segment
#pragma unroll
      for (int q = 0; q < NDYW; ++q)
        ws_dma16(dy_rsrc, sb + (wave + 4 * q) * 1024, dy_p[q] < lim ? dy_voff[q] : 0xFFFFFFF0u, soff);
    }
    {
      // the patch is input row iy = 2 oy + kh - 2 of sample n, pixels 2 x0 - 2 + pp; relative to the resource base (row xrow_b - 2,
      // pixel -2) that is row (n H + 2 oy - xrow_b) + kh, pixel 2 x0 + pp -- never negative
      const int iy = 2 * oy + kh - 2;
      const bool row_ok = (unsigned)iy < (unsigned)p.H;
      const long long rel = ((long long)n * p.H + 2 * oy - xrow_b + kh) * p.W + 2 * x0;
      const unsigned soff = (unsigned)(rel * p.x_cs * 2);
      const int lo = 2 - 2 * x0, hi = p.W - 2 * x0 + 2;             // patch pixel pp is image column 2 x0 - 2 + pp
#pragma unroll
      for (int q = 0; q < NXW; ++q) {
        int j = wave + 4 * q;
        j = j < NX ? j : NX - 1;
        const bool ok = row_ok && x_p[q] >= lo && x_p[q] < hi;
        ws_dma16(x_rsrc, sb + DYB + j * 1024, ok ? x_voff[q] : 0xFFFFFFF0u, soff);
      }
    }
  };

  // ---- fragment reads: inline asm, LDS counter managed by hand (see wgrad_tr.hip)
  constexpr int NR = 2 * (TM + TN);                 // tr reads per k-step
  constexpr int NH1 = NR / 2;
  static_assert(NH1 <= 15, "lgkmcnt is a 4-bit counter");
  constexpr int KSTEPS = TW / 16;
  static_assert(KSTEPS % 2 == 0, "fragment sets alternate by k-step parity");
  ws_s16x4 fr[2][NR];                               // [set][read]: reads 2i, 2i+1 = a[i] (lo, hi); 2TM + 2j, +1 = b[j]
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
#define WS_READ(DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
  // a: 4 pixel rows one pixel apart, +4 pixels for the hi half, +16 per k-step; b: rows two patch pixels apart, +8 / +32 patch pixels
#define WS_READS(SET, KS, R0, R1, ABASE, BBASE)                                                            \
  {                                                                                                        \
    _Pragma("unroll") for (int r = (R0); r < (R1); ++r) {                                                  \
      if (r < 2 * TM) {                                                                                    \
        const int i = r >> 1, hi = r & 1;                                                                  \
        WS_READ(fr[SET][r], ABASE, (KS) * (16 * RDY * 16) + i * 64 + hi * (4 * RDY * 16));                 \
      } else {                                                                                             \
        const int j = (r - 2 * TM) >> 1, hi = r & 1;                                                       \
        WS_READ(fr[SET][r], BBASE[j], (KS) * (32 * RX * 16) + hi * (8 * RX * 16));                         \
      }                                                                                                    \
    }                                                                                                      \
  }
#define WS_FRAG(SET, R) __builtin_bit_cast(ws_bf16x8, __builtin_shufflevector(fr[SET][2 * (R)], fr[SET][2 * (R) + 1], 0, 1, 2, 3, 4, 5, 6, 7))
#define WS_MMAS(SET, M0, M1)                                                                               \
  {                                                                                                        \
    _Pragma("unroll") for (int m = (M0); m < (M1); ++m) {                                                  \
      const int i = m / TN, j = m - i * TN;                                                                \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WS_FRAG(SET, i), WS_FRAG(SET, TM + j), acc[i][j], 0, 0, 0); \
    }                                                                                                      \
    if ((M1) == TM * TN) {                                                                                 \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                       \
        if (bias_i == i) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WS_FRAG(SET, i), ones, acc_b, 0, 0, 0); \
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
    WS_READS(0, 0, 0, NR, a_addr, b_addr)                 // first k-step of the first tile
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
          WS_READS(nxt, ks + 1, 0, NH1, a_addr, b_addr)
          __builtin_amdgcn_s_waitcnt(WAIT_H1);             // set `cur` has landed
        } else {
          // last k-step of the tile: every LDS read of this tile has been issued; once they are back the stage is
          // free, and tile t+1 must have landed before its first fragments are fetched
          if (t + 1 < t_end) {
            if (more) __builtin_amdgcn_s_waitcnt(WAIT_RUN);   // lgkmcnt(0) + this wave's DMA of tile t+1
            else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            WS_READS(nxt, 0, 0, NH1, a_next, b_next)
            __builtin_amdgcn_s_waitcnt(WAIT_H1);
          } else {
            __builtin_amdgcn_s_waitcnt(WAIT_L0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        WS_MMAS(cur, 0, (TM * TN) / 2)
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < KSTEPS) {
          WS_READS(nxt, ks + 1, NH1, NR, a_addr, b_addr)
        } else if (t + 1 < t_end) {
          WS_READS(nxt, 0, NH1, NR, a_next, b_next)
        }
        __builtin_amdgcn_sched_barrier(0);
        WS_MMAS(cur, (TM * TN) / 2, TM * TN)
        __builtin_amdgcn_sched_barrier(0);
      }
      a_addr = a_next;
#pragma unroll
      for (int j = 0; j < TN; ++j) b_addr[j] = b_next[j];
      rb = nb;
      wb = wb == NS - 1 ? 0 : wb + 1;
    }
  }
#undef WS_READ
#undef WS_READS
#undef WS_FRAG
#undef WS_MMAS

  // D[i = cout][j = ci]: col = lane&31 (ci), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (cout)
  if (bias_i >= 0 && l31 == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = co0 + bias_i * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
      if (co < p.Cout) p.bias_ws[(size_t)s * p.Cout + co] = acc_b[e];
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int ci = b_chunk[j] * 32 + l31;
    if (ci >= p.ci_real) continue;
    float* wsp = p.ws + ((size_t)s * 16 + kh * 4 + b_kw[j]) * p.Cout * p.CinTot;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + i * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
        if (co < p.Cout) wsp[(size_t)co * p.CinTot + p.ci_base + ci] = acc[i][j][e];
      }
  }
}

// 1 / 2: the shape class the kernel serves (the conditions wgrad_s2_try applies), 0: none
static int wgrad_s2_class(int Cout, int x_C, int x_cs, int x_co, int dy_cs, int dy_co, int N, int H, int W) {
  const char* env = hrv::env("HRV_WGRAD_S2");
  if (env && env[0] == '0') return 0;
  if ((dy_cs | dy_co | x_cs | x_co | x_C) & 7) return 0;                         // 16-byte DMA granules
  const int Ho = H / 2 + 1, Wo = W / 2 + 1;
  if ((long long)N * Ho * Wo < 8192 || Wo < 32) return 0;
  const int gpt = (x_C + 31) / 32;
  if (gpt == 2 && Cout % 128 == 0) return 1;            // 64 -> 128 (model1): 128 couts x (4 kw x 2 chunks)
  if (gpt == 4 && Cout % 64 == 0) return 2;             // 128 -> 256 (model2): 64 couts x (4 kw x 4 chunks)
  return 0;
}
int wgrad_s2_serves(int Cout, int x_C, int x_cs, int x_co, int dy_cs, int dy_co, int N, int H, int W) {
  return wgrad_s2_class(Cout, x_C, x_cs, x_co, dy_cs, dy_co, N, H, W) != 0 ? 1 : 0;
}

// Host side.  Returns 1 when the kernel was launched (partials in `workspace`, *S_out slabs), 0 when the shape is not one it
// serves (the caller falls back to conv_wgrad_bf16_kernel), < 0 on error.
int wgrad_s2_try(const void* dy, int dy_cs, int dy_co, int Cout, const void* x, int x_C, int x_cs, int x_co, int x_C_real, int ci_base,
                 int CinTot, int N, int H, int W, int Ho, int Wo, float* workspace, long long workspace_bytes, float* dbias, hipStream_t st,
                 int* S_out) {
  if (Ho != H / 2 + 1 || Wo != W / 2 + 1) return 0;
  const int c_ = wgrad_s2_class(Cout, x_C, x_cs, x_co, dy_cs, dy_co, N, H, W);
  if (c_ == 0) return 0;
  const int cls = c_ - 1, tm = c_ == 1 ? 4 : 2;
  const int gpt = (x_C + 31) / 32;
  WgradS2Params p;
  p.dy = dy; p.dy_cs = dy_cs; p.dy_co = dy_co; p.Cout = Cout;
  p.x = x; p.x_cs = x_cs; p.x_co = x_co; p.x_C = x_C;
  p.N = N; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
  p.CinTot = CinTot; p.ci_base = ci_base; p.ci_real = x_C_real;
  p.gpt = gpt;
  p.co_tiles = Cout / (32 * tm);
  p.tiles_per_row = (Wo + 63) / 64;
  p.n_tiles = N * Ho * p.tiles_per_row;
  const int jobs = p.co_tiles * 4;
  // one block per CU (a block owns 120-150 KB of LDS): the grid must not exceed the CU count
  const int n_cu = persistent_cus();
  int S = n_cu / jobs;
  if (S > p.n_tiles / 8) S = p.n_tiles / 8;
  if (S > 256) S = 256;
  if (S < 1) S = 1;
  // per-tile scalar offsets are relative to the slab's first row: the slab's extent must fit 31 bits
  const long long slab_rows = (long long)p.n_tiles / S / p.tiles_per_row + 4;
  if ((2 * slab_rows + 8) * W * (long long)x_cs * 2 >= 0x7FF00000LL || slab_rows * Wo * (long long)dy_cs * 2 >= 0x7FF00000LL) return 0;
  const long long need = ((long long)S * 16 * Cout * CinTot + 256LL * Cout) * 4;
  if (workspace_bytes < need) {
    set_error("wgrad_s2: workspace too small (%lld < %lld)", workspace_bytes, need);
    return HRV_ERR_ARG;
  }
  p.S = S; p.ws = workspace;
  p.bias_ws = dbias ? workspace + (size_t)S * 16 * Cout * CinTot : nullptr;
  const int nblk = jobs * S;
  if (cls == 0) hipLaunchKernelGGL((conv_wgrad_s2_kernel<4, 2, 2>), dim3(nblk), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((conv_wgrad_s2_kernel<2, 4, 4>), dim3(nblk), dim3(256), 0, st, p);
  int rc = check_launch("conv_wgrad_s2_kernel");
  if (rc) return rc;
  *S_out = S;
  return 1;
}

}  // namespace hrv
