// SPADE gamma|beta convolution (network_generator.py:117-121: conv_gamma / conv_beta, 3x3 over the 128-channel actv)
// with the modulate epilogue IN(x + noise) * (1 + gamma) + beta (+ LeakyReLU), and its data gradient
// d(actv) = conv^T([dgamma | dbeta]) * relu'(actv) -- the two largest convolution families of one train_generator.py
// iteration.  One persistent block per CU:
//
//   * 256 threads = 4 waves, one per SIMD (up to 512 registers each).  A block owns a 16x16-pixel tile and ALL the layer's columns: the 18x18
//     halo patch of the source (128 bf16 channels per chunk, 90 KB) is DMA'd into LDS once per tile, and the columns
//     run in passes of 4 or 5 column tiles of 32 (160 = 5 x 32 for the 80-channel norms, 288 = 4 + 5 for the 144-channel
//     ones: no padded columns -- the generic patch tiles issue 160 columns as 128 + 64 and load the patch twice).
//   * wave w multiplies tile rows 4w .. 4w+3 (64 pixels) by every column of the pass: 2 x NTP accumulator tiles of 32x32
//     (per 16-k step 2 A + NTP B fragment reads feed 2 NTP MFMAs: the LDS is ~35 % busy at the full MFMA rate; with
//     eight waves of 32 pixels it was ~60 % and the measured main loop ran at 41 % of the MFMA rate).
//     A fragments come from the patch (tap = pixel offset), B fragments from a 3-stage weight ring.  Reads run one k-step
//     ahead of the MFMAs across K-tile boundaries: the barrier that publishes the next weight tile sits in front of the
//     last k-step's MFMAs.  The next tile's patch and first weight tiles are requested when the main loop ends and fly
//     under the epilogue.
//   * weights are packed (hrv_spade_gb_pack_dev) in FRAGMENT order: [pass][K-tile][column tile][k-step][lane][8 bf16],
//     so a stage is one linear DMA copy and a B fragment read is lane-linear (conflict-free, immediate offsets).  A stage
//     holds 64 k-values x (NTP x 32) columns (16 / 20 KB) and feeds 256 pixels: half the L2 -> LDS weight bytes per FLOP
//     of the 128-pixel tiles, and one tile's DMA stays in flight across the fence-less barrier (counted vmcnt).
//   * swapped-operand MFMA (D[cout][pixel]): a lane holds 4 consecutive channels of one pixel, gamma and beta of the
//     same channel in the same lane (paired column tiles; the 16-channel tail pairs inside ONE tile: rows 0-15 gamma,
//     16-31 beta).  Results leave through a per-wave LDS scratch so the global stores run along the channels.
//   * the epilogue's x / noise (forward) or mask (data gradient) values are requested right after the prologue's DMAs
//     (they are the oldest requests of the pass: the counted vmcnt waits of the weight stream never wait longer for
//     them than the prologue does); the per-channel constants sit in LDS from the start of the pass.
//
// LDS: 92,160 (patch, 18 rows x 20 pixels x 256 B) + 3 x 20,480 (weight ring) + 2,048 (constants) = 155,648 bytes -> one block per CU.
#include <string.h>

#include <type_traits>
#include <utility>

#include "conv_params.h"

namespace hrv {

// compile-time loop (the scheduling hints take literal arguments)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void gb_store16(f32x4 v, rsrc_t r, unsigned voff) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, (int)voff, 0, 0);
}
#else
__device__ inline void gb_store16(f32x4, rsrc_t, unsigned) {}
#endif

// registers 4g .. 4g+3 of an accumulator tile (4 consecutive output channels of the lane's pixel)
__device__ __forceinline__ f32x4 gb_acc4(const f32x16& a, int g) {
  f32x4 r;
  r[0] = a[4 * g]; r[1] = a[4 * g + 1]; r[2] = a[4 * g + 2]; r[3] = a[4 * g + 3];
  return r;
}

template <typename F, int... Is>
__device__ __forceinline__ void gb_static_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

constexpr int GB_MAXP = 16;
constexpr int GB_PW = 20;                      // patch row pitch in pixels (18 + 2: a row is five 4-pixel DMA pieces)
constexpr int GB_PATCH_B = 18 * GB_PW * 256;
constexpr int GB_SBMAX = 5 * 4096;
constexpr int GB_CB_OFF = GB_PATCH_B + 3 * GB_SBMAX;
constexpr int GB_LDS = GB_CB_OFF + 2048;
constexpr int GB_CV = 80;                      // channels per pass the constant vectors hold

struct GbParams {
  const void* src; int src_cs, src_co, C; unsigned src_bytes;   // bf16 NHWC source, C % 16 == 0; src_bytes: ONE image
  int N, H, W, M;
  const void* wp; unsigned w_bytes;
  int npass;
  int ntp[GB_MAXP];         // column tiles of 32 per pass (4 or 5)
  int tile0[GB_MAXP];       // first column tile of the pass
  unsigned woff[GB_MAXP];   // byte offset of the pass's weight block
  int nchunk, KT;           // 128-channel chunks of the source; K-tiles (chunk, tap, 64-k half) per pass
  int m_tiles;
  // forward (SPADE modulate) epilogue
  const float* sx; int sx_cs, sx_co, sx_f32, sC;
  const float *smean, *srstd, *sz, *sns, *bg, *bb;
  void* g1p; int g1_bf16;
  int act; float slope;
  void* out; int out_cs, out_co, out_f32;
  // data-gradient epilogue
  const void* mask; int mask_cs, mask_co;
  unsigned long long* tlog;
  int stagger_ticks;        // 100 MHz ticks between the phase groups' starts (0: none)
};

struct GbPlan {
  int npass, ntp[GB_MAXP], tile0[GB_MAXP], nchunk, KT;
  unsigned woff[GB_MAXP];
  long long bytes;
};

// mode 0: forward, columns = (gamma32 | beta32) pairs (+ one 16|16 tail tile) of C norm channels over `hid` source channels;
// mode 1: data gradient, columns = `hid` actv channels over the 2*Cp channels of [dgamma | dbeta]
static bool gb_plan(int mode, int C, int Cp, int hid, GbPlan& pl) {
  memset(&pl, 0, sizeof(pl));
  int NT, Cs;
  if (mode == 0) {
    if (hid != 128 || C % 16 != 0 || (C % 32 != 0 && C % 32 != 16)) return false;
    NT = 2 * (C / 32) + (C % 32 ? 1 : 0);
    Cs = hid;
  } else {
    if (hid != 128 || Cp % 8 != 0 || (2 * Cp) % 16 != 0) return false;
    NT = hid / 32;
    Cs = 2 * Cp;
  }
  // column passes: 4 tiles each, then (NT % 4 == 2) one pass of 2, then (16-channel tail) one pass of 5 = two pairs + the tail
  const int n5 = NT & 1;
  const int rest = NT - 5 * n5;
  if (rest < 0) return false;
  const int n2 = (rest % 4 == 2) ? 1 : 0;
  const int n4 = (rest - 2 * n2) / 4;
  if (n4 + n2 + n5 > GB_MAXP || n4 + n2 + n5 < 1) return false;
  pl.npass = n4 + n2 + n5;
  pl.nchunk = (Cs + 127) / 128;
  {
    const int last = Cs - 128 * (pl.nchunk - 1);     // the kernel's K-tiles are 64-k halves; a last chunk of 32 runs two-step tiles
    if (last != 32 && last != 64 && last != 128) return false;
    if (mode == 0 && last != 128) return false;
  }
  int KT = 0;
  for (int c = 0; c < pl.nchunk; ++c) {
    const int kc = Cs - 128 * c < 128 ? Cs - 128 * c : 128;
    KT += 9 * ((kc + 63) / 64);
  }
  pl.KT = KT;
  long long off = 0;
  int t0 = 0;
  for (int i = 0; i < pl.npass; ++i) {
    pl.ntp[i] = i < n4 ? 4 : (i < n4 + n2 ? 2 : 5);
    pl.tile0[i] = t0;
    t0 += pl.ntp[i];
    pl.woff[i] = (unsigned)off;
    off += (long long)KT * pl.ntp[i] * 4096;
  }
  pl.bytes = off;
  return off < (long long)0xFFFFFFF0;
}

// ------------------------------------------------------------------------------------------------ weight packer
struct GbPackParams {
  GbPlan pl;
  int mode, C, Cp, hid;
  const float* wg;   // [C][hid][3][3]
  const float* wb;
  unsigned short* out;
};

__global__ __launch_bounds__(256) void gb_pack_kernel(const GbPackParams p) {
  // one thread per 16-byte group: (pass, K-tile q, piece = (column tile j, k-step s), lane)
  const long long G = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = p.pl.bytes / 16;
  if (G >= total) return;
  int pass = 0;
  for (int i = 1; i < p.pl.npass; ++i)
    if (G * 16 >= (long long)p.pl.woff[i]) pass = i;
  const int ntp = p.pl.ntp[pass];
  long long r = G - (long long)p.pl.woff[pass] / 16;
  const int lane = (int)(r & 63);
  r >>= 6;
  const int piece = (int)(r % (ntp * 4));
  const int q = (int)(r / (ntp * 4));
  const int j = piece >> 2, s = piece & 3;
  const int l31 = lane & 31, lh = lane >> 5;
  const int Cs = p.mode == 0 ? p.hid : 2 * p.Cp;
  // K-tile -> (chunk, tap, half): every chunk but the last holds 18 K-tiles
  int chunk = q / 18;
  if (chunk > p.pl.nchunk - 1) chunk = p.pl.nchunk - 1;
  const int rq = q - 18 * chunk;
  const int kc = Cs - 128 * chunk < 128 ? Cs - 128 * chunk : 128;
  const int nhalf = (kc + 63) / 64;
  const int tap = rq / nhalf, half = rq - tap * nhalf;
  const int kh = tap / 3, kw = tap - 3 * kh;
  const int k0 = chunk * 128 + half * 64 + s * 16 + lh * 8;
  const int jt = p.pl.tile0[pass] + j;
  unsigned short v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    float w = 0.f;
    if (p.mode == 0) {
      const bool tail = (p.C % 32) != 0 && jt == 2 * (p.C / 32);
      int c;
      bool beta;
      if (tail) { beta = l31 >= 16; c = (jt >> 1) * 32 + (l31 & 15); }
      else { beta = (jt & 1) != 0; c = (jt >> 1) * 32 + l31; }
      if (c < p.C && k < p.hid) w = (beta ? p.wb : p.wg)[(((size_t)c * p.hid + k) * 3 + kh) * 3 + kw];
    } else {
      // dX[y][x] = sum_taps dY[y + kh' - 1][x + kw' - 1] * W[.][.][2 - kh'][2 - kw']
      const int col = jt * 32 + l31;               // actv channel
      const bool beta = k >= p.Cp;
      const int c = beta ? k - p.Cp : k;           // norm channel
      if (k < 2 * p.Cp && c < p.C && col < p.hid)
        w = (beta ? p.wb : p.wg)[(((size_t)c * p.hid + col) * 3 + (2 - kh)) * 3 + (2 - kw)];
    }
    v[e] = f2bf(w);
  }
  uint4 o;
  o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
  o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
  reinterpret_cast<uint4*>(p.out)[G] = o;
}

// ------------------------------------------------------------------------------------------------ the kernel
// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] bits 3:0, expcnt bits 6:4, lgkmcnt bits 11:8, vmcnt[5:4] bits 15:14)
constexpr int gb_wait(int vm) { return (vm & 15) | (7 << 4) | (0 << 8) | ((vm >> 4) << 14); }   // vmcnt(vm) lgkmcnt(0)

// Everything one (tile, pass) needs before its main loop: the halo patch (chunk 0) and weight tiles 0, 1 of the pass.
// Issued by the PREVIOUS pass's epilogue (after the barrier that frees the patch and ring stages 0 / 1), so the loads fly
// while that epilogue computes and stores.
template <int NTP>
struct GbIssue {
  static constexpr int NP = NTP * 4;        // 1-KB pieces (column tile, k-step) per weight stage
  static constexpr int NB = NP / 4;         // DMA instructions per wave per stage (4 waves)
  static constexpr int SB = NP * 1024;
  static __device__ __forceinline__ void b_dma(const GbParams& p, unsigned char* smem, unsigned wbase, int q, int st, int wave, int lane) {
    const rsrc_t w_rsrc = make_rsrc(p.wp, p.w_bytes);
    const unsigned soff0 = wbase + (unsigned)q * (unsigned)SB;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = wave + 4 * i;          // wave-uniform
      dma16(w_rsrc, reinterpret_cast<float*>(smem + GB_PATCH_B + st * GB_SBMAX + idx * 1024), (unsigned)lane * 16u,
            soff0 + (unsigned)idx * 1024u);
    }
  }
  // one instruction = 4 consecutive halo pixels of one patch row x 16 groups of 8 channels (90 instructions per patch,
  // instruction u = 5 * row + piece, dealt round-robin to the waves); the 16-byte groups of a pixel are XOR-swizzled by
  // (hx & 15) on the SOURCE side: tap-shifted b128 fragment reads of 16 adjacent pixels are bank-disjoint.  Per
  // instruction the lane offset is (lane constant) + (scalar): no division, nothing worth hoisting.
  static __device__ __forceinline__ void patch_dma(const GbParams& p, unsigned char* smem, int pt_n, int pt_y0, int pt_x0, int chunk,
                                                   int wave, int lane) {
    // buffer resource of the tile's IMAGE (serving batches of 16 x 1024x768 x 384 channels exceed one 32-bit range)
    const rsrc_t a_rsrc = make_rsrc(reinterpret_cast<const char*>(p.src) + (size_t)pt_n * p.src_bytes, p.src_bytes);
    const int dma_dx = lane >> 4, dma_s = (lane & 15) ^ (lane >> 4);
    const int kc = p.C - 128 * chunk;
#pragma unroll 1
    for (int u = wave; u < 90; u += 4) {
      const int hy = (u * 205) >> 10, i4 = (u - 5 * hy) << 2;        // u / 5, 4 * (u % 5)   (u < 90)
      const int y = pt_y0 - 1 + hy, x = pt_x0 - 1 + i4 + dma_dx;
      const int g = dma_s ^ (i4 & 15);
      const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W && g * 8 < kc;
      const unsigned off = ((unsigned)(y * p.W + x) * (unsigned)p.src_cs + (unsigned)(p.src_co + chunk * 128 + g * 8)) * 2u;
      dma16(a_rsrc, reinterpret_cast<float*>(smem + u * 1024), ok ? off : 0xFFFFFFF0u, 0u);
    }
  }
  // piece i (0..NB-1) of this wave's share of weight tile q -> ring stage st
  static __device__ __forceinline__ void b_dma1(const GbParams& p, unsigned char* smem, unsigned wbase, int q, int st, int wave, int lane,
                                                int i) {
    const rsrc_t w_rsrc = make_rsrc(p.wp, p.w_bytes);
    const int idx = wave + 4 * i;
    dma16(w_rsrc, reinterpret_cast<float*>(smem + GB_PATCH_B + st * GB_SBMAX + idx * 1024), (unsigned)lane * 16u,
          wbase + (unsigned)q * (unsigned)SB + (unsigned)idx * 1024u);
  }
  static __device__ __forceinline__ void issue(const GbParams& p, unsigned char* smem, int pass, bool with_patch, int pt_n, int pt_y0,
                                               int pt_x0, int wave, int lane) {
    if (with_patch) patch_dma(p, smem, pt_n, pt_y0, pt_x0, 0, wave, lane);
    const unsigned wbase = p.woff[pass];
    b_dma(p, smem, wbase, 0, 0, wave, lane);
    if (p.KT > 1) b_dma(p, smem, wbase, 1, 1, wave, lane);
  }
};

struct GbTile { int n, y0, x0; };
__device__ __forceinline__ GbTile gb_tile(const GbParams& p, int bid) {
  const int tx = (p.W + 15) >> 4, ty = (p.H + 15) >> 4;
  const int mt = xcd_remap(bid, p.m_tiles);
  GbTile t;
  t.n = mt / (tx * ty);
  const int rr = mt - t.n * (tx * ty);
  t.y0 = (rr / tx) << 4;
  t.x0 = (rr % tx) << 4;
  return t;
}

// One (tile, pass).  On entry its patch / weight tiles 0, 1 are in flight or landed (GbIssue::issue); ``nxt_*`` describe
// what to issue for the next (tile, pass) of this block once the main loop is done (nxt_pass < 0: nothing).
template <int NTP, int EPI, bool HALF>
__device__ __forceinline__ void gb_pass(const GbParams& p, const int pass, unsigned char* const smem, const GbTile T,
                                        const int bid, const bool first, const bool last, const int nxt_pass, const bool nxt_patch,
                                        const GbTile NT_) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  using IS = GbIssue<NTP>;
  constexpr int NB = IS::NB;
  constexpr int NPAIR = NTP / 2, TAIL = NTP & 1;
  static_assert(NPAIR * 32 + TAIL * 16 <= GB_CV, "constant vectors");
  unsigned char* const patch = smem;
  unsigned char* const bst = smem + GB_PATCH_B;
  float* const cbuf = reinterpret_cast<float*>(smem + GB_CB_OFF);
  const int KT = p.KT;
  const int tile0 = p.tile0[pass];
  const unsigned wbase = p.woff[pass];
  const int pt_n = T.n, pt_y0 = T.y0, pt_x0 = T.x0;

  // ---- this lane's pixels (two: tile rows 4 w + (l31 >> 4) and + 2) and fragment addresses
  const int ty = 4 * wave + (l31 >> 4), tx = l31 & 15;
  const int px = pt_x0 + tx;
  int pidx[2];
  bool pix_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int py = pt_y0 + ty + 2 * i;
    pix_ok[i] = py < p.H && px < p.W;
    pidx[i] = pix_ok[i] ? (pt_n * p.H + py) * p.W + px : 0;
  }
  const unsigned char* const a_lb = patch + (ty * GB_PW + tx) * 256;
  const unsigned char* const b_lb = bst + lane * 16;

  f32x16 acc[2][NTP];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTP; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- the pass's per-channel constants -> LDS (read by the epilogue only)
  const int cb0 = (tile0 >> 1) * 32;          // first norm channel of this pass (forward)
  if constexpr (EPI == 1) {
    // bias gamma | bias beta | noise scale | mean | rstd.  Scalar loads: the bias / noise-scale parameters are views into
    // the fused optimizer's flat buffer (4-byte aligned)
    // per channel: 1 + bias_gamma | bias_beta | rstd | noise_scale * rstd | -mean * rstd, so that
    //   IN(x + z ns) = x * rstd + (z * (ns rstd) - mean rstd)
    for (int t = tid; t < GB_CV; t += 256) {
      const int c = cb0 + t;
      float b1 = 1.f, b2 = 0.f, rs = 0.f, nr = 0.f, mr = 0.f;
      if (c < p.sC) {
        b1 = 1.f + p.bg[c];
        b2 = p.bb[c];
        rs = p.srstd[(size_t)pt_n * p.sC + c];
        nr = p.sns ? p.sns[c] * rs : 0.f;
        mr = -p.smean[(size_t)pt_n * p.sC + c] * rs;
      }
      cbuf[t] = b1; cbuf[GB_CV + t] = b2; cbuf[2 * GB_CV + t] = rs; cbuf[3 * GB_CV + t] = nr; cbuf[4 * GB_CV + t] = mr;
    }
  }
  // the patch and weight tiles 0, 1 (issued before this call) have landed
  __builtin_amdgcn_s_waitcnt(gb_wait(0));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (p.tlog && tid == 0 && first) p.tlog[(size_t)bid * 8 + 1] = wall_clock64();

  // epilogue operands: requested now (the oldest requests of the pass: the counted vmcnt waits of the weight stream
  // never wait for them longer than the prologue did), used after the main loop
  [[maybe_unused]] f32x4 xv[2][NPAIR][4];
  [[maybe_unused]] f32x4 xt[2][2];
  [[maybe_unused]] float zv[2] = {0.f, 0.f};
  [[maybe_unused]] u16x4 mv[2][NTP][4];
  if constexpr (EPI == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          xv[i][pr][g] = ld4rt<true>(p.sx, (size_t)pidx[i] * p.sx_cs + p.sx_co + cb0 + pr * 32 + 8 * g + 4 * lh, p.sx_f32);
      if constexpr (TAIL != 0) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
          xt[i][g] = ld4rt<true>(p.sx, (size_t)pidx[i] * p.sx_cs + p.sx_co + cb0 + NPAIR * 32 + 8 * g + 4 * lh, p.sx_f32);
      }
      if (p.sz) {
        const int py = pt_y0 + ty + 2 * i;
        zv[i] = p.sz[((size_t)pt_n * p.W + (pix_ok[i] ? px : 0)) * p.H + (pix_ok[i] ? py : 0)];
      }
    }
  } else {
    if (p.mask) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTP; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            mv[i][j][g] = *reinterpret_cast<const u16x4*>(reinterpret_cast<const unsigned short*>(p.mask) +
                                                          (size_t)pidx[i] * p.mask_cs + p.mask_co + (tile0 + j) * 32 + 8 * g + 4 * lh);
    }
  }

  // ---- main loop over the K-tiles (chunk, tap, 64-k half).  Fragment reads run one k-step ahead of the MFMAs, ACROSS
  // K-tiles: the wait + barrier that publishes weight tile q+1 sits in front of the LAST k-step's MFMAs of tile q (all
  // of this wave's reads of tile q are complete there), so tile q+1's first fragments are read under those MFMAs.
  int it_chunk = 0, it_tap = 0, it_half = 0;
  int kc = p.C < 128 ? p.C : 128;             // channels of the current chunk
  int nhalf = (kc + 63) >> 6;
  int rb = 0, wb = 2;
  f32x4 fa[2][2], fb[2][NTP];
  // fragment source of the K-tile the iterator points at
  const unsigned char* Ap;
  unsigned ax;
  const unsigned char* Bp;
  auto point = [&]() {
    const int kh = (it_tap * 11) >> 5, kw = it_tap - 3 * kh;
    Ap = a_lb + (kh * GB_PW + kw) * 256;
    ax = (unsigned)(((((tx + kw) & 15) ^ lh) << 4) ^ (it_half << 7));
    Bp = b_lb + rb * GB_SBMAX;
  };
#define GB_READ(SET, S)                                                                                    \
  {                                                                                                        \
    fa[SET][0] = *reinterpret_cast<const f32x4*>(Ap + (ax ^ (unsigned)((S) << 5)));                        \
    fa[SET][1] = *reinterpret_cast<const f32x4*>(Ap + 2 * GB_PW * 256 + (ax ^ (unsigned)((S) << 5)));      \
    _Pragma("unroll") for (int j = 0; j < NTP; ++j)                                                        \
        fb[SET][j] = *reinterpret_cast<const f32x4*>(Bp + (j * 4 + (S)) * 1024);                            \
  }
#define GB_MMA(SET)                                                                                        \
  {                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < NTP; ++j) {                                                      \
      acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[SET][j]),         \
                                                          __builtin_bit_cast(bf16x8, fa[SET][0]), acc[0][j], 0, 0, 0); \
      acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[SET][j]),         \
                                                          __builtin_bit_cast(bf16x8, fa[SET][1]), acc[1][j], 0, 0, 0); \
    }                                                                                                      \
  }
  // issue order of one k-step: the NTP + 2 fragment reads of the next step spread between this step's 2 NTP MFMAs
  // (left alone, the scheduler sinks every read next to its MFMA and waits lgkmcnt(0) per MFMA)
#define GB_ORDER(NDMA)                                                                                     \
  {                                                                                                        \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
    _Pragma("unroll") for (int j = 0; j < NTP; ++j) {                                                      \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                   \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
    }                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < (NDMA); ++j) {                                                   \
      __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                                                   \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
    }                                                                                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, NTP - 1 - (NDMA), 0);                                      \
  }
  point();
  GB_READ(0, 0)
  // K-tile q with NKS k-steps of 16 (4: a full 64-k half; 2: the 32-channel last chunk of a data-gradient source);
  // NEXT: a K-tile follows.  (The MFMA sequences appear once per instantiation, in straight-line code: accumulators that
  // flow through diverging branches made the register allocator shuffle them between the two register files.)
  auto ktile = [&](const int q, auto nks_c, auto next_c, auto more_c) {
    constexpr int NKS = decltype(nks_c)::value;
    constexpr bool NEXT = decltype(next_c)::value;       // K-tile q+1 exists
    constexpr bool more = decltype(more_c)::value;       // K-tile q+2 exists (compile time: no branch splits the k-step's schedule)
    // weight tile q+2 -> the stage K-tile q-1 read (every wave has passed that barrier).  Its NB DMA instructions are
    // dealt over the first k-steps, each behind a few MFMAs: issued back to back at the top of the K-tile they held this
    // wave's (in-order) instruction stream for ~100 cycles apiece with the matrix pipe drained -- one wave per SIMD has
    // nobody to cover that.
    gb_static_for(std::make_integer_sequence<int, NKS - 1>{}, [&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      constexpr int NS = NKS - 1;                       // k-steps that carry DMA pieces
      constexpr int d0 = (NB * s) / NS, d1 = (NB * (s + 1)) / NS;
      GB_READ((s + 1) & 1, s + 1)
      if constexpr (more) {
#pragma unroll
        for (int i = d0; i < d1; ++i) IS::b_dma1(p, smem, wbase, q + 2, wb, wave, lane, i);
      }
      GB_MMA(s & 1)
      GB_ORDER(more ? d1 - d0 : 0)
    });
    if constexpr (NEXT) {
      // last k-step: advance to K-tile q+1, publish its weights, read its first fragments under this step's MFMAs
      [[maybe_unused]] bool new_chunk = false;
      if (++it_half == nhalf) {
        it_half = 0;
        if (++it_tap == 9) {
          it_tap = 0;
          ++it_chunk;
          new_chunk = true;
        }
      }
      rb = rb == 2 ? 0 : rb + 1;
      wb = wb == 2 ? 0 : wb + 1;
      asm volatile("" ::: "memory");
      if (more) __builtin_amdgcn_s_waitcnt(gb_wait(NB));
      else __builtin_amdgcn_s_waitcnt(gb_wait(0));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr (EPI == 2) {
        // next 128-channel chunk of the source: every wave's reads of the patch are complete (lgkmcnt(0) above); its DMA
        // flies under this step's MFMAs
        if (new_chunk) {
          kc = p.C - 128 * it_chunk;
          kc = kc < 128 ? kc : 128;
          nhalf = (kc + 63) >> 6;
          IS::patch_dma(p, smem, pt_n, pt_y0, pt_x0, it_chunk, wave, lane);
        } else {
          point();
          GB_READ(0, 0)
        }
        GB_MMA((NKS - 1) & 1)
        if (new_chunk) {
          asm volatile("" ::: "memory");
          __builtin_amdgcn_s_waitcnt(gb_wait(0));
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          point();
          GB_READ(0, 0)
        }
      } else {
        point();
        GB_READ(0, 0)
        GB_MMA((NKS - 1) & 1)
        GB_ORDER(0)
      }
    } else {
      GB_MMA((NKS - 1) & 1)
    }
  };
  using C4 = std::integral_constant<int, 4>;
  using C2 = std::integral_constant<int, 2>;
  using Y = std::true_type;
  using N = std::false_type;
  if constexpr (!HALF) {
#pragma unroll 1
    for (int q = 0; q < KT - 2; ++q) ktile(q, C4{}, Y{}, Y{});
    ktile(KT - 2, C4{}, Y{}, N{});
    ktile(KT - 1, C4{}, N{}, N{});
  } else {
    // a 32-channel last chunk (data gradient: 2 Cp = 160, 288, ...): the last 9 K-tiles are two-step tiles
#pragma unroll 1
    for (int q = 0; q < KT - 9; ++q) ktile(q, C4{}, Y{}, Y{});
#pragma unroll 1
    for (int q = KT - 9; q < KT - 2; ++q) ktile(q, C2{}, Y{}, Y{});
    ktile(KT - 2, C2{}, Y{}, N{});
    ktile(KT - 1, C2{}, N{}, N{});
  }
#undef GB_READ
#undef GB_MMA
#undef GB_ORDER
  if (p.tlog && tid == 0 && last) p.tlog[(size_t)bid * 8 + 2] = wall_clock64();

  // ---- epilogue.  D layout (swapped operands): lane -> pixel l31; regs 4g..4g+3 -> channels 8g + 4 lh + (0..3) of the tile
  __syncthreads();                             // every wave is done with the patch and the weight ring
  // the next (tile, pass) of this block: its patch and first two weight tiles fly while this epilogue computes and stores
  if (nxt_pass >= 0) IS::issue(p, smem, nxt_pass, nxt_patch, NT_.n, NT_.y0, NT_.x0, wave, lane);
  // (the epilogue's index arithmetic depends on the lane id only: hidden from the optimiser behind an empty asm, or it is
  //  hoisted out of the persistent tile loop and lives -- spilled -- through the main loop)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  constexpr int SCS = 36;                      // scratch row stride in floats (32 channels + 4: conflict-free)
  const int l31e = lane_e & 31, lhe = lane_e >> 5;
  // per-wave scratch, 32 pixels x 32 channels, in ring stage 2 (stages 0 / 1 are being refilled for the next pass)
  float* const scr = reinterpret_cast<float*>(smem + GB_PATCH_B + 2 * GB_SBMAX) + wave * (32 * SCS);
  static_assert(4 * 32 * SCS * 4 <= GB_SBMAX, "epilogue scratch fits one ring stage");
  // Stores go through buffer resources of this tile's IMAGE (32-bit byte offsets; an out-of-image row gets an offset
  // beyond num_records and the hardware drops the store: no branches, no 64-bit address arithmetic -- the first version
  // of this epilogue was ~8000 instructions per wave and took as long as the main loop).
  const int oes = p.out_f32 ? 4 : 2;
  const size_t img_px = (size_t)p.H * p.W;
  const rsrc_t o_rsrc = make_rsrc(reinterpret_cast<const char*>(p.out) + (size_t)pt_n * img_px * p.out_cs * oes,
                                  (unsigned)(img_px * p.out_cs * oes));
  // byte offset of channel 0 of row r (0..31) of this wave's half i in a tensor of `cs` channels of `es` bytes
  auto row_off = [&](int i, int r, int cs, int es) -> unsigned {
    const int y = pt_y0 + 4 * wave + 2 * i + (r >> 4), x = pt_x0 + (r & 15);
    return (y < p.H && x < p.W) ? (unsigned)((y * p.W + x) * cs) * (unsigned)es : 0xFFFFFFF0u;
  };
  // scratch rows (NG 16-byte groups of 8 channels per pixel, NG = 2 or 4) -> global, along the channels
  auto copy_out = [&](const int i, auto ng_c, const rsrc_t rs, const int dcs, const int dco, const bool f32) {
    constexpr int NG = decltype(ng_c)::value;
    constexpr int SH = NG == 4 ? 2 : 1;        // log2(NG)
    if (!f32) {
#pragma unroll
      for (int k = 0; k < NG / 2; ++k) {       // bf16 rows: 32 pixels x NG pieces of 8 channels
        const int t = lane_e + 64 * k, r = t >> SH, kk = t & (NG - 1);
        const f32x4 lo = *reinterpret_cast<const f32x4*>(scr + r * SCS + kk * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(scr + r * SCS + kk * 8 + 4);
        const unsigned ro = row_off(i, r, dcs, 2);
        gb_store16(pack_bf16x8(lo, hi), rs, ro == 0xFFFFFFF0u ? ro : ro + (unsigned)(dco + kk * 8) * 2u);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NG; ++k) {           // fp32 rows: 32 pixels x 2 NG pieces of 4 channels
        const int t = lane_e + 64 * k, r = t >> (SH + 1), kk = t & (2 * NG - 1);
        const f32x4 v = *reinterpret_cast<const f32x4*>(scr + r * SCS + kk * 4);
        const unsigned ro = row_off(i, r, dcs, 4);
        gb_store16(v, rs, ro == 0xFFFFFFF0u ? ro : ro + (unsigned)(dco + kk * 4) * 4u);
      }
    }
  };
  // ---- bf16 staging shared by both epilogues: the two 32-pixel halves of the wave go through the scratch TOGETHER (two
  // buffers of 32 rows x 64 B + pad), rows leave 16 bytes (8 channels) per lane along the channels
  typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
  unsigned char* const sb0 = smem + GB_PATCH_B + 2 * GB_SBMAX + wave * 5120;
  static_assert(4 * 5120 <= GB_SBMAX, "epilogue scratch fits one ring stage");
  constexpr int RS = 80;                     // scratch row stride in bytes (32 bf16 channels + 16)
  // pixel (in-image index, or -1) of the scratch rows this lane stores: 4-group rows (lane >> 2) + 16 k, 2-group rows lane >> 1
  int pp4[2][2], pp2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int r = (lane_e >> 2) + 16 * k;
      const int y = pt_y0 + 4 * wave + 2 * i + (r >> 4), x = pt_x0 + (r & 15);
      pp4[i][k] = (y < p.H && x < p.W) ? y * p.W + x : -1;
    }
    const int r = lane_e >> 1;
    const int y = pt_y0 + 4 * wave + 2 * i + (r >> 4), x = pt_x0 + (r & 15);
    pp2[i] = (y < p.H && x < p.W) ? y * p.W + x : -1;
  }
  // scratch rows of both halves -> global: 16 bytes (8 channels) per lane, along the channels
  auto rows_out = [&](auto ng_c, const rsrc_t rs, const int dcs, const int dco) {
    constexpr int NG = decltype(ng_c)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned char* sb = sb0 + i * 2560;
      if constexpr (NG == 4) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int r = (lane_e >> 2) + 16 * k, kk = lane_e & 3;
          const f32x4 v = *reinterpret_cast<const f32x4*>(sb + r * RS + kk * 16);
          gb_store16(v, rs, pp4[i][k] < 0 ? 0xFFFFFFF0u : (unsigned)(pp4[i][k] * dcs + dco + kk * 8) * 2u);
        }
      } else {
        const int r = lane_e >> 1, kk = lane_e & 1;
        const f32x4 v = *reinterpret_cast<const f32x4*>(sb + r * RS + kk * 16);
        gb_store16(v, rs, pp2[i] < 0 ? 0xFFFFFFF0u : (unsigned)(pp2[i] * dcs + dco + kk * 8) * 2u);
      }
    }
  };
  if constexpr (EPI == 1) {
    // Both outputs are bf16 (the activation feeds matrix cores and its own LeakyReLU mask; (1 + gamma) is read once, by
    // the normalisation backward).  The two 32-pixel halves of the wave go through the scratch TOGETHER (two buffers of
    // 32 rows x 64 B + pad): a group's LDS write -> read -> store chain is exposed once per group, not once per half,
    // and the per-channel constants are fetched once for both.
    const rsrc_t g_rsrc = make_rsrc(reinterpret_cast<const char*>(p.g1p) + (size_t)pt_n * img_px * p.sC * 2,
                                    p.g1p ? (unsigned)(img_px * p.sC * 2) : 0u);
    // act(v) = max(v, v * sl): LeakyReLU (sl = slope), ReLU (0), none (1) -- the SPADE sites use LeakyReLU / none
    const float sl = p.act == HRV_ACT_LRELU ? p.slope : (p.act == HRV_ACT_RELU ? 0.f : 1.f);
    // NG * 8 channels starting at local channel lc0: modulate, stage, store the activation, then (1 + gamma)
    auto group = [&](const int lc0, auto ng_c, auto&& gam, auto&& bet, auto&& xin) {
      constexpr int NG = decltype(ng_c)::value;
      f32x4 g1r[2][NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int lc = lc0 + 8 * g + 4 * lhe;
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(cbuf + lc);
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(cbuf + GB_CV + lc);
        const f32x4 rs = *reinterpret_cast<const f32x4*>(cbuf + 2 * GB_CV + lc);
        const f32x4 nr = *reinterpret_cast<const f32x4*>(cbuf + 3 * GB_CV + lc);
        const f32x4 mr = *reinterpret_cast<const f32x4*>(cbuf + 4 * GB_CV + lc);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          // whole-vector arithmetic: pairs of lanes' channels go through the packed fp32 instructions
          const f32x4 xn = xin(i, g) * rs + (nr * zv[i] + mr);
          g1r[i][g] = gam(i, g) + b1;
          const f32x4 t = xn * g1r[i][g] + (bet(i, g) + b2);
          const f32x4 ts = t * sl;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(t[e], ts[e]);
          *reinterpret_cast<bf16x4v*>(sb0 + i * 2560 + l31e * RS + 16 * g + 8 * lhe) = __builtin_convertvector(v, bf16x4v);
        }
      }
      const int cb = cb0 + lc0;                // first norm channel of the group
      // same wave wrote and reads: LDS operations of a wave complete in order
      rows_out(ng_c, o_rsrc, p.out_cs, p.out_co + cb);
      if (p.g1p) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            *reinterpret_cast<bf16x4v*>(sb0 + i * 2560 + l31e * RS + 16 * g + 8 * lhe) = __builtin_convertvector(g1r[i][g], bf16x4v);
        rows_out(ng_c, g_rsrc, p.sC, cb);
      }
    };
    using G4 = std::integral_constant<int, 4>;
    using G2 = std::integral_constant<int, 2>;
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr)
      group(pr * 32, G4{}, [&](int i, int g) { return gb_acc4(acc[i][2 * pr], g); },
            [&](int i, int g) { return gb_acc4(acc[i][2 * pr + 1], g); }, [&](int i, int g) { return xv[i][pr][g]; });
    if constexpr (TAIL != 0)
      group(NPAIR * 32, G2{}, [&](int i, int g) { return gb_acc4(acc[i][NTP - 1], g); },
            [&](int i, int g) { return gb_acc4(acc[i][NTP - 1], g + 2); }, [&](int i, int g) { return xt[i][g]; });
  } else {
    using G4 = std::integral_constant<int, 4>;
    if (!p.out_f32) {
      // bf16 d(actv) (its only reader is conv_shared's weight gradient): both halves staged together
#pragma unroll
      for (int j = 0; j < NTP; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            f32x4 v = gb_acc4(acc[i][j], g);
            if (p.mask) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = bf2f(mv[i][j][g][e]) > 0.f ? v[e] : v[e] * p.slope;
            }
            *reinterpret_cast<bf16x4v*>(sb0 + i * 2560 + l31e * RS + 16 * g + 8 * lhe) = __builtin_convertvector(v, bf16x4v);
          }
        rows_out(G4{}, o_rsrc, p.out_cs, p.out_co + (tile0 + j) * 32);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTP; ++j) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float a = acc[i][j][4 * g + e];
              if (p.mask) a = bf2f(mv[i][j][g][e]) > 0.f ? a : a * p.slope;
              v[e] = a;
            }
            *reinterpret_cast<f32x4*>(scr + l31e * SCS + 8 * g + 4 * lhe) = v;
          }
          copy_out(i, G4{}, o_rsrc, p.out_cs, p.out_co + (tile0 + j) * 32, true);
        }
    }
  }
  if (p.tlog && tid == 0 && last) p.tlog[(size_t)bid * 8 + 6] = wall_clock64();      // every store of the tile is issued
  // (the caller's next gb_pass waits vmcnt(0) + barrier before touching the ring: this pass's scratch reads are done then)
}

// One launch covers the passes [pass0, pass1) of the plan, all of NTP column tiles (a single-chunk source's patch stays
// resident across them: the epilogue's scratch lives in the weight ring).
template <int NTP, int EPI, bool HALF>
__global__ __launch_bounds__(256) void spade_gb_kernel(const GbParams p, const int pass0, const int pass1) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[GB_LDS];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const bool repatch = p.nchunk > 1;           // a multi-chunk source's patch holds its LAST chunk when a pass ends
  // Staggered start.  Every tile is a compute phase (main loop, HBM idle) followed by a memory phase (epilogue stores,
  // the next tile's patch and x: ~280 KB per tile); blocks that start together stay in lock step, so all 256 CUs hit HBM
  // at once and then leave it idle (measured: a 14 us memory phase = 256 x 283 KB at HBM speed, next to a 19 us main
  // loop).  Four phase groups, a quarter of a tile apart, spread the memory phases over the compute phases.
  if (p.stagger_ticks > 0) {
    const unsigned long long t_go = wall_clock64() + (unsigned long long)((blockIdx.x >> 3) & 3) * (unsigned)p.stagger_ticks;
    while (wall_clock64() < t_go) __builtin_amdgcn_s_sleep(32);
  }
  if ((int)blockIdx.x < p.m_tiles) {
    const GbTile T0 = gb_tile(p, blockIdx.x);
    GbIssue<NTP>::issue(p, smem, pass0, true, T0.n, T0.y0, T0.x0, wave, lane);
  }
#pragma unroll 1
  for (int bid = blockIdx.x; bid < p.m_tiles; bid += gridDim.x) {
    if (p.tlog && threadIdx.x == 0) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      p.tlog[(size_t)bid * 8 + 0] = wall_clock64();
      p.tlog[(size_t)bid * 8 + 4] = ((unsigned long long)xcc << 32) | hw;
      p.tlog[(size_t)bid * 8 + 5] = blockIdx.x;
    }
    const GbTile T = gb_tile(p, bid);
    const int nbid = bid + gridDim.x;
    const GbTile TN = gb_tile(p, nbid < p.m_tiles ? nbid : bid);
#pragma unroll 1
    for (int pass = pass0; pass < pass1; ++pass) {
      const bool lastp = pass == pass1 - 1;
      const int nxt_pass = !lastp ? pass + 1 : (nbid < p.m_tiles ? pass0 : -1);
      gb_pass<NTP, EPI, HALF>(p, pass, smem, T, bid, pass == pass0, lastp, nxt_pass, lastp || repatch, lastp ? TN : T);
    }
    if (p.tlog) {
      __builtin_amdgcn_s_waitcnt(gb_wait(0));
      if (threadIdx.x == 0) p.tlog[(size_t)bid * 8 + 3] = wall_clock64();
    }
  }
}

}  // namespace hrv

using namespace hrv;

extern "C" int64_t hrv_spade_gb_packed_bytes(int32_t mode, int32_t C, int32_t Cp, int32_t hid) {
  GbPlan pl;
  if (!gb_plan(mode, C, Cp, hid, pl)) return -1;
  return pl.bytes;
}

extern "C" int hrv_spade_gb_supported(int32_t mode, int32_t C, int32_t Cp, int32_t hid, int32_t N, int32_t H, int32_t W) {
  GbPlan pl;
  if (!gb_plan(mode, C, Cp, hid, pl)) return 0;
  const int64_t tiles = (int64_t)N * ((H + 15) / 16) * ((W + 15) / 16);
  // fewer tiles than CUs: the generic tiles (more, smaller blocks; split-K) fill the chip better.  HRV_SPADE_GB_MIN_TILES: the threshold
  const char* e = hrv::env("HRV_SPADE_GB_MIN_TILES");
  int tmin = e ? atoi(e) : 256;
  if (tmin < 1) tmin = 256;
  return tiles >= tmin ? 1 : 0;
}

extern "C" int hrv_spade_gb_pack_dev(int32_t mode, const float* w_gamma, const float* w_beta, int32_t C, int32_t Cp, int32_t hid,
                                     void* out, hrv_stream_t stream) {
  HRV_REQUIRE(w_gamma && w_beta && out, "spade_gb_pack: null pointer");
  GbPackParams pp;
  HRV_REQUIRE(gb_plan(mode, C, Cp, hid, pp.pl), "spade_gb_pack: unsupported shape (mode %d, C %d, Cp %d, hid %d)", mode, C, Cp, hid);
  HRV_REQUIRE(((uintptr_t)out & 15) == 0, "spade_gb_pack: out must be 16-byte aligned");
  pp.mode = mode; pp.C = C; pp.Cp = Cp; pp.hid = hid;
  pp.wg = w_gamma; pp.wb = w_beta; pp.out = (unsigned short*)out;
  const long long groups = pp.pl.bytes / 16;
  hipLaunchKernelGGL(gb_pack_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pp);
  return check_launch("gb_pack_kernel");
}

extern "C" int hrv_spade_gb_bf16(const hrv_spade_gb_t* d, hrv_stream_t stream) {
  HRV_REQUIRE(d != nullptr, "spade_gb: null descriptor");
  HRV_REQUIRE(d->mode == 0 || d->mode == 1, "spade_gb: mode");
  GbPlan pl;
  HRV_REQUIRE(gb_plan(d->mode, d->C, d->Cp, d->hid, pl), "spade_gb: unsupported shape (mode %d, C %d, Cp %d, hid %d)", d->mode,
              d->C, d->Cp, d->hid);
  HRV_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && (int64_t)d->N * d->H * d->W < ((int64_t)1 << 31), "spade_gb: bad extent");
  HRV_REQUIRE(d->src && d->w_packed && d->out, "spade_gb: null pointer");
  const int Cs = d->mode == 0 ? d->hid : 2 * d->Cp;
  HRV_REQUIRE(d->src_cstride % 8 == 0 && d->src_coff % 8 == 0 && d->src_coff + Cs <= d->src_cstride, "spade_gb: source slice");
  const int64_t sbytes = (int64_t)d->H * d->W * d->src_cstride * 2;      // one image: the kernel addresses the source per image
  HRV_REQUIRE(sbytes < (int64_t)0xFFFFFFF0, "spade_gb: one image of the source exceeds the 32-bit buffer range (%lld bytes)", (long long)sbytes);
  HRV_REQUIRE((((uintptr_t)d->src | (uintptr_t)d->w_packed | (uintptr_t)d->out) & 15) == 0, "spade_gb: 16-byte alignment");
  GbParams p;
  memset(&p, 0, sizeof(p));
  p.src = d->src; p.src_cs = d->src_cstride; p.src_co = d->src_coff; p.C = Cs; p.src_bytes = (unsigned)sbytes;
  p.N = d->N; p.H = d->H; p.W = d->W; p.M = d->N * d->H * d->W;
  p.wp = d->w_packed; p.w_bytes = (unsigned)pl.bytes;
  p.npass = pl.npass; p.nchunk = pl.nchunk; p.KT = pl.KT;
  for (int i = 0; i < pl.npass; ++i) { p.ntp[i] = pl.ntp[i]; p.tile0[i] = pl.tile0[i]; p.woff[i] = pl.woff[i]; }
  p.m_tiles = d->N * ((d->H + 15) / 16) * ((d->W + 15) / 16);
  p.out = d->out; p.out_cs = d->out_cstride; p.out_co = d->out_coff; p.out_f32 = d->out_f32;
  p.act = d->act; p.slope = d->act_slope;
  const int oal = d->out_f32 ? 4 : 8;
  HRV_REQUIRE(d->out_cstride % oal == 0 && d->out_coff % oal == 0, "spade_gb: out slice must be 16-byte aligned");
  // diag only (tools/gb_bench.py): per-tile phase timestamps into a device buffer the TOOL owns -- 8 u64 per tile, handed
  // over through hrv_diag_set_tlog (never read from the environment: a stray variable must not turn the hottest
  // kernel of the iteration into a scribbler on arbitrary device memory)
  p.tlog = diag_tlog(p.m_tiles);
  int grid = persistent_cus();
  if (grid > p.m_tiles) grid = p.m_tiles;
  p.stagger_ticks = 0;      // (staggered block starts measured no gain: the epilogue is issue-bound, not HBM-bound)
  if (d->mode == 0) {
    HRV_REQUIRE(d->x && d->mean && d->rstd && d->bias_gamma && d->bias_beta, "spade_gb: null epilogue pointer");
    HRV_REQUIRE((d->noise_z == nullptr) == (d->noise_scale == nullptr), "spade_gb: noise_z/noise_scale go together");
    HRV_REQUIRE(d->out_cstride >= d->out_coff + d->C, "spade_gb: out slice");
    HRV_REQUIRE(d->out_f32 == 0 && (d->g1p == nullptr || d->g1p_bf16 == 1), "spade_gb: the forward stores bf16 (out and 1 + gamma)");
    HRV_REQUIRE((int64_t)d->H * d->W * d->out_cstride * 2 < (int64_t)0xFFFFFFF0, "spade_gb: one image of `out` exceeds 4 GB");
    HRV_REQUIRE(d->x_cstride % 4 == 0 && d->x_coff % 4 == 0 && d->x_coff + d->C <= d->x_cstride,
                "spade_gb: x slice");
    HRV_REQUIRE((((uintptr_t)d->x | (uintptr_t)d->g1p) & 15) == 0, "spade_gb: x / g1p must be 16-byte aligned");
    HRV_REQUIRE(d->stat_stride == d->C, "spade_gb: stat_stride must equal C (C %% 16 == 0: no channel padding)");
    p.sx = (const float*)d->x; p.sx_cs = d->x_cstride; p.sx_co = d->x_coff; p.sx_f32 = d->x_f32; p.sC = d->stat_stride;
    p.smean = d->mean; p.srstd = d->rstd; p.sz = d->noise_z; p.sns = d->noise_scale; p.bg = d->bias_gamma; p.bb = d->bias_beta;
    p.g1p = d->g1p; p.g1_bf16 = d->g1p_bf16;
    // the passes of equal width share a launch (the patch stays resident across them): 4-tile passes, a 2-tile pass, the 5-tile tail pass
    for (int a = 0; a < pl.npass;) {
      int b = a;
      while (b < pl.npass && pl.ntp[b] == pl.ntp[a]) ++b;
      if (pl.ntp[a] == 4) hipLaunchKernelGGL((spade_gb_kernel<4, 1, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, a, b);
      else if (pl.ntp[a] == 2) hipLaunchKernelGGL((spade_gb_kernel<2, 1, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, a, b);
      else hipLaunchKernelGGL((spade_gb_kernel<5, 1, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, a, b);
      a = b;
    }
  } else {
    HRV_REQUIRE(d->out_cstride >= d->out_coff + d->hid, "spade_gb: out slice");
    HRV_REQUIRE(d->mask == nullptr || (d->mask_cstride % 4 == 0 && d->mask_coff % 4 == 0 && ((uintptr_t)d->mask & 7) == 0),
                "spade_gb: mask slice");
    p.mask = d->mask; p.mask_cs = d->mask_cstride; p.mask_co = d->mask_coff;
    if ((p.C & 127) == 32) hipLaunchKernelGGL((spade_gb_kernel<4, 2, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, 0, 1);
    else hipLaunchKernelGGL((spade_gb_kernel<4, 2, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, 0, 1);
  }
  return check_launch("spade_gb_kernel");
}
