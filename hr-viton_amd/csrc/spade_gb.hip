// SPADE gamma|beta convolution (network_generator.py:117-121: conv_gamma / conv_beta, 3x3 over the 128-channel actv)
// with the modulate epilogue IN(x + noise) * (1 + gamma) + beta (+ LeakyReLU), and its data gradient
// d(actv) = conv^T([dgamma | dbeta]) * relu'(actv) -- the two largest convolution families of one train_generator.py
// iteration.  One persistent block per CU:
//
//   * 512 threads = 8 waves (two per SIMD).  A block owns a 16x16-pixel tile and ALL the layer's columns: the 18x18
//     halo patch of the source (128 bf16 channels per chunk, 90 KB) is DMA'd into LDS once per tile, and the columns
//     run in passes of 4 or 5 column tiles of 32 (160 = 5 x 32 for the 80-channel norms, 288 = 4 + 5 for the 144-channel
//     ones: no padded columns -- the generic patch tiles issue 160 columns as 128 + 64 and load the patch twice).
//   * wave w multiplies tile rows 2w, 2w+1 (32 pixels) by every column of the pass: NTP accumulator tiles of 32x32.
//     A fragments come from the patch (tap = pixel offset), B fragments from a 3-stage weight ring.
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

#include "conv_params.h"

namespace hrv {

constexpr int GB_MAXP = 16;
constexpr int GB_PW = 20;                      // patch row pitch in pixels (18 + 2: a row is five 4-pixel DMA pieces)
constexpr int GB_PATCH_B = 18 * GB_PW * 256;
constexpr int GB_SBMAX = 5 * 4096;
constexpr int GB_CB_OFF = GB_PATCH_B + 3 * GB_SBMAX;
constexpr int GB_LDS = GB_CB_OFF + 2048;
constexpr int GB_CV = 80;                      // channels per pass the constant vectors hold

struct GbParams {
  const void* src; int src_cs, src_co, C; unsigned src_bytes;   // bf16 NHWC source, C % 16 == 0
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
  int n5 = NT & 1;
  if ((NT - 5 * n5) < 0 || (NT - 5 * n5) % 4 != 0) return false;
  const int n4 = (NT - 5 * n5) / 4;
  if (n4 + n5 > GB_MAXP || n4 + n5 < 1) return false;
  pl.npass = n4 + n5;
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
  for (int i = 0; i < pl.npass; ++i) {
    pl.ntp[i] = (i == pl.npass - 1 && n5) ? 5 : 4;
    pl.tile0[i] = 4 * i;
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

template <int NTP, int EPI>
__device__ __forceinline__ void gb_pass(const GbParams& p, const int pass, unsigned char* const smem, const int pt_n,
                                        const int pt_y0, const int pt_x0, const int bid, const bool load_patch0,
                                        const bool first, const bool last) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  constexpr int NP = NTP * 4;               // 1-KB pieces (column tile, k-step) per weight stage
  constexpr int NB = (NP + 7) / 8;          // DMA instructions per wave per stage
  constexpr int SB = NP * 1024;
  constexpr int NPAIR = NTP / 2, TAIL = NTP & 1;
  static_assert(NPAIR * 32 + TAIL * 16 <= GB_CV, "constant vectors");
  unsigned char* const patch = smem;
  unsigned char* const bst = smem + GB_PATCH_B;
  float* const cbuf = reinterpret_cast<float*>(smem + GB_CB_OFF);
  const rsrc_t a_rsrc = make_rsrc(p.src, p.src_bytes);
  const rsrc_t w_rsrc = make_rsrc(p.wp, p.w_bytes);
  const unsigned wbase = p.woff[pass];
  const int KT = p.KT;
  const int tile0 = p.tile0[pass];

  auto b_dma = [&](int q, int st) {
    const unsigned soff0 = wbase + (unsigned)q * (unsigned)SB;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int idx = wave + 8 * i;                // wave-uniform; a wrapped index re-writes identical bytes (benign)
      if (idx >= NP) idx -= NP;
      dma16(w_rsrc, reinterpret_cast<float*>(bst + st * GB_SBMAX + idx * 1024), (unsigned)lane * 16u, soff0 + (unsigned)idx * 1024u);
    }
  };
  // one instruction = 4 consecutive halo pixels of one patch row x 16 groups of 8 channels (90 instructions per patch,
  // instruction u = 5 * row + piece, dealt round-robin to the 8 waves); the 16-byte groups of a pixel are XOR-swizzled
  // by (hx & 15) on the SOURCE side: tap-shifted b128 fragment reads of 16 adjacent pixels are bank-disjoint.  Per
  // instruction the lane offset is (lane constant) + (scalar): no division, nothing worth hoisting.
  const int dma_dx = lane >> 4, dma_s = (lane & 15) ^ (lane >> 4);
  auto patch_dma = [&](int chunk) {
    const int kc = p.C - 128 * chunk;
#pragma unroll 1
    for (int u = wave; u < 90; u += 8) {
      const int hy = (u * 205) >> 10, i4 = (u - 5 * hy) << 2;        // u / 5, 4 * (u % 5)   (u < 90)
      const int y = pt_y0 - 1 + hy, x = pt_x0 - 1 + i4 + dma_dx;
      const int g = dma_s ^ (i4 & 15);
      const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W && g * 8 < kc;
      const unsigned off = ((unsigned)((pt_n * p.H + y) * p.W + x) * (unsigned)p.src_cs + (unsigned)(p.src_co + chunk * 128 + g * 8)) * 2u;
      dma16(a_rsrc, reinterpret_cast<float*>(patch + u * 1024), ok ? off : 0xFFFFFFF0u, 0u);
    }
  };

  // ---- this lane's pixel and fragment addresses
  const int ty = 2 * wave + (l31 >> 4), tx = l31 & 15;
  const int py = pt_y0 + ty, px = pt_x0 + tx;
  const bool pix_ok = py < p.H && px < p.W;
  const int pidx = pix_ok ? (pt_n * p.H + py) * p.W + px : 0;
  const unsigned char* const a_lb = patch + (ty * GB_PW + tx) * 256;
  const unsigned char* const b_lb = bst + lane * 16;

  f32x16 acc[NTP];
#pragma unroll
  for (int j = 0; j < NTP; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  // ---- prologue: (patch chunk 0,) weight tiles 0 and 1, the pass's per-channel constants
  if (load_patch0) patch_dma(0);
  b_dma(0, 0);
  if (KT > 1) b_dma(1, 1);
  const int cb0 = (tile0 >> 1) * 32;          // first norm channel of this pass (forward)
  if constexpr (EPI == 1) {
    // bias gamma | bias beta | noise scale | mean | rstd of the pass's channels -> LDS (read by the epilogue only).
    // Scalar loads: the bias / noise-scale parameters are views into the fused optimizer's flat buffer (4-byte aligned)
    if (tid < 5 * GB_CV) {
      const int v = tid / GB_CV, c = cb0 + (tid - v * GB_CV);
      float val = 0.f;
      if (c < p.sC) {
        if (v == 0) val = p.bg[c];
        else if (v == 1) val = p.bb[c];
        else if (v == 2) val = p.sns ? p.sns[c] : 0.f;
        else if (v == 3) val = p.smean[(size_t)pt_n * p.sC + c];
        else val = p.srstd[(size_t)pt_n * p.sC + c];
      }
      cbuf[tid] = val;
    }
  }
  if (KT > 1) __builtin_amdgcn_s_waitcnt(gb_wait(NB));
  else __builtin_amdgcn_s_waitcnt(gb_wait(0));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (p.tlog && tid == 0 && first) p.tlog[(size_t)bid * 8 + 1] = wall_clock64();

  // epilogue operands requested inside the main loop
  [[maybe_unused]] f32x4 xv[NPAIR][4];
  [[maybe_unused]] f32x4 xt[2];
  [[maybe_unused]] float zv = 0.f;
  [[maybe_unused]] u16x4 mv[NTP][4];
  auto prefetch_epi = [&]() {
    if constexpr (EPI == 1) {
#pragma unroll
      for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          xv[pr][g] = ld4rt<true>(p.sx, (size_t)pidx * p.sx_cs + p.sx_co + cb0 + pr * 32 + 8 * g + 4 * lh, p.sx_f32);
      if constexpr (TAIL != 0) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
          xt[g] = ld4rt<true>(p.sx, (size_t)pidx * p.sx_cs + p.sx_co + cb0 + NPAIR * 32 + 8 * g + 4 * lh, p.sx_f32);
      }
      if (p.sz) zv = p.sz[((size_t)pt_n * p.W + (pix_ok ? px : 0)) * p.H + (pix_ok ? py : 0)];
    } else {
      if (p.mask) {
#pragma unroll
        for (int j = 0; j < NTP; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            mv[j][g] = *reinterpret_cast<const u16x4*>(reinterpret_cast<const unsigned short*>(p.mask) +
                                                       (size_t)pidx * p.mask_cs + p.mask_co + (tile0 + j) * 32 + 8 * g + 4 * lh);
      }
    }
  };

  // ---- main loop over the K-tiles (chunk, tap, 64-k half)
  int it_chunk = 0, it_tap = 0, it_half = 0;
  int kc = p.C < 128 ? p.C : 128;             // channels of the current chunk
  int nhalf = (kc + 63) >> 6;
  int rb = 0, wb = 2;
  prefetch_epi();
  // K-tile q with NKS k-steps of 16 (4: a full 64-k half; 2: the 32-channel last chunk of a data-gradient source)
  auto ktile = [&](const int q, auto nks_c) {
    constexpr int NKS = decltype(nks_c)::value;
    const bool more = q + 2 < KT;
    if (more) b_dma(q + 2, wb);               // that stage was read in K-tile q-1: every wave has passed the barrier
    {
      const int kh = (it_tap * 11) >> 5, kw = it_tap - 3 * kh;
      const unsigned char* const Ap = a_lb + (kh * GB_PW + kw) * 256;
      const unsigned ax = (unsigned)(((((tx + kw) & 15) ^ lh) << 4) ^ (it_half << 7));
      const unsigned char* const Bp = b_lb + rb * GB_SBMAX;
      f32x4 fa[2], fb[2][NTP];
#define GB_READ(SET, S)                                                                                    \
      {                                                                                                    \
        fa[SET] = *reinterpret_cast<const f32x4*>(Ap + (ax ^ (unsigned)((S) << 5)));                       \
        _Pragma("unroll") for (int j = 0; j < NTP; ++j)                                                    \
            fb[SET][j] = *reinterpret_cast<const f32x4*>(Bp + (j * 4 + (S)) * 1024);                        \
      }
#define GB_MMA(SET)                                                                                        \
      {                                                                                                    \
        _Pragma("unroll") for (int j = 0; j < NTP; ++j)                                                    \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[SET][j]),      \
                                                             __builtin_bit_cast(bf16x8, fa[SET]), acc[j], 0, 0, 0); \
      }
      GB_READ(0, 0)
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        if (s + 1 < NKS) GB_READ((s + 1) & 1, s + 1)
        GB_MMA(s & 1)
      }
      // issue order (the scheduler otherwise sinks every fragment read next to its MFMA and waits lgkmcnt(0) per MFMA):
      // the first k-step's reads, then per k-step the next step's NTP + 1 reads spread between this step's NTP MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, NTP + 1, 0);
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        if (s + 1 < NKS) {
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
          for (int j = 1; j < NTP; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, NTP, 0);
        }
      }
#undef GB_READ
#undef GB_MMA
    }
    // advance the K-tile iterator
    bool new_chunk = false;
    if (++it_half == nhalf) {
      it_half = 0;
      if (++it_tap == 9) {
        it_tap = 0;
        ++it_chunk;
        new_chunk = true;
      }
    }
    if (q + 1 < KT) {
      asm volatile("" ::: "memory");
      if (new_chunk) {
        // next 128-channel chunk of the source: every wave is done with the patch; the in-flight weight tiles stay
        __builtin_amdgcn_s_waitcnt(gb_wait(0));
        __builtin_amdgcn_s_barrier();
        kc = p.C - 128 * it_chunk;
        kc = kc < 128 ? kc : 128;
        nhalf = (kc + 63) >> 6;
        patch_dma(it_chunk);
        __builtin_amdgcn_s_waitcnt(gb_wait(0));
      } else {
        if (more) __builtin_amdgcn_s_waitcnt(gb_wait(NB));
        else __builtin_amdgcn_s_waitcnt(gb_wait(0));
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    rb = rb == 2 ? 0 : rb + 1;
    wb = wb == 2 ? 0 : wb + 1;
  };
  // full 64-k K-tiles first; a 32-channel last chunk (data gradient: 2 Cp = 160, 288, ...) ends with 9 two-step tiles
  const int KT4 = (p.C & 127) == 32 ? KT - 9 : KT;
#pragma unroll 1
  for (int q = 0; q < KT4; ++q) ktile(q, std::integral_constant<int, 4>{});
  if constexpr (EPI == 2) {
#pragma unroll 1
    for (int q = KT4; q < KT; ++q) ktile(q, std::integral_constant<int, 2>{});
  }
  if (p.tlog && tid == 0 && last) p.tlog[(size_t)bid * 8 + 2] = wall_clock64();

  // ---- epilogue.  D layout (swapped operands): lane -> pixel l31; regs 4g..4g+3 -> channels 8g + 4 lh + (0..3) of the tile
  __syncthreads();                             // every wave is done with the patch and the weight ring
  // (the epilogue's index arithmetic depends on the lane id only: hidden from the optimiser behind an empty asm, or it is
  //  hoisted out of the persistent tile loop and lives -- spilled -- through the main loop)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  constexpr int SCS = 36;                       // scratch row stride in floats (32 channels + 4: conflict-free)
  const int l31e = lane_e & 31, lhe = lane_e >> 5;
  // per-wave scratch, 32 pixels x 32 channels, inside the (now idle) weight ring: the patch survives for the next pass
  float* const scr = reinterpret_cast<float*>(smem + GB_PATCH_B) + wave * (32 * SCS);
  static_assert(8 * 32 * SCS * 4 <= 3 * GB_SBMAX, "epilogue scratch fits the weight ring");
  auto row_pix = [&](int r) -> int {           // pixel index of row r (0..31) of this wave, or -1
    const int y = pt_y0 + 2 * wave + (r >> 4), x = pt_x0 + (r & 15);
    return (y < p.H && x < p.W) ? (pt_n * p.H + y) * p.W + x : -1;
  };
  if constexpr (EPI == 1) {
    // NCH channels of the group starting at local channel lc0 / norm channel cb: modulate, stage, store along the channels
    auto group = [&](const int lc0, const int NG, auto&& gam, auto&& bet, auto&& xin) {
      f32x4 g1r[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g >= NG) break;
        const int lc = lc0 + 8 * g + 4 * lhe;
        const f32x4 bgv = *reinterpret_cast<const f32x4*>(cbuf + lc);
        const f32x4 bbv = *reinterpret_cast<const f32x4*>(cbuf + GB_CV + lc);
        const f32x4 ns4 = *reinterpret_cast<const f32x4*>(cbuf + 2 * GB_CV + lc);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(cbuf + 3 * GB_CV + lc);
        const f32x4 rs = *reinterpret_cast<const f32x4*>(cbuf + 4 * GB_CV + lc);
        const f32x4 x4 = xin(g);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = x4[e] + zv * ns4[e];
          g1r[g][e] = 1.f + gam(g, e) + bgv[e];
          v[e] = apply_act((x - mu[e]) * rs[e] * g1r[g][e] + (bet(g, e) + bbv[e]), p.act, p.slope);
        }
        *reinterpret_cast<f32x4*>(scr + l31e * SCS + 8 * g + 4 * lhe) = v;
      }
      const int cb = cb0 + lc0;                // first norm channel of the group; NG * 8 channels
      // same wave wrote and reads: LDS operations of a wave complete in order.  ``which`` 0: the activation, 1: (1 + gamma)
      auto copy_out = [&](const int which) {
        void* const dst = which == 0 ? p.out : p.g1p;
        const int dcs = which == 0 ? p.out_cs : p.sC, dco = which == 0 ? p.out_co : 0;
        const bool f32 = which == 0 ? p.out_f32 != 0 : p.g1_bf16 == 0;
        if (!f32) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {        // bf16 rows: 32 pixels x NG pieces of 8 channels
            const int t = lane_e + 64 * k;
            if (t < 32 * NG) {
              const int r = t / NG, kk = t - r * NG;
              const int po = row_pix(r);
              if (po >= 0) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(scr + r * SCS + kk * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(scr + r * SCS + kk * 8 + 4);
                *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned short*>(dst) + (size_t)po * dcs + dco + cb + kk * 8) = pack_bf16x8(lo, hi);
              }
            }
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {        // fp32 rows: 32 pixels x 2 NG pieces of 4 channels
            const int t = lane_e + 64 * k;
            if (t < 64 * NG) {
              const int r = t / (2 * NG), kk = t - r * (2 * NG);
              const int po = row_pix(r);
              if (po >= 0)
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dst) + (size_t)po * dcs + dco + cb + kk * 4) =
                    *reinterpret_cast<const f32x4*>(scr + r * SCS + kk * 4);
            }
          }
        }
      };
      copy_out(0);
      if (p.g1p) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (g >= NG) break;
          *reinterpret_cast<f32x4*>(scr + l31e * SCS + 8 * g + 4 * lhe) = g1r[g];
        }
        copy_out(1);
      }
    };
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr)
      group(pr * 32, 4, [&](int g, int e) { return acc[2 * pr][4 * g + e]; }, [&](int g, int e) { return acc[2 * pr + 1][4 * g + e]; },
            [&](int g) { return xv[pr][g]; });
    if constexpr (TAIL != 0)
      group(NPAIR * 32, 2, [&](int g, int e) { return acc[NTP - 1][4 * g + e]; },
            [&](int g, int e) { return acc[NTP - 1][4 * (g + 2) + e]; }, [&](int g) { return xt[g]; });
  } else {
#pragma unroll
    for (int j = 0; j < NTP; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = acc[j][4 * g + e];
          if (p.mask) a = bf2f(mv[j][g][e]) > 0.f ? a : a * p.slope;
          v[e] = a;
        }
        *reinterpret_cast<f32x4*>(scr + l31e * SCS + 8 * g + 4 * lhe) = v;
      }
      const int cb = (tile0 + j) * 32;
      if (p.out_f32) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int t = lane_e + 64 * k, r = t >> 3, kk = t & 7;
          const int po = row_pix(r);
          if (po >= 0)
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)po * p.out_cs + p.out_co + cb + kk * 4) =
                *reinterpret_cast<const f32x4*>(scr + r * SCS + kk * 4);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int t = lane_e + 64 * k, r = t >> 2, kk = t & 3;
          const int po = row_pix(r);
          if (po >= 0) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(scr + r * SCS + kk * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(scr + r * SCS + kk * 8 + 4);
            *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned short*>(p.out) + (size_t)po * p.out_cs + p.out_co + cb + kk * 8) =
                pack_bf16x8(lo, hi);
          }
        }
      }
    }
  }
  __syncthreads();                             // the scratch lives in the weight ring: the next pass / tile refills it
}

// One launch covers the passes [pass0, pass1) of the plan, all of NTP column tiles (a single-chunk source's patch stays
// resident across them: the epilogue's scratch lives in the weight ring).
template <int NTP, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void spade_gb_kernel(const GbParams p, const int pass0, const int pass1) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[GB_LDS];
  const int tx = (p.W + 15) >> 4, ty = (p.H + 15) >> 4;
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
    const int mt = xcd_remap(bid, p.m_tiles);
    const int pt_n = mt / (tx * ty);
    const int rr = mt - pt_n * (tx * ty);
    const int pt_y0 = (rr / tx) << 4, pt_x0 = (rr % tx) << 4;
#pragma unroll 1
    for (int pass = pass0; pass < pass1; ++pass)
      gb_pass<NTP, EPI>(p, pass, smem, pt_n, pt_y0, pt_x0, bid, pass == pass0 || p.nchunk > 1, pass == pass0, pass == pass1 - 1);
    if (p.tlog) {
      __builtin_amdgcn_s_waitcnt(gb_wait(0));
      if (threadIdx.x == 0) p.tlog[(size_t)bid * 8 + 3] = wall_clock64();
    }
  }
}

static int gb_n_cu() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  return n_cu;
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
  return tiles >= 256 ? 1 : 0;      // fewer tiles than CUs: the generic tiles (more, smaller blocks; split-K) fill the chip better
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
  const int64_t sbytes = (int64_t)d->N * d->H * d->W * d->src_cstride * 2;
  HRV_REQUIRE(sbytes < (int64_t)0xFFFFFFF0, "spade_gb: source exceeds the 32-bit buffer range (%lld bytes)", (long long)sbytes);
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
  {
    const char* e = getenv("HRV_PATCH_TLOG");      // diag only: device buffer (hex address) for per-tile phase timestamps
    p.tlog = e ? (unsigned long long*)strtoull(e, nullptr, 16) : nullptr;
  }
  int grid = gb_n_cu();
  if (grid > p.m_tiles) grid = p.m_tiles;
  if (d->mode == 0) {
    HRV_REQUIRE(d->x && d->mean && d->rstd && d->bias_gamma && d->bias_beta, "spade_gb: null epilogue pointer");
    HRV_REQUIRE((d->noise_z == nullptr) == (d->noise_scale == nullptr), "spade_gb: noise_z/noise_scale go together");
    HRV_REQUIRE(d->out_cstride >= d->out_coff + d->C, "spade_gb: out slice");
    HRV_REQUIRE(d->x_cstride % 4 == 0 && d->x_coff % 4 == 0 && d->x_coff + d->C <= d->x_cstride,
                "spade_gb: x slice");
    HRV_REQUIRE((((uintptr_t)d->x | (uintptr_t)d->g1p) & 15) == 0, "spade_gb: x / g1p must be 16-byte aligned");
    HRV_REQUIRE(d->stat_stride == d->C, "spade_gb: stat_stride must equal C (C %% 16 == 0: no channel padding)");
    p.sx = (const float*)d->x; p.sx_cs = d->x_cstride; p.sx_co = d->x_coff; p.sx_f32 = d->x_f32; p.sC = d->stat_stride;
    p.smean = d->mean; p.srstd = d->rstd; p.sz = d->noise_z; p.sns = d->noise_scale; p.bg = d->bias_gamma; p.bb = d->bias_beta;
    p.g1p = d->g1p; p.g1_bf16 = d->g1p_bf16;
    // the passes of 4 column tiles in one launch, the 5-tile tail pass (16-channel tail) in another
    const int n4 = pl.ntp[pl.npass - 1] == 5 ? pl.npass - 1 : pl.npass;
    if (n4 > 0) hipLaunchKernelGGL((spade_gb_kernel<4, 1>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, 0, n4);
    if (n4 < pl.npass) hipLaunchKernelGGL((spade_gb_kernel<5, 1>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, n4, pl.npass);
  } else {
    HRV_REQUIRE(d->out_cstride >= d->out_coff + d->hid, "spade_gb: out slice");
    HRV_REQUIRE(d->mask == nullptr || (d->mask_cstride % 4 == 0 && d->mask_coff % 4 == 0 && ((uintptr_t)d->mask & 7) == 0),
                "spade_gb: mask slice");
    p.mask = d->mask; p.mask_cs = d->mask_cstride; p.mask_co = d->mask_coff;
    hipLaunchKernelGGL((spade_gb_kernel<4, 2>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, 0, 1);
  }
  return check_launch("spade_gb_kernel");
}
