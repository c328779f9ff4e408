// SPADENorm forward, fused end to end (network_generator.py:93-121):
//
//     actv   = ReLU(conv_shared(seg))                       3x3, label_nc (7) -> 128        (:97-102, :116)
//     gamma  = conv_gamma(actv),  beta = conv_beta(actv)    3x3, 128 -> C each              (:117-118)
//     out    = act(IN(x + noise) * (1 + gamma) + beta)                                      (:104-110, :120-121)
//
// in ONE kernel: ``actv`` is computed inside the block from the 8-channel label patch and never travels through HBM
// (the training forward may still WRITE it once, for the backward; the no_grad forward of the discriminator step and
// inference never do).  What that buys over spade_gb.hip (which reads a materialised actv):
//
//   * the 92 KB actv halo patch of a 16x16-pixel tile is gone from LDS: the block keeps ONE 64-channel half of it (41 KB),
//     produced by 60 extra MFMAs per wave (K = 9 taps x 8 label channels, +8 % of the tile's matrix work) from a 6.4 KB
//     label patch, so a block needs 81,472 B of LDS and <= 256 registers and TWO blocks share a CU: one block's epilogue
//     (VALU + stores, matrix pipes idle), prologue and barrier waits run under the other block's MFMAs -- with one
//     block per CU the epilogue alone was a quarter of the tile time;
//   * no conv_shared launch, no 256 B/pixel read of actv (x 1.27 halo) per norm and forward.
//
// Block = 256 threads = 4 waves, a 16x16-pixel tile, all columns of a pass (<= 5 column tiles of 32); wave w owns tile
// rows 4w..4w+3 (2 x 32 pixels) like spade_gb.hip.  The K loop runs over the two 64-channel halves of actv:
//
//     item stream through a 3-stage LDS ring (10 KB stages, LDS-DMA, fragment-ordered):
//        [Wshared half 0] [W k-tile 0 .. 17] [Wshared half 1] [W k-tile 18 .. 35]        k-tile = (half, tap, 32 k)
//     while item i is consumed, item i+2 is requested into the stage item i-1 left; a counted s_waitcnt + one barrier per
//     item publishes item i+1.  "Consuming" a Wshared item = computing that half of the 18x18 actv patch (bias, ReLU,
//     zero outside the image, bf16) into LDS; consuming a k-tile = 2 k-steps of 2 x NTP MFMAs (32x32x16 bf16, swapped
//     operands: a lane ends up with 4 consecutive channels of one pixel, gamma and beta of a channel in the same lane).
//
// LDS (81,472 B): ring 3 x 10,240 | label patch 20x20 px x 16 B (7,168) | actv half patch 18x18 px x 128 B (41,472,
// 16-byte groups XOR-swizzled by (hx >> 1) & 7: the tap-shifted b128 fragment reads and the 8-byte producer writes are
// bank-conflict free) | per-channel constants + conv_shared bias (2,112).  The epilogue's staging scratch lives in the ring.
#include <string.h>

#include <type_traits>
#include <utility>

#include "conv_params.h"

namespace hrv {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void gf_store16(f32x4 v, rsrc_t r, unsigned voff) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, (int)voff, 0, 0);
}
#else
__device__ inline void gf_store16(f32x4, rsrc_t, unsigned) {}
#endif

__device__ __forceinline__ f32x4 gf_acc4(const f32x16& a, int g) {
  f32x4 r;
  r[0] = a[4 * g]; r[1] = a[4 * g + 1]; r[2] = a[4 * g + 2]; r[3] = a[4 * g + 3];
  return r;
}

constexpr int GF_MAXP = 16;
constexpr int GF_SB = 10240;                            // ring stage: 32 k x 160 columns = one conv_shared half (80 k x 64 columns)
constexpr int GF_SEG_OFF = 3 * GF_SB;                   // 30,720
constexpr int GF_SEG_B = 7168;                          // 20 x 20 label pixels x 16 B = 6,400, DMA'd as 7 x 1 KB
constexpr int GF_PATCH_OFF = GF_SEG_OFF + GF_SEG_B;     // 37,888
constexpr int GF_PP = 18;                               // actv patch pitch in pixels
constexpr int GF_NPIX = 18 * 18;                        // 324
constexpr int GF_PATCH_B = GF_NPIX * 128;               // 41,472
constexpr int GF_CB_OFF = GF_PATCH_OFF + GF_PATCH_B;    // 79,360
constexpr int GF_CV = 80;                               // norm channels per pass the constant vectors hold
constexpr int GF_CB_B = 5 * GF_CV * 4 + 128 * 4;        // 2,112
constexpr int GF_LDS = GF_CB_OFF + GF_CB_B;             // 81,472: two blocks per CU (2 x 64 allocation granules of 1,280 B)
static_assert(2 * ((GF_LDS + 1279) / 1280) * 1280 <= 160 * 1024, "two blocks per CU");
constexpr int GF_WSH_B = 2 * GF_SB;                     // packed conv_shared weights, both halves
constexpr int GF_HDR_B = GF_WSH_B + 512;                // + its bias (128 fp32)
constexpr int GF_KT = 36;                               // k-tiles per pass: 2 halves x 9 taps x 2

struct GfParams {
  const void* seg; int seg_H, seg_W, seg_shift; unsigned seg_bytes;     // bf16 [N][seg_H][seg_W][8]; seg_bytes: ONE image
  int N, H, W;
  const void* wp; unsigned w_bytes;
  int npass;
  int ntp[GF_MAXP];         // column tiles of 32 per pass (2, 4 or 5)
  int tile0[GF_MAXP];       // first column tile of the pass
  unsigned woff[GF_MAXP];   // byte offset of the pass's k-tile stream in the packed weights
  int m_tiles;
  const float* sx; int sx_cs, sx_co, sx_f32, sC;
  // x = cat(nearest_up2(lo), hi), never materialised (sx_up_c > 0, fp32): channels [0, sx_up_c) from sx = lo [N][H/2][W/2][sx_cs]
  // at (y >> 1, x >> 1), the others from sx2 = hi [N][H][W][sx2_cs]
  const float* sx2; int sx2_cs, sx2_co, sx_up_c;
  const float *smean, *srstd, *sz, *sns, *bg, *bb;
  void* g1p;
  int act; float slope;
  void* out; int out_cs, out_co;
  void* actv; int actv_cs, actv_co;        // optional: ReLU(conv_shared(seg)) as bf16 NHWC (training forward, for the backward)
  unsigned long long* tlog;
  int pp;                   // one (tile, pass) per unit of work (see spade_fused_kernel)
};

struct GfPlan {
  int npass, ntp[GF_MAXP], tile0[GF_MAXP];
  unsigned woff[GF_MAXP];
  long long bytes;
};

// columns = (gamma32 | beta32) pairs (+ one 16|16 tail tile) of C norm channels; passes of 4 column tiles, then one of 2,
// then (16-channel tail) one of 5 = two pairs + the tail
static bool gf_plan(int C, GfPlan& pl) {
  memset(&pl, 0, sizeof(pl));
  if (C < 32 || C % 16 != 0) return false;
  const int NT = 2 * (C / 32) + (C % 32 ? 1 : 0);
  const int n5 = NT & 1;
  const int rest = NT - 5 * n5;
  if (rest < 0) return false;
  const int n2 = (rest % 4 == 2) ? 1 : 0;
  const int n4 = (rest - 2 * n2) / 4;
  if (n4 + n2 + n5 > GF_MAXP || n4 + n2 + n5 < 1) return false;
  pl.npass = n4 + n2 + n5;
  long long off = GF_HDR_B;
  int t0 = 0;
  for (int i = 0; i < pl.npass; ++i) {
    pl.ntp[i] = i < n4 ? 4 : (i < n4 + n2 ? 2 : 5);
    pl.tile0[i] = t0;
    t0 += pl.ntp[i];
    pl.woff[i] = (unsigned)off;
    off += (long long)GF_KT * pl.ntp[i] * 2048;
  }
  pl.bytes = off;
  return off < (long long)0xFFFFFFF0;
}

// ------------------------------------------------------------------------------------------------ weight packer
struct GfPackParams {
  GfPlan pl;
  int C, label_nc;
  const float* wsh;   // conv_shared.weight [128][label_nc][3][3]
  const float* bsh;   // conv_shared.bias [128]
  const float* wg;    // conv_gamma.weight [C][128][3][3]
  const float* wb;
  unsigned short* out;
};

__global__ __launch_bounds__(256) void gf_pack_kernel(const GfPackParams p) {
  // one thread per 16-byte group
  const long long G = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = p.pl.bytes / 16;
  if (G >= total) return;
  uint4 o;
  if (G < GF_WSH_B / 16) {
    // conv_shared as a GEMM over K = (tap, label channel): [half][column tile (2)][k-step (5)][lane][8 bf16];
    // lane & 31 -> actv channel, lane >> 5 -> tap 2 s + (lane >> 5) (tap 9 does not exist: zeros), element -> label channel
    int r = (int)G;
    const int lane = r & 63;
    r >>= 6;
    const int half = r / 10, piece = r - 10 * half;
    const int ct = piece / 5, s = piece - 5 * ct;
    const int c = half * 64 + ct * 32 + (lane & 31);
    const int t = 2 * s + (lane >> 5);
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float w = 0.f;
      if (t < 9 && e < p.label_nc) w = p.wsh[((size_t)c * p.label_nc + e) * 9 + t];
      v[e] = f2bf(w);
    }
    o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
    o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
  } else if (G < GF_HDR_B / 16) {
    const int i = (int)(G - GF_WSH_B / 16) * 4;
    o.x = __builtin_bit_cast(unsigned, p.bsh[i]); o.y = __builtin_bit_cast(unsigned, p.bsh[i + 1]);
    o.z = __builtin_bit_cast(unsigned, p.bsh[i + 2]); o.w = __builtin_bit_cast(unsigned, p.bsh[i + 3]);
  } else {
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
    const int l31 = lane & 31, lh = lane >> 5;
    // k-tile -> (half, tap, 32-k sub-tile)
    const int half = kt / 18, rq = kt - 18 * half;
    const int tap = rq >> 1, sub = rq & 1;
    const int kh = tap / 3, kw = tap - 3 * kh;
    const int k0 = half * 64 + sub * 32 + s * 16 + lh * 8;
    const int jt = p.pl.tile0[pass] + j;
    const bool tail = (p.C % 32) != 0 && jt == 2 * (p.C / 32);
    int c;
    bool beta;
    if (tail) { beta = l31 >= 16; c = (jt >> 1) * 32 + (l31 & 15); }
    else { beta = (jt & 1) != 0; c = (jt >> 1) * 32 + l31; }
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float w = 0.f;
      if (c < p.C) w = (beta ? p.wb : p.wg)[(((size_t)c * 128 + k0 + e) * 3 + kh) * 3 + kw];
      v[e] = f2bf(w);
    }
    o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
    o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
  }
  reinterpret_cast<uint4*>(p.out)[G] = o;
}

// ------------------------------------------------------------------------------------------------ the kernel
// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] bits 3:0, expcnt bits 6:4, lgkmcnt bits 11:8, vmcnt[5:4] bits 15:14)
constexpr int gf_wait(int vm) { return (vm & 15) | (7 << 4) | (0 << 8) | ((vm >> 4) << 14); }   // vmcnt(vm) lgkmcnt(0)

struct GfTile { int n, y0, x0; };
__device__ __forceinline__ GfTile gf_tile(const GfParams& p, int bid) {
  const int tx = (p.W + 15) >> 4, ty = (p.H + 15) >> 4;
  const int mt = xcd_remap(bid, p.m_tiles);
  GfTile t;
  t.n = mt / (tx * ty);
  const int rr = mt - t.n * (tx * ty);
  t.y0 = (rr / tx) << 4;
  t.x0 = (rr % tx) << 4;
  return t;
}

typedef __bf16 gf_bf16x4 __attribute__((ext_vector_type(4)));

// The head of a (tile, pass): what its first barrier waits for -- the label patch (first pass of a tile), the conv_shared
// weights of half 0 (item 0 -> stage 0) and k-tile 0 (item 1 -> stage 1).  Issued by the PREVIOUS pass right after its main
// loop (ring and label patch are free then), so the loads fly under that pass's epilogue and are OLDER than its stores:
// the counted wait at the top of this pass does not wait for the stores to drain.
template <int NTP>
__device__ __forceinline__ void gf_head(const GfParams& p, const int pass, unsigned char* const smem, const GfTile T, const bool load_seg,
                                        const int wave, const int lane) {
  constexpr int NPW = NTP * 2, NBW = (NPW + 3) / 4;
  unsigned char* const ring = smem;
  unsigned char* const segp = smem + GF_SEG_OFF;
  const rsrc_t w_rsrc = make_rsrc(p.wp, p.w_bytes);
  if (load_seg) {
    // label patch: halo pixel (j, i) of the 20 x 20 patch = level pixel (y0 - 2 + j, x0 - 2 + i) = label-map pixel
    // (that << seg_shift) -- F.interpolate(segmap, size=x.size()[2:], mode='nearest') of network_generator.py:115 for the
    // power-of-two ratios of the generator; out of the image: zeros (conv_shared's zero padding)
    const rsrc_t s_rsrc = make_rsrc(reinterpret_cast<const char*>(p.seg) + (size_t)T.n * p.seg_bytes, p.seg_bytes);
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      // 7 pieces of 64 pixels: waves 0..2 issue two; wave 3's second instruction re-loads its first piece (same bytes to the
      // same place: every wave issues the same number of instructions -- the vmcnt arithmetic is the same in every wave)
      const int u = wave + 4 * uu < 7 ? wave + 4 * uu : wave;
      const int q = u * 64 + lane;
      const int j = (q * 3277) >> 16, i = q - 20 * j;                    // q / 20, q % 20 (q < 448)
      const int y = T.y0 - 2 + j, x = T.x0 - 2 + i;
      const bool ok = q < 400 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const unsigned off = (unsigned)((y << p.seg_shift) * p.seg_W + (x << p.seg_shift)) * 16u;
      dma16(s_rsrc, reinterpret_cast<float*>(segp + u * 1024), ok ? off : 0xFFFFFFF0u, 0u);
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {                                          // item 0: conv_shared half 0 -> stage 0
    int idx = wave + 4 * k;
    idx = idx < 10 ? idx : idx - 2;
    dma16(w_rsrc, reinterpret_cast<float*>(ring + idx * 1024), (unsigned)lane * 16u, (unsigned)idx * 1024u);
  }
#pragma unroll
  for (int k = 0; k < NBW; ++k) {                                        // item 1: k-tile 0 -> stage 1
    int idx = wave + 4 * k;
    idx = idx < NPW ? idx : idx - (NPW == 10 ? 2 : 4);
    dma16(w_rsrc, reinterpret_cast<float*>(ring + GF_SB + idx * 1024), (unsigned)lane * 16u, p.woff[pass] + (unsigned)idx * 1024u);
  }
}

// One (tile, pass).  Its head is in flight or landed (gf_head).  ``load_consts``: the per-channel constants in LDS belong to
// another (image, pass).  ``wait_all``: nothing of a previous pass is in flight behind the head (first pass of the block).
// ``nxt_*``: the (tile, pass) whose head this pass issues after its main loop (nxt_pass < 0: none).
template <int NTP>
__device__ __forceinline__ void gf_pass(const GfParams& p, const int pass, unsigned char* const smem, const GfTile T, const int bid,
                                        const bool load_consts, const bool wait_all, const bool save_actv, const bool first, const bool last,
                                        const int nxt_pass, const GfTile NT_, const bool nxt_seg) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  constexpr int NPW = NTP * 2;                               // 1-KB pieces (column tile, k-step) of a k-tile
  constexpr int NBW = (NPW + 3) / 4;                         // DMA instructions per wave per k-tile (NTP 5: 3, two of the 12 re-load a piece)
  constexpr int NBS = 3;                                     // ... per conv_shared half (10 pieces)
  constexpr int NPAIR = NTP / 2, TAIL = NTP & 1;
  constexpr int NST1 = 4 * NPAIR + 2 * TAIL, NST2 = 2 * NST1;   // global stores of one epilogue per wave (out; out + (1 + gamma))
  constexpr int NSA = 8;                                     // global stores of one saved actv half per wave
  static_assert(NPAIR * 32 + TAIL * 16 <= GF_CV, "constant vectors");
  static_assert(NPW * 1024 <= GF_SB, "ring stage");
  unsigned char* const ring = smem;
  unsigned char* const segp = smem + GF_SEG_OFF;
  unsigned char* const patch = smem + GF_PATCH_OFF;
  float* const cbuf = reinterpret_cast<float*>(smem + GF_CB_OFF);
  float* const bsh = cbuf + 5 * GF_CV;
  const rsrc_t w_rsrc = make_rsrc(p.wp, p.w_bytes);
  const unsigned wbase = p.woff[pass];
  const int tile0 = p.tile0[pass];
  const int pt_n = T.n, pt_y0 = T.y0, pt_x0 = T.x0;

  // ---- stream items -> ring stages.  Piece k of this wave: index wave + 4 k, folded back by 2 when the item has no such
  // piece (10 pieces over 4 waves: waves 2 / 3 re-load pieces 8 / 9 -- same bytes to the same place; every wave issues
  // the same number of instructions, so the DMA sits in straight-line code between the MFMAs and the vmcnt arithmetic is
  // the same in every wave)
  auto dma_w = [&](const int kt, const int st, const int k) {              // k-tile kt (0..35) of this pass
    int idx = wave + 4 * k;
    idx = idx < NPW ? idx : idx - (NPW == 10 ? 2 : 4);
    dma16(w_rsrc, reinterpret_cast<float*>(ring + st * GF_SB + idx * 1024), (unsigned)lane * 16u,
          wbase + (unsigned)kt * (unsigned)(NPW * 1024) + (unsigned)idx * 1024u);
  };
  auto dma_s = [&](const int half, const int st, const int k) {            // conv_shared weights of a half
    int idx = wave + 4 * k;
    idx = idx < 10 ? idx : idx - 2;
    dma16(w_rsrc, reinterpret_cast<float*>(ring + st * GF_SB + idx * 1024), (unsigned)lane * 16u,
          (unsigned)half * (unsigned)GF_SB + (unsigned)idx * 1024u);
  };

  // every wave is done with the previous (tile, pass): its epilogue's staging scratch lives in the patch, its constants in cbuf
  __builtin_amdgcn_s_waitcnt(gf_wait(63));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (p.tlog && tid == 0 && first) p.tlog[(size_t)bid * 8 + 0] = wall_clock64();

  // ---- this lane's pixels (two: tile rows 4 w + (l31 >> 4) and + 2) and fragment addresses
  const int ty = 4 * wave + (l31 >> 4), tx = l31 & 15;
  const unsigned char* const a_lb = patch + (ty * GF_PP + tx) * 128;
  const unsigned char* const b_lb = ring + lane * 16;

  f32x16 acc[2][NTP];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTP; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- the pass's per-channel constants -> LDS: 1 + bias_gamma | bias_beta | rstd | noise_scale * rstd | -mean * rstd,
  // so that IN(x + z ns) = x * rstd + (z * (ns rstd) - mean rstd); and conv_shared's bias
  const int cb0 = (tile0 >> 1) * 32;          // first norm channel of this pass
  if (load_consts) {
    for (int t = tid; t < GF_CV; t += 256) {
      const int c = cb0 + t;
      float b1 = 1.f, b2 = 0.f, rs = 0.f, nr = 0.f, mr = 0.f;
      if (c < p.sC) {
        b1 = 1.f + p.bg[c];
        b2 = p.bb[c];
        rs = p.srstd[(size_t)pt_n * p.sC + c];
        nr = p.sns ? p.sns[c] * rs : 0.f;
        mr = -p.smean[(size_t)pt_n * p.sC + c] * rs;
      }
      cbuf[t] = b1; cbuf[GF_CV + t] = b2; cbuf[2 * GF_CV + t] = rs; cbuf[3 * GF_CV + t] = nr; cbuf[4 * GF_CV + t] = mr;
    }
  }

  // ---- one half of the actv patch: ReLU(conv_shared(label patch)), zero outside the image, bf16 -> LDS.
  // GEMM per 32 patch pixels: D[channel][pixel] += Wsh[channel][k] * S[k][pixel], k = (tap, label channel), 5 k-steps of
  // 16 (= 2 taps): a lane's B operand is the 16-byte label pixel under tap 2 s + (lane >> 5).
  auto conv_shared = [&](const int half, const int st) {
    // (index arithmetic that depends on the lane id only is hidden from the optimiser behind an empty asm: hoisted out of the
    //  persistent tile loop it would live -- spilled -- through the main loop)
    int lane_c = lane;
    asm volatile("" : "+v"(lane_c));
    const int l31 = lane_c & 31, lh = lane_c >> 5;
    const unsigned char* const wst = ring + st * GF_SB + lane_c * 16;
#pragma nounroll
    for (int pt = wave; pt < (GF_NPIX + 31) / 32; pt += 4) {
      // (nothing of this loop may be hoisted or software-pipelined across iterations: in the second half the 32 NTP
      //  accumulators of the main loop are live, and a hoisted weight fragment set costs 40 registers)
      asm volatile("" ::: "memory");
      const int P = 32 * pt + l31;
      const int Pc = P < GF_NPIX ? P : GF_NPIX - 1;
      const int hy = (Pc * 3641) >> 16, hx = Pc - GF_PP * hy;           // Pc / 18, Pc % 18 (Pc < 324)
      f32x16 c0, c1;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bsh + half * 64 + 8 * g + 4 * lh);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bsh + half * 64 + 32 + 8 * g + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) { c0[4 * g + e] = b0[e]; c1[4 * g + e] = b1[e]; }
      }
      // this lane's label pixel under tap t = 2 s + lh (tap 9: zero weights, any finite operand -- tap 8 again)
      const unsigned char* const sp = segp + (hy * 20 + hx) * 16;
      auto tap_off = [&](const int s) {
        int t = 2 * s + lh;
        t = t < 9 ? t : 8;
        const int kh = (t * 11) >> 5, kw = t - 3 * kh;
        return (kh * 20 + kw) * 16;
      };
      f32x4 fb_, fa0, fa1;
      fb_ = *reinterpret_cast<const f32x4*>(sp + tap_off(0));
      fa0 = *reinterpret_cast<const f32x4*>(wst);
      fa1 = *reinterpret_cast<const f32x4*>(wst + 5 * 1024);
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        f32x4 nb = fb_, n0 = fa0, n1 = fa1;
        if (s < 4) {
          nb = *reinterpret_cast<const f32x4*>(sp + tap_off(s + 1));
          n0 = *reinterpret_cast<const f32x4*>(wst + (s + 1) * 1024);
          n1 = *reinterpret_cast<const f32x4*>(wst + (5 + s + 1) * 1024);
        }
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa0), __builtin_bit_cast(bf16x8, fb_), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa1), __builtin_bit_cast(bf16x8, fb_), c1, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        fb_ = nb; fa0 = n0; fa1 = n1;
      }
      const int y = pt_y0 - 1 + hy, x = pt_x0 - 1 + hx;
      const float keep = ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) ? 1.f : 0.f;   // gamma|beta's zero padding
      if (P < GF_NPIX) {
        const unsigned sw = (unsigned)(hx >> 1) & 7u;
        unsigned char* const dst = patch + P * 128 + lh * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v0 = gf_acc4(c0, g), v1 = gf_acc4(c1, g);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f) * keep; v1[e] = fmaxf(v1[e], 0.f) * keep; }
          *reinterpret_cast<gf_bf16x4*>(dst + ((((unsigned)g) ^ sw) << 4)) = __builtin_convertvector(v0, gf_bf16x4);
          *reinterpret_cast<gf_bf16x4*>(dst + ((((unsigned)(4 + g)) ^ sw) << 4)) = __builtin_convertvector(v1, gf_bf16x4);
        }
      }
    }
  };
  // the tile's 16 x 16 interior of the half just produced -> global (training forward: the backward reads actv)
  auto store_actv = [&](const int half) {
    const size_t img_px = (size_t)p.H * p.W;
    const rsrc_t a_rsrc = make_rsrc(reinterpret_cast<const char*>(p.actv) + (size_t)pt_n * img_px * p.actv_cs * 2,
                                    (unsigned)(img_px * p.actv_cs * 2));
    int tid_s = tid;
    asm volatile("" : "+v"(tid_s));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid_s + 256 * i;
      const int pxl = idx >> 3, g = idx & 7;
      const int hy = (pxl >> 4) + 1, hx = (pxl & 15) + 1;
      const f32x4 v = *reinterpret_cast<const f32x4*>(patch + (hy * GF_PP + hx) * 128 + ((((unsigned)g) ^ ((unsigned)(hx >> 1) & 7u)) << 4));
      const int y = pt_y0 + hy - 1, x = pt_x0 + hx - 1;
      const unsigned off = (y < p.H && x < p.W) ? (unsigned)((y * p.W + x) * p.actv_cs + p.actv_co + half * 64 + g * 8) * 2u : 0xFFFFFFF0u;
      gf_store16(v, a_rsrc, off);
    }
  };

  // The head (label patch, items 0 and 1) has landed.  Behind it in this wave's queue sit only the previous pass's epilogue
  // stores (NST of them: they need not drain) -- unless this pass loaded constants or is the block's first
  if (wait_all || load_consts) __builtin_amdgcn_s_waitcnt(gf_wait(0));
  else if (p.g1p) __builtin_amdgcn_s_waitcnt(gf_wait(NST2));
  else __builtin_amdgcn_s_waitcnt(gf_wait(NST1));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int k = 0; k < NBW; ++k) dma_w(1, 2, k);              // item 2: k-tile 1 -> stage 2
  conv_shared(0, 0);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(gf_wait(63));                   // (LDS writes only: the k-tile in flight stays in flight)
  __builtin_amdgcn_s_barrier();                              // half 0 of the patch is published; stage 0 is free
  asm volatile("" ::: "memory");
  if (p.tlog && tid == 0 && first) p.tlog[(size_t)bid * 8 + 1] = wall_clock64();

  // ---- main loop over the k-tiles (half, tap, 32-k sub-tile).  Fragment reads run one k-step ahead of the MFMAs, ACROSS
  // k-tiles: the wait + barrier that publishes the next item sits in front of the LAST k-step's MFMAs of this one (all of
  // this wave's reads of it are complete there), so the next k-tile's first fragments are read under those MFMAs.
  int kt = 0;                     // current k-tile
  int it_tap = 0, it_sub = 0;
  int rb = 1, wb = 0;             // ring stage of the current k-tile / the stage to fill next
  // A fragments (pixels) are double-buffered, B fragments (weights) are refilled IN PLACE: fragment j of the next k-step is
  // read right behind the two MFMAs that consume fragment j of this one (2 NTP MFMAs = 64 NTP cycles ahead of its use) --
  // 2 x 8 + 4 NTP fragment registers instead of 2 x (8 + 4 NTP): with 32 NTP accumulators the wave has to stay under 256
  f32x4 fa[2][2], fb[NTP];
  const unsigned char* Ap;
  unsigned ax;
  const unsigned char* Bp;
  auto point = [&]() {
    const int kh = (it_tap * 11) >> 5, kw = it_tap - 3 * kh;
    Ap = a_lb + (kh * GF_PP + kw) * 128;
    ax = (unsigned)(((4 * it_sub + lh) ^ (((tx + kw) >> 1) & 7)) << 4);
    Bp = b_lb + rb * GF_SB;
  };
#define GF_READ_A(SET, S)                                                                                  \
  {                                                                                                        \
    fa[SET][0] = *reinterpret_cast<const f32x4*>(Ap + (ax ^ (unsigned)((S) << 5)));                        \
    fa[SET][1] = *reinterpret_cast<const f32x4*>(Ap + 2 * GF_PP * 128 + (ax ^ (unsigned)((S) << 5)));      \
  }
#define GF_READ_B(J, S) fb[J] = *reinterpret_cast<const f32x4*>(Bp + ((J) * 2 + (S)) * 1024);
  // the 2 NTP MFMAs of one k-step over fa[SET]; behind pair j: the refill of fb[j] with k-step SN of the stage Bp points
  // at (REFILL), and -- behind the first NDMA pairs -- one DMA instruction of the item requested under this k-tile
#define GF_STEP(SET, REFILL, SN, DMA_STMT, NDMA)                                                           \
  {                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < NTP; ++j) {                                                      \
      acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[j]),              \
                                                          __builtin_bit_cast(bf16x8, fa[SET][0]), acc[0][j], 0, 0, 0); \
      acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[j]),              \
                                                          __builtin_bit_cast(bf16x8, fa[SET][1]), acc[1][j], 0, 0, 0); \
      if constexpr (REFILL) { GF_READ_B(j, SN) }                                                           \
      if (j < (NDMA)) { const int k = j; DMA_STMT }                                                        \
    }                                                                                                      \
    if constexpr ((NDMA) > NTP) {                                                                          \
      _Pragma("unroll") for (int k = NTP; k < (NDMA); ++k) { DMA_STMT }                                    \
    }                                                                                                      \
  }
  // issue order of one k-step: the two A reads of the next step first, then (2 MFMAs, 1 B refill[, 1 DMA]) per column tile
  // (left alone, the scheduler sinks every read next to its consumer and waits lgkmcnt(0) per MFMA)
#define GF_ORDER(NA, REFILL, NDMA)                                                                         \
  {                                                                                                        \
    if constexpr ((NA) > 0) __builtin_amdgcn_sched_group_barrier(0x100, (NA), 0);                          \
    _Pragma("unroll") for (int j = 0; j < NTP; ++j) {                                                      \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                   \
      if constexpr (REFILL) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                             \
      if (j < (NDMA)) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                                   \
    }                                                                                                      \
    if constexpr ((NDMA) > NTP) __builtin_amdgcn_sched_group_barrier(0x010, (NDMA) - NTP, 0);              \
  }
  // One k-tile (2 k-steps).  On entry fa[0] / fb hold its k-step 0.  ISSUE: what this k-tile requests into stage wb -- 0
  // nothing, 1 k-tile kt + 2, 2 the conv_shared weights of half 1.  NEXT: a k-tile follows directly (its k-step 0 is read
  // under this one's last MFMAs, behind the counted wait + barrier that publishes it: vmcnt(WAITN)); otherwise the caller
  // publishes what comes next.
  auto ktile = [&](auto issue_c, auto next_c, auto waitn_c) {
    constexpr int ISSUE = decltype(issue_c)::value;
    constexpr bool NEXT = decltype(next_c)::value;
    constexpr int WAITN = decltype(waitn_c)::value;
    constexpr int NDMA = ISSUE == 0 ? 0 : (ISSUE == 1 ? NBW : NBS);
    GF_READ_A(1, 1)
    if constexpr (ISSUE == 1) { GF_STEP(0, true, 1, dma_w(kt + 2, wb, k);, NDMA) }
    else if constexpr (ISSUE == 2) { GF_STEP(0, true, 1, dma_s(1, wb, k);, NDMA) }
    else { GF_STEP(0, true, 1, ;, 0) }
    GF_ORDER(2, true, NDMA)
    // advance
    ++kt;
    if (++it_sub == 2) { it_sub = 0; if (++it_tap == 9) it_tap = 0; }
    rb = rb == 2 ? 0 : rb + 1;
    wb = wb == 2 ? 0 : wb + 1;
    if constexpr (NEXT) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(gf_wait(WAITN));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      point();
      GF_READ_A(0, 0)
      GF_STEP(1, true, 0, ;, 0)
      GF_ORDER(2, true, 0)
    } else {
      GF_STEP(1, false, 0, ;, 0)
    }
  };
  // k-step 0 of the k-tile the iterator points at -> fa[0] / fb
  auto first_frags = [&]() {
    point();
    GF_READ_A(0, 0)
#pragma unroll
    for (int j = 0; j < NTP; ++j) { GF_READ_B(j, 0) }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using WN = std::integral_constant<int, NBW>;
  using WS = std::integral_constant<int, NBS>;
  using Y = std::true_type;
  using N = std::false_type;
  // half 0: k-tiles 0..17 = items 1..18; the item requested under k-tile kt is item kt + 3: k-tile kt + 2 -- or, under
  // k-tile 16, the conv_shared weights of half 1 (item 19), and under k-tile 17 k-tile 18 (item 20).
  // Training forward: a half of actv leaves for HBM at the END of its main loop (the patch still holds it), so its NSA
  // stores per wave are the YOUNGEST entries of the queue at the next two publishes (vmcnt counts them, nobody waits for
  // them) and are long done when the third comes.
  first_frags();
  ktile(I1{}, Y{}, WN{});                                   // k-tile 0 (requests k-tile 2)
#pragma unroll 1
  for (int q = 1; q < 16; ++q) ktile(I1{}, Y{}, WN{});      // k-tiles 1..15 (request 3..17)
  ktile(I2{}, Y{}, WS{});                                   // k-tile 16: requests conv_shared half 1
  {                                                         // k-tile 17: requests k-tile 18 (kt + 1 here: the stream skips an item)
    GF_READ_A(1, 1)
    GF_STEP(0, true, 1, dma_w(18, wb, k);, NBW)
    GF_ORDER(2, true, NBW)
    ++kt;
    it_sub = 0; it_tap = 0;
    rb = rb == 2 ? 0 : rb + 1;
    wb = wb == 2 ? 0 : wb + 1;
    GF_STEP(1, false, 0, ;, 0)
  }
  if (save_actv) store_actv(0);
  // every wave is done with half 0 of the patch; the conv_shared weights of half 1 (requested one item ago) have landed
  asm volatile("" ::: "memory");
  if (save_actv) __builtin_amdgcn_s_waitcnt(gf_wait(NBW + NSA));
  else __builtin_amdgcn_s_waitcnt(gf_wait(NBW));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
    // item 19 sits in stage rb; under it: request k-tile 19 (item 21) into the stage k-tile 17 left
#pragma unroll
    for (int k = 0; k < NBW; ++k) dma_w(19, wb, k);
    conv_shared(1, rb);
    rb = rb == 2 ? 0 : rb + 1;
    wb = wb == 2 ? 0 : wb + 1;
  }
  // publish: half 1 of the patch (LDS writes of every wave) and k-tile 18 (requested two items ago)
  asm volatile("" ::: "memory");
  if (save_actv) __builtin_amdgcn_s_waitcnt(gf_wait(NBW + NSA));
  else __builtin_amdgcn_s_waitcnt(gf_wait(NBW));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // half 1: k-tiles 18..35; under k-tile kt request k-tile kt + 2 (<= 35)
  first_frags();
  ktile(I1{}, Y{}, WN{});                                   // k-tile 18 (requests 20)
#pragma unroll 1
  for (int q = 19; q < 34; ++q) ktile(I1{}, Y{}, WN{});     // k-tiles 19..33 (request 21..35)
  ktile(I0{}, Y{}, I0{});                                   // k-tile 34
  ktile(I0{}, N{}, I0{});                                   // k-tile 35
  if (save_actv) store_actv(1);
#undef GF_READ_A
#undef GF_READ_B
#undef GF_STEP
#undef GF_ORDER
  if (p.tlog && tid == 0 && last) p.tlog[(size_t)bid * 8 + 2] = wall_clock64();

  // ---- epilogue.  D layout (swapped operands): lane -> pixel l31; regs 4g..4g+3 -> channels 8g + 4 lh + (0..3) of the tile
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(gf_wait(63));
  __builtin_amdgcn_s_barrier();                // every wave is done with the patch and the weight ring
  asm volatile("" ::: "memory");
  // the next (tile, pass) of this block: its head flies while this epilogue computes and stores
  if (nxt_pass >= 0) gf_head<NTP>(p, nxt_pass, smem, NT_, nxt_seg, wave, lane);
  // (the epilogue's index arithmetic depends on the lane id only: hidden from the optimiser behind an empty asm, or it is
  //  hoisted out of the persistent tile loop and lives -- spilled -- through the main loop)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int l31e = lane_e & 31, lhe = lane_e >> 5;
  const size_t img_px = (size_t)p.H * p.W;
  // this lane's two pixels (tile rows 4 w + (l31 >> 4) and + 2)
  const int tye = 4 * wave + (l31e >> 4), pxe = pt_x0 + (l31e & 15);
  int pidx[2], pidx_lo[2];
  bool pix_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int py = pt_y0 + tye + 2 * i;
    pix_ok[i] = py < p.H && pxe < p.W;
    pidx[i] = pix_ok[i] ? (pt_n * p.H + py) * p.W + pxe : 0;
    pidx_lo[i] = pix_ok[i] ? (pt_n * (p.H >> 1) + (py >> 1)) * (p.W >> 1) + (pxe >> 1) : 0;
  }
  // Stores go through buffer resources of this tile's IMAGE (32-bit byte offsets; an out-of-image row gets an offset
  // beyond num_records and the hardware drops the store: no branches, no 64-bit address arithmetic)
  const rsrc_t o_rsrc = make_rsrc(reinterpret_cast<const char*>(p.out) + (size_t)pt_n * img_px * p.out_cs * 2, (unsigned)(img_px * p.out_cs * 2));
  const rsrc_t g_rsrc = make_rsrc(reinterpret_cast<const char*>(p.g1p) + (size_t)pt_n * img_px * p.sC * 2,
                                  p.g1p ? (unsigned)(img_px * p.sC * 2) : 0u);
  // bf16 staging: the two 32-pixel halves of the wave go through the scratch TOGETHER (two buffers of 32 rows x 64 B + pad),
  // rows leave 16 bytes (8 channels) per lane along the channels
  unsigned char* const sb0 = patch + wave * 5120;      // (the ring and the label patch are being refilled for the next pass)
  static_assert(4 * 5120 <= GF_PATCH_B, "epilogue scratch fits the patch");
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
          gf_store16(v, rs, pp4[i][k] < 0 ? 0xFFFFFFF0u : (unsigned)(pp4[i][k] * dcs + dco + kk * 8) * 2u);
        }
      } else {
        const int r = lane_e >> 1, kk = lane_e & 1;
        const f32x4 v = *reinterpret_cast<const f32x4*>(sb + r * RS + kk * 16);
        gf_store16(v, rs, pp2[i] < 0 ? 0xFFFFFFF0u : (unsigned)(pp2[i] * dcs + dco + kk * 8) * 2u);
      }
    }
  };
  float zv[2] = {0.f, 0.f};
  if (p.sz) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int py = pt_y0 + tye + 2 * i;
      zv[i] = p.sz[((size_t)pt_n * p.W + (pix_ok[i] ? pxe : 0)) * p.H + (pix_ok[i] ? py : 0)];
    }
  }
  // act(v) = max(v, v * sl): LeakyReLU (sl = slope), ReLU (0), none (1) -- the SPADE sites use LeakyReLU / none
  const float sl = p.act == HRV_ACT_LRELU ? p.slope : (p.act == HRV_ACT_RELU ? 0.f : 1.f);
  // x of NG groups of 8 channels starting at local channel lc0, both pixels of the lane
  auto load_x = [&](const int lc0, auto ng_c, f32x4 (&xr)[2][4]) {
    constexpr int NG = decltype(ng_c)::value;
    if (p.sx_up_c > 0) {  // the group's channels lie on one side of the split (both are multiples of 16)
      const bool lo = cb0 + lc0 < p.sx_up_c;
      const float* const base = lo ? p.sx : p.sx2;
      const int cs = lo ? p.sx_cs : p.sx2_cs, c0 = lo ? p.sx_co + cb0 + lc0 : p.sx2_co + cb0 + lc0 - p.sx_up_c;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < NG; ++g)
          xr[i][g] = ld4e<false>(base, (size_t)(lo ? pidx_lo[i] : pidx[i]) * cs + c0 + 8 * g + 4 * lhe);
    } else if (p.sx_f32) {       // (one branch per group of loads, not one per load)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < NG; ++g)
          xr[i][g] = ld4e<false>(p.sx, (size_t)pidx[i] * p.sx_cs + p.sx_co + cb0 + lc0 + 8 * g + 4 * lhe);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < NG; ++g)
          xr[i][g] = ld4e<true>(p.sx, (size_t)pidx[i] * p.sx_cs + p.sx_co + cb0 + lc0 + 8 * g + 4 * lhe);
    }
  };
  // NG * 8 channels starting at local channel lc0: modulate, stage, store the activation, then (1 + gamma)
  auto group = [&](const int lc0, auto ng_c, auto&& gam, auto&& bet, const f32x4 (&xr)[2][4]) {
    constexpr int NG = decltype(ng_c)::value;
    f32x4 g1r[2][NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int lc = lc0 + 8 * g + 4 * lhe;
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(cbuf + lc);
      const f32x4 b2 = *reinterpret_cast<const f32x4*>(cbuf + GF_CV + lc);
      const f32x4 rs = *reinterpret_cast<const f32x4*>(cbuf + 2 * GF_CV + lc);
      const f32x4 nr = *reinterpret_cast<const f32x4*>(cbuf + 3 * GF_CV + lc);
      const f32x4 mr = *reinterpret_cast<const f32x4*>(cbuf + 4 * GF_CV + lc);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x4 xn = xr[i][g] * rs + (nr * zv[i] + mr);
        g1r[i][g] = gam(i, g) + b1;
        const f32x4 t = xn * g1r[i][g] + (bet(i, g) + b2);
        const f32x4 ts = t * sl;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(t[e], ts[e]);
        *reinterpret_cast<gf_bf16x4*>(sb0 + i * 2560 + l31e * RS + 16 * g + 8 * lhe) = __builtin_convertvector(v, gf_bf16x4);
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
          *reinterpret_cast<gf_bf16x4*>(sb0 + i * 2560 + l31e * RS + 16 * g + 8 * lhe) = __builtin_convertvector(g1r[i][g], gf_bf16x4);
      rows_out(ng_c, g_rsrc, p.sC, cb);
    }
  };
  using G4 = std::integral_constant<int, 4>;
  using G2 = std::integral_constant<int, 2>;
  // x is fetched group by group, just in time (32 registers; with 32 NTP accumulators still live a second set in flight
  // spilled): the load latency is the co-resident block's to hide
  f32x4 xa[2][4];
  if constexpr (NPAIR >= 1) {
    load_x(0, G4{}, xa);
    group(0, G4{}, [&](int i, int g) { return gf_acc4(acc[i][0], g); }, [&](int i, int g) { return gf_acc4(acc[i][1], g); }, xa);
  }
  if constexpr (NPAIR >= 2) {
    load_x(32, G4{}, xa);
    group(32, G4{}, [&](int i, int g) { return gf_acc4(acc[i][2], g); }, [&](int i, int g) { return gf_acc4(acc[i][3], g); }, xa);
  }
  if constexpr (TAIL != 0) {
    load_x(NPAIR * 32, G2{}, xa);
    group(NPAIR * 32, G2{}, [&](int i, int g) { return gf_acc4(acc[i][NTP - 1], g); },
          [&](int i, int g) { return gf_acc4(acc[i][NTP - 1], g + 2); }, xa);
  }
  if (p.tlog && tid == 0 && last) p.tlog[(size_t)bid * 8 + 6] = wall_clock64();      // every store of the tile is issued
}

template <int NTP>
__global__ __launch_bounds__(256, 2) void spade_fused_kernel(const GfParams p, const int pass0, const int pass1) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[GF_LDS];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  // conv_shared's bias: the same for every tile and pass
  if (threadIdx.x < 128)
    reinterpret_cast<float*>(smem + GF_CB_OFF)[5 * GF_CV + threadIdx.x] =
        reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.wp) + GF_WSH_B)[threadIdx.x];
  // A unit of work = one tile with the launch's passes pass0 .. pass1 one after the other (label patch loaded once) -- or, p.pp (fewer
  // tiles than resident blocks: the 128 x 96 / 64 x 48 levels), ONE (tile, pass): the passes of a tile run on different CUs at once
  const int npg = pass1 - pass0;
  const int units = p.pp ? p.m_tiles * npg : p.m_tiles;
  auto unit_tile = [&](const int u) { return p.pp ? u / npg : u; };
  auto unit_pass = [&](const int u) { return p.pp ? pass0 + u % npg : pass0; };
  if ((int)blockIdx.x < units) gf_head<NTP>(p, unit_pass(blockIdx.x), smem, gf_tile(p, unit_tile(blockIdx.x)), true, wave, lane);
  int c_n = -1, c_pass = -1;                   // (image, pass) of the constants in LDS
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
    const GfTile T = gf_tile(p, bid);
    const int nu = u + gridDim.x;
    const GfTile TN = gf_tile(p, unit_tile(nu < units ? nu : u));
    const int pa = unit_pass(u), pb = p.pp ? pa + 1 : pass1;
#pragma unroll 1
    for (int pass = pa; pass < pb; ++pass) {
      const bool lastp = pass == pb - 1;
      const int nxt_pass = !lastp ? pass + 1 : (nu < units ? unit_pass(nu) : -1);
      const bool lc = c_n != T.n || c_pass != pass;
      c_n = T.n; c_pass = pass;
      gf_pass<NTP>(p, pass, smem, T, bid, lc, u == (int)blockIdx.x && pass == pa, p.actv != nullptr && pass == 0, pass == pa, lastp,
                   nxt_pass, lastp ? TN : T, lastp);
    }
    if (p.tlog) {
      __builtin_amdgcn_s_waitcnt(gf_wait(0));
      if (threadIdx.x == 0) p.tlog[(size_t)bid * 8 + 3] = wall_clock64();
    }
  }
}

}  // namespace hrv

using namespace hrv;

extern "C" int64_t hrv_spade_fused_packed_bytes(int32_t C) {
  GfPlan pl;
  if (!gf_plan(C, pl)) return -1;
  return pl.bytes;
}

extern "C" int hrv_spade_fused_supported(int32_t C, int32_t hid, int32_t label_nc, int32_t N, int32_t H, int32_t W) {
  GfPlan pl;
  if (hid != 128 || label_nc < 1 || label_nc > 8 || !gf_plan(C, pl)) return 0;
  const int64_t tiles = (int64_t)N * ((H + 15) / 16) * ((W + 15) / 16);
  // two blocks per CU: fewer tiles leave half the slots empty.  HRV_SPADE_FUSED_MIN_TILES_X4: the threshold in quarter-tiles per CU
  // (default 8 = two tiles per CU; A/B at 3 -- the 128 x 96 level at 4 images -- in profiles/r05_ab_fused_threshold.txt)
  const char* e = hrv::env("HRV_SPADE_FUSED_MIN_TILES_X4");
  int q4 = e ? atoi(e) : 8;
  if (q4 < 1) q4 = 8;
  // (round 6: counted in UNITS of work -- with fewer tiles than resident blocks the passes of a tile spread over the CUs)
  const int64_t units = tiles < 2 * (int64_t)persistent_cus() ? tiles * pl.npass : tiles;
  return 4 * units >= q4 * (int64_t)persistent_cus() ? 1 : 0;
}

extern "C" int hrv_spade_fused_pack_dev(const float* w_shared, const float* b_shared, int32_t label_nc, const float* w_gamma,
                                        const float* w_beta, int32_t C, void* out, hrv_stream_t stream) {
  HRV_REQUIRE(w_shared && b_shared && w_gamma && w_beta && out, "spade_fused_pack: null pointer");
  HRV_REQUIRE(label_nc >= 1 && label_nc <= 8, "spade_fused_pack: label_nc %d (1..8)", label_nc);
  GfPackParams pp;
  HRV_REQUIRE(gf_plan(C, pp.pl), "spade_fused_pack: unsupported norm width %d", C);
  HRV_REQUIRE(((uintptr_t)out & 15) == 0 && ((uintptr_t)b_shared & 3) == 0, "spade_fused_pack: alignment");
  pp.C = C; pp.label_nc = label_nc;
  pp.wsh = w_shared; pp.bsh = b_shared; pp.wg = w_gamma; pp.wb = w_beta; pp.out = (unsigned short*)out;
  const long long groups = pp.pl.bytes / 16;
  hipLaunchKernelGGL(gf_pack_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pp);
  return check_launch("gf_pack_kernel");
}

extern "C" int hrv_spade_fused_bf16(const hrv_spade_fused_t* d, hrv_stream_t stream) {
  HRV_REQUIRE(d != nullptr, "spade_fused: null descriptor");
  GfPlan pl;
  HRV_REQUIRE(gf_plan(d->C, pl), "spade_fused: unsupported norm width %d", d->C);
  HRV_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && (int64_t)d->N * d->H * d->W < ((int64_t)1 << 31), "spade_fused: bad extent");
  HRV_REQUIRE(d->seg && d->w_packed && d->out && d->x && d->mean && d->rstd && d->bias_gamma && d->bias_beta, "spade_fused: null pointer");
  HRV_REQUIRE(d->seg_shift >= 0 && d->seg_shift < 16 && d->seg_H == (d->H << d->seg_shift) && d->seg_W == (d->W << d->seg_shift),
              "spade_fused: the label map must be [N, H << shift, W << shift, 8] bf16 (got %d x %d for %d x %d, shift %d)", d->seg_H, d->seg_W,
              d->H, d->W, d->seg_shift);
  const int64_t sbytes = (int64_t)d->seg_H * d->seg_W * 16;
  HRV_REQUIRE(sbytes < (int64_t)0xFFFFFFF0, "spade_fused: one label image exceeds the 32-bit buffer range");
  HRV_REQUIRE((((uintptr_t)d->seg | (uintptr_t)d->w_packed | (uintptr_t)d->out | (uintptr_t)d->x | (uintptr_t)d->g1p | (uintptr_t)d->actv) & 15) == 0,
              "spade_fused: 16-byte alignment");
  HRV_REQUIRE((d->noise_z == nullptr) == (d->noise_scale == nullptr), "spade_fused: noise_z/noise_scale go together");
  HRV_REQUIRE(d->out_cstride % 8 == 0 && d->out_coff % 8 == 0 && d->out_cstride >= d->out_coff + d->C, "spade_fused: out slice");
  HRV_REQUIRE((int64_t)d->H * d->W * d->out_cstride * 2 < (int64_t)0xFFFFFFF0, "spade_fused: one image of `out` exceeds 4 GB");
  if (d->x_up_channels > 0) {
    HRV_REQUIRE(d->x_f32 && d->x2 && d->x_up_channels % 16 == 0 && d->x_up_channels < d->C && d->H % 2 == 0 && d->W % 2 == 0 &&
                    d->x_cstride % 4 == 0 && d->x_coff % 4 == 0 && d->x_coff + d->x_up_channels <= d->x_cstride && d->x2_cstride % 4 == 0 &&
                    d->x2_coff % 4 == 0 && d->x2_coff + (d->C - d->x_up_channels) <= d->x2_cstride && ((uintptr_t)d->x2 & 15) == 0,
                "spade_fused: upsampled x (%d of %d channels, %d x %d)", d->x_up_channels, d->C, d->H, d->W);
  } else {
    HRV_REQUIRE(d->x_cstride % 4 == 0 && d->x_coff % 4 == 0 && d->x_coff + d->C <= d->x_cstride, "spade_fused: x slice");
  }
  HRV_REQUIRE(d->actv == nullptr || (d->actv_cstride % 8 == 0 && d->actv_coff % 8 == 0 && d->actv_coff + 128 <= d->actv_cstride &&
                                     (int64_t)d->H * d->W * d->actv_cstride * 2 < (int64_t)0xFFFFFFF0),
              "spade_fused: actv slice");
  GfParams p;
  memset(&p, 0, sizeof(p));
  p.seg = d->seg; p.seg_H = d->seg_H; p.seg_W = d->seg_W; p.seg_shift = d->seg_shift; p.seg_bytes = (unsigned)sbytes;
  p.N = d->N; p.H = d->H; p.W = d->W;
  p.wp = d->w_packed; p.w_bytes = (unsigned)pl.bytes;
  p.npass = pl.npass;
  for (int i = 0; i < pl.npass; ++i) { p.ntp[i] = pl.ntp[i]; p.tile0[i] = pl.tile0[i]; p.woff[i] = pl.woff[i]; }
  p.m_tiles = d->N * ((d->H + 15) / 16) * ((d->W + 15) / 16);
  p.sx = (const float*)d->x; p.sx_cs = d->x_cstride; p.sx_co = d->x_coff; p.sx_f32 = d->x_f32; p.sC = d->C;
  p.sx2 = (const float*)d->x2; p.sx2_cs = d->x2_cstride; p.sx2_co = d->x2_coff; p.sx_up_c = d->x_up_channels;
  p.smean = d->mean; p.srstd = d->rstd; p.sz = d->noise_z; p.sns = d->noise_scale; p.bg = d->bias_gamma; p.bb = d->bias_beta;
  p.g1p = d->g1p;
  p.act = d->act; p.slope = d->act_slope;
  p.out = d->out; p.out_cs = d->out_cstride; p.out_co = d->out_coff;
  p.actv = d->actv; p.actv_cs = d->actv_cstride; p.actv_co = d->actv_coff;
  p.tlog = diag_tlog(p.m_tiles);
  const int cap = 2 * persistent_cus();
  p.pp = p.m_tiles < cap ? 1 : 0;
  if (p.pp) p.tlog = nullptr;          // (the timeline's slots are per tile)
  // the passes of equal width share a launch: 4-tile passes, a 2-tile pass, the 5-tile tail pass
  for (int a = 0; a < pl.npass;) {
    int b = a;
    while (b < pl.npass && pl.ntp[b] == pl.ntp[a]) ++b;
    const long long units = p.pp ? (long long)p.m_tiles * (b - a) : p.m_tiles;
    const int grid = units < cap ? (int)units : cap;
    if (pl.ntp[a] == 4) hipLaunchKernelGGL((spade_fused_kernel<4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, a, b);
    else if (pl.ntp[a] == 2) hipLaunchKernelGGL((spade_fused_kernel<2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, a, b);
    else hipLaunchKernelGGL((spade_fused_kernel<5>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, a, b);
    a = b;
  }
  return check_launch("spade_fused_kernel");
}
