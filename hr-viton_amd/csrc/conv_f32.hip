// Implicit-GEMM NHWC convolution on the gfx950 fp32 matrix cores.
//
//   M = N*Ho*Wo output pixels, Ncols = Cout, K = KH*KW*sum(C_src)
//   D[pixel][cout] = sum_k A[pixel][k] * B[k][cout]
//
// * A is gathered on the fly from up to 4 NHWC sources (channel concatenation,
//   optional nearest-x2 upsample and LeakyReLU folded into the gather): no
//   im2col buffer, no torch.cat copy ever touches HBM.
// * B is the pre-packed weight [kt][CoutPad][16] (hrv_conv2d_pack_weight_f32),
//   so a block's B tile is one contiguous BN*16-float run.
// * Both tiles are staged through LDS as [row][16 + 4 pad] so every lane
//   fetches its 4 k-values with one conflict-free ds_read_b128; the k index is
//   permuted consistently on A and B (lane half h takes k = kq*8 + h*4 + e),
//   which is legal because the MFMA sums over k.
// * v_mfma_f32_32x32x2_f32: exact fp32 (a k-ordered fmaf chain), 157 TFLOP/s
//   peak, 64 cycles per instruction per SIMD -> a wave with TM*TN >= 2
//   independent accumulators keeps its SIMD's matrix pipe full; global loads for
//   K-tile t+1 are in flight (registers) while tile t is multiplied, one
//   __syncthreads per K-tile (double-buffered LDS).
// * Epilogue (fused): out = act(acc*scale[c] + shift[c] + residual).
//
// Reference ops replaced: see include/hrviton_hip.h (hrv_conv2d_nhwc_f32).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "conv_params.h"

namespace hrv {

// One output tile of the implicit GEMM.  ``bid``: dispatch-order id of the tile (blockIdx.x for the one-tile-per-block
// launches; the persistent patch launches walk bid = blockIdx.x, + gridDim.x, ...).
template <int TM, int TN, int WM, int WN, int VAR, bool BF, int RB = 64>
__device__ __forceinline__ void conv_mfma_tile(const ConvParams& p, const int bid) {
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 MMA waves per block");
  constexpr int NT = 64 * WM * WN;   // threads that share the gather / the MMA wave grid
  // VAR bit 5: wave specialisation.  The block carries WM*WN extra LOADER waves: they issue every LDS-DMA
  // instruction and wait for it, the MMA waves only read fragments and issue MFMAs -- a wave that issues its own
  // DMA cannot issue MFMAs meanwhile (in-order issue; ~100 cycles per DMA instruction inside an MFMA stream).
  constexpr bool SPEC = (VAR & 32) != 0;
  static_assert(RB == 64 || RB == 128, "K-tile row = 64 or 128 bytes");
  constexpr int ES = BF ? 2 : 4;     // element size in bytes
  constexpr int EPG = 16 / ES;       // elements per 16-byte gather group
  constexpr int GPR = RB / 16;       // 16-byte groups per K-tile row
  constexpr int BKE = GPR * EPG;     // k-values per K-tile row
  constexpr bool GLDS = (VAR & 4) != 0;  // operands staged by LDS-DMA (global_load_lds_dwordx4), no register hop
  static_assert(!GLDS || (BF && RB == 128), "the LDS-DMA variant exists for the bf16 engine with 128-byte rows");
  // bf16 matrix cores over fp32 tensors (mixed-precision TRAINING, the reference's --fp16 / apex O1 role): the
  // gather loads 8 fp32 channels (two float4s) per 16-byte LDS group and rounds them with v_cvt_pk_bf16_f32;
  // weights are packed as bf16; accumulation, epilogue, residual and output stay fp32.
  constexpr bool SF = (VAR & 8) != 0;
  static_assert(!SF || (BF && !GLDS), "fp32-source variant: bf16 MFMA, register-staged");
  // LDS row stride in floats.  Register-staged: row + 16 B pad (conflict-free b128 reads).  LDS-DMA: the
  // image is lane-linear (dest = wave base + lane*16), so rows are dense and the 16-byte groups of a row
  // are XOR-swizzled by ((row >> 1) & 7) on the SOURCE address and on the fragment read instead.
  constexpr int LS = GLDS ? RB / 4 : RB / 4 + 4;
  constexpr int BK = RB / 4;         // floats per packed-weight row
  constexpr int KQ = RB / 32;        // 32-byte fragment steps per row (2 lane-halves x 16 B)
  constexpr int RPP = NT / GPR;      // tile rows covered by one pass of the block's threads
  constexpr int BM = 32 * TM * WM;
  constexpr int BN = 32 * TN * WN;
  constexpr int AR = BM / RPP;                      // 16-byte A loads per thread per K-tile
  constexpr int BR = (BN * GPR + NT - 1) / NT;      // 16-byte B loads per thread per K-tile
  constexpr bool SWAP = (VAR & 1) != 0;
  constexpr bool PIPE = (VAR & 2) != 0;
  // VAR bit 4 (LDS-DMA only): THREE K-tile stages in LDS -- two tiles' DMA stay in flight across a fence-less
  // barrier (counted vmcnt), so the load latency is hidden inside the block instead of by a second resident block
  constexpr int ST = (VAR & 16) ? 3 : 2;
  static_assert(ST == 2 || GLDS, "the 3-stage pipeline exists for the LDS-DMA variant");
  static_assert(!SPEC || ST == 3, "wave specialisation rides on the 3-stage LDS-DMA pipeline");
  // VAR bit 6: PATCH mode for 3x3 stride-1 'same' convolutions over ONE bf16 source with Cin % 128 == 0.  The
  // implicit-GEMM gather re-reads every activation pixel from L2 once per tap (9x); here the block's 16x16 output
  // pixels are a 2-D tile whose 18x18 halo patch (128 channels) is DMA'd into LDS ONCE per 128-channel chunk and
  // all 9 taps x 2 K-tiles read their A fragments from it -- only the weight tiles stream (3 LDS stages, counted
  // vmcnt).  L2 -> LDS bytes per block tile drop from 18 x 48 KB to 81 KB + 18 x 16 KB (2.3x fewer).
  constexpr bool PATCH = (VAR & 64) != 0;
  static_assert(!PATCH || (GLDS && !SPEC && ((BM == 256 && BN == 128 && NT == 512 && ST == 3) ||
                                            (BM == 128 && (BN == 128 || BN == 64) && NT == 256 && (ST == 2 || BN == 64)))),
                "patch mode: 16x16 pixels x 128 columns with 8 waves and 3 weight stages (132 KB LDS, one block per "
                "CU), or 8x16 pixels with 4 waves and 2 weight stages (78 KB: two blocks per CU overlap each "
                "other's patch load and epilogue; 128 or 64 columns)");
  constexpr int TH = BM / 16;                 // tile rows (pixels); tile width is 16
  constexpr int PW = 18, PPIX = (TH + 2) * PW;   // halo patch, pixels
  constexpr int PFL = 64;                     // floats per patch pixel (128 bf16 channels)
  __shared__ __attribute__((aligned(16))) float smem[PATCH ? (PPIX * PFL + ST * BN * LS) : ST * (BM + BN) * LS];

  // loader waves mirror the MMA waves' thread ids: the gather distribution below is written for NT threads
  const int tid = SPEC ? (int)(threadIdx.x & (NT - 1)) : (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): LDS-DMA bases stay scalar
  [[maybe_unused]] const bool loader = SPEC && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) >= WM * WN;
  const int wm = wave / WN;
  const int wn = wave % WN;

  const int lid_all = xcd_remap(bid, p.m_tiles * p.n_tiles * p.splitk);
  const int lid = lid_all / p.splitk;       // the splits of one tile are neighbours (same XCD)
  const int ks = lid_all - lid * p.splitk;
  const int mt = lid / p.n_tiles;
  const int nt = lid - mt * p.n_tiles;
  const int m0 = mt * BM;
  const int n0 = p.n0_base + nt * BN;
  // patch mode: mt = (image, tile row, tile column) of a 16x16 output tile
  [[maybe_unused]] int pt_n = 0, pt_y0 = 0, pt_x0 = 0;
  if constexpr (PATCH) {
    const int tx = (p.W + 15) >> 4, ty = (p.H + TH - 1) / TH;
    pt_n = mt / (tx * ty);
    const int r = mt - pt_n * (tx * ty);
    pt_y0 = (r / tx) * TH;
    pt_x0 = (r % tx) << 4;
  }
  // GEMM row of this block -> output pixel index (>= p.M: no such pixel)
  auto row2pix = [&](int row) -> int {
    if constexpr (PATCH) {
      const int y = pt_y0 + (row >> 4), x = pt_x0 + (row & 15);
      return (y < p.H && x < p.W) ? (pt_n * p.H + y) * p.W + x : p.M;
    } else {
      return m0 + row;
    }
  };
  const int kt_begin = (int)(((long long)p.KT * ks) / p.splitk);
  const int kt_end = (int)(((long long)p.KT * (ks + 1)) / p.splitk);

  // ---- per-thread gather coordinates (fixed for the whole K loop) ----
  const int a_c4 = tid % GPR;  // which 16-byte group of the K-tile row
  const int a_row = tid / GPR; // + RPP*r
  int a_n[AR], a_hi0[AR], a_wi0[AR];
  bool a_ok[AR];
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    const int pidx = m0 + a_row + RPP * r;
    a_ok[r] = pidx < p.M;
    const int pp = a_ok[r] ? pidx : 0;
    const int n = pp / (p.Ho * p.Wo);
    const int rem = pp - n * (p.Ho * p.Wo);
    const int ho = rem / p.Wo;
    const int wo = rem - ho * p.Wo;
    a_n[r] = n;
    a_hi0[r] = ho * p.stride - p.pad;
    a_wi0[r] = wo * p.stride - p.pad_w;
  }

  // LDS-DMA gather state: logical channel group of this thread (the swizzle term (row >> 1) & 7 does not
  // depend on r: rows of one thread are RPP = 32 apart), pixel index of tap (0,0), per-row tap-validity
  // bit mask, byte offset of the current (tap 0, source) run, constant weight-tile offsets.
  [[maybe_unused]] const int g_ch = (a_c4 ^ ((a_row >> 1) & 7)) * EPG;
  [[maybe_unused]] int a_pix[AR];
  [[maybe_unused]] unsigned a_mask[AR], a_off[AR], b_voff[BR];
  [[maybe_unused]] bool new_run = true;
  [[maybe_unused]] const float* s_ptr = p.src[0].ptr;
  [[maybe_unused]] int s_C = 0, s_cs = 0, s_co = 0, s_up = 0, s_chunks = 1;
  [[maybe_unused]] unsigned s_bytes = 0;
  [[maybe_unused]] rsrc_t w_rsrc = make_rsrc(p.wp, p.w_bytes);
  if constexpr (GLDS) {
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      a_pix[r] = (a_n[r] * p.H + a_hi0[r]) * p.W + a_wi0[r];
      a_off[r] = 0;
      unsigned m = 0;
      if (p.KH * p.KW <= 32)
        for (int t = 0; t < p.KH * p.KW; ++t) {
          const int hi = a_hi0[r] + t / p.KW, wi = a_wi0[r] + t % p.KW;
          m |= (a_ok[r] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) ? (1u << t) : 0u;
        }
      a_mask[r] = m;
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      const int idx = tid + NT * j;
      const int row = idx / GPR, slot = idx % GPR;
      b_voff[j] = (unsigned)(row * RB + ((slot ^ ((row >> 1) & 7)) * 16));
    }
  }

  // K-tile iterator state: (tap kh,kw) x (source s) x (chunk c), positioned at kt_begin
  int it_kh, it_kw, it_s = 0, it_c;
  {
    const int tap = kt_begin / p.chunks_total;
    int r = kt_begin - tap * p.chunks_total;
    it_kh = tap / p.KW;
    it_kw = tap - it_kh * p.KW;
#pragma unroll
    for (int q = 0; q < HRV_MAX_SRC - 1; ++q)
      if (it_s == q && q < p.nsrc - 1 && r >= p.src[q].chunks) {
        r -= p.src[q].chunks;
        ++it_s;
      }
    it_c = r;
  }

  f32x4 a_reg[AR];
  f32x4 b_reg[BR];

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int l31 = lane & 31;
  const int lh = lane >> 5;

// ---- global -> registers for K-tile KTN (stays in flight while the MFMAs run).
// Branch-free (uniform per-field selects; loads from a valid address then select) so
// the steady-state loop body is a single scheduling region.
#define HRV_LOAD_TILE(KTN)                                                                                   \
  {                                                                                                          \
    const float* s_ptr = p.src[0].ptr;                                                                       \
    int s_C = p.src[0].C, s_cs = p.src[0].cstride, s_co = p.src[0].coff, s_up = p.src[0].up_shift;          \
    _Pragma("unroll") for (int q = 1; q < HRV_MAX_SRC; ++q) {                                                \
      const bool sel = it_s == q;                                                                            \
      s_ptr = sel ? p.src[q].ptr : s_ptr;                                                                    \
      s_C = sel ? p.src[q].C : s_C;                                                                          \
      s_cs = sel ? p.src[q].cstride : s_cs;                                                                  \
      s_co = sel ? p.src[q].coff : s_co;                                                                     \
      s_up = sel ? p.src[q].up_shift : s_up;                                                                 \
    }                                                                                                        \
    const int c = it_c * BKE + a_c4 * EPG;                                                                   \
    const bool c_ok = c < s_C;                                                                               \
    /* up_shift > 0: source is 2^up smaller (nearest upsample); < 0: 2^-up larger (nearest downsample) */    \
    const int sh_r = s_up > 0 ? s_up : 0, sh_l = s_up < 0 ? -s_up : 0;                                       \
    const int Hs = (p.H >> sh_r) << sh_l, Ws = (p.W >> sh_r) << sh_l;                                        \
    const int cc = c_ok ? c : 0;                                                                             \
    _Pragma("unroll") for (int r = 0; r < AR; ++r) {                                                         \
      const int hi = a_hi0[r] + it_kh;                                                                       \
      const int wi = a_wi0[r] + it_kw;                                                                       \
      const bool ok = a_ok[r] && c_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;       \
      /* clamp instead of branching: the address is always valid, the value is zeroed below */              \
      const int hic = min(max(hi, 0), p.H - 1), wic = min(max(wi, 0), p.W - 1);                              \
      const unsigned off =                                                                                   \
          ((unsigned)(a_n[r] * Hs + ((hic >> sh_r) << sh_l)) * Ws + ((wic >> sh_r) << sh_l)) * s_cs + s_co + cc; \
      if constexpr (SF) {                                                                                    \
        const float* g = s_ptr + off;                                                                        \
        const bool ok2 = ok && (c + 4 < s_C);                                                                \
        f32x4 lo = *reinterpret_cast<const f32x4*>(g);                                                       \
        f32x4 hi4 = *reinterpret_cast<const f32x4*>(g + (ok2 ? 4 : 0));                                      \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                      \
          lo[e] = ok ? lo[e] : 0.f;                                                                          \
          hi4[e] = ok2 ? hi4[e] : 0.f;                                                                       \
        }                                                                                                    \
        a_reg[r] = pack_bf16x8(lo, hi4);                                                                     \
      } else {                                                                                               \
        f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(s_ptr) + (size_t)off * ES);  \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;                                \
        a_reg[r] = v;                                                                                        \
      }                                                                                                      \
    }                                                                                                        \
    const float* wt = p.wp + ((size_t)(KTN)*p.CoutPad + n0) * BK; /* 64-byte rows in both modes */            \
    _Pragma("unroll") for (int j = 0; j < BR; ++j) {                                                         \
      const int idx = tid + NT * j;                                                                          \
      b_reg[j] = *reinterpret_cast<const f32x4*>(wt + ((BN * GPR) % NT == 0 || idx < BN * GPR ? idx : 0) * 4);  \
    }                                                                                                        \
  }

// ---- global -> LDS directly (LDS-DMA, `buffer_load_dwordx4 ... offen lds`) for K-tile KTN into buffer BUF.
// Lane l of a wave-instruction lands at (wave-uniform base) + 16*l: the thread that owns LDS slot (row, slot)
// fetches the logical 16-byte group slot ^ ((row >> 1) & 7) of that row.  Padding / out-of-range lanes pass
// an offset beyond the buffer resource's num_records: the hardware range check writes ZEROS to LDS for them
// (tools/probes/buffer_lds_oob.hip), so no zero page and no clamping.  Per K-tile VALU cost of the fast path
// (sources without nearest up/down-sampling): one bit test + one add + one select per row; everything else
// is wave-uniform (SALU) or hoisted to the start of a (tap, source) run.
#define HRV_DMA_SRC()                                                                                        \
  {                                                                                                          \
    s_ptr = p.src[0].ptr; s_C = p.src[0].C; s_cs = p.src[0].cstride; s_co = p.src[0].coff;                   \
    s_up = p.src[0].up_shift; s_bytes = p.src[0].bytes; s_chunks = p.src[0].chunks;                          \
    _Pragma("unroll") for (int q = 1; q < HRV_MAX_SRC; ++q) {                                                \
      const bool sel = it_s == q;                                                                            \
      s_ptr = sel ? p.src[q].ptr : s_ptr;                                                                    \
      s_C = sel ? p.src[q].C : s_C;                                                                          \
      s_cs = sel ? p.src[q].cstride : s_cs;                                                                  \
      s_co = sel ? p.src[q].coff : s_co;                                                                     \
      s_up = sel ? p.src[q].up_shift : s_up;                                                                 \
      s_bytes = sel ? p.src[q].bytes : s_bytes;                                                              \
      s_chunks = sel ? p.src[q].chunks : s_chunks;                                                           \
    }                                                                                                        \
  }
// iterator advance of the LDS-DMA loop: the current source's fields live in scalars and are reloaded only
// when the source changes (never, for the single-source convolutions that dominate the path)
#define HRV_DMA_ADVANCE()                                                                                    \
  {                                                                                                          \
    new_run = false;                                                                                         \
    if (++it_c == s_chunks) {                                                                                \
      it_c = 0;                                                                                              \
      new_run = true;                                                                                        \
      if (++it_s == p.nsrc) {                                                                                \
        it_s = 0;                                                                                            \
        if (++it_kw == p.KW) {                                                                               \
          it_kw = 0;                                                                                         \
          ++it_kh;                                                                                           \
        }                                                                                                    \
      }                                                                                                      \
      if (p.nsrc > 1) HRV_DMA_SRC()                                                                          \
    }                                                                                                        \
  }
#define HRV_DMA_TILE(KTN, BUF)                                                                               \
  {                                                                                                          \
    const rsrc_t a_rsrc = make_rsrc(s_ptr, s_bytes);                                                         \
    float* Abuf = smem + (BUF) * (BM + BN) * LS;                                                             \
    const bool c_ok = it_c * BKE + g_ch < s_C;                                                               \
    if (s_up == 0 && p.KH * p.KW <= 32) {                                                                    \
      if (new_run) {                                                                                         \
        _Pragma("unroll") for (int r = 0; r < AR; ++r)                                                       \
            a_off[r] = ((unsigned)a_pix[r] * (unsigned)s_cs + (unsigned)(s_co + g_ch)) * (unsigned)ES;       \
      }                                                                                                      \
      const unsigned delta = (unsigned)(((it_kh * p.W + it_kw) * s_cs + it_c * BKE) * ES);                   \
      const int tap = it_kh * p.KW + it_kw;                                                                  \
      _Pragma("unroll") for (int r = 0; r < AR; ++r) {                                                       \
        const bool ok = c_ok && ((a_mask[r] >> tap) & 1u);                                                   \
        const unsigned voff = ok ? a_off[r] + delta : 0xFFFFFFF0u;                                           \
        dma16(a_rsrc, Abuf + (RPP * r + wave * (64 / GPR)) * LS, voff, 0u);                                  \
      }                                                                                                      \
    } else {                                                                                                 \
      const int sh_r = s_up > 0 ? s_up : 0, sh_l = s_up < 0 ? -s_up : 0;                                     \
      const int Hs = (p.H >> sh_r) << sh_l, Ws = (p.W >> sh_r) << sh_l;                                      \
      _Pragma("unroll") for (int r = 0; r < AR; ++r) {                                                       \
        const int hi = a_hi0[r] + it_kh;                                                                     \
        const int wi = a_wi0[r] + it_kw;                                                                     \
        const bool ok = a_ok[r] && c_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;     \
        const unsigned off = ((unsigned)(a_n[r] * Hs + ((hi >> sh_r) << sh_l)) * Ws + ((wi >> sh_r) << sh_l)) * s_cs + \
                             s_co + it_c * BKE + g_ch;                                                       \
        const unsigned voff = ok ? off * ES : 0xFFFFFFF0u;                                                   \
        dma16(a_rsrc, Abuf + (RPP * r + wave * (64 / GPR)) * LS, voff, 0u);                                  \
      }                                                                                                      \
    }                                                                                                        \
    const unsigned w_soff = (unsigned)(((KTN)*p.CoutPad + n0) * RB);                                         \
    float* Bbuf = Abuf + BM * LS;                                                                            \
    _Pragma("unroll") for (int j = 0; j < BR; ++j) {                                                         \
      dma16(w_rsrc, Bbuf + ((NT * j + 64 * wave) / GPR) * LS, b_voff[j], w_soff);                            \
    }                                                                                                        \
  }

#define HRV_ADVANCE_ITER()                                                                  \
  {                                                                                         \
    int chunks = p.src[0].chunks;                                                           \
    _Pragma("unroll") for (int q = 1; q < HRV_MAX_SRC; ++q) chunks = it_s == q ? p.src[q].chunks : chunks; \
    if (++it_c == chunks) {                                                                 \
      it_c = 0;                                                                             \
      if (++it_s == p.nsrc) {                                                               \
        it_s = 0;                                                                           \
        if (++it_kw == p.KW) {                                                              \
          it_kw = 0;                                                                        \
          ++it_kh;                                                                          \
        }                                                                                   \
      }                                                                                     \
    }                                                                                       \
  }

#define HRV_STORE_TILE(BUF)                                                                             \
  {                                                                                                     \
    float* Asw = smem + (BUF) * (BM + BN) * LS;                                                         \
    float* Bsw = Asw + BM * LS;                                                                         \
    _Pragma("unroll") for (int r = 0; r < AR; ++r)                                                      \
        *reinterpret_cast<f32x4*>(Asw + (a_row + RPP * r) * LS + a_c4 * 4) = a_reg[r];                  \
    _Pragma("unroll") for (int j = 0; j < BR; ++j) {                                                    \
      const int idx = tid + NT * j;                                                                     \
      if ((BN * GPR) % NT == 0 || idx < BN * GPR)                                                       \
        *reinterpret_cast<f32x4*>(Bsw + (idx / GPR) * LS + (idx % GPR) * 4) = b_reg[j];                 \
    }                                                                                                   \
  }

#define HRV_READ_FRAGS(BUF)                                                                                  \
  {                                                                                                          \
    if constexpr (GLDS) {                                                                                    \
      /* rows are multiples of 32 apart: the swizzle term depends on l31 only */                             \
      const int sw = (l31 >> 1) & 7;                                                                         \
      const float* As = smem + (BUF) * (BM + BN) * LS + (wm * TM * 32 + l31) * LS;                           \
      const float* Bs = smem + (BUF) * (BM + BN) * LS + BM * LS + (wn * TN * 32 + l31) * LS;                 \
      _Pragma("unroll") for (int kq = 0; kq < KQ; ++kq) {                                                    \
        const int go = ((kq * 2 + lh) ^ sw) * 4;                                                             \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[kq][i] =                                           \
            *reinterpret_cast<const f32x4*>(As + i * 32 * LS + go);                                          \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[kq][j] =                                           \
            *reinterpret_cast<const f32x4*>(Bs + j * 32 * LS + go);                                          \
      }                                                                                                      \
    } else {                                                                                                 \
      const float* As = smem + (BUF) * (BM + BN) * LS + (wm * TM * 32 + l31) * LS + lh * 4;                  \
      const float* Bs = smem + (BUF) * (BM + BN) * LS + BM * LS + (wn * TN * 32 + l31) * LS + lh * 4;        \
      _Pragma("unroll") for (int kq = 0; kq < KQ; ++kq) {                                                    \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[kq][i] =                                           \
            *reinterpret_cast<const f32x4*>(As + i * 32 * LS + kq * 8);                                      \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[kq][j] =                                           \
            *reinterpret_cast<const f32x4*>(Bs + j * 32 * LS + kq * 8);                                      \
      }                                                                                                      \
    }                                                                                                        \
  }

#define HRV_MMA1(I, J, AV, BV)                                                              \
  acc[I][J] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(BV, AV, acc[I][J], 0, 0, 0)       \
                   : __builtin_amdgcn_mfma_f32_32x32x2f32(AV, BV, acc[I][J], 0, 0, 0);

#define HRV_MMA_FRAGS()                                                                              \
  {                                                                                                  \
    if constexpr (BF) {                                                                              \
      /* v_mfma_f32_32x32x16_bf16: lane half h supplies k = 8h..8h+7 of each 16-wide step */          \
      _Pragma("unroll") for (int kq = 0; kq < KQ; ++kq)                                              \
          _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
              _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                       \
        const bf16x8 av = __builtin_bit_cast(bf16x8, fa[kq][i]);                                     \
        const bf16x8 bv = __builtin_bit_cast(bf16x8, fb[kq][j]);                                     \
        acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, acc[i][j], 0, 0, 0)       \
                         : __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i][j], 0, 0, 0);      \
      }                                                                                              \
    } else {                                                                                         \
      _Pragma("unroll") for (int kq = 0; kq < KQ; ++kq)                                              \
          _Pragma("unroll") for (int e = 0; e < 4; ++e)                                              \
              _Pragma("unroll") for (int i = 0; i < TM; ++i)                                         \
                  _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                   \
        HRV_MMA1(i, j, fa[kq][i][e], fb[kq][j][e])                                                   \
      }                                                                                              \
    }                                                                                                \
  }

  f32x4 fa[KQ][TM], fb[KQ][TN];
  // patch tiles with the vector epilogue: tile constants travel through registers + LDS (see the patch prologue)
  constexpr bool PCST = PATCH && SWAP;
  [[maybe_unused]] f32x4 tile_cst = (f32x4)(0.f);
  // SPADE epilogue on the patch tiles: the wave's x (and noise) values are requested in the PROLOGUE as well -- they are
  // the oldest requests of the tile, so the counted vmcnt waits of the weight stream never wait longer for them than the
  // prologue's own vmcnt(0) does, and the epilogue starts with its data in registers (32 + TM registers through the loop)
  constexpr bool PXV = PCST && TN == 2;
  [[maybe_unused]] f32x4 xv[PXV ? TM : 1][4];
  [[maybe_unused]] float zv[PXV ? TM : 1];

  if constexpr (PATCH) {
    // ---- patch mode main loop (see the VAR bit 6 note above)
    float* patch = smem;
    float* bst = smem + PPIX * PFL;                        // ST weight stages of BN rows x 128 B
    const SrcDev& S = p.src[0];
    const rsrc_t a_rsrc = make_rsrc(S.ptr, S.bytes);
    const int c64 = S.C >> 6;                              // 64-channel K-tiles per tap
    const int nchunk = S.C >> 7;                           // 128-channel patch chunks
    const int KTOT = 18 * nchunk;                          // weight tiles in patch order: chunk, tap, half
    constexpr int NB = BR;                                 // weight DMA instructions per wave per K-tile
    constexpr int WAIT_B1 = (NB & 15) | (7 << 4) | (0 << 8) | ((NB >> 4) << 14);   // vmcnt(NB) lgkmcnt(0)
    constexpr int WAIT_0 = 0 | (7 << 4) | (0 << 8);                                // vmcnt(0) lgkmcnt(0)
    // one DMA instruction = 4 consecutive halo pixels x 16 groups of 8 channels; instruction t of wave w is
    // t*8 + w (81 instructions per patch).  The 16-byte groups of a pixel are XOR-swizzled by (hx & 15) on the
    // SOURCE side, which makes the tap-shifted b128 fragment reads of 16 horizontally adjacent pixels bank-disjoint
#define HRV_PATCH_DMA(CHUNK)                                                                               \
    {                                                                                                      \
      for (int t = wave; t < PPIX / 4; t += NT / 64) {                                                     \
        const int P = 4 * t + (lane >> 4), s16 = lane & 15;                                                \
        const int hy = P / PW, hx = P - hy * PW;                                                           \
        const int y = pt_y0 - 1 + hy, x = pt_x0 - 1 + hx;                                                  \
        const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;                        \
        const int g = s16 ^ (hx & 15);                                                                     \
        const unsigned off = ((unsigned)((pt_n * p.H + y) * p.W + x) * (unsigned)S.cstride +               \
                              (unsigned)(S.coff + (CHUNK)*128 + g * 8)) * 2u;                              \
        dma16(a_rsrc, patch + t * 256, ok ? off : 0xFFFFFFF0u, 0u);                                        \
      }                                                                                                    \
    }
#define HRV_PATCH_KT(Q) ((((Q) % 18) >> 1) * c64 + ((Q) / 18) * 2 + ((Q)&1))
#define HRV_PATCH_BDMA(Q, BUF)                                                                             \
    {                                                                                                      \
      const unsigned w_soff = (unsigned)((HRV_PATCH_KT(Q) * p.CoutPad + n0) * RB);                         \
      float* Bbuf = bst + (BUF)*BN * LS;                                                                   \
      _Pragma("unroll") for (int j = 0; j < BR; ++j)                                                       \
          dma16(w_rsrc, Bbuf + ((NT * j + 64 * wave) / GPR) * LS, b_voff[j], w_soff);                      \
    }
    // Tile constants of the epilogue (bias / scale of the BN columns; SPADE: noise scale, mean, rstd of the tile's BN/2
    // channels): fetched NOW by the first lanes, parked in 4 registers through the main loop and spread through LDS
    // when the epilogue starts -- the epilogue then has no dependent global load except the pixel data, which it
    // issues in one batch.  (It was 9 us (dense) / 15 us (SPADE, training) of a 25-33 us tile, longer than the main
    // loop: up to 11 loads per channel group, each waited for before the next -- tools/patch_timeline.py.)
    if constexpr (PCST) {
      const int t4 = tid * 4;
      const float* src = nullptr;
      if (p.epi == 1) {
        const int cb0 = (n0 >> 6) * 32;                   // first channel of this tile's gamma|beta pairs
        if (tid < BN / 4) src = p.shift + n0 + t4;                                                     // gamma|beta bias
        else if (tid < 3 * BN / 8) { const int c = cb0 + t4 - BN; if (p.sns && c < p.sC) src = p.sns + c; }
        else if (tid < BN / 2) { const int c = cb0 + t4 - 3 * BN / 2; if (c < p.sC) src = p.smean + (size_t)pt_n * p.sC + c; }
        else if (tid < 5 * BN / 8) { const int c = cb0 + t4 - 2 * BN; if (c < p.sC) src = p.srstd + (size_t)pt_n * p.sC + c; }
      } else {
        if (tid < BN / 4) { if (p.scale && n0 + t4 < p.Cout) src = p.scale + n0 + t4; }
        else if (tid < BN / 2) { const int c = n0 + t4 - BN; if (p.shift && c < p.Cout) src = p.shift + c; }
      }
      tile_cst = src ? *reinterpret_cast<const f32x4*>(src) : ((p.epi != 1 && tid < BN / 4) ? (f32x4)(1.f) : (f32x4)(0.f));
    }
    if constexpr (PXV) {
      if (p.epi == 1) {
        const int cb = ((n0 + wn * 64) >> 6) * 32;       // channel base of this wave's gamma|beta pair
        const int HWo = p.Ho * p.Wo;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int px = row2pix((wm * TM + i) * 32 + l31);
          const int ps = px < p.M ? px : 0;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c0 = cb + 8 * g + 4 * lh;
            xv[i][g] = ld4rt<BF>(p.sx, (size_t)ps * p.sx_cs + p.sx_co + (c0 < p.sC ? c0 : 0), p.sx_f32);
          }
          zv[i] = 0.f;
          if (p.sz) {
            const int rem = ps - pt_n * HWo;
            const int h = rem / p.Wo, w = rem - h * p.Wo;
            zv[i] = p.sz[((size_t)pt_n * p.Wo + w) * p.Ho + h];
          }
        }
      }
    }
    HRV_PATCH_DMA(0)
    HRV_PATCH_BDMA(0, 0)
    if (ST == 3 && KTOT > 1) {
      HRV_PATCH_BDMA(1, 1)
      __builtin_amdgcn_s_waitcnt(WAIT_B1);
    } else {
      __builtin_amdgcn_s_waitcnt(WAIT_0);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (p.tlog && tid == 0) p.tlog[(size_t)bid * 8 + 1] = wall_clock64();      // patch + first weight tiles have landed
    int rb = 0, wb = ST - 1;
    for (int q = 0; q < KTOT; ++q) {
      const bool more = q + ST - 1 < KTOT;
      if (more) HRV_PATCH_BDMA(q + ST - 1, wb)
      {
        const int rem = q % 18, tap = rem >> 1, half = rem & 1;
        const int kh = tap / 3, kw = tap - kh * 3;
        const int sw = (l31 >> 1) & 7;
        const float* Bs = bst + rb * BN * LS + (wn * TN * 32 + l31) * LS;
        _Pragma("unroll") for (int kq = 0; kq < KQ; ++kq) {
          _Pragma("unroll") for (int i = 0; i < TM; ++i) {
            const int r = (wm * TM + i) * 32 + l31;
            const int hx = (r & 15) + kw;
            const int pix = ((r >> 4) + kh) * PW + hx;
            fa[kq][i] = *reinterpret_cast<const f32x4*>(patch + pix * PFL + (((half * 8 + kq * 2 + lh) ^ (hx & 15)) * 4));
          }
          _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[kq][j] =
              *reinterpret_cast<const f32x4*>(Bs + j * 32 * LS + (((kq * 2 + lh) ^ sw) * 4));
        }
      }
      HRV_MMA_FRAGS()
      if (q + 1 < KTOT) {
        asm volatile("" ::: "memory");
        if (ST == 3 && more) __builtin_amdgcn_s_waitcnt(WAIT_B1);
        else __builtin_amdgcn_s_waitcnt(WAIT_0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if ((q + 1) % 18 == 0) {
          // next 128-channel chunk: every wave has passed the barrier, the patch is free
          HRV_PATCH_DMA((q + 1) / 18)
          __builtin_amdgcn_s_waitcnt(WAIT_0);
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
      }
      rb = rb == ST - 1 ? 0 : rb + 1;
      wb = wb == ST - 1 ? 0 : wb + 1;
    }
#undef HRV_PATCH_DMA
#undef HRV_PATCH_KT
#undef HRV_PATCH_BDMA
    if (p.tlog && tid == 0) p.tlog[(size_t)bid * 8 + 2] = wall_clock64();      // main loop done (phase timeline, diag only)
  } else if constexpr (GLDS && ST == 3) {
    static_assert((BN * GPR) % NT == 0, "LDS-DMA B tile: every wave-instruction must be full");
    // every wave issues exactly AR + BR DMA instructions per K-tile (masked lanes use out-of-range offsets), so
    // "tile k+1 has landed, tile k+2 may still fly" is s_waitcnt vmcnt(AR + BR).  gfx9 encoding: vmcnt[3:0] in
    // bits 3:0, vmcnt[5:4] in bits 15:14, expcnt (bits 6:4) and lgkmcnt (bits 11:8) at their maxima = no wait.
    constexpr int NDMA = AR + BR;
    static_assert(NDMA < 64, "vmcnt is a 6-bit counter");
    constexpr int WAIT_ONE = (NDMA & 15) | (7 << 4) | (0 << 8) | ((NDMA >> 4) << 14);   // vmcnt(NDMA) lgkmcnt(0)
    constexpr int WAIT_ALL = 0 | (7 << 4) | (0 << 8);                                   // vmcnt(0) lgkmcnt(0)
    constexpr int WAIT_LDS = 0x3F | (7 << 4) | (0 << 8) | (3 << 14);                    // lgkmcnt(0) only
    const bool dma_wave = !SPEC || loader, mma_wave = !SPEC || !loader;
    const bool two = kt_begin + 1 < kt_end;
    if (dma_wave) {
      HRV_DMA_SRC()
      HRV_DMA_TILE(kt_begin, 0)
      HRV_DMA_ADVANCE()
      if (two) {
        HRV_DMA_TILE(kt_begin + 1, 1)
        HRV_DMA_ADVANCE()
        __builtin_amdgcn_s_waitcnt(WAIT_ONE);
      } else {
        __builtin_amdgcn_s_waitcnt(WAIT_ALL);
      }
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int rb = 0, wb = 2;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const bool more = kt + 2 < kt_end;
      if (dma_wave && more) {
        HRV_DMA_TILE(kt + 2, wb)   // buffer wb was read in iteration kt-1: every wave passed that barrier
        HRV_DMA_ADVANCE()
      }
      if (mma_wave) {
        HRV_READ_FRAGS(rb)
        HRV_MMA_FRAGS()
      }
      if (kt + 1 < kt_end) {
        asm volatile("" ::: "memory");
        if (dma_wave) {
          if (more) __builtin_amdgcn_s_waitcnt(WAIT_ONE);
          else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
        } else {
          __builtin_amdgcn_s_waitcnt(WAIT_LDS);   // this wave's fragment reads are done before the buffer is refilled
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      rb = rb == 2 ? 0 : rb + 1;
      wb = wb == 2 ? 0 : wb + 1;
    }
    if (SPEC && loader) return;   // the epilogue belongs to the MMA waves (no barrier after this point)
  } else if constexpr (GLDS) {
    static_assert((BN * GPR) % NT == 0, "LDS-DMA B tile: every wave-instruction must be full");
    HRV_DMA_SRC()
    HRV_DMA_TILE(kt_begin, kt_begin & 1)
    HRV_DMA_ADVANCE()
    __syncthreads();   // carries vmcnt(0): the DMA has landed before any wave reads the tile
    for (int kt = kt_begin; kt < kt_end - 1; ++kt) {
      HRV_DMA_TILE(kt + 1, (kt + 1) & 1)   // in flight while tile kt is multiplied
      HRV_DMA_ADVANCE()
      HRV_READ_FRAGS(kt & 1)
      HRV_MMA_FRAGS()
      __syncthreads();
    }
    HRV_READ_FRAGS((kt_end - 1) & 1)
    HRV_MMA_FRAGS()
  } else {
  // prologue: fetch + stage the first K-tile of this block's range
  HRV_LOAD_TILE(kt_begin)
  HRV_ADVANCE_ITER()
  HRV_STORE_TILE(kt_begin & 1)
  __syncthreads();

  // steady state: tile kt is multiplied while tile kt+1 travels global -> regs -> LDS
  for (int kt = kt_begin; kt < kt_end - 1; ++kt) {
    if (PIPE) {
      HRV_READ_FRAGS(kt & 1)
      HRV_LOAD_TILE(kt + 1)
      HRV_MMA_FRAGS()
      // scheduling recipe for this region: fragment reads first, then every MFMA is
      // followed by a slice of the gather's address arithmetic / load issue.
      __builtin_amdgcn_sched_group_barrier(0x100, KQ * (TM + TN), 0);
#pragma unroll
      for (int m = 0; m < (BF ? 1 : 4) * KQ * TM * TN; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
        if (m < AR + BR) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    } else {
      HRV_LOAD_TILE(kt + 1)
      HRV_READ_FRAGS(kt & 1)
      HRV_MMA_FRAGS()
    }
    HRV_ADVANCE_ITER()
    HRV_STORE_TILE((kt + 1) & 1)
    __syncthreads();
  }
  // last tile: nothing left to fetch
  HRV_READ_FRAGS((kt_end - 1) & 1)
  HRV_MMA_FRAGS()
  }

#undef HRV_DMA_TILE
#undef HRV_DMA_SRC
#undef HRV_DMA_ADVANCE
#undef HRV_LOAD_TILE
#undef HRV_ADVANCE_ITER
#undef HRV_STORE_TILE
#undef HRV_READ_FRAGS
#undef HRV_MMA1
#undef HRV_MMA_FRAGS

  if (p.splitk > 1) {
    // raw partial sums -> workspace; hrv::splitk_reduce_kernel applies the epilogue
    float* wsp = p.ws + (size_t)ks * p.M * p.CoutPad;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (SWAP) {
          const int pidx = row2pix((wm * TM + i) * 32 + l31);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c0 = n0 + (wn * TN + j) * 32 + 8 * g + 4 * lh;
            if (pidx < p.M) {
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
              *reinterpret_cast<f32x4*>(wsp + (size_t)pidx * p.CoutPad + c0) = v;
            }
          }
        } else {
          const int c = n0 + (wn * TN + j) * 32 + l31;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int pidx = row2pix((wm * TM + i) * 32 + 4 * lh + (e & 3) + 8 * (e >> 2));
            if (pidx < p.M) wsp[(size_t)pidx * p.CoutPad + c] = acc[i][j][e];
          }
        }
      }
    return;
  }
  // ---- fused epilogue: out = act(acc * scale[c] + shift[c] + residual)
  if (!SWAP) {
    // D layout: col = lane&31 (cout), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = n0 + (wn * TN + j) * 32 + l31;
      const bool c_ok = c < p.Cout;
      const float sc = (c_ok && p.scale) ? p.scale[c] : 1.f;
      const float sh = (c_ok && p.shift) ? p.shift[c] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int prow0 = (wm * TM + i) * 32 + 4 * lh;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int pidx = row2pix(prow0 + (e & 3) + 8 * (e >> 2));
          if (c_ok && pidx < p.M) {
            float v = acc[i][j][e] * sc + sh;
            const size_t opix = out_pixel(p, pidx);
            if (p.res) v = res_combine(v, ld1rt<BF>(p.res, opix * p.res_cs + p.res_co + c, p.res_f32), p.res_mode, p.slope);
            v = apply_act(v, p.act, p.slope);
            if (!p.out_up) {
              st1rt<BF>(p.out, opix * p.out_cs + p.out_co + c, v, p.out_f32);
            } else {
              const int n = pidx / (p.Ho * p.Wo), rem = pidx - n * (p.Ho * p.Wo);
              const int h = rem / p.Wo, w = rem - h * p.Wo;
              const size_t o = (((size_t)n * 2 * p.Ho + 2 * h) * 2 * p.Wo + 2 * w) * p.out_cs + p.out_co + c;
              st1rt<BF>(p.out, o, v, p.out_f32); st1rt<BF>(p.out, o + p.out_cs, v, p.out_f32);
              st1rt<BF>(p.out, o + (size_t)2 * p.Wo * p.out_cs, v, p.out_f32); st1rt<BF>(p.out, o + (size_t)2 * p.Wo * p.out_cs + p.out_cs, v, p.out_f32);
            }
          }
        }
      }
    }
  } else {
    // D layout: col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (cout):
    // regs 4g..4g+3 are 4 consecutive output channels of this lane's pixel.
    // (host side only selects this variant when Cout, out/residual strides and offsets are
    //  multiples of 4, so every group of 4 channels is stored as one 16-byte access)
    if (p.epi == 1) {
      // SPADE: tiles come in (gamma | beta) pairs of the same 32 channels, so this lane holds
      // gamma and beta of its pixel for the same 4 channels in acc[i][2q] / acc[i][2q+1].
      if constexpr (PCST && TN == 2) {
        // patch tiles: constants from LDS, every pixel load of the wave issued before the first use
        __syncthreads();                                 // every wave is done with the operand stages
        float* cbuf = smem;
        if (tid < 5 * BN / 8) *reinterpret_cast<f32x4*>(cbuf + 4 * tid) = tile_cst;
        const int col0 = n0 + wn * 64;                   // first gamma column of this wave's pair
        const int cb = (col0 >> 6) * 32;                 // its channel base
        const int HWo = p.Ho * p.Wo;
        int pidx[TM];
        bool okp[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          pidx[i] = row2pix((wm * TM + i) * 32 + l31);
          okp[i] = pidx[i] < p.M;
        }
        __syncthreads();                                 // the constants are in LDS
        // results leave through a per-wave LDS scratch so that the global stores run along the channels: 64-byte (bf16
        // result) / 128-byte ((1 + gamma), fp32 result) runs per pixel instead of 8- / 16-byte pieces at the pixel stride
        constexpr int SCS = 36;                          // scratch row stride, floats (32 channels + 4: conflict-free)
        float* scr = smem + 5 * BN / 2 + wave * (2 * 32 * SCS);   // [v | 1+gamma] x 32 pixels, behind the constants
        static_assert(sizeof(smem) >= (size_t)(5 * BN / 2 + (NT / 64) * 2 * 32 * SCS) * 4, "SPADE epilogue scratch");
        const int lc = wn * 32;                          // this wave's first channel within the tile
        const bool bf_out = BF && !p.out_f32;
        const bool staged = ((uintptr_t)p.out & 15) == 0 && ((p.out_cs | p.out_co) & 7) == 0 && (p.sC & 3) == 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int cl = lc + 8 * g + 4 * lh, c0 = cb + 8 * g + 4 * lh;
            const bool c_ok = c0 < p.sC;
            const f32x4 bg = *reinterpret_cast<const f32x4*>(cbuf + wn * 64 + 8 * g + 4 * lh);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(cbuf + wn * 64 + 32 + 8 * g + 4 * lh);
            const f32x4 ns4 = *reinterpret_cast<const f32x4*>(cbuf + BN + cl);
            const f32x4 mu = *reinterpret_cast<const f32x4*>(cbuf + 3 * BN / 2 + cl);
            const f32x4 rs = *reinterpret_cast<const f32x4*>(cbuf + 2 * BN + cl);
            f32x4 v = (f32x4)(0.f), g1 = (f32x4)(0.f);
            if (c_ok && okp[i]) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float x = xv[i][g][e] + zv[i] * ns4[e];
                g1[e] = 1.f + acc[i][0][4 * g + e] + bg[e];
                const float bet = acc[i][1][4 * g + e] + bb[e];
                v[e] = apply_act((x - mu[e]) * rs[e] * g1[e] + bet, p.act, p.slope);
              }
              if (!staged) {
                if (p.sg1p) *reinterpret_cast<f32x4*>(p.sg1p + (size_t)pidx[i] * p.sC + c0) = g1;
                st4rt<BF>(p.out, (size_t)pidx[i] * p.out_cs + p.out_co + c0, v, p.out_f32);
              }
            }
            if (staged) {
              *reinterpret_cast<f32x4*>(scr + l31 * SCS + 8 * g + 4 * lh) = v;
              *reinterpret_cast<f32x4*>(scr + 32 * SCS + l31 * SCS + 8 * g + 4 * lh) = g1;
            }
          }
          if (staged) {
            // same wave wrote and reads: LDS operations of a wave complete in order
#pragma unroll
            for (int k = 0; k < 4; ++k) {                  // 32 pixels x 8 chunks of 4 channels
              const int t = lane + 64 * k, px = t >> 3, kk = t & 7;
              const int po = row2pix((wm * TM + i) * 32 + px);
              const bool ok = po < p.M && cb + kk * 4 < p.sC;
              if (p.sg1p && ok)
                *reinterpret_cast<f32x4*>(p.sg1p + (size_t)po * p.sC + cb + kk * 4) =
                    *reinterpret_cast<const f32x4*>(scr + 32 * SCS + px * SCS + kk * 4);
              if (!bf_out && ok)
                *reinterpret_cast<f32x4*>(p.out + (size_t)po * p.out_cs + p.out_co + cb + kk * 4) =
                    *reinterpret_cast<const f32x4*>(scr + px * SCS + kk * 4);
            }
            if (bf_out) {
#pragma unroll
              for (int k = 0; k < 2; ++k) {                // 32 pixels x 4 chunks of 8 channels
                const int t = lane + 64 * k, px = t >> 2, kk = t & 3;
                const int po = row2pix((wm * TM + i) * 32 + px);
                const f32x4 lo = *reinterpret_cast<const f32x4*>(scr + px * SCS + kk * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(scr + px * SCS + kk * 8 + 4);
                if (po < p.M && cb + kk * 8 < p.sC)
                  *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned short*>(p.out) + (size_t)po * p.out_cs + p.out_co + cb + kk * 8) =
                      pack_bf16x8(lo, hi);
              }
            }
          }
        }
        return;
      }
      if constexpr (TN % 2 == 0) {
        const int HWo = p.Ho * p.Wo;
#pragma unroll
        for (int q = 0; q < TN / 2; ++q) {
          const int col0 = n0 + (wn * TN + 2 * q) * 32;  // first gamma column of the pair
          const int cb = (col0 >> 6) * 32;               // channel base of the pair
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c0 = cb + 8 * g + 4 * lh;
            const bool c_ok = c0 < p.sC;
            const int cs = c_ok ? c0 : 0;
            const int colg = c_ok ? col0 + 8 * g + 4 * lh : 0;
            const f32x4 bg = *reinterpret_cast<const f32x4*>(p.shift + colg);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(p.shift + colg + (c_ok ? 32 : 0));
            const f32x4 ns4 = p.sns ? *reinterpret_cast<const f32x4*>(p.sns + cs) : (f32x4)(0.f);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              const int pidx = row2pix((wm * TM + i) * 32 + l31);
              if (c_ok && pidx < p.M) {
                const int n = pidx / HWo;
                f32x4 x = ld4rt<BF>(p.sx, (size_t)pidx * p.sx_cs + p.sx_co + c0, p.sx_f32);
                if (p.sz) {
                  const int rem = pidx - n * HWo;
                  const int h = rem / p.Wo, w = rem - h * p.Wo;
                  x += p.sz[((size_t)n * p.Wo + w) * p.Ho + h] * ns4;
                }
                const f32x4 mu = *reinterpret_cast<const f32x4*>(p.smean + (size_t)n * p.sC + c0);
                const f32x4 rs = *reinterpret_cast<const f32x4*>(p.srstd + (size_t)n * p.sC + c0);
                f32x4 v, g1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  g1[e] = 1.f + acc[i][2 * q][4 * g + e] + bg[e];
                  const float bet = acc[i][2 * q + 1][4 * g + e] + bb[e];
                  v[e] = apply_act((x[e] - mu[e]) * rs[e] * g1[e] + bet, p.act, p.slope);
                }
                if (p.sg1p) *reinterpret_cast<f32x4*>(p.sg1p + (size_t)pidx * p.sC + c0) = g1;
                st4rt<BF>(p.out, (size_t)pidx * p.out_cs + p.out_co + c0, v, p.out_f32);
              }
            }
          }
        }
      }
      return;
    }
    // Dense outputs: the finished 32-pixel x 32-channel tile goes through a per-wave LDS scratch so that the global
    // stores run along the channels (128-byte runs per pixel for fp32, 64 for bf16) instead of 16 / 8 bytes per lane
    // at the pixel stride -- the stores of the wide memory-bound layers (conv_shared: 384 columns, K = 72) were 8-byte
    // pieces 768 bytes apart.  Pad channels are never written (a slice may sit inside a wider tensor).
    constexpr int SCR = 36;                                  // scratch row stride in floats (32 + 4: conflict-free)
    const bool bf_out = BF && !p.out_f32;
    // (out_up: the fused nearest x2 upsample of the result -- fp32 only -- writes each staged row to its 2x2 block)
    const bool lds_store = !SPEC && (!p.out_up || !bf_out) && p.out_step != 2 && ((uintptr_t)p.out & 15) == 0 &&
                           (!bf_out || ((p.Cout | p.out_cs | p.out_co) & 7) == 0);
    if (lds_store) {
      static_assert(sizeof(smem) >= (size_t)(NT / 64) * 32 * SCR * 4, "epilogue scratch must fit the operand stages");
      __syncthreads();                                       // every wave is done reading the operand stages
      float* scr = smem + wave * (32 * SCR);
      // patch tiles: scale / shift of the tile's columns come from LDS (parked in registers since the prologue) and
      // the residual of every accumulator group is requested before the first one is used
      [[maybe_unused]] float* cbuf = smem + (NT / 64) * 32 * SCR;
      [[maybe_unused]] f32x4 rv[TM][4];                      // (one 32-column group at a time: two blocks per CU need <= 256 registers)
      if constexpr (PCST) {
        static_assert(sizeof(smem) >= (size_t)((NT / 64) * 32 * SCR + 2 * BN) * 4, "epilogue scratch + constants");
        if (tid < BN / 2) *reinterpret_cast<f32x4*>(cbuf + 4 * tid) = tile_cst;
        __syncthreads();                                     // the constants are in LDS
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int cj = n0 + (wn * TN + j) * 32;              // first channel of this 32-column tile
        if constexpr (PCST) {
          if (p.res) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              const int px = row2pix((wm * TM + i) * 32 + l31);
              const int ps = px < p.M ? px : 0;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int c0 = cj + 8 * g + 4 * lh;
                rv[i][g] = ld4rt<BF>(p.res, (size_t)ps * p.res_cs + p.res_co + (c0 < p.Cout ? c0 : 0), p.res_f32);
              }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int pidx = row2pix((wm * TM + i) * 32 + l31);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c0 = cj + 8 * g + 4 * lh;
            const bool c_ok = c0 < p.Cout;
            const int cs = c_ok ? c0 : 0;
            f32x4 sc, sh;
            if constexpr (PCST) {
              sc = *reinterpret_cast<const f32x4*>(cbuf + (wn * TN + j) * 32 + 8 * g + 4 * lh);
              sh = *reinterpret_cast<const f32x4*>(cbuf + BN + (wn * TN + j) * 32 + 8 * g + 4 * lh);
            } else {
              sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + cs) : (f32x4)(1.f);
              sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + cs) : (f32x4)(0.f);
            }
            f32x4 v = (f32x4)(0.f);
            if (c_ok && pidx < p.M) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] * sc[e] + sh[e];
              if (p.res) {
                f32x4 r4;
                if constexpr (PCST) r4 = rv[i][g];
                else r4 = ld4rt<BF>(p.res, (size_t)pidx * p.res_cs + p.res_co + c0, p.res_f32);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = res_combine(v[e], r4[e], p.res_mode, p.slope);
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act, p.slope);
            }
            *reinterpret_cast<f32x4*>(scr + l31 * SCR + 8 * g + 4 * lh) = v;
          }
          // same wave wrote and reads: LDS operations of a wave complete in order (the compiler waits lgkmcnt)
          if (bf_out) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {                    // 32 pixels x 4 chunks of 8 channels
              const int t = lane + 64 * k, px = t >> 2, kk = t & 3;
              const int po = row2pix((wm * TM + i) * 32 + px);
              const f32x4 lo = *reinterpret_cast<const f32x4*>(scr + px * SCR + kk * 8);
              const f32x4 hi = *reinterpret_cast<const f32x4*>(scr + px * SCR + kk * 8 + 4);
              if (po < p.M && cj + kk * 8 < p.Cout)
                *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned short*>(p.out) + (size_t)po * p.out_cs + p.out_co + cj + kk * 8) =
                    pack_bf16x8(lo, hi);
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                    // 32 pixels x 8 chunks of 4 channels
              const int t = lane + 64 * k, px = t >> 3, kk = t & 7;
              const int po = row2pix((wm * TM + i) * 32 + px);
              const f32x4 v = *reinterpret_cast<const f32x4*>(scr + px * SCR + kk * 4);
              if (po < p.M && cj + kk * 4 < p.Cout) {
                if (!p.out_up) {
                  *reinterpret_cast<f32x4*>(p.out + (size_t)po * p.out_cs + p.out_co + cj + kk * 4) = v;
                } else {
                  const int n = po / (p.Ho * p.Wo), rem = po - n * (p.Ho * p.Wo);
                  const int h = rem / p.Wo, w = rem - h * p.Wo;
                  float* o = p.out + (((size_t)n * 2 * p.Ho + 2 * h) * 2 * p.Wo + 2 * w) * p.out_cs + p.out_co + cj + kk * 4;
                  *reinterpret_cast<f32x4*>(o) = v;
                  *reinterpret_cast<f32x4*>(o + p.out_cs) = v;
                  *reinterpret_cast<f32x4*>(o + (size_t)2 * p.Wo * p.out_cs) = v;
                  *reinterpret_cast<f32x4*>(o + (size_t)2 * p.Wo * p.out_cs + p.out_cs) = v;
                }
              }
            }
          }
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = n0 + (wn * TN + j) * 32 + 8 * g + 4 * lh;
        const bool c_ok = c0 < p.Cout;
        const int cs = c_ok ? c0 : 0;
        const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + cs) : (f32x4)(1.f);
        const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + cs) : (f32x4)(0.f);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int pidx = row2pix((wm * TM + i) * 32 + l31);
          if (c_ok && pidx < p.M) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] * sc[e] + sh[e];
            const size_t opix = out_pixel(p, pidx);
            if (p.res) {
              const f32x4 r4 = ld4rt<BF>(p.res, opix * p.res_cs + p.res_co + c0, p.res_f32);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = res_combine(v[e], r4[e], p.res_mode, p.slope);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act, p.slope);
            if (!p.out_up) {
              st4rt<BF>(p.out, opix * p.out_cs + p.out_co + c0, v, p.out_f32);
            } else {
              const int n = pidx / (p.Ho * p.Wo), rem = pidx - n * (p.Ho * p.Wo);
              const int h = rem / p.Wo, w = rem - h * p.Wo;
              const size_t o = (((size_t)n * 2 * p.Ho + 2 * h) * 2 * p.Wo + 2 * w) * p.out_cs + p.out_co + c0;
              st4rt<BF>(p.out, o, v, p.out_f32);
              st4rt<BF>(p.out, o + p.out_cs, v, p.out_f32);
              st4rt<BF>(p.out, o + (size_t)2 * p.Wo * p.out_cs, v, p.out_f32);
              st4rt<BF>(p.out, o + (size_t)2 * p.Wo * p.out_cs + p.out_cs, v, p.out_f32);
            }
          }
        }
      }
    }
  }
}

template <int TM, int TN, int WM, int WN, int VAR, bool BF, int RB = 64>
__global__ __launch_bounds__(64 * WM * WN * ((VAR & 32) ? 2 : 1), ((VAR & 64) && WM * WN == 4) ? 2 : 1) void conv_mfma_kernel(
    const ConvParams p) {
  if constexpr ((VAR & 64) != 0) {
    // patch tiles: PERSISTENT blocks (grid = resident slots).  A one-tile-per-block grid of these 77-KB-LDS blocks spent
    // ~80 us of a 400 us launch (6144 tiles) on workgroup dispatch alone -- the same launch with every DMA, fragment
    // read, MFMA and the epilogue switched off (tools/patch_ablation.sh, HRV_PATCH_DBG=31).
    const int total = p.m_tiles * p.n_tiles;
    for (int bid = blockIdx.x; bid < total; bid += gridDim.x) {
      if (p.tlog && threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.tlog[(size_t)bid * 8 + 0] = wall_clock64();
        p.tlog[(size_t)bid * 8 + 4] = ((unsigned long long)xcc << 32) | hw;
        p.tlog[(size_t)bid * 8 + 5] = blockIdx.x;
      }
      conv_mfma_tile<TM, TN, WM, WN, VAR, BF, RB>(p, bid);
      if (p.tlog) {
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0 << 8));   // timeline only: this thread's stores have left
        if (threadIdx.x == 0) p.tlog[(size_t)bid * 8 + 3] = wall_clock64();
      }
      __syncthreads();      // the next tile's DMA overwrites the LDS this tile's epilogue staged through
    }
  } else {
    conv_mfma_tile<TM, TN, WM, WN, VAR, BF, RB>(p, blockIdx.x);
  }
}

// Split-K second stage: fixed-order sum of the partial tiles + the standard epilogue.
template <bool BF>
__global__ void splitk_reduce_kernel(const ConvParams p) {
  const int C4 = (p.Cout + 3) / 4;
  const size_t total = (size_t)p.M * C4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(idx % C4);
    const int pidx = (int)(idx / C4);
    f32x4 v = (f32x4)(0.f);
    for (int s = 0; s < p.splitk; ++s)
      v += *reinterpret_cast<const f32x4*>(p.ws + ((size_t)s * p.M + pidx) * p.CoutPad + g * 4);
    int on = 0, oh = 0, ow = 0;
    if (p.out_up) {
      on = pidx / (p.Ho * p.Wo);
      const int rem = pidx - on * (p.Ho * p.Wo);
      oh = rem / p.Wo;
      ow = rem - oh * p.Wo;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = g * 4 + e;
      if (c >= p.Cout) break;
      float t = v[e] * (p.scale ? p.scale[c] : 1.f) + (p.shift ? p.shift[c] : 0.f);
      const size_t opix = out_pixel(p, pidx);
      if (p.res) t = res_combine(t, ld1rt<BF>(p.res, opix * p.res_cs + p.res_co + c, p.res_f32), p.res_mode, p.slope);
      t = apply_act(t, p.act, p.slope);
      if (!p.out_up) {
        st1rt<BF>(p.out, opix * p.out_cs + p.out_co + c, t, p.out_f32);
      } else {
        const size_t o = (((size_t)on * 2 * p.Ho + 2 * oh) * 2 * p.Wo + 2 * ow) * p.out_cs + p.out_co + c;
        st1rt<BF>(p.out, o, t, p.out_f32); st1rt<BF>(p.out, o + p.out_cs, t, p.out_f32);
        st1rt<BF>(p.out, o + (size_t)2 * p.Wo * p.out_cs, t, p.out_f32); st1rt<BF>(p.out, o + (size_t)2 * p.Wo * p.out_cs + p.out_cs, t, p.out_f32);
      }
    }
  }
}

// One thread per output element, raw OIHW weights: a device-side cross-check.
__global__ void conv_f32_naive_kernel(const ConvParams p, const float* __restrict__ w, int Cin_real_total,
                                      const int* __restrict__ real_c /*[nsrc]*/) {
  const size_t total = (size_t)p.M * p.Cout;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(idx % p.Cout);
    const int pidx = (int)(idx / p.Cout);
    const int n = pidx / (p.Ho * p.Wo);
    const int rem = pidx - n * (p.Ho * p.Wo);
    const int ho = rem / p.Wo, wo = rem - (rem / p.Wo) * p.Wo;
    float acc = 0.f;
    for (int kh = 0; kh < p.KH; ++kh)
      for (int kw = 0; kw < p.KW; ++kw) {
        const int hi = ho * p.stride - p.pad + kh, wi = wo * p.stride - p.pad_w + kw;
        if (hi < 0 || hi >= p.H || wi < 0 || wi >= p.W) continue;
        int cbase = 0;
        for (int s = 0; s < p.nsrc; ++s) {
          const SrcDev sd = p.src[s];
          const int sr = sd.up_shift > 0 ? sd.up_shift : 0, sl = sd.up_shift < 0 ? -sd.up_shift : 0;
          const int Hs = (p.H >> sr) << sl, Ws = (p.W >> sr) << sl;
          const float* px = sd.ptr + ((size_t)(n * Hs + ((hi >> sr) << sl)) * Ws + ((wi >> sr) << sl)) * sd.cstride + sd.coff;
          for (int c = 0; c < real_c[s]; ++c) {
            float x = px[c];
            if (sd.pre_act == HRV_ACT_LRELU) x = x > 0.f ? x : x * p.pre_slope;
            acc = fmaf(x, w[(((size_t)co * Cin_real_total + cbase + c) * p.KH + kh) * p.KW + kw], acc);
          }
          cbase += real_c[s];
        }
      }
    float v = acc * (p.scale ? p.scale[co] : 1.f) + (p.shift ? p.shift[co] : 0.f);
    const size_t opix = out_pixel(p, pidx);
    if (p.res) v = res_combine(v, p.res[opix * p.res_cs + p.res_co + co], p.res_mode, p.slope);
    v = apply_act(v, p.act, p.slope);
    if (!p.out_up) {
      p.out[opix * p.out_cs + p.out_co + co] = v;
    } else {
      float* o = p.out + (((size_t)n * 2 * p.Ho + 2 * ho) * 2 * p.Wo + 2 * wo) * p.out_cs + p.out_co + co;
      o[0] = v; o[p.out_cs] = v;
      o[(size_t)2 * p.Wo * p.out_cs] = v; o[(size_t)2 * p.Wo * p.out_cs + p.out_cs] = v;
    }
  }
}

struct TileCfg {
  int TM, TN, WM, WN;
  int RB;  // bytes per K-tile row: 64 (16 fp32 / 32 bf16 k-values) or 128 (bf16 engine only: 64 k-values)
};
// id -> (TM,TN,WM,WN); BM = 32*TM*WM, BN = 32*TN*WN
static const TileCfg kCfgs[] = {
    {2, 2, 2, 2, 64},   // 0: 128 x 128
    {1, 3, 4, 1, 64},   // 1: 128 x 96
    {2, 3, 4, 1, 64},   // 2: 256 x 96
    {2, 1, 4, 1, 64},   // 3: 256 x 32
    {2, 2, 4, 1, 64},   // 4: 256 x 64
    {1, 1, 4, 1, 64},   // 5: 128 x 32
    {1, 2, 4, 1, 64},   // 6: 128 x 64
    {4, 2, 2, 2, 64},   // 7: 256 x 128
    {2, 2, 2, 2, 128},  // 8: 128 x 128, 64 bf16 k-values per K-tile (bf16 engine only)
    {1, 2, 4, 1, 128},  // 9: 128 x 64,  64 bf16 k-values per K-tile (bf16 engine only)
    {4, 2, 2, 2, 128},  // 10: 256 x 128 (wave tile 128 x 64: 32 MFMAs per K-tile), LDS-DMA, 96 KB LDS, 1 block / CU
    {2, 4, 2, 2, 128},  // 11: 128 x 256 (wave tile 64 x 128)
    {2, 2, 4, 2, 128},  // 12: 256 x 128 with EIGHT waves (wave tile 64 x 64, two waves per SIMD), LDS-DMA, 96 KB LDS
    {2, 2, 4, 2, 128},  // 13: the same tile with THREE LDS stages (144 KB): two tiles' DMA in flight per block
    {4, 2, 2, 2, 128},  // 14: 256 x 128, 4 MMA waves (wave tile 128 x 64) + 4 LOADER waves, three LDS stages
    {2, 2, 2, 2, 128},  // 15: 128 x 128, 4 MMA waves (wave tile 64 x 64) + 4 LOADER waves, three LDS stages (96 KB)
    {2, 2, 4, 2, 128},  // 16: PATCH mode -- 16x16-pixel x 128-column tile, 18x18 halo patch resident in LDS (3x3 s1 only)
    {2, 2, 2, 2, 128},  // 17: PATCH mode -- 8x16-pixel x 128-column tile, 10x18 halo patch, two blocks per CU
    {1, 2, 4, 1, 128},  // 18: PATCH mode -- 8x16-pixel x 64-column tile (column counts that are odd multiples of 64)
    {2, 2, 4, 1, 128},  // 19: WIDE PATCH mode (conv_patchw.hip) -- 16x16-pixel tile x up to 192 columns per block, one block
                        //     per CU; weights packed for 64-column tiles (this entry only carries BM = 256, BN = 64)
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

static int cfg_bm(int c) { return 32 * kCfgs[c].TM * kCfgs[c].WM; }
static int cfg_bn(int c) { return 32 * kCfgs[c].TN * kCfgs[c].WN; }
static int cfg_rb(int c) { return kCfgs[c].RB; }

// Split-K factor for a launch that would otherwise leave most of the 256 CUs idle
// (tocg levels 3-4 and the generator's 8x6..32x24 blocks: few pixels, K up to 9360).
static int pick_splitk(int nblk, int KT, bool allowed) {
  // target >= 4 blocks per CU: below that the 4-wave blocks cannot hide each other's barriers
  if (!allowed || nblk >= 768 || KT < 16) return 1;
  int s = (1024 + nblk - 1) / nblk;
  if (s > 32) s = 32;
  if (s > KT / 8) s = KT / 8;
  return s < 2 ? 1 : s;
}

static int fill_params(const hrv_conv2d_t* d, ConvParams& p, bool need_packed, bool bf = false) {
  const bool srcf = bf && d && ((d->mixed_flags >> 3) & 1);   // fp32 sources rounded to bf16 in the gather
  const int cm = (bf && !srcf) ? 8 : 4;     // channel granularity = one 16-byte gather group
  HRV_REQUIRE(d != nullptr, "conv2d: null descriptor");
  const int rb = (need_packed && d->tile_cfg >= 0 && d->tile_cfg < kNumCfgs) ? cfg_rb(d->tile_cfg) : 64;
  HRV_REQUIRE(bf || rb == 64, "conv2d: tile_cfg=%d (128-byte K-tile rows) exists on the bf16 engine only", d->tile_cfg);
  const int bke = rb / (bf ? 2 : 4);  // k-values per K-tile
  HRV_REQUIRE(d->nsrc >= 1 && d->nsrc <= HRV_MAX_SRC, "conv2d: nsrc=%d out of range", d->nsrc);
  HRV_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0, "conv2d: bad extent");
  HRV_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "conv2d: bad kernel geometry");
  const int padw = d->pad_w_plus1 > 0 ? d->pad_w_plus1 - 1 : d->pad;
  HRV_REQUIRE(d->free_extent ||
                  (d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1 && d->Wo == (d->W + 2 * padw - d->KW) / d->stride + 1),
              "conv2d: Ho/Wo (%d,%d) inconsistent with H,W,k,stride,pad", d->Ho, d->Wo);
  HRV_REQUIRE(d->out_step == 0 || (d->out_step == 2 && d->out_up_shift == 0 && !d->spade && d->out_off_h >= 0 &&
                                   d->out_off_w >= 0 && 2 * (d->Ho - 1) + d->out_off_h < d->out_H &&
                                   2 * (d->Wo - 1) + d->out_off_w < d->out_W),
              "conv2d: out_step/out_off/out_H/out_W inconsistent");
  HRV_REQUIRE(d->res_mode == 0 || d->res_mode == 1, "conv2d: res_mode");
  HRV_REQUIRE(d->Cout > 0 && d->out != nullptr, "conv2d: bad output");
  HRV_REQUIRE((int64_t)d->N * d->Ho * d->Wo < (int64_t)1 << 31, "conv2d: too many output pixels");
  memset(&p, 0, sizeof(p));
  p.bf16 = bf ? 1 : 0;
  p.out_f32 = bf ? (d->mixed_flags & 1) : 1;
  p.res_f32 = bf ? ((d->mixed_flags >> 1) & 1) : 1;
  p.sx_f32 = bf ? ((d->mixed_flags >> 2) & 1) : 1;
  p.src_f32 = srcf ? 1 : 0;
  HRV_REQUIRE(!srcf || rb == 128, "conv2d: fp32-source bf16 mode needs a 128-byte-row tile (cfg 8/9)");
  p.nsrc = d->nsrc;
  int chunks_total = 0;
  bool dma_ok = bf && rb == 128 && !srcf;   // LDS-DMA staging: every operand must fit a 32-bit buffer resource
  for (int i = 0; i < d->nsrc; ++i) {
    const hrv_src_t& s = d->src[i];
    HRV_REQUIRE(s.ptr != nullptr, "conv2d: src[%d] null", i);
    HRV_REQUIRE(s.C > 0 && s.C % cm == 0 && s.cstride % cm == 0 && s.coff % cm == 0 && s.coff + s.C <= s.cstride,
                "conv2d: src[%d] channels C=%d cstride=%d coff=%d must be multiples of %d and in range", i, s.C,
                s.cstride, s.coff, cm);
    HRV_REQUIRE(((uintptr_t)s.ptr & 15) == 0, "conv2d: src[%d] not 16-byte aligned", i);
    HRV_REQUIRE(s.up_shift >= -7 && s.up_shift <= 1 && (s.up_shift != 1 || (d->H % 2 == 0 && d->W % 2 == 0)),
                "conv2d: src[%d] up_shift=%d out of range", i, s.up_shift);
    HRV_REQUIRE(!need_packed || s.pre_act == HRV_ACT_NONE, "conv2d: src[%d] pre_act is not supported by the MFMA gather "
                "(fuse the activation into the producer's epilogue)", i);
    {
      const int sr = s.up_shift > 0 ? s.up_shift : 0, sl = s.up_shift < 0 ? -s.up_shift : 0;
      const int64_t elems = (int64_t)d->N * ((d->H >> sr) << sl) * ((d->W >> sr) << sl) * s.cstride;
      HRV_REQUIRE(elems < ((int64_t)1 << 32), "conv2d: src[%d] has %lld elements; the gather uses 32-bit offsets", i,
                  (long long)elems);
    }
    p.src[i].ptr = (const float*)s.ptr;
    p.src[i].C = s.C;
    p.src[i].cstride = s.cstride;
    p.src[i].coff = s.coff;
    p.src[i].up_shift = s.up_shift;
    p.src[i].pre_act = s.pre_act;
    p.src[i].chunks = (s.C + bke - 1) / bke;
    chunks_total += p.src[i].chunks;
    {
      const int sr = s.up_shift > 0 ? s.up_shift : 0, sl = s.up_shift < 0 ? -s.up_shift : 0;
      const int64_t bytes = (int64_t)d->N * ((d->H >> sr) << sl) * ((d->W >> sr) << sl) * s.cstride * ((bf && !srcf) ? 2 : 4);
      p.src[i].bytes = bytes < (int64_t)0xFFFFFFF0 ? (unsigned)bytes : 0u;
      if (bytes >= (int64_t)0xFFFFFFF0) dma_ok = false;
    }
  }
  p.N = d->N; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
  p.M = d->N * d->Ho * d->Wo;
  p.Cout = d->Cout;
  p.chunks_total = chunks_total;
  p.KT = d->KH * d->KW * chunks_total;
  p.scale = d->scale; p.shift = d->shift;
  p.res = (const float*)d->residual; p.res_cs = d->res_cstride; p.res_co = d->res_coff;
  p.act = d->act; p.slope = d->act_slope; p.pre_slope = 0.2f;
  p.out = (float*)d->out; p.out_cs = d->out_cstride; p.out_co = d->out_coff;
  p.pad_w = padw;
  p.out_step = d->out_step; p.out_oh = d->out_off_h; p.out_ow = d->out_off_w; p.out_H = d->out_H; p.out_W = d->out_W;
  p.res_mode = d->res_mode;
  HRV_REQUIRE(d->out_up_shift == 0 || (d->out_up_shift == 1 && !d->spade), "conv2d: out_up_shift must be 0 or 1");
  p.out_up = d->out_up_shift;
  if (d->spade) {
    const hrv_spade_epi_t& e = *d->spade;
    HRV_REQUIRE(need_packed, "conv2d: the SPADE epilogue exists on the MFMA engine only");
    HRV_REQUIRE(e.x && e.mean && e.rstd && d->shift, "conv2d/spade: null pointer");
    HRV_REQUIRE(e.C > 0 && e.C % 4 == 0 && e.x_cstride % 4 == 0 && e.x_coff % 4 == 0 && e.x_coff + e.C <= e.x_cstride,
                "conv2d/spade: channels must be multiples of 4 and in range");
    HRV_REQUIRE(d->Cout == (e.C + 31) / 32 * 64, "conv2d/spade: Cout must be 2*ceil32(C) = %d", (e.C + 31) / 32 * 64);
    HRV_REQUIRE(d->stride == 1 && d->Ho == d->H && d->Wo == d->W, "conv2d/spade: 'same' stride-1 geometry only");
    HRV_REQUIRE((e.noise_z == nullptr) == (e.noise_scale == nullptr), "conv2d/spade: noise_z/noise_scale go together");
    HRV_REQUIRE(d->out_cstride % 4 == 0 && d->out_coff % 4 == 0 && d->out_cstride >= d->out_coff + e.C,
                "conv2d/spade: out slice");
    HRV_REQUIRE((((uintptr_t)e.x | (uintptr_t)e.mean | (uintptr_t)e.rstd | (uintptr_t)e.noise_scale |
                  (uintptr_t)d->shift | (uintptr_t)d->out) & 15) == 0, "conv2d/spade: 16-byte alignment");
    p.epi = 1; p.sx = e.x; p.sx_cs = e.x_cstride; p.sx_co = e.x_coff; p.sC = e.C;
    p.smean = e.mean; p.srstd = e.rstd; p.sz = e.noise_z; p.sns = e.noise_scale; p.sg1p = e.g1p_out;
    p.res = nullptr; p.scale = nullptr;
  } else {
    HRV_REQUIRE(d->out_cstride >= d->out_coff + d->Cout, "conv2d: out slice out of range");
  }
  HRV_REQUIRE(d->residual == nullptr || d->res_cstride >= d->res_coff + d->Cout, "conv2d: residual slice out of range");
  if (need_packed) {
    HRV_REQUIRE(d->tile_cfg >= 0 && d->tile_cfg < kNumCfgs, "conv2d: tile_cfg=%d invalid", d->tile_cfg);
    HRV_REQUIRE(d->w_packed != nullptr && ((uintptr_t)d->w_packed & 15) == 0, "conv2d: w_packed null/unaligned");
    const int bm = cfg_bm(d->tile_cfg), bn = cfg_bn(d->tile_cfg);
    p.wp = (const float*)d->w_packed;
    p.n_tiles = (d->Cout + bn - 1) / bn;
    p.CoutPad = p.n_tiles * bn;
    {
      const int64_t wb = (int64_t)p.KT * p.CoutPad * rb;
      p.w_bytes = (dma_ok && wb < (int64_t)0xFFFFFFF0) ? (unsigned)wb : 0u;
    }
    p.m_tiles = (p.M + bm - 1) / bm;
    p.splitk = 1;
    const char* ev = hrv::env("HRV_CONV_SPLITK");   // 0 disables, N forces (A/B measurements)
    int want = pick_splitk(p.m_tiles * p.n_tiles, p.KT, d->spade == nullptr);
    if (ev) want = atoi(ev) > 0 ? atoi(ev) : 1;
    if (d->spade) want = 1;
    if (want > p.KT) want = p.KT;
    if (want > 1) {
      const int64_t need = (int64_t)want * p.M * p.CoutPad * (int64_t)sizeof(float);
      if (d->workspace && d->workspace_bytes >= need && (((uintptr_t)d->workspace) & 15) == 0) {
        p.splitk = want;
        p.ws = (float*)d->workspace;
      }  // no/too small workspace: run unsplit (correct, slower)
    }
  } else {
    p.splitk = 1;
  }
  return HRV_OK;
}

// Product default variant; HRV_CONV_VARIANT (0..3) overrides it for A/B measurements
// (variants 2/3 = software-pipelined body exist for fp32 only).
constexpr int kDefaultVariant = 1;

template <int TM, int TN, int WM, int WN, int RB = 64>
static int launch_cfg(const ConvParams& p, hipStream_t st) {
  const int nblk = p.m_tiles * p.n_tiles * p.splitk;
  const char* ev = hrv::env("HRV_CONV_VARIANT");
  int var = ev ? atoi(ev) : kDefaultVariant;
  const int oesz = p.out_f32 ? 4 : 2, resz = p.res_f32 ? 4 : 2;
  const bool vec_ok = (p.Cout & 3) == 0 && ((p.out_cs | p.out_co) & 3) == 0 && (!p.res || ((p.res_cs | p.res_co) & 3) == 0) &&
                      (((uintptr_t)p.scale | (uintptr_t)p.shift) & 15) == 0 && (((uintptr_t)p.out) & (4 * oesz - 1)) == 0 &&
                      (((uintptr_t)p.res) & (4 * resz - 1)) == 0;
  if (p.epi == 1) {
    if (TN % 2 != 0) { set_error("conv2d/spade: tile_cfg must have an even TN (cfg 0, 4, 6 or 7)"); return HRV_ERR_ARG; }
    var |= 1;  // the SPADE epilogue lives in the swapped-operand layout
  } else if (!vec_ok) {
    var &= ~1;  // scalar epilogue for odd channel counts / unaligned slices
  }
  if (p.bf16) {
    if constexpr (RB == 128 && (32 * TN * WN * 8) % 256 == 0) {
      // 128-byte rows: LDS-DMA staging is the default for the 64-column tile (440-580 vs 350-370 TFLOP/s
      // register-staged); the 128x128 tile measures the same either way (~470-505: both are bound by
      // moving 32 KB per K-tile through the CU's texture path) and keeps the register pipeline.
      // HRV_CONV_GLDS=0/1 forces it off/on (A/B measurements: profiles/r01_conv_bench_bf16_glds.txt).
      const char* eg = hrv::env("HRV_CONV_GLDS");
      const bool want = eg ? atoi(eg) != 0 : (32 * TN * WN == 64 || TM * TN >= 8);
      const bool glds = want && p.w_bytes != 0;
      if (p.src_f32) {
        if (var & 1) hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, 9, true, RB>), dim3(nblk), dim3(64 * WM * WN), 0, st, p);
        else hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, 8, true, RB>), dim3(nblk), dim3(64 * WM * WN), 0, st, p);
      } else if (glds) {
        if (var & 1) hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, 5, true, RB>), dim3(nblk), dim3(64 * WM * WN), 0, st, p);
        else hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, 4, true, RB>), dim3(nblk), dim3(64 * WM * WN), 0, st, p);
      } else {
        if (var & 1) hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, 1, true, RB>), dim3(nblk), dim3(64 * WM * WN), 0, st, p);
        else hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, 0, true, RB>), dim3(nblk), dim3(64 * WM * WN), 0, st, p);
      }
    } else {
      if (var & 1) hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, 1, true, RB>), dim3(nblk), dim3(64 * WM * WN), 0, st, p);
      else hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, 0, true, RB>), dim3(nblk), dim3(64 * WM * WN), 0, st, p);
    }
  } else if constexpr (RB != 64) {
    set_error("conv2d: 128-byte K-tile rows exist on the bf16 engine only");
    return HRV_ERR_ARG;
  } else {
    switch (var) {
      case 0: hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, 0, false>), dim3(nblk), dim3(64 * WM * WN), 0, st, p); break;
      case 1: hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, 1, false>), dim3(nblk), dim3(64 * WM * WN), 0, st, p); break;
      case 2: hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, 2, false>), dim3(nblk), dim3(64 * WM * WN), 0, st, p); break;
      case 3: hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, 3, false>), dim3(nblk), dim3(64 * WM * WN), 0, st, p); break;
      default: set_error("conv2d: HRV_CONV_VARIANT=%d invalid", var); return HRV_ERR_ARG;
    }
  }
  int rc = check_launch("conv_mfma_kernel");
  if (rc || p.splitk <= 1) return rc;
  const size_t total = (size_t)p.M * ((p.Cout + 3) / 4);
  const size_t gsz = (total + 255) / 256;
  const dim3 g((unsigned)(gsz > 4096 ? 4096 : gsz));
  if (p.bf16) hipLaunchKernelGGL(splitk_reduce_kernel<true>, g, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(splitk_reduce_kernel<false>, g, dim3(256), 0, st, p);
  return check_launch("splitk_reduce_kernel");
}

// Eight-wave LDS-DMA tiles (cfg 12/13): bf16 storage only (sources, weights); no register-staged fallback is
// instantiated for them.
template <int TM, int TN, int WM, int WN, bool ST3, bool SPECW = false>
static int launch_cfg8w(const ConvParams& p, hipStream_t st) {
  const int nblk = p.m_tiles * p.n_tiles * p.splitk;
  const int oesz = p.out_f32 ? 4 : 2, resz = p.res_f32 ? 4 : 2;
  const bool vec_ok = (p.Cout & 3) == 0 && ((p.out_cs | p.out_co) & 3) == 0 && (!p.res || ((p.res_cs | p.res_co) & 3) == 0) &&
                      (((uintptr_t)p.scale | (uintptr_t)p.shift) & 15) == 0 && (((uintptr_t)p.out) & (4 * oesz - 1)) == 0 &&
                      (((uintptr_t)p.res) & (4 * resz - 1)) == 0;
  if (!p.bf16 || p.src_f32 || p.w_bytes == 0) {
    set_error("conv2d: tile_cfg 12/13 (eight-wave LDS-DMA) needs bf16-stored sources within the 32-bit buffer range");
    return HRV_ERR_ARG;
  }
  constexpr int V = 4 | (ST3 ? 16 : 0) | (SPECW ? 32 : 0);
  constexpr int NTB = 64 * WM * WN * (SPECW ? 2 : 1);
  if (vec_ok || p.epi == 1)   // swapped-operand vector epilogue (the SPADE epilogue lives there)
    hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, V | 1, true, 128>), dim3(nblk), dim3(NTB), 0, st, p);
  else                        // scalar epilogue for odd channel counts / unaligned slices
    hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, V, true, 128>), dim3(nblk), dim3(NTB), 0, st, p);
  int rc = check_launch("conv_mfma_kernel");
  if (rc || p.splitk <= 1) return rc;
  const size_t total = (size_t)p.M * ((p.Cout + 3) / 4);
  const size_t gsz = (total + 255) / 256;
  const dim3 g((unsigned)(gsz > 4096 ? 4096 : gsz));
  hipLaunchKernelGGL(splitk_reduce_kernel<true>, g, dim3(256), 0, st, p);
  return check_launch("splitk_reduce_kernel");
}

// cfg 16: patch mode (VAR bit 6).  3x3, stride 1, pad 1 ('same'), ONE bf16-stored source with C % 128 == 0 read at its
// own resolution, bf16 packed weights, no split-K, no strided scatter.
static bool patch_eligible(const ConvParams& p) {
  return p.bf16 && !p.src_f32 && p.nsrc == 1 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.pad_w == 1 &&
         p.Ho == p.H && p.Wo == p.W && p.src[0].up_shift == 0 && p.src[0].C % 128 == 0 && p.src[0].bytes != 0 &&
         p.w_bytes != 0 && p.out_step != 2;
}

// ``n0_base`` / ``n_cols_tiles``: cover only the column tiles [n0_base / BNP, + n_cols_tiles) (0: all of them)
template <int TMP, int TNP, int WMP, int WNP, int STV>
static int launch_patch(const ConvParams& p0, hipStream_t st, int n0_base = 0, int n_cols_tiles = 0) {
  constexpr int BNP = 32 * TNP * WNP;
  if (!patch_eligible(p0) || (n_cols_tiles == 0 && p0.CoutPad % BNP != 0) || n0_base + n_cols_tiles * BNP > p0.CoutPad) {
    set_error("conv2d: tile_cfg 16-18 (patch mode) needs a 3x3 stride-1 'same' convolution over one bf16-stored source with "
              "C %% 128 == 0");
    return HRV_ERR_ARG;
  }
  ConvParams p = p0;
  p.splitk = 1;
  constexpr int THP = TMP * WMP * 32 / 16;   // tile rows: BM / 16
  p.m_tiles = p.N * ((p.H + THP - 1) / THP) * ((p.W + 15) / 16);
  p.n_tiles = n_cols_tiles > 0 ? n_cols_tiles : p.CoutPad / BNP;
  p.n0_base = n0_base;
  const int nblk = p.m_tiles * p.n_tiles;
  const int oesz = p.out_f32 ? 4 : 2, resz = p.res_f32 ? 4 : 2;
  const bool vec_ok = (p.Cout & 3) == 0 && ((p.out_cs | p.out_co) & 3) == 0 && (!p.res || ((p.res_cs | p.res_co) & 3) == 0) &&
                      (((uintptr_t)p.scale | (uintptr_t)p.shift) & 15) == 0 && (((uintptr_t)p.out) & (4 * oesz - 1)) == 0 &&
                      (((uintptr_t)p.res) & (4 * resz - 1)) == 0;
  constexpr int V = 4 | STV | 64;
  // persistent grid: the resident slots (LDS-bound: 160 KB / block) of every CU.  HRV_PATCH_PERSIST=0: one tile per block
  const int n_cu = persistent_cus();
  p.tlog = diag_tlog(p.m_tiles);       // diag only (hrv_diag_set_tlog): per-tile phase timestamps
  constexpr int PATCH_LDS = ((TMP * WMP * 32 / 16 + 2) * 18 * 64 + ((STV & 16) ? 3 : 2) * BNP * 32) * 4;
  const int per_cu = PATCH_LDS <= 80 * 1024 ? 2 : 1;
  const char* ep = hrv::env("HRV_PATCH_PERSIST");
  int grid = (ep && ep[0] == '0') ? nblk : n_cu * per_cu;
  if (grid > nblk) grid = nblk;
  if (vec_ok || p.epi == 1)
    hipLaunchKernelGGL((conv_mfma_kernel<TMP, TNP, WMP, WNP, V | 1, true, 128>), dim3(grid), dim3(64 * WMP * WNP), 0, st, p);
  else
    hipLaunchKernelGGL((conv_mfma_kernel<TMP, TNP, WMP, WNP, V, true, 128>), dim3(grid), dim3(64 * WMP * WNP), 0, st, p);
  return check_launch("conv_mfma_kernel[patch]");
}

static int launch_any(int tile_cfg, const ConvParams& p, hipStream_t st) {
  switch (tile_cfg) {
    case 0: return launch_cfg<2, 2, 2, 2>(p, st);
    case 1: return launch_cfg<1, 3, 4, 1>(p, st);
    case 2: return launch_cfg<2, 3, 4, 1>(p, st);
    case 3: return launch_cfg<2, 1, 4, 1>(p, st);
    case 4: return launch_cfg<2, 2, 4, 1>(p, st);
    case 5: return launch_cfg<1, 1, 4, 1>(p, st);
    case 6: return launch_cfg<1, 2, 4, 1>(p, st);
    case 7: return launch_cfg<4, 2, 2, 2>(p, st);
    case 8: return launch_cfg<2, 2, 2, 2, 128>(p, st);
    case 9: return launch_cfg<1, 2, 4, 1, 128>(p, st);
    case 10: return launch_cfg<4, 2, 2, 2, 128>(p, st);
    case 11: return launch_cfg<2, 4, 2, 2, 128>(p, st);
    case 12: return launch_cfg8w<2, 2, 4, 2, false>(p, st);
    case 13: return launch_cfg8w<2, 2, 4, 2, true>(p, st);
    case 14: return launch_cfg8w<4, 2, 2, 2, true, true>(p, st);
    case 15: return launch_cfg8w<2, 2, 2, 2, true, true>(p, st);
    case 16: return launch_patch<2, 2, 4, 2, 16>(p, st);
    case 17: return launch_patch<2, 2, 2, 2, 0>(p, st);
    case 18: {
      // 64-column patch tile: THREE weight stages still fit two blocks per CU (46 KB patch + 3 x 8 KB): one tile's DMA stays
      // in flight across the barrier (counted vmcnt) instead of vmcnt(0) per K-tile.  HRV_CONV_PATCH_ST3=0: two stages (A/B)
      const char* e3 = hrv::env("HRV_CONV_PATCH_ST3");
      if (e3 && e3[0] == '0') return launch_patch<1, 2, 4, 1, 0>(p, st);
      // an odd number (>= 3) of 64-column tiles: the even part runs on the 128-column tile (the halo patch is loaded once
      // per 128 columns instead of once per 64), the last 64 columns here.  HRV_CONV_PATCH_SPLIT=0: 64-column tiles only
      const char* es = hrv::env("HRV_CONV_PATCH_SPLIT");
      const int c64 = p.CoutPad / 64;
      if (c64 >= 3 && !(es && es[0] == '0')) {
        const int rc = launch_patch<2, 2, 2, 2, 0>(p, st, 0, c64 / 2);
        if (rc) return rc;
        return launch_patch<1, 2, 4, 1, 16>(p, st, (c64 / 2) * 128, c64 - 2 * (c64 / 2));
      }
      return launch_patch<1, 2, 4, 1, 16>(p, st);
    }
    case 19: return launch_patchw(p, st);
  }
  set_error("conv2d: tile_cfg=%d invalid", tile_cfg);
  return HRV_ERR_ARG;
}

}  // namespace hrv

using namespace hrv;

extern "C" int hrv_conv2d_tile_bn(int32_t c) { return (c >= 0 && c < kNumCfgs) ? cfg_bn(c) : -1; }
extern "C" int hrv_conv2d_tile_bm(int32_t c) { return (c >= 0 && c < kNumCfgs) ? cfg_bm(c) : -1; }
// bytes per packed K-tile row of the bf16 engine for this tile (64: 32 k-values, 128: 64 k-values)
extern "C" int hrv_conv2d_tile_row_bytes(int32_t c) { return (c >= 0 && c < kNumCfgs) ? cfg_rb(c) : -1; }

extern "C" int hrv_conv2d_pick_tile(int64_t M, int32_t Cout) {
  // 1) output-channel tile: least padding waste, ties -> wider tile (more A reuse)
  const int bns[4] = {128, 96, 64, 32};
  int best_bn = 32;
  int64_t best_pad = INT64_MAX;
  for (int i = 0; i < 4; ++i) {
    const int64_t padded = (int64_t)((Cout + bns[i] - 1) / bns[i]) * bns[i];
    if (padded < best_pad) { best_pad = padded; best_bn = bns[i]; }
  }
  // 2) pixel tile: the 128-row tiles (3-6 resident waves per SIMD) beat the 256-row ones on every
  //    layer of the path (tools/conv_bench.py on MI355X: 110-122 vs 97-118 TFLOP/s): the fp32 MFMA is
  //    slow enough (64 cycles) that LDS traffic is irrelevant and occupancy hides the barrier.
  (void)M;
  switch (best_bn) {
    case 128: return 0;
    case 96: return 1;
    case 64: return 6;
    default: return 5;
  }
}

extern "C" int64_t hrv_conv2d_packed_elems(int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc, const int32_t* srcC,
                                           int32_t tile_cfg) {
  if (tile_cfg < 0 || tile_cfg >= kNumCfgs || cfg_rb(tile_cfg) != 64 || nsrc < 1 || nsrc > HRV_MAX_SRC || !srcC) return -1;
  constexpr int BK = HOST_BK;
  const int bn = cfg_bn(tile_cfg);
  const int64_t cpad = (int64_t)((Cout + bn - 1) / bn) * bn;
  int64_t chunks = 0;
  for (int i = 0; i < nsrc; ++i) chunks += (srcC[i] + BK - 1) / BK;
  return (int64_t)KH * KW * chunks * cpad * BK;
}

extern "C" int hrv_conv2d_pack_weight_f32(const float* w, int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc,
                                          const int32_t* srcC, const int32_t* srcC_real, int32_t tile_cfg,
                                          float* out) {
  HRV_REQUIRE(w && out && srcC && srcC_real, "pack_weight: null pointer");
  HRV_REQUIRE(tile_cfg >= 0 && tile_cfg < kNumCfgs && cfg_rb(tile_cfg) == 64, "pack_weight: bad tile_cfg %d", tile_cfg);
  HRV_REQUIRE(nsrc >= 1 && nsrc <= HRV_MAX_SRC, "pack_weight: bad nsrc");
  constexpr int BK = HOST_BK;
  const int bn = cfg_bn(tile_cfg);
  const int64_t cpad = (int64_t)((Cout + bn - 1) / bn) * bn;
  int cin_real = 0, chunks_total = 0;
  for (int i = 0; i < nsrc; ++i) {
    HRV_REQUIRE(srcC_real[i] > 0 && srcC_real[i] <= srcC[i] && srcC[i] % 4 == 0, "pack_weight: bad channel counts");
    cin_real += srcC_real[i];
    chunks_total += (srcC[i] + BK - 1) / BK;
  }
  const int64_t total = (int64_t)KH * KW * chunks_total * cpad * BK;
  memset(out, 0, sizeof(float) * total);
  for (int kh = 0; kh < KH; ++kh)
    for (int kw = 0; kw < KW; ++kw) {
      int chunk0 = 0, cbase = 0;
      for (int s = 0; s < nsrc; ++s) {
        const int chunks = (srcC[s] + BK - 1) / BK;
        for (int c = 0; c < srcC_real[s]; ++c) {
          const int64_t kt = (int64_t)(kh * KW + kw) * chunks_total + chunk0 + c / BK;
          float* dst = out + (kt * cpad) * BK + (c % BK);
          const float* srcw = w + ((int64_t)(cbase + c) * KH + kh) * KW + kw;
          const int64_t ostride = (int64_t)cin_real * KH * KW;
          for (int co = 0; co < Cout; ++co) dst[(int64_t)co * BK] = srcw[co * ostride];
        }
        chunk0 += chunks;
        cbase += srcC_real[s];
      }
    }
  return HRV_OK;
}

extern "C" int64_t hrv_conv2d_workspace_bytes(const hrv_conv2d_t* d) {
  if (!d || d->tile_cfg < 0 || d->tile_cfg >= kNumCfgs || d->spade) return 0;
  const int bm = cfg_bm(d->tile_cfg), bn = cfg_bn(d->tile_cfg);
  const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
  const int n_tiles = (d->Cout + bn - 1) / bn;
  const int m_tiles = (int)((M + bm - 1) / bm);
  // K-tile count as the fp32 engine sees it (the bf16 engine has at most as many -> its split factor
  // is never larger, so the workspace sized here always suffices)
  int chunks = 0;
  for (int i = 0; i < d->nsrc && i < HRV_MAX_SRC; ++i) chunks += (d->src[i].C + HOST_BK - 1) / HOST_BK;
  const int KT = d->KH * d->KW * chunks;
  const char* ev = hrv::env("HRV_CONV_SPLITK");
  int s = pick_splitk(m_tiles * n_tiles, KT, true);
  if (ev) s = atoi(ev) > 0 ? atoi(ev) : 1;
  if (s > KT) s = KT;
  return s > 1 ? (int64_t)s * M * n_tiles * bn * (int64_t)sizeof(float) : 0;
}

static int conv2d_one(const hrv_conv2d_t* d, hipStream_t stream, bool bf) {
  ConvParams p;
  int rc = fill_params(d, p, true, bf);
  if (rc) return rc;
  return launch_any(d->tile_cfg, p, stream);
}

// The gather addresses a source with 32-bit element offsets.  A batch whose largest source exceeds 2^32 elements
// (serving batches: 16 x 1024x768 x 384 channels) is issued as consecutive launches over sub-batches of whole
// images -- convolution rows never cross an image, so only the base pointers move.  Forward geometry only.
static int conv2d_any(const hrv_conv2d_t* d, hipStream_t stream, bool bf) {
  HRV_REQUIRE(d != nullptr, "conv2d: null descriptor");
  HRV_REQUIRE(d->nsrc >= 1 && d->nsrc <= HRV_MAX_SRC && d->N > 0 && d->H > 0 && d->W > 0, "conv2d: bad geometry");
  int64_t per_img = 0;
  for (int i = 0; i < d->nsrc; ++i) {
    const hrv_src_t& s = d->src[i];
    const int sr = s.up_shift > 0 ? s.up_shift : 0, sl = s.up_shift < 0 ? -s.up_shift : 0;
    const int64_t e = (int64_t)((d->H >> sr) << sl) * ((d->W >> sr) << sl) * s.cstride;
    per_img = e > per_img ? e : per_img;
  }
  const int64_t out_px = (int64_t)d->Ho * d->Wo;
  // HRV_CONV_MAX_BATCH caps the images per launch so that the tests can exercise the sub-batch path on small tensors.
  // The limit is on BYTES (the LDS-DMA staging addresses a source through a 32-bit buffer resource; the patch-mode
  // tiles have no other staging), which also keeps the element offsets of the register-staged gather in range.
  const bool srcf0 = bf && (d->mixed_flags & 8);
  const int64_t lim = (int64_t)0xFFFFFFF0 / ((bf && !srcf0) ? 2 : 4);
  int64_t cap = d->N;
  if (const char* e = hrv::env("HRV_CONV_MAX_BATCH")) {
    const long long v = atoll(e);
    if (v > 0 && v < cap) cap = v;
  }
  const bool fits = cap == d->N && d->N * per_img < lim && d->N * out_px < ((int64_t)1 << 31);
  const bool plain_fwd = d->out_step <= 1 && d->free_extent == 0 && d->res_mode == 0;
  if (fits || !plain_fwd || per_img <= 0 || out_px <= 0) return conv2d_one(d, stream, bf);
  int64_t nb = (lim - 1) / per_img;
  const int64_t nb2 = (((int64_t)1 << 31) - 1) / out_px;
  nb = nb < nb2 ? nb : nb2;
  nb = nb < cap ? nb : cap;
  HRV_REQUIRE(nb >= 1, "conv2d: one image alone exceeds the 32-bit gather range");
  const bool srcf = bf && (d->mixed_flags & 8);
  const size_t es_src = bf && !srcf ? 2 : 4, es_out = bf && !(d->mixed_flags & 1) ? 2 : 4;
  const size_t es_res = bf && !(d->mixed_flags & 2) ? 2 : 4, es_x = bf && !(d->mixed_flags & 4) ? 2 : 4;
  const int64_t out_img_px = ((int64_t)d->Ho << d->out_up_shift) * ((int64_t)d->Wo << d->out_up_shift);
  for (int64_t n0 = 0; n0 < d->N; n0 += nb) {
    hrv_conv2d_t c = *d;
    hrv_spade_epi_t e;
    c.N = (int32_t)((d->N - n0) < nb ? (d->N - n0) : nb);
    for (int i = 0; i < d->nsrc; ++i) {
      const hrv_src_t& s = d->src[i];
      const int sr = s.up_shift > 0 ? s.up_shift : 0, sl = s.up_shift < 0 ? -s.up_shift : 0;
      const int64_t img = (int64_t)((d->H >> sr) << sl) * ((d->W >> sr) << sl) * s.cstride;
      c.src[i].ptr = (const char*)s.ptr + (size_t)(n0 * img) * es_src;
    }
    c.out = (char*)d->out + (size_t)(n0 * out_img_px * d->out_cstride) * es_out;
    if (d->residual) c.residual = (const char*)d->residual + (size_t)(n0 * out_px * d->res_cstride) * es_res;
    if (d->spade) {
      e = *d->spade;
      e.x = (const float*)((const char*)e.x + (size_t)(n0 * out_px * e.x_cstride) * es_x);
      e.mean += n0 * e.C;
      e.rstd += n0 * e.C;
      if (e.noise_z) e.noise_z += n0 * out_px;
      if (e.g1p_out) e.g1p_out += n0 * out_px * e.C;
      c.spade = &e;
    }
    const int rc = conv2d_one(&c, stream, bf);
    if (rc) return rc;
  }
  return HRV_OK;
}

extern "C" int hrv_conv2d_nhwc_f32(const hrv_conv2d_t* d, hrv_stream_t stream) {
  return conv2d_any(d, (hipStream_t)stream, false);
}

extern "C" int hrv_conv2d_nhwc_bf16(const hrv_conv2d_t* d, hrv_stream_t stream) {
  return conv2d_any(d, (hipStream_t)stream, true);
}

static inline unsigned short host_f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

extern "C" int64_t hrv_conv2d_packed_elems_bf16(int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc, const int32_t* srcC,
                                                int32_t tile_cfg) {
  if (tile_cfg < 0 || tile_cfg >= kNumCfgs || nsrc < 1 || nsrc > HRV_MAX_SRC || !srcC) return -1;
  const int bn = cfg_bn(tile_cfg);
  const int64_t cpad = (int64_t)((Cout + bn - 1) / bn) * bn;
  int64_t chunks = 0;
  const int bke = cfg_rb(tile_cfg) / 2;
  for (int i = 0; i < nsrc; ++i) chunks += (srcC[i] + bke - 1) / bke;
  return (int64_t)KH * KW * chunks * cpad * bke;
}

// HOST packer for the bf16 engine: [kt][CoutPad][bke] bf16 (uint16), kt = (tap, source, bke-channel chunk),
// bke = 32 (64-byte K-tile rows) or 64 (tile_cfg 8/9: 128-byte rows)
extern "C" int hrv_conv2d_pack_weight_bf16(const float* w, int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc,
                                           const int32_t* srcC, const int32_t* srcC_real, int32_t tile_cfg,
                                           uint16_t* out) {
  HRV_REQUIRE(w && out && srcC && srcC_real, "pack_weight_bf16: null pointer");
  HRV_REQUIRE(tile_cfg >= 0 && tile_cfg < kNumCfgs, "pack_weight_bf16: bad tile_cfg %d", tile_cfg);
  HRV_REQUIRE(nsrc >= 1 && nsrc <= HRV_MAX_SRC, "pack_weight_bf16: bad nsrc");
  const int bn = cfg_bn(tile_cfg);
  const int bke = cfg_rb(tile_cfg) / 2;  // k-values per K-tile row
  const int64_t cpad = (int64_t)((Cout + bn - 1) / bn) * bn;
  int cin_real = 0, chunks_total = 0;
  for (int i = 0; i < nsrc; ++i) {
    // multiples of 8 for bf16 tensors; multiples of 4 when the sources are fp32 tensors rounded in the gather
    HRV_REQUIRE(srcC_real[i] > 0 && srcC_real[i] <= srcC[i] && srcC[i] % 4 == 0, "pack_weight_bf16: bad channel counts");
    cin_real += srcC_real[i];
    chunks_total += (srcC[i] + bke - 1) / bke;
  }
  const int64_t total = (int64_t)KH * KW * chunks_total * cpad * bke;
  memset(out, 0, sizeof(uint16_t) * total);
  const int64_t ostride = (int64_t)cin_real * KH * KW;
  for (int kh = 0; kh < KH; ++kh)
    for (int kw = 0; kw < KW; ++kw) {
      int chunk0 = 0, cbase = 0;
      for (int s = 0; s < nsrc; ++s) {
        for (int c = 0; c < srcC_real[s]; ++c) {
          const int64_t kt = (int64_t)(kh * KW + kw) * chunks_total + chunk0 + c / bke;
          uint16_t* dst = out + (kt * cpad) * bke + (c % bke);
          const float* srcw = w + ((int64_t)(cbase + c) * KH + kh) * KW + kw;
          for (int co = 0; co < Cout; ++co) dst[(int64_t)co * bke] = host_f2bf(srcw[co * ostride]);
        }
        chunk0 += (srcC[s] + bke - 1) / bke;
        cbase += srcC_real[s];
      }
    }
  return HRV_OK;
}

extern "C" int hrv_conv2d_naive_nhwc_f32(const hrv_conv2d_t* d, hrv_stream_t stream) {
  ConvParams p;
  int rc = fill_params(d, p, false);
  if (rc) return rc;
  HRV_REQUIRE(d->w_oihw != nullptr, "conv2d_naive: w_oihw null");
  int real_c[HRV_MAX_SRC];
  int cin = 0;
  for (int i = 0; i < d->nsrc; ++i) {
    real_c[i] = d->src[i].C_real > 0 ? d->src[i].C_real : d->src[i].C;
    cin += real_c[i];
  }
  hipStream_t st = (hipStream_t)stream;
  int* d_real = nullptr;
  if (hipMalloc(&d_real, sizeof(real_c)) != hipSuccess) { set_error("conv2d_naive: hipMalloc"); return HRV_ERR_LAUNCH; }
  (void)hipMemcpyAsync(d_real, real_c, sizeof(real_c), hipMemcpyHostToDevice, st);
  const size_t total = (size_t)p.M * p.Cout;
  const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  hipLaunchKernelGGL(conv_f32_naive_kernel, dim3(blocks), dim3(256), 0, st, p, d->w_oihw, cin, d_real);
  rc = check_launch("conv_f32_naive_kernel");
  (void)hipStreamSynchronize(st);
  (void)hipFree(d_real);
  return rc;
}
