// Wide patch-mode convolution (tile_cfg 19): 3x3 stride-1 'same' convolution over ONE bf16-stored source with
// C % 128 == 0 -- the SPADE gamma|beta convolutions (network_generator.py:117-121: actv 128 -> 2C) with their fused
// modulate epilogue, and any other convolution of that geometry.
//
// What limits the 8x16-pixel x 128-column patch tiles of conv_f32.hip (cfg 17/18, ~31 % MFMA-busy,
// profiles/r01_pmc_patch_conv.txt) is the weight stream: every block re-streams its 128 columns x K of weights from
// L2 for only 128 pixels, waits vmcnt(0) + barrier per 64-k tile, and column counts like 160 need two column tiles
// (the halo patch is loaded twice).  Here
//   * a block owns a 16x16-pixel tile x ALL of its (up to 192) columns: the 18x18x128-channel halo patch (83 KB) is
//     DMA'd into LDS once, the weights stream once per 256 pixels (half the bytes per FLOP) in 64-k tiles of
//     (32 TN) x 128 bytes through THREE LDS stages with a counted vmcnt across a fence-less s_barrier;
//   * 4 waves, one per SIMD, each 64 pixels (4 tile rows) x 32 TN columns: 2 x TN accumulator tiles (192 registers at
//     TN = 6); per 16-k step a wave reads 2 A + TN B fragments for 2 TN MFMAs (0.67 KB of LDS per MFMA);
//   * one wave per SIMD has nobody to hide its latencies, so the fragment reads are inline-asm ds_read_b128 issued a
//     k-step ahead in two halves around the MFMAs, with hand-counted lgkmcnt (same scheme as wgrad_tr.hip);
//   * swapped-operand MFMA (D[cout][pixel]): a lane holds 4 consecutive output channels of one pixel, so the SPADE
//     epilogue (gamma | beta column pairs) and the vector epilogue are those of conv_f32.hip.
// LDS: 82,944 (patch) + 3 x 24,576 (weights, TN = 6) = 156,672 bytes -> one block per CU.
#include <utility>

#include "conv_params.h"

namespace hrv {

// compile-time loop: the epilogues index the accumulator tiles with constants (a `#pragma unroll` loop over TN = 6
// column tiles exceeds the pragma-unroll size limit, stays a loop, and the accumulators end up in scratch memory)
template <typename F, int... Is>
__device__ __forceinline__ void pw_static_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

#if defined(__HIP_DEVICE_COMPILE__)
#define PW_READ128(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#else
#define PW_READ128(DST, ADDR, OFF) DST = f32x4{0.f, 0.f, 0.f, 0.f}
#endif

template <int TN>
__global__ __launch_bounds__(256) void conv_patchw_kernel(const ConvParams p, const int n0) {
  constexpr int PW = 18, PPIX = PW * PW;          // halo patch, pixels
  constexpr int PATCH_B = PPIX * 256;             // bytes: 128 bf16 channels per pixel
  constexpr int BN = 32 * TN;
  constexpr int BSTAGE = BN * 128;                // one 64-k weight tile: BN rows x 128 bytes
  constexpr int NS = 3;
  constexpr int NB = TN;                          // weight DMA instructions per wave per K-tile (BN * 8 slots / 256 lanes)
  constexpr int WAIT_RUN = (NB & 15) | (7 << 4) | (0 << 8) | ((NB >> 4) << 14);   // vmcnt(NB) lgkmcnt(0)
  constexpr int WAIT_ALL = 0 | (7 << 4) | (0 << 8);                                // vmcnt(0) lgkmcnt(0)
  constexpr int NR = 2 + TN;                      // 16-byte fragment reads per k-step: 2 A (pixels) + TN B (columns)
  constexpr int NH1 = NR / 2, NH2 = NR - NH1;
  constexpr int WAIT_H1 = 0x3F | (7 << 4) | (NH1 << 8) | (3 << 14);                // lgkmcnt(NH1), vmcnt untouched
  constexpr int WAIT_L0 = 0x3F | (7 << 4) | (0 << 8) | (3 << 14);                  // lgkmcnt(0)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[PATCH_B + NS * BSTAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;

  // block -> (image, tile row, tile column)
  const int mt = xcd_remap(blockIdx.x, p.m_tiles);
  const int tx = (p.W + 15) >> 4, ty = (p.H + 15) >> 4;
  const int pt_n = mt / (tx * ty);
  const int rr = mt - pt_n * (tx * ty);
  const int pt_y0 = (rr / tx) << 4, pt_x0 = (rr % tx) << 4;
  auto row2pix = [&](int row) -> int {
    const int y = pt_y0 + (row >> 4), x = pt_x0 + (row & 15);
    return (y < p.H && x < p.W) ? (pt_n * p.H + y) * p.W + x : p.M;
  };

  const SrcDev& S = p.src[0];
  const rsrc_t a_rsrc = make_rsrc(S.ptr, S.bytes);
  const rsrc_t w_rsrc = make_rsrc(p.wp, p.w_bytes);
  const int c64 = S.C >> 6, nchunk = S.C >> 7;
  const int KTOT = 18 * nchunk;                   // weight tiles in patch order: chunk, tap, half

  // weight DMA: instruction j of this wave covers LDS rows 32 j + 8 wave .. + 7 (8 slots of 16 bytes per row); the
  // 16-byte groups of a row are XOR-swizzled by (row >> 1) & 7 on the SOURCE side (conflict-free b128 fragment reads)
  unsigned b_voff[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int idx = tid + 256 * j;
    const int row = idx >> 3, slot = idx & 7;
    b_voff[j] = (unsigned)(row * 128 + ((slot ^ ((row >> 1) & 7)) * 16));
  }
  unsigned char* const patch = smem;
  unsigned char* const bst = smem + PATCH_B;

  auto patch_dma = [&](int chunk) {
    // one instruction = 4 consecutive halo pixels x 16 groups of 8 channels, group slots XOR-swizzled by (hx & 15)
    for (int t = wave; t < PPIX / 4; t += 4) {
      const int P = 4 * t + (lane >> 4), s16 = lane & 15;
      const int hy = P / PW, hx = P - hy * PW;
      const int y = pt_y0 - 1 + hy, x = pt_x0 - 1 + hx;
      const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const int g = s16 ^ (hx & 15);
      const unsigned off = ((unsigned)((pt_n * p.H + y) * p.W + x) * (unsigned)S.cstride + (unsigned)(S.coff + chunk * 128 + g * 8)) * 2u;
      dma16(a_rsrc, reinterpret_cast<float*>(patch + t * 1024), ok ? off : 0xFFFFFFF0u, 0u);
    }
  };
  auto b_dma = [&](int q, int buf) {
    const int rem = q % 18, kt = (rem >> 1) * c64 + (q / 18) * 2 + (rem & 1);
    const unsigned w_soff = (unsigned)((kt * p.CoutPad + n0) * 128);
    unsigned char* Bbuf = bst + buf * BSTAGE;
#pragma unroll
    for (int j = 0; j < NB; ++j) dma16(w_rsrc, reinterpret_cast<float*>(Bbuf + (256 * j + 64 * wave) * 16), b_voff[j], w_soff);
  };

  // ---- fragment addresses (LDS byte addresses)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // A: GEMM row r = 64 wave + 32 i + l31 -> tile pixel (4 wave + 2 i + (l31 >> 4), l31 & 15); tap (kh, kw) and i are
  // immediates; the channel-group slot ((8 half + 2 ks + lh) ^ hx) with hx = (l31 & 15) + kw needs one XOR per step
  const unsigned a_lb = lds0 + (unsigned)(((wave * 4 + (l31 >> 4)) * PW + (l31 & 15)) * 256);
  unsigned a_hx[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) a_hx[kw] = (unsigned)(((((l31 & 15) + kw) & 15) ^ lh) << 4);
  // B: row nt*32 + l31, slot ((2 ks + lh) ^ ((l31 >> 1) & 7))
  unsigned b_k[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) b_k[ks] = lds0 + (unsigned)(PATCH_B + l31 * 128 + (((2 * ks + lh) ^ ((l31 >> 1) & 7)) << 4));

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 fr[2][NR];                                 // [set][read]: 0, 1 = A tiles; 2 + j = B tile j
  // reads [R0, R1) of k-step KS (0..3) of weight tile Q (tap / half decoded at compile time where Q is) from stage SB
#define PW_READS(SET, R0, R1, TAP, HALF, KS, SBOFF)                                                        \
  {                                                                                                        \
    const int kh_ = (TAP) / 3, kw_ = (TAP) - 3 * ((TAP) / 3);                                              \
    const unsigned aa = a_lb + (a_hx[kw_] ^ (unsigned)((2 * (KS) + 8 * (HALF)) << 4));                     \
    const unsigned bb = b_k[KS] + (SBOFF);                                                                 \
    _Pragma("unroll") for (int r = (R0); r < (R1); ++r) {                                                  \
      if (r < 2) {                                                                                         \
        PW_READ128(fr[SET][r], aa, (kh_ * PW + kw_) * 256 + r * (2 * PW * 256));                           \
      } else {                                                                                             \
        PW_READ128(fr[SET][r], bb, (r - 2) * 4096);                                                        \
      }                                                                                                    \
    }                                                                                                      \
  }
#define PW_MMAS(SET, M0, M1)                                                                               \
  {                                                                                                        \
    _Pragma("unroll") for (int m = (M0); m < (M1); ++m) {                                                  \
      const int i = m / TN, j = m - i * TN;                                                                \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[SET][2 + j]),      \
                                                          __builtin_bit_cast(bf16x8, fr[SET][i]), acc[i][j], 0, 0, 0); \
    }                                                                                                      \
  }

  // ---- prologue: patch chunk 0 + weight tiles 0, 1
  patch_dma(0);
  b_dma(0, 0);
  if (KTOT > 1) {
    b_dma(1, 1);
    __builtin_amdgcn_s_waitcnt(WAIT_RUN);
  } else {
    __builtin_amdgcn_s_waitcnt(WAIT_ALL);
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  int rb = 0, wb = NS - 1;
  for (int ch = 0; ch < nchunk; ++ch) {
    {
      const unsigned sb0 = (unsigned)(rb * BSTAGE);
      PW_READS(0, 0, NR, 0, 0, 0, sb0)               // first k-step of the chunk (tap 0, half 0)
    }
#pragma unroll
    for (int qq = 0; qq < 18; ++qq) {                 // (tap, half) of this chunk: compile-time unrolled
      const int q = ch * 18 + qq;
      const bool more = q + NS - 1 < KTOT;
      if (more) b_dma(q + NS - 1, wb);               // that stage was read in tile q-1: every wave passed the barrier
      const unsigned sb = (unsigned)(rb * BSTAGE);
      const int nb = rb == NS - 1 ? 0 : rb + 1;
      const unsigned sbn = (unsigned)(nb * BSTAGE);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks < 3) {
          PW_READS(nxt, 0, NH1, qq >> 1, qq & 1, ks + 1, sb)
          __builtin_amdgcn_s_waitcnt(WAIT_H1);
        } else if (qq < 17) {
          // last k-step of this weight tile: its LDS reads are all issued; tile q+1 must have landed
          if (more) __builtin_amdgcn_s_waitcnt(WAIT_RUN);
          else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          PW_READS(nxt, 0, NH1, (qq + 1) >> 1, (qq + 1) & 1, 0, sbn)
          __builtin_amdgcn_s_waitcnt(WAIT_H1);
        } else {
          __builtin_amdgcn_s_waitcnt(WAIT_L0);       // end of the chunk: nothing to prefetch across the patch reload
        }
        __builtin_amdgcn_sched_barrier(0);
        PW_MMAS(cur, 0, TN)
        __builtin_amdgcn_sched_barrier(0);
        if (ks < 3) {
          PW_READS(nxt, NH1, NR, qq >> 1, qq & 1, ks + 1, sb)
        } else if (qq < 17) {
          PW_READS(nxt, NH1, NR, (qq + 1) >> 1, (qq + 1) & 1, 0, sbn)
        }
        __builtin_amdgcn_sched_barrier(0);
        PW_MMAS(cur, TN, 2 * TN)
        __builtin_amdgcn_sched_barrier(0);
      }
      rb = nb;
      wb = wb == NS - 1 ? 0 : wb + 1;
    }
    if (ch + 1 < nchunk) {
      // next 128-channel chunk: every wave is done with the patch; the in-flight weight tiles belong to that chunk
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      patch_dma(ch + 1);
      __builtin_amdgcn_s_waitcnt(WAIT_ALL);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
#undef PW_READS
#undef PW_MMAS

  // ---- epilogue.  D layout (swapped operands): col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (cout):
  // regs 4g..4g+3 are 4 consecutive output channels of this lane's pixel.
  if (p.epi == 1) {
    // SPADE: column tiles come in (gamma | beta) pairs of the same 32 channels
    static_assert(TN % 2 == 0, "gamma | beta pairs");
    const int HWo = p.Ho * p.Wo;
    pw_static_for(std::make_integer_sequence<int, TN / 2>{}, [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const int col0 = n0 + 2 * q * 32;              // first gamma column of the pair
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
        for (int i = 0; i < 2; ++i) {
          const int pidx = row2pix(wave * 64 + i * 32 + l31);
          if (c_ok && pidx < p.M) {
            const int n = pidx / HWo;
            f32x4 x = ld4rt<true>(p.sx, (size_t)pidx * p.sx_cs + p.sx_co + c0, p.sx_f32);
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
            st4rt<true>(p.out, (size_t)pidx * p.out_cs + p.out_co + c0, v, p.out_f32);
          }
        }
      }
    });
    return;
  }
  pw_static_for(std::make_integer_sequence<int, TN>{}, [&](auto jc) {
    constexpr int j = decltype(jc)::value;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c0 = n0 + j * 32 + 8 * g + 4 * lh;
      const bool c_ok = c0 < p.Cout;
      const int cs = c_ok ? c0 : 0;
      const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + cs) : (f32x4)(1.f);
      const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + cs) : (f32x4)(0.f);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int pidx = row2pix(wave * 64 + i * 32 + l31);
        if (c_ok && pidx < p.M) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] * sc[e] + sh[e];
          if (p.res) {
            const f32x4 r4 = ld4rt<true>(p.res, (size_t)pidx * p.res_cs + p.res_co + c0, p.res_f32);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = res_combine(v[e], r4[e], p.res_mode, p.slope);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act, p.slope);
          st4rt<true>(p.out, (size_t)pidx * p.out_cs + p.out_co + c0, v, p.out_f32);
        }
      }
    }
  });
}

// tile_cfg 19.  The packed weight has CoutPad = multiple of 64 columns ([kt][CoutPad][64 bf16], bn = 64); the columns
// are covered by launches of 192 / 128 / 64 columns (a launch = every pixel tile x that column range; the halo patch
// is reloaded per launch, so ranges are as wide as the accumulators allow).
int launch_patchw(const ConvParams& p0, hipStream_t st) {
  const ConvParams& q = p0;
  const bool ok = q.bf16 && !q.src_f32 && q.nsrc == 1 && q.KH == 3 && q.KW == 3 && q.stride == 1 && q.pad == 1 && q.pad_w == 1 &&
                  q.Ho == q.H && q.Wo == q.W && q.src[0].up_shift == 0 && q.src[0].C % 128 == 0 && q.src[0].bytes != 0 &&
                  q.w_bytes != 0 && q.out_step != 2 && q.out_up == 0 && q.CoutPad % 64 == 0;
  if (!ok) {
    set_error("conv2d: tile_cfg 19 (wide patch mode) needs a 3x3 stride-1 'same' convolution over one bf16-stored source with "
              "C %% 128 == 0 and a weight packed for 64-column tiles");
    return HRV_ERR_ARG;
  }
  const int oesz = q.out_f32 ? 4 : 2, resz = q.res_f32 ? 4 : 2;
  const bool vec_ok = (q.Cout & 3) == 0 && ((q.out_cs | q.out_co) & 3) == 0 && (!q.res || ((q.res_cs | q.res_co) & 3) == 0) &&
                      (((uintptr_t)q.scale | (uintptr_t)q.shift) & 15) == 0 && (((uintptr_t)q.out) & (4 * oesz - 1)) == 0 &&
                      (((uintptr_t)q.res) & (4 * resz - 1)) == 0;
  if (!vec_ok && q.epi != 1) {
    set_error("conv2d: tile_cfg 19 needs 4-channel-aligned outputs (vector epilogue)");
    return HRV_ERR_ARG;
  }
  ConvParams p = p0;
  p.splitk = 1;
  p.m_tiles = p.N * ((p.H + 15) / 16) * ((p.W + 15) / 16);
  for (int n0 = 0; n0 < p.CoutPad;) {
    const int left = p.CoutPad - n0;
    if (left >= 192 && left != 256) {       // 256 = 128 + 128 (two equal launches) rather than 192 + 64
      hipLaunchKernelGGL((conv_patchw_kernel<6>), dim3(p.m_tiles), dim3(256), 0, st, p, n0);
      n0 += 192;
    } else if (left >= 128) {
      hipLaunchKernelGGL((conv_patchw_kernel<4>), dim3(p.m_tiles), dim3(256), 0, st, p, n0);
      n0 += 128;
    } else {
      hipLaunchKernelGGL((conv_patchw_kernel<2>), dim3(p.m_tiles), dim3(256), 0, st, p, n0);
      n0 += 64;
    }
    const int rc = check_launch("conv_patchw_kernel");
    if (rc) return rc;
  }
  return HRV_OK;
}

}  // namespace hrv
