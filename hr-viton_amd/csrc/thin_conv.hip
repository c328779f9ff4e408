// Thin convolutions of the 1024x768 level (mixed-precision training): 3x3 / 1x1 stride-1 'same' convolutions with at most 96
// channels on either side over a bf16-STORED source -- SPADEResBlock conv_0 / conv_1 / conv_s of up_4 and conv_img
// (network_generator.py:141-143,201), forward and data gradient.
//
// These layers are HBM-bound (conv_0 of up_4: 288 bytes per pixel for 46 kFLOP), and the implicit-GEMM engine serves them
// badly: its gather re-reads every pixel from L2 once per tap, the narrowest tile pads 32 columns to 64 and K to multiples
// of 64, and every 64-k tile costs a barrier (measured 100-190 TFLOP/s = 1.2-1.4 TB/s, 4-6x off the HBM roofline).  Here
//   * the WHOLE weight (<= 56 KB as bf16) is converted from the fp32 parameter (x 1/sigma for spectral norm) into LDS once
//     per persistent block, in MFMA-fragment order -- no packing launch, no weight streaming, no per-K-tile barrier;
//   * a tile is 8x16 output pixels; its halo patch [pixel][channel] travels global -> LDS by LDS-DMA (pad channels and
//     out-of-image pixels arrive as zeros via out-of-range offsets); pixel rows are padded to an ODD number of 16-byte
//     slots, which makes the b128 fragment reads of 16 horizontally adjacent pixels bank-disjoint;
//   * each wave owns 32 pixels x all (<= 96) columns: per 16-k step one A fragment + TN weight fragments, TN MFMAs
//     (swapped operands: a lane ends up with 4 consecutive output channels of one pixel -> 16-byte epilogue accesses);
//   * the next tile's patch DMA is issued before the current tile's epilogue, and two blocks are resident per CU.
// Epilogue: out = act((acc + shift[c]) (+ residual | * act'(residual))), fp32 or bf16 output.
#include <stdlib.h>

#include "conv_params.h"

namespace hrv {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float thin_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 thin_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned thin_pk_bf16(float a, float b) {     // v_cvt_pk_bf16_f32 (round to nearest even)
  const thin_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, thin_bf16x2));
}

struct ThinParams {
  const void* src; int N, H, W, CK, cs, co; unsigned src_bytes;     // bf16 source, CK channels (multiple of 8)
  const float* w; int Cout, Cin, KH, KW; const float* sigma; float wscale; int transposed;   // fp32 OIHW parameter
  int NC;                                   // output columns (forward: Cout, data gradient: Cin)
  const float* shift;
  const void* res; int res_cs, res_co, res_f32, res_mode;
  int act; float slope;
  void* out; int out_cs, out_co, out_f32;
  int pad, tiles, tx, ty;
  int full;                                 // every tile lies inside the image (H % 8 == 0, W % 16 == 0)
};

// WIDE (SPADE's conv_shared of a whole block as ONE 1x1 convolution over the tap-expanded label map, 72 -> 3 x 128
// columns over every pixel of the 1024x768 / 512x384 levels: 768 bytes written per pixel for 55 kFLOP, HBM-write-bound):
//   * all 32 * TN columns are real and there is no residual (host contract); the bias sits in LDS;
//   * lanes l and l + 32 exchange 4-channel groups (v_permlane32_swap) so that a lane stores 8 consecutive channels,
//     16 bytes, per instruction;
//   * with full tiles every wave issues exactly 2 * TN stores after the next patch's DMA: the loop waits for the DMA
//     with a counted vmcnt and lets the stores drain under the next tile's MFMAs (one block per CU: 61 KB of weights).
template <int TN, int KB, int KS, bool WIDE = false>
__global__ __launch_bounds__(256) void thin_conv_kernel(const ThinParams p) {
  constexpr int KH = KS == 9 ? 3 : 1, KW = KH;
  constexpr int PWD = 16 + KW - 1, PHT = 8 + KH - 1, PPIX = PWD * PHT;
  constexpr int SL = (2 * KB) | 1;                  // 16-byte slots per patch pixel: >= 2 KB (= KB*16 channels), odd
  constexpr int PS = SL * 16;                       // patch pixel stride, bytes
  constexpr int NP = 32 * TN;
  constexpr int WBYTES = KS * KB * NP * 32;         // weights: [tap][kb][n][16 k] bf16
  constexpr int NI = (PPIX * SL + 63) / 64;         // patch DMA instructions per tile
  constexpr int NQ = (NI + 3) / 4;                  // per wave
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const wl = smem;
  unsigned char* const patch = smem + ((WBYTES + 1023) & ~1023);
  float* const bias_l = reinterpret_cast<float*>(patch + NI * 1024);      // WIDE only: [32 * TN]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- weights -> LDS (bf16, fragment order; the 16-byte halves of a row are swapped for rows with bit 3 set so that the
  // b128 reads of 16 consecutive columns cover all 64 banks)
  {
    const float inv = p.wscale / (p.sigma ? p.sigma[0] : 1.f);
    for (int idx = tid; idx < KS * KB * NP * 16; idx += 256) {
      const int kk = idx & 15;
      int t = idx >> 4;
      const int n = t % NP; t /= NP;
      const int kb = t % KB, tap = t / KB;
      const int k = kb * 16 + kk;
      float v = 0.f;
      if (k < p.CK && n < p.NC) {
        const int kh = tap / KW, kw = tap - kh * KW;
        v = p.transposed ? p.w[(((size_t)k * p.Cin + n) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)]
                         : p.w[(((size_t)n * p.Cin + k) * KH + kh) * KW + kw];
        v *= inv;
      }
      const int half = (kk >> 3) ^ ((n >> 3) & 1);
      *reinterpret_cast<unsigned short*>(wl + ((tap * KB + kb) * NP + n) * 32 + half * 16 + (kk & 7) * 2) = f2bf(v);
    }
  }

  if constexpr (WIDE)
    for (int c = tid; c < NP; c += 256) bias_l[c] = p.shift ? p.shift[c] : 0.f;

  // ---- patch DMA lane constants: slot -> (patch pixel, 16-byte channel group)
  const rsrc_t rs = make_rsrc(p.src, p.src_bytes);
  int d_py[NQ], d_px[NQ];
  unsigned d_off[NQ];
  bool d_ok[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    int j = wave + 4 * q;
    j = j < NI ? j : NI - 1;
    const int slot = 64 * j + lane;
    const int pix = slot / SL, s = slot - pix * SL;
    const int py = pix / PWD, px = pix - py * PWD;
    d_ok[q] = pix < PPIX && 8 * s < p.CK;
    d_py[q] = py; d_px[q] = px;
    d_off[q] = (unsigned)(((py * p.W + px) * p.cs + 8 * s) * 2);
  }
  auto tile_origin = [&](int t, int& n, int& y0, int& x0) {
    n = t / (p.tx * p.ty);
    const int r = t - n * (p.tx * p.ty);
    y0 = (r / p.tx) * 8; x0 = (r % p.tx) * 16;
  };
  auto issue = [&](int t) {
    int n, y0, x0;
    tile_origin(t, n, y0, x0);
    const unsigned base = (unsigned)((((n * p.H + y0 - p.pad) * p.W + x0 - p.pad) * p.cs + p.co) * 2);   // may wrap: masked lanes only
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      int j = wave + 4 * q;
      j = j < NI ? j : NI - 1;
      const int y = y0 - p.pad + d_py[q], x = x0 - p.pad + d_px[q];
      const bool ok = d_ok[q] && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      dma16(rs, reinterpret_cast<float*>(patch + j * 1024), ok ? base + d_off[q] : 0xFFFFFFF0u, 0u);
    }
  };

  // fragment addresses
  const unsigned char* const a_lane = patch + ((2 * wave + (l31 >> 4)) * PWD + (l31 & 15)) * PS + lh * 16;
  const unsigned char* const b_lane = wl + l31 * 32 + (lh ^ ((l31 >> 3) & 1)) * 16;

  int t = blockIdx.x;
  if (t < p.tiles) issue(t);
  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0 << 8));     // vmcnt(0) lgkmcnt(0): this wave's DMA and weight stores
  __syncthreads();
  for (; t < p.tiles; t += gridDim.x) {
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int tap = 0; tap < KS; ++tap) {
      const int kh = tap / KW, kw = tap - kh * KW;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(a_lane + (kh * PWD + kw) * PS + kb * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(b_lane + ((tap * KB + kb) * NP + j * 32) * 32);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), acc[j], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                  // every wave is done with the patch
    const int tn = t + gridDim.x;
    if (tn < p.tiles) issue(tn);                      // the next tile's patch travels while this tile's results are written

    // ---- epilogue.  D (swapped operands): col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (column)
    {
      int n, y0, x0;
      tile_origin(t, n, y0, x0);
      const int y = y0 + 2 * wave + (l31 >> 4), x = x0 + (l31 & 15);
      if constexpr (WIDE) {
        if (y < p.H && x < p.W) {
          const float act_sl = p.act == HRV_ACT_NONE ? 1.f : (p.act == HRV_ACT_RELU ? 0.f : p.slope);
          unsigned short* const orow = reinterpret_cast<unsigned short*>(p.out) + (((size_t)n * p.H + y) * p.W + x) * p.out_cs + p.out_co + 8 * lh;
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              unsigned q[2][2];      // [g = 2h, 2h + 1][2 packed bf16 pairs]
#pragma unroll
              for (int gg = 0; gg < 2; ++gg) {
                const int g = 2 * h + gg;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_l + j * 32 + 8 * g + 4 * lh);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float t = acc[j][4 * g + e] + b4[e];
                  v[e] = fmaxf(t, t * act_sl);          // none (1) / ReLU (0) / LeakyReLU (slope): one form, no branches
                }
                q[gg][0] = thin_pk_bf16(v[0], v[1]);
                q[gg][1] = thin_pk_bf16(v[2], v[3]);
              }
              // lanes < 32 end up with (own g0, partner's g0) = channels 16h + 0..7, lanes >= 32 with (partner's g1, own g1)
              // = channels 16h + 8..15 of their pixel
              const auto s0 = __builtin_amdgcn_permlane32_swap(q[0][0], q[1][0], false, false);
              const auto s1 = __builtin_amdgcn_permlane32_swap(q[0][1], q[1][1], false, false);
              u32x4 o;
              o[0] = s0[0]; o[1] = s1[0]; o[2] = s0[1]; o[3] = s1[1];
              *reinterpret_cast<u32x4*>(orow + j * 32 + 16 * h) = o;
            }
        }
      } else
      if (y < p.H && x < p.W) {
        const size_t pix = ((size_t)n * p.H + y) * p.W + x;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c0 = j * 32 + 8 * g + 4 * lh;
            if (c0 < p.NC) {
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = acc[j][4 * g + e] + ((p.shift && c0 + e < p.NC) ? p.shift[c0 + e] : 0.f);
              if (p.res) {
                const f32x4 r4 = ld4rt<true>(reinterpret_cast<const float*>(p.res), pix * p.res_cs + p.res_co + c0, p.res_f32);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = res_combine(v[e], r4[e], p.res_mode, p.slope);
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (c0 + e < p.NC) ? apply_act(v[e], p.act, p.slope) : 0.f;   // pad channels stay zero
              st4rt<true>(reinterpret_cast<float*>(p.out), pix * p.out_cs + p.out_co + c0, v, p.out_f32);
            }
          }
      }
    }
    if (WIDE && p.full) {
      // exactly 2 * TN stores follow the patch DMA in this wave's queue (in-order completion): the patch has landed, the
      // stores drain under the next tile's MFMAs
      constexpr int NS = 2 * TN;
      static_assert(NS < 64, "vmcnt is six bits");
      __builtin_amdgcn_s_waitcnt((NS & 15) | (7 << 4) | (0 << 8) | ((NS >> 4) << 14));
    } else {
      __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0 << 8));   // the next patch has landed (and this tile's stores left)
    }
    __syncthreads();
  }
}

template <int TN, int KB, int KS, bool WIDE = false>
static int thin_launch(const ThinParams& p, hipStream_t st) {
  constexpr int KH = KS == 9 ? 3 : 1;
  constexpr int PPIX = (16 + KH - 1) * (8 + KH - 1), SL = (2 * KB) | 1;
  constexpr int WB = (KS * KB * 32 * TN * 32 + 1023) & ~1023;
  constexpr int PB = ((PPIX * SL + 63) / 64) * 1024;
  constexpr int LDS = WB + PB + (WIDE ? 32 * TN * 4 : 0);
  static_assert(LDS <= 160 * 1024, "thin conv: LDS");
  static bool attr_done_dev[HRV_MAX_DEVICES] = {false};      // the attribute is per device (and per instantiation: this is a template)
  bool& attr_done = attr_done_dev[current_device()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_conv_kernel<TN, KB, KS, WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) !=
        hipSuccess) {
      set_error("thin_conv: hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) failed", LDS);
      return HRV_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const int n_cu = persistent_cus();
  int per_cu = (160 * 1024) / LDS;
  per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
  int grid = n_cu * per_cu;
  if (grid > p.tiles) grid = p.tiles;
  hipLaunchKernelGGL((thin_conv_kernel<TN, KB, KS, WIDE>), dim3(grid), dim3(256), LDS, st, p);
  return check_launch("thin_conv_kernel");
}

}  // namespace hrv

using namespace hrv;

extern "C" int hrv_thin_conv_supported(int32_t KH, int32_t KW, int32_t src_channels, int32_t out_columns) {
  if (!((KH == 3 && KW == 3) || (KH == 1 && KW == 1))) return 0;
  const int kb = (src_channels + 15) / 16, tn = (out_columns + 31) / 32;
  const bool k9 = KH == 3;
  // instantiated (TN, KB): up_4 conv_0 forward (1,5) / data gradient (3,2), conv_1 and conv_img (1,2), conv_s forward (1,5) / dgrad (3,2)
  // kb 1: the 9-channel stem, conv_img's data gradient; (tn 2, kb 1): VGG19 features.0 (3 -> 64 over every pixel)
  // (tn 2, kb 4): VGG19 features.2 (64 -> 64 over every pixel, bf16-stored activations)
  if (k9) return (tn == 1 && (kb == 5 || kb == 2 || kb == 1)) || (tn == 3 && kb == 2) || (tn == 2 && (kb == 1 || kb == 4));
  // (tn 12, kb 5): a SPADEResBlock's three conv_shared as one 1x1 over the tap-expanded label map (72 -> 384), see WIDE
  return (tn == 1 && kb == 5) || (tn == 3 && kb == 2) || (tn == 12 && kb == 5 && out_columns == 384);
}

extern "C" int hrv_thin_conv_bf16(const hrv_thin_conv_t* d, hrv_stream_t stream) {
  HRV_REQUIRE(d && d->src && d->w_oihw && d->out, "thin_conv: null pointer");
  HRV_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cout > 0 && d->Cin > 0, "thin_conv: extent");
  HRV_REQUIRE(d->mode == 0 || d->mode == 1, "thin_conv: mode 0 (forward) or 1 (data gradient)");
  const int CK = d->mode == 0 ? d->Cin : d->Cout, NC = d->mode == 0 ? d->Cout : d->Cin;
  HRV_REQUIRE(d->src_channels >= CK && d->src_channels % 8 == 0 && d->src_cstride % 8 == 0 && d->src_coff % 8 == 0 &&
                  d->src_coff + d->src_channels <= d->src_cstride && ((uintptr_t)d->src & 15) == 0,
              "thin_conv: the bf16 source needs 8-channel granules");
  HRV_REQUIRE(hrv_thin_conv_supported(d->KH, d->KW, d->src_channels, NC), "thin_conv: shape (k=%d, %d source channels, %d columns) "
              "is not one of the instantiated thin layers", d->KH, d->src_channels, NC);
  HRV_REQUIRE(d->out_cstride % 4 == 0 && d->out_coff % 4 == 0 && d->out_coff + ((NC + 3) & ~3) <= d->out_cstride,
              "thin_conv: output slice (4-channel groups)");
  HRV_REQUIRE(!d->residual || (d->res_cstride % 4 == 0 && d->res_coff % 4 == 0), "thin_conv: residual slice");
  HRV_REQUIRE(d->res_mode == 0 || d->res_mode == 1, "thin_conv: res_mode");
  const int64_t bytes = (int64_t)d->N * d->H * d->W * d->src_cstride * 2;
  HRV_REQUIRE(bytes < (int64_t)0xFFFFFFF0, "thin_conv: source larger than the 32-bit buffer range");
  ThinParams p;
  p.src = d->src; p.N = d->N; p.H = d->H; p.W = d->W; p.CK = d->src_channels; p.cs = d->src_cstride; p.co = d->src_coff;
  p.src_bytes = (unsigned)bytes;
  p.w = d->w_oihw; p.Cout = d->Cout; p.Cin = d->Cin; p.KH = d->KH; p.KW = d->KW; p.sigma = d->sigma; p.wscale = d->wscale;
  p.transposed = d->mode;
  p.CK = CK;           // channels with weights (the rest of src_channels, if any, is padding and multiplies zeros)
  p.NC = NC;
  p.shift = d->shift;
  p.res = d->residual; p.res_cs = d->res_cstride; p.res_co = d->res_coff; p.res_f32 = d->res_bf16 ? 0 : 1; p.res_mode = d->res_mode;
  p.act = d->act; p.slope = d->act_slope;
  p.out = d->out; p.out_cs = d->out_cstride; p.out_co = d->out_coff; p.out_f32 = d->out_bf16 ? 0 : 1;
  p.pad = d->KH / 2;
  p.tx = (d->W + 15) / 16; p.ty = (d->H + 7) / 8; p.tiles = d->N * p.tx * p.ty;
  p.full = (d->W % 16 == 0 && d->H % 8 == 0) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  const int kb = (d->src_channels + 15) / 16, tn = (NC + 31) / 32;
  if (d->KH == 3) {
    if (tn == 1 && kb == 5) return thin_launch<1, 5, 9>(p, st);
    if (tn == 1 && kb == 2) return thin_launch<1, 2, 9>(p, st);
    if (tn == 1 && kb == 1) return thin_launch<1, 1, 9>(p, st);
    if (tn == 3 && kb == 2) return thin_launch<3, 2, 9>(p, st);
    if (tn == 2 && kb == 1) return thin_launch<2, 1, 9>(p, st);
    if (tn == 2 && kb == 4) return thin_launch<2, 4, 9>(p, st);
  } else {
    if (tn == 1 && kb == 5) return thin_launch<1, 5, 1>(p, st);
    if (tn == 3 && kb == 2) return thin_launch<3, 2, 1>(p, st);
    if (tn == 12 && kb == 5) {
      HRV_REQUIRE(d->act == HRV_ACT_NONE || d->act == HRV_ACT_RELU || (d->act == HRV_ACT_LRELU && d->act_slope >= 0.f && d->act_slope <= 1.f),
                  "thin_conv: the 384-column layer takes no activation, ReLU or LeakyReLU");
      HRV_REQUIRE(d->mode == 0 && !d->residual && d->out_bf16 && d->out_cstride % 8 == 0 && d->out_coff % 8 == 0 &&
                      ((uintptr_t)d->out & 15) == 0,
                  "thin_conv: the 384-column layer is forward-only, without residual, into a bf16 tensor with 8-channel granules");
      return thin_launch<12, 5, 1, true>(p, st);
    }
  }
  set_error("thin_conv: no instantiation");
  return HRV_ERR_ARG;
}
