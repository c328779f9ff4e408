// Shared by the convolution kernels (conv_f32.hip: the generic implicit-GEMM engine; conv_patchw.hip: the wide
// patch-mode kernel): launch parameters, element-type helpers, LDS-DMA primitives.
#pragma once
#include "hrv_common.h"

namespace hrv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HOST_BK = 16;   // fp32 k-values per 64-byte K-tile row (host-side packing of the default tiles)

struct SrcDev {
  const float* ptr;
  int C, cstride, coff, up_shift, pre_act, chunks;
  unsigned bytes;  // extent of the tensor in bytes (buffer resource of the LDS-DMA gather)
};

struct ConvParams {
  SrcDev src[HRV_MAX_SRC];
  int nsrc;
  int N, H, W, Ho, Wo, KH, KW, stride, pad;
  int M;        // N*Ho*Wo
  int Cout, CoutPad;
  int chunks_total;  // sum over sources of ceil(C/16)
  int KT;            // KH*KW*chunks_total
  int m_tiles, n_tiles;
  int n0_base;       // first output column of this launch (patch tiles: a layer's columns may be covered by launches of different tile widths)
  const float* wp;
  const float* scale;
  const float* shift;
  const float* res;
  int res_cs, res_co;
  int act;
  float slope;
  float pre_slope;
  float* out;
  int out_cs, out_co;
  int splitk;  // >1: K range split over `splitk` blocks per tile; raw partials go to `ws`
  float* ws;   // [splitk][M][CoutPad]
  int pad_w;   // horizontal padding (pad is the vertical one)
  int out_up;  // 1: replicate every result to its 2x2 block of a (2Ho x 2Wo) output
  int out_step, out_oh, out_ow, out_H, out_W;  // out_step 2: scatter to (2h+oh, 2w+ow) of an out_H x out_W output
  int bf16;      // 1: sources / weights are bf16 (accumulate + stats fp32)
  int out_f32, res_f32, sx_f32;  // bf16 mode: these tensors are fp32 instead of bf16
  int src_f32;   // bf16 mode: the conv SOURCES are fp32 and are rounded to bf16 while being staged
  int res_mode;  // 0: + residual; 1: * (residual > 0 ? 1 : slope)   (activation derivative, backward)
  unsigned w_bytes;       // size of the packed weight (LDS-DMA buffer resource; 0: LDS-DMA not usable)
  // SPADE epilogue (epi == 1)
  int epi;
  const float* sx;
  int sx_cs, sx_co, sC;
  const float* smean;
  const float* srstd;
  const float* sz;
  const float* sns;
  float* sg1p;  // optional (1+gamma) output, dense [M][sC]
  unsigned long long* tlog;   // diag only (HRV_PATCH_TLOG, tools/patch_timeline.py): per-tile phase timestamps of the patch tiles
};

// VAR bit0: swapped-operand MFMA (D[cout][pixel]) -> each lane owns 4 consecutive
//           output channels of one pixel -> float4 epilogue loads/stores.
// VAR bit1: software-pipelined body: fragment ds_reads first, next tile's global
//           loads issued between them and the MFMAs, scheduler hints interleave
//           the address arithmetic with the matrix pipe.
// Output pixel index of GEMM row `pidx`: dense, or the (2h+oh, 2w+ow) scatter of one phase of a
// stride-2 data gradient.
__device__ __forceinline__ size_t out_pixel(const ConvParams& p, int pidx) {
  if (p.out_step != 2) return (size_t)pidx;
  const int n = pidx / (p.Ho * p.Wo), rem = pidx - n * (p.Ho * p.Wo);
  const int h = rem / p.Wo, w = rem - h * p.Wo;
  return ((size_t)n * p.out_H + 2 * h + p.out_oh) * p.out_W + 2 * w + p.out_ow;
}

__device__ __forceinline__ float res_combine(float v, float r, int mode, float slope) {
  return mode == 0 ? v + r : v * (r > 0.f ? 1.f : slope);
}

// ---- element-type helpers: BF = activations / weights / residual / output are bf16 (fp32 accumulate,
// fp32 scale/shift/statistics); the LDS tiles and the 16-byte gather are byte-identical in both modes
// (a K-tile row is 64 bytes: 16 fp32 or 32 bf16 k-values).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even
  unsigned u = __builtin_bit_cast(unsigned, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
// 8 fp32 -> 8 bf16 (round to nearest even, v_cvt_pk_bf16_f32), returned as the 16 raw bytes of one LDS group
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 pack_bf16x8(f32x4 lo, f32x4 hi) {
  f32x4 r;
  const f32x2 p0 = {lo[0], lo[1]}, p1 = {lo[2], lo[3]}, p2 = {hi[0], hi[1]}, p3 = {hi[2], hi[3]};
  r[0] = __builtin_bit_cast(float, __builtin_convertvector(p0, bf16x2));
  r[1] = __builtin_bit_cast(float, __builtin_convertvector(p1, bf16x2));
  r[2] = __builtin_bit_cast(float, __builtin_convertvector(p2, bf16x2));
  r[3] = __builtin_bit_cast(float, __builtin_convertvector(p3, bf16x2));
  return r;
}

template <bool BF>
__device__ __forceinline__ f32x4 ld4e(const float* base, size_t idx) {
  if constexpr (BF) {
    const u16x4 h = *reinterpret_cast<const u16x4*>(reinterpret_cast<const unsigned short*>(base) + idx);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = bf2f(h[e]);
    return v;
  } else {
    return *reinterpret_cast<const f32x4*>(base + idx);
  }
}
template <bool BF>
__device__ __forceinline__ void st4e(float* base, size_t idx, f32x4 v) {
  if constexpr (BF) {
    u16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = f2bf(v[e]);
    *reinterpret_cast<u16x4*>(reinterpret_cast<unsigned short*>(base) + idx) = h;
  } else {
    *reinterpret_cast<f32x4*>(base + idx) = v;
  }
}
template <bool BF>
__device__ __forceinline__ float ld1e(const float* base, size_t idx) {
  if constexpr (BF) return bf2f(reinterpret_cast<const unsigned short*>(base)[idx]);
  else return base[idx];
}
template <bool BF>
__device__ __forceinline__ void st1e(float* base, size_t idx, float v) {
  if constexpr (BF) reinterpret_cast<unsigned short*>(base)[idx] = f2bf(v);
  else base[idx] = v;
}

// In bf16 mode the conv SOURCES and WEIGHTS are bf16 (compile time); the output, the residual and the
// SPADE x tensor may each be bf16 or fp32 (run-time flags): tensors that feed an InstanceNorm (block
// inputs / residual stream) stay fp32, tensors that only feed convolutions are bf16.
template <bool BF>
__device__ __forceinline__ f32x4 ld4rt(const float* base, size_t idx, int is_f32) {
  if constexpr (BF) { if (!is_f32) return ld4e<true>(base, idx); }
  return ld4e<false>(base, idx);
}
template <bool BF>
__device__ __forceinline__ float ld1rt(const float* base, size_t idx, int is_f32) {
  if constexpr (BF) { if (!is_f32) return ld1e<true>(base, idx); }
  return ld1e<false>(base, idx);
}
template <bool BF>
__device__ __forceinline__ void st4rt(float* base, size_t idx, f32x4 v, int is_f32) {
  if constexpr (BF) { if (!is_f32) { st4e<true>(base, idx, v); return; } }
  st4e<false>(base, idx, v);
}
template <bool BF>
__device__ __forceinline__ void st1rt(float* base, size_t idx, float v, int is_f32) {
  if constexpr (BF) { if (!is_f32) { st1e<true>(base, idx, v); return; } }
  st1e<false>(base, idx, v);
}

// LDS-DMA primitives.  The buffer-resource type and builtins exist in the device pass only; the host pass
// (which still parses kernel bodies to emit launch stubs) sees inert stand-ins.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
// 16 bytes per lane: global[base + voff + soff] -> LDS[lds (wave-uniform) + 16*lane]; offsets past `bytes` store 0
__device__ __forceinline__ void dma16(rsrc_t r, float* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
#else
struct rsrc_t { int unused; };
__device__ inline rsrc_t make_rsrc(const void*, unsigned) { return rsrc_t{0}; }
__device__ inline void dma16(rsrc_t, float*, unsigned, unsigned) {}
#endif


// conv_patchw.hip: 16x16-pixel patch tiles x up to 192 columns per block, one block per CU (tile_cfg 19)
int launch_patchw(const ConvParams& p, hipStream_t st);

}  // namespace hrv
