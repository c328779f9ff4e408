// Probe: the SUSTAINED bf16 matrix rate and shader clock of this MI355X under load (VERDICT r5 "do this" #1b).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_roof mfma_roof.hip      Run: ./mfma_roof [seconds per variant, default 2]
//
// Every variant launches 256-thread blocks (4 waves, one per SIMD) on all 256 CUs, B blocks per CU, back to back for the stated
// time, and reports TFLOP/s over the second half of the run together with the shader clock the chip held (s_memtime ticks per
// 100-MHz wall_clock64 tick, sampled by one lane per block).  The variants walk from the datasheet condition to the condition of
// conv_p2.hip / spade_fused.hip's main loops:
//   reg-zero     register-resident v_mfma_f32_32x32x16_bf16, all-zero operands (no toggling: the datasheet's number)
//   reg-rand     the same with 8 x 8 distinct random operand fragments cycled through (operand buses toggle as in a real kernel)
//   lds-rand     operands re-read from LDS in front of every MFMA group at conv_p2<4>'s ratio: 6 ds_read_b128 per 8 MFMAs
//   lds-bar      + one s_barrier per 16 MFMAs (one per k-tile)
//   lds-bar-dma  + the weight ring's refill: 2 LDS-DMA pieces of 1 KB per wave per 16 MFMAs from an L2-resident stream
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
  } while (0)

__device__ __forceinline__ f32x16 mfma(f32x4 a, f32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct Clk { unsigned long long cyc, wall; };

// V: 0 reg (operands from `src`: zero or random), 2 lds, 3 lds + barrier, 4 lds + barrier + dma
template <int V>
__global__ __launch_bounds__(256, 2) void roof_kernel(const f32x4* __restrict__ src, const void* wstream, unsigned wbytes, int iters, float* sink,
                                                      Clk* clk) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  unsigned long long c0 = 0, w0 = 0;
  if (tid == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
  if constexpr (V == 0) {
    f32x4 fa[8], fb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { fa[k] = src[(k * 64 + lane)]; fb[k] = src[((8 + k) * 64 + lane)]; }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        // one "k-step" of conv_p2<4>: 2 A fragments x 4 B fragments = 8 MFMAs
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0][j] = mfma(fb[(k + j) & 7], fa[k], acc[0][j]);
          acc[1][j] = mfma(fb[(k + j) & 7], fa[k + 1], acc[1][j]);
        }
      }
      if ((it & 63) == 63) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] *= 1e-3f;
      }
    }
  } else {
    // 48 KB of random operand bytes in LDS (the ring + one patch buffer); fragment reads walk it like the tap-shifted reads do
    for (int i = tid; i < 48 * 1024 / 16; i += 256) reinterpret_cast<f32x4*>(smem)[i] = src[i & 1023];
    __syncthreads();
    const rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wstream), 0, wbytes, 0x00020000);
    unsigned woff = (unsigned)(blockIdx.x & 63) * 8192u;
    const unsigned char* const a_l = smem + 24576 + wave * 4096 + lane * 16;
    const unsigned char* const b_l = smem + lane * 16;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
      // 32 MFMAs = 4 k-steps = 2 k-tiles per iteration
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const unsigned ao = (unsigned)(((it * 4 + ks) * 64) & 1023);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(a_l + ao);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(a_l + 2048 + ao);
        f32x4 b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f32x4*>(b_l + ((((it * 4 + ks) * 4 + j) * 1024) & 16383));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0][j] = mfma(b[j], a0, acc[0][j]);
          acc[1][j] = mfma(b[j], a1, acc[1][j]);
        }
        if constexpr (V >= 4) {
          if ((ks & 1) == 0) {
            // the ring's refill: 2 pieces of 1 KB per wave per k-tile into stage 2 (never read above: contents irrelevant, traffic real)
#pragma unroll
            for (int q = 0; q < 2; ++q)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(smem + 16384 + (wave * 2 + q) * 1024), 16,
                                                       (unsigned)lane * 16u, woff + (unsigned)(wave * 2 + q) * 1024u, 0, 0);
            woff += 8192u;
            woff = woff + 8192u <= wbytes ? woff : 0u;
          }
        }
        if constexpr (V >= 3) {
          if (ks & 1) {
            if constexpr (V >= 4) __builtin_amdgcn_s_waitcnt((2 & 15) | (7 << 4) | (15 << 8));      // vmcnt(2): the previous k-tile's pieces landed
            __builtin_amdgcn_s_barrier();
          }
        }
      }
      if ((it & 31) == 31) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] *= 1e-3f;
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  if (s == 12345.678f) sink[tid] = s;
  if (tid == 0) {
    clk[blockIdx.x].cyc = __builtin_readcyclecounter() - c0;
    clk[blockIdx.x].wall = wall_clock64() - w0;
  }
}

template <int V>
static void run(const char* name, const f32x4* src, const void* wstream, unsigned wbytes, float* sink, Clk* clk, int bpc, double seconds) {
  const int grid = 256 * bpc;
  const size_t lds = V == 0 ? 0 : 48 * 1024;
  const int mf_per_iter = 32;
  int iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // calibrate to ~25 ms per launch
  for (int pass = 0; pass < 2; ++pass) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((roof_kernel<V>), dim3(grid), dim3(256), lds, 0, src, wstream, wbytes, iters, sink, clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (pass == 0) iters = (int)(iters * 25.0 / ms) + 1;
  }
  const double flop_per_launch = (double)grid * 4 * iters * mf_per_iter * (2.0 * 32 * 32 * 16);
  double total_ms = 0, half_ms = 0; int n = 0, nh = 0;
  std::vector<Clk> h(grid);
  double mhz_sum = 0; int mhz_n = 0;
  while (total_ms < seconds * 1e3) {
    CK(hipEventRecord(e0));
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((roof_kernel<V>), dim3(grid), dim3(256), lds, 0, src, wstream, wbytes, iters, sink, clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    total_ms += ms; n += 4;
    if (total_ms >= seconds * 5e2) {
      half_ms += ms; nh += 4;
      CK(hipMemcpy(h.data(), clk, sizeof(Clk) * grid, hipMemcpyDeviceToHost));
      for (int b = 0; b < grid; b += 7)
        if (h[b].wall) { mhz_sum += (double)h[b].cyc / (double)h[b].wall * 100.0; ++mhz_n; }
    }
  }
  const double tf = flop_per_launch * nh / (half_ms * 1e-3) / 1e12;
  printf("%-12s blocks/CU %d  %8.1f TFLOP/s  = %.3f of 2500   shader clock %.0f MHz   (%d launches of %.1f ms, second half of %.1f s)\n", name, bpc,
         tf, tf / 2500.0, mhz_n ? mhz_sum / mhz_n : 0.0, nh, half_ms / nh, total_ms * 1e-3);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  f32x4 *zero, *rnd; float* sink; Clk* clk; void* wstream;
  const unsigned wbytes = 8u << 20;
  CK(hipMalloc(&zero, 16 * 1024 * 16)); CK(hipMalloc(&rnd, 16 * 1024 * 16)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&clk, sizeof(Clk) * 1024));
  CK(hipMalloc(&wstream, wbytes));
  CK(hipMemset(zero, 0, 16 * 1024 * 16));
  std::vector<uint16_t> h(16 * 1024 * 8);
  srand(1);
  for (auto& v : h) {
    // bf16 of a value in (-2, 2) with a random mantissa
    const float f = ((rand() & 0xFFFF) / 32768.0f - 1.0f) * 2.0f;
    uint32_t u; memcpy(&u, &f, 4);
    v = (uint16_t)(u >> 16);
  }
  CK(hipMemcpy(rnd, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  std::vector<uint16_t> hw(wbytes / 2);
  for (auto& v : hw) { const float f = ((rand() & 0xFFFF) / 32768.0f - 1.0f) * 0.1f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
  CK(hipMemcpy(wstream, hw.data(), wbytes, hipMemcpyHostToDevice));
  printf("# v_mfma_f32_32x32x16_bf16 sustained rate, 256 CUs, %.1f s per line; peak priced at 2500 TFLOP/s (2.4 GHz)\n", seconds);
  for (int bpc = 1; bpc <= 2; ++bpc) {
    run<0>("reg-zero", zero, wstream, wbytes, sink, clk, bpc, seconds);
    run<0>("reg-rand", rnd, wstream, wbytes, sink, clk, bpc, seconds);
    run<2>("lds-rand", rnd, wstream, wbytes, sink, clk, bpc, seconds);
    run<3>("lds-bar", rnd, wstream, wbytes, sink, clk, bpc, seconds);
    run<4>("lds-bar-dma", rnd, wstream, wbytes, sink, clk, bpc, seconds);
  }
  return 0;
}
