// Probe: lane mapping and bank behaviour of ds_read_b64_tr_b16 (gfx950).  Build: hipcc --offload-arch=gfx950 -O3 -o tr_read_probe tr_read_probe.hip
// Part 1: LDS element e holds the bf16-encoded integer e; lane l reads 8 bytes at byte offset 8*l.  The printed
//         table says, for every lane and result slot, which LDS element arrived there.
// Part 2: cycles per wave-instruction for candidate [pixel][channel] images (4 waves, 16 reads per iteration).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ s16x4 tr_read(const unsigned short* lds_ptr) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lds_ptr);
}

__global__ void map_kernel(unsigned short* out) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const s16x4 v = tr_read(lds + 4 * threadIdx.x);
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (unsigned short)v[e];
}

// variant: 0 = row stride 256 B, no swizzle; 1 = row stride 256 B, 64-B quarter XOR (p & 3); 2 = [*][4 px][16 ch] 128-B blocks
//          3 = row stride 256 B, 16-B group XOR (p & 15) (the forward patch's swizzle); 4 = row stride 272 B (padded)
template <int V>
__global__ void time_kernel(long long* cyc, unsigned* sink, int iters) {
  __shared__ unsigned short lds[32768];
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, i = lane & 15;
  // operand fragment of a 32x32x16 MFMA: group g -> channels 16*(g&1).., pixels 8*(g>>1) + 4*r + (i>>2)
  unsigned acc = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = ((r >> 1) * 16 + 8 * (g >> 1) + 4 * (r & 1) + (i >> 2) + wave * 3 + it) & 63;   // pixel row
      const int c = 16 * (g & 1) + 4 * (i & 3) + 32 * (wave & 3);                                     // channel (bf16 index)
      int off;
      if (V == 0) off = p * 128 + c;
      else if (V == 1) off = p * 128 + ((((c >> 5) ^ (p & 3)) << 5) | (c & 31));
      else if (V == 2) off = ((p >> 2) * 8 + (c >> 4)) * 64 + (p & 3) * 16 + (c & 15);
      else if (V == 3) off = p * 128 + ((((c >> 3) ^ (p & 15)) << 3) | (c & 7));
      else off = p * 136 + c;
      const s16x4 v = tr_read(lds + off);
      acc += (unsigned)v[0] + (unsigned)v[3];
    }
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int V>
static void run_time(const char* name) {
  long long* cyc; unsigned* sink;
  hipMalloc(&cyc, 256 * 4 * sizeof(long long)); hipMalloc(&sink, 256 * 256 * 4);
  const int iters = 2000;
  hipLaunchKernelGGL(time_kernel<V>, dim3(256), dim3(256), 0, 0, cyc, sink, iters);
  hipLaunchKernelGGL(time_kernel<V>, dim3(256), dim3(256), 0, 0, cyc, sink, iters);
  hipDeviceSynchronize();
  long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 1024; ++i) s += (double)h[i];
  printf("%-46s %.2f clock64 ticks per tr-read wave-instruction (4 waves/CU streaming)\n", name, s / 1024 / iters / 16);
  hipFree(cyc); hipFree(sink);
}

int main() {
  unsigned short* d; hipMalloc(&d, 256 * 2);
  hipLaunchKernelGGL(map_kernel, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("ds_read_b64_tr_b16: lane l reads LDS elements 4l..4l+3 (8 bytes at offset 8l); result[lane][slot] = source element\n");
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d   (source lane,slot: %d.%d %d.%d %d.%d %d.%d)\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2],
                                     h[4 * l + 3], h[4 * l] / 4, h[4 * l] % 4, h[4 * l + 1] / 4, h[4 * l + 1] % 4, h[4 * l + 2] / 4, h[4 * l + 2] % 4,
                                     h[4 * l + 3] / 4, h[4 * l + 3] % 4);
  run_time<0>("rows 256 B, linear");
  run_time<1>("rows 256 B, 64-B quarter ^ (p&3)");
  run_time<2>("[4 px][16 ch] 128-B blocks");
  run_time<3>("rows 256 B, 16-B group ^ (p&15)");
  run_time<4>("rows 272 B (padded)");
  return 0;
}
