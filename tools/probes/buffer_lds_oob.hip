// Probe: does `buffer_load_dwordx4 ... offen lds` write ZEROS to LDS for lanes whose offset is out of range
// of the buffer resource (num_records)?  The bf16 conv engine's LDS-DMA gather relies on it for padding.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* src, float* out, unsigned nbytes) {
  __shared__ float smem[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) smem[i] = 7.0f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x & 1) voff = 0xFFFFFFF0u;
  if (threadIdx.x == 2) voff = nbytes - 8;   // straddles the end: partially out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
  __syncthreads();
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = smem[threadIdx.x * 4 + e];
}
int main() {
  float *src, *out, h[256], hs[256];
  for (int i = 0; i < 256; ++i) hs[i] = 100.f + i;
  hipMalloc(&src, 1024); hipMalloc(&out, 1024);
  hipMemcpy(src, hs, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, 1024u);
  hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
  for (int t = 0; t < 6; ++t) printf("lane %d: %g %g %g %g\n", t, h[t * 4], h[t * 4 + 1], h[t * 4 + 2], h[t * 4 + 3]);
  int zeros = 0, stale = 0;
  for (int t = 1; t < 64; t += 2) for (int e = 0; e < 4; ++e) { zeros += h[t * 4 + e] == 0.f; stale += h[t * 4 + e] == 7.f; }
  printf("odd (OOB) lanes: %d zero words, %d stale words of 128\n", zeros, stale);
  return 0;
}
