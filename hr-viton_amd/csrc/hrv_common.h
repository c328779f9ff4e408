// Shared host/device helpers for the gfx950 HR-VITON kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/hrviton_hip.h"

namespace hrv {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return HRV_ERR_LAUNCH;
  }
  return HRV_OK;
}

// ---- per-DEVICE facts (one process per GPU is the deployment, but the library must not assume it: nothing here is keyed
// on "the process")
constexpr int HRV_MAX_DEVICES = 64;
int current_device();      // hipGetDevice, clamped into [0, HRV_MAX_DEVICES)
int device_cus();          // multiProcessorCount of the current device (cached per device)
// The CU count persistent kernels (grid = one block per CU, each owning most of the LDS) size themselves to: device_cus()
// minus the CUs reserved for kernels that must run CONCURRENTLY with them -- RCCL's collective kernels during a
// data-parallel backward would otherwise queue behind 40 us .. 1 ms persistent blocks (hrv_set_reserved_cus /
// HRV_RESERVE_CUS, default 0).
int persistent_cus();
// diagnostic per-tile timeline buffer (hrv_diag_set_tlog; nullptr in production): 8 u64 per tile, owned by the measuring tool
unsigned long long* diag_tlog(long long tiles);

#define HRV_REQUIRE(cond, ...)     \
  do {                             \
    if (!(cond)) {                 \
      hrv::set_error(__VA_ARGS__); \
      return HRV_ERR_ARG;          \
    }                              \
  } while (0)

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case HRV_ACT_RELU: return v > 0.f ? v : 0.f;
    case HRV_ACT_LRELU: return v > 0.f ? v : v * slope;
    case HRV_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// Bijective XCD-aware remap: hardware places block b on XCD b % 8; give each XCD
// a contiguous range of logical tiles so neighbouring tiles (shared halo rows,
// shared weight panels) hit the same 4 MiB L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int xcd = bid & 7;
  const int q = nblk >> 3, r = nblk & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// A/B and diagnostic switches (HRV_*): the environment is read ONCE per name and cached -- after a name's first use no launch path
// calls getenv().  hrv_diag_reload_env() drops the cache (tests / tools that flip a switch inside one process).  Defined in
// sample.hip; the returned pointer stays valid until the next reload.
const char* env(const char* name);

// Launch geometry shared by the per-(sample, channel) reduction kernels (InstanceNorm statistics, SPADE / InstanceNorm
// backward): grid = (pixel slabs, samples, channel chunks).  A block covers <= NORM_GCAP groups of 4 channels, so the
// 1040-channel / 64x48-pixel levels of the generator spread over 5 x 24 x N blocks instead of 6 x N (they ran at
// 50-200 GB/s), and slabs are >= 128 pixels (fixed-order second stage over <= 256 slab partials).
constexpr int NORM_GCAP = 64;
inline int norm_slab_cap() {      // (hrv::env is the cache: no static copy here that hrv_diag_reload_env could not reach)
  const char* e = hrv::env("HRV_NORM_SLABS_MAX");
  const int cap = e ? atoi(e) : 256;
  return cap < 1 ? 256 : cap;
}
inline int norm_slabs(int HW) {
  const int nb = (HW + 127) / 128, cap = norm_slab_cap();
  return nb < 1 ? 1 : (nb > cap ? cap : nb);
}
inline int norm_chunks(int C4) { return (C4 + NORM_GCAP - 1) / NORM_GCAP; }

// out[c] (+)= sum over rows of part[r][c]: 16 channels x 16 row-groups per block, double accumulation, fixed combination
// order (a thread per channel walking all rows serially was 15-111 us per call).  A template so that every translation
// unit that launches it carries its own instance (separate compilation, no relocatable device code).
__device__ __forceinline__ void sum_rows_block(const int block, const float* __restrict__ part, int rows, int C,
                                               float* __restrict__ out, int accumulate) {
  __shared__ double red[16][16];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = block * 16 + cl;
  double s = 0.0;
  if (c < C)
    for (int r = rg; r < rows; r += 16) s += (double)part[(size_t)r * C + c];
  red[rg][cl] = s;
  __syncthreads();
  if (threadIdx.x < 16 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][cl];
    out[c] = accumulate ? out[c] + (float)t : (float)t;
  }
}

template <int = 0>
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ part, int rows, int C, float* __restrict__ out, int accumulate) {
  sum_rows_block(blockIdx.x, part, rows, C, out, accumulate);
}

}  // namespace hrv
