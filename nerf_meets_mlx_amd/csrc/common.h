// Shared helpers for the gfx950 kernels behind include/nerf_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/nerf_hip.h"

#define WAVE 64

namespace nerf {

// last-error text (thread-compatible, one process per GPU)
char* err_buf();
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// grid sizing for HBM-bound kernels: enough workgroups to fill 256 CUs, grid-stride the rest
static inline int grid_for(int64_t work_items, int block, int max_blocks = 256 * 8) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

// ---- wave64 scans / reductions over DPP-free shuffles (HBM-bound callers: not the limiter)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ float wave_scan_incl(float v, int lane) {
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    float t = __shfl_up(v, o, WAVE);
    if (lane >= o) v += t;
  }
  return v;
}
__device__ __forceinline__ double wave_scan_incl(double v, int lane) {
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    double t = __shfl_up(v, o, WAVE);
    if (lane >= o) v += t;
  }
  return v;
}
// inclusive suffix sum (lane i gets sum over lanes >= i)
__device__ __forceinline__ float wave_rscan_incl(float v, int lane) {
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    float t = __shfl_down(v, o, WAVE);
    if (lane + o < WAVE) v += t;
  }
  return v;
}

}  // namespace nerf

#define NERF_REQUIRE(cond, code, ...) \
  do {                                \
    if (!(cond)) return nerf::fail(code, __VA_ARGS__); \
  } while (0)
