// Shared helpers for the gfx950 kernels behind include/nerf_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <mutex>
#include <utility>

#include "../../include/nerf_hip.h"

#define WAVE 64

namespace nerf {

// last-error text (thread-compatible, one process per GPU)
char* err_buf();
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// One-time host set-up per DEVICE (the opt-in to dynamic LDS above 64 KiB: hipFuncSetAttribute applies to the device that is
// current when it is called, and a process may move between devices).  Thread-safe: std::call_once -- a second thread
// launching the same kernel returns from run() only after the first one's attribute call has completed (a plain bool flag set
// before / after the call let it launch in between).
struct DevOnce {
  std::once_flag flag[64];
  template <class F> void run(F&& f) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
    std::call_once(flag[d], std::forward<F>(f));
  }
};

// grid sizing for HBM-bound kernels: enough workgroups to fill 256 CUs, grid-stride the rest
static inline int grid_for(int64_t work_items, int block, int max_blocks = 256 * 8) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

// ---- wave64 scans / reductions.  float: DPP (v_add_f32 ... row_shr / row_shl / row_bcast: the data path of the
// VALU, no LDS crossbar round trip per step as with ds_bpermute shuffles); double: shuffles (the one double scan,
// the importance sampler's CDF, must keep its summation order: its result is compared bit for bit with torch-CPU's).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or_zero(float v) {          // lanes without a source (and masked rows) read 0.0f
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// inclusive prefix sum across the 64 lanes: Hillis-Steele inside each row of 16 (row_shr 1, 2, 4, 8), then the last lane
// of row 0 / 2 into rows 1 / 3 (row_bcast:15), then lane 31 into rows 2 and 3 (row_bcast:31)
__device__ __forceinline__ float wave_scan_incl(float v, int lane) {
  (void)lane;
  v += dpp_or_zero<0x111, 0xf>(v);
  v += dpp_or_zero<0x112, 0xf>(v);
  v += dpp_or_zero<0x114, 0xf>(v);
  v += dpp_or_zero<0x118, 0xf>(v);
  v += dpp_or_zero<0x142, 0xa>(v);
  v += dpp_or_zero<0x143, 0xc>(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_scan_incl(v, 0)), 63));
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
// inclusive prefix sum across the 64 lanes (double: shuffles, see above)
__device__ __forceinline__ double wave_scan_incl(double v, int lane) {
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    double t = __shfl_up(v, o, WAVE);
    if (lane >= o) v += t;
  }
  return v;
}
// inclusive suffix sum (lane i gets the sum over lanes >= i): row_shl 1, 2, 4, 8 inside each row of 16, then the totals of
// the rows above (they sit in the first lane of each row: three v_readlane) are added per row
__device__ __forceinline__ float wave_rscan_incl(float v, int lane) {
  v += dpp_or_zero<0x101, 0xf>(v);
  v += dpp_or_zero<0x102, 0xf>(v);
  v += dpp_or_zero<0x104, 0xf>(v);
  v += dpp_or_zero<0x108, 0xf>(v);
  const int vi = __builtin_bit_cast(int, v);
  const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
  const float t2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32));
  const float t3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
  const int row = lane >> 4;
  const float above = row == 0 ? (t1 + (t2 + t3)) : row == 1 ? (t2 + t3) : row == 2 ? t3 : 0.0f;
  return v + above;
}

}  // namespace nerf

#define NERF_REQUIRE(cond, code, ...) \
  do {                                \
    if (!(cond)) return nerf::fail(code, __VA_ARGS__); \
  } while (0)
