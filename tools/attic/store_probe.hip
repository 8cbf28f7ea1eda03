// What does one 1 KiB fragment store cost a wave that is otherwise issuing MFMAs?  (diagnostic, not part of the library)
//   hipcc -O3 --offload-arch=gfx950 tools/store_probe.hip -o tools/diag/store_probe && tools/diag/store_probe
//
// The training forward / backward chain take 124 k cycles per 256-sample pass against 92 k without their fragment
// stores: the store cost adds up serially to the MFMA time, and de-phasing the waves three different
// ways did not hide it.  This probe isolates the pattern: 8 waves per CU (2 per SIMD) run a loop of 16 dependent-free
// v_mfma_f32_32x32x16_bf16 per iteration and, per iteration, K stores of 16 B per lane (1 KiB per wave-instruction),
//   mode 0: no stores            mode 1: data from VGPRs (global_store_dwordx4 v, v[4])
//   mode 2: data from AGPRs (global_store_dwordx4 v, a[4]) -- if the serialisation were a VGPR-read-port conflict
//           between the store's data export and the MFMA operand reads, this form would overlap
//   mode 3: data from VGPRs, non-temporal
// Stores go to a small L2-resident region (the cost under test is inside the CU, not HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int K>
__global__ void __launch_bounds__(512, 2) probe(u32x4* __restrict__ dst, int iters, float* sink) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 8 + (threadIdx.x >> 6);
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
  u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  u32x4* p = dst + (size_t)(wave & 2047) * 64 * 8 + lane;             // 8 KiB per wave slot, 16 MiB region: L2-resident
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
      if (MODE != 0 && (m % (16 / K)) == 0) {
        const int k = m / (16 / K);
        v[0] += it;                                                  // keep the data live and changing
        if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p + 64 * (k & 7)), "v"(v) : "memory");
        else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p + 64 * (k & 7)), "a"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p + 64 * (k & 7)), "v"(v) : "memory");
      }
    }
  }
  float s = 0.0f;
  for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
  if (s == 12345.678f) *sink = s;
}

template <int MODE, int K>
static void run(const char* name, u32x4* dst, float* sink) {
  const int iters = 4000, grid = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, K>), dim3(grid), dim3(512), 0, 0, dst, 200, sink);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, K>), dim3(grid), dim3(512), 0, 0, dst, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  // per SIMD: 2 waves x 16 MFMAs x 32 cycles = 1024 MFMA cycles per iteration at full rate
  const double us_per_iter = best * 1e3 / iters;
  printf("%-34s K=%d stores / 16 MFMAs: %.3f ms, %.1f ns per iteration per wave-pair (MFMA floor at 2.4 GHz: 426.7 ns), %.1f TFLOP/s\n",
         name, MODE ? K : 0, best, us_per_iter * 1e3, 2.0 * 32 * 32 * 16 * 16 * 8 * grid * iters / (best * 1e-3) / 1e12);
}

int main() {
  u32x4* dst; float* sink;
  hipMalloc(&dst, (size_t)2048 * 8 * 1024);
  hipMalloc(&sink, 4);
  run<0, 1>("no stores", dst, sink);
  run<1, 1>("stores from VGPRs", dst, sink);
  run<2, 1>("stores from AGPRs", dst, sink);
  run<1, 2>("stores from VGPRs", dst, sink);
  run<2, 2>("stores from AGPRs", dst, sink);
  run<3, 2>("stores from VGPRs, nt", dst, sink);
  run<1, 4>("stores from VGPRs", dst, sink);
  run<2, 4>("stores from AGPRs", dst, sink);
  run<0, 1>("no stores (again)", dst, sink);
  return 0;
}
