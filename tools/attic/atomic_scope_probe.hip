// Probe (round 5): do integer / float global atomics run faster at a narrower scope, or when every table slice is only ever touched
// by ONE XCD (HW_REG_XCC_ID)?  The configs[4] table-gradient scatter is bound by the request rate of device-scope atomics, which
// execute at the memory side (MI355X_MICROARCH.md "Global float atomics"); if workgroup-scope atomics on XCD-private lines ran in
// the XCD's L2 instead, an XCD-partitioned scatter would be worth building.
//   hipcc --offload-arch=gfx950 -O3 -o tools/diag/atomic_scope_probe tools/atomic_scope_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7; }
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// every thread: N adds to pseudo-random entries (pairs of adjacent lanes share a 16-byte entry pair like the two features of a hash-grid
// entry); PART: the entry's top 3 index bits are the issuing XCD's id (a slice is private to an XCD)
template <typename T, int SCOPE, bool PART>
__global__ void scatter(T* table, unsigned entries_log2, int n_adds, unsigned seed) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned xcc = xcc_id();
  const unsigned mask = (1u << entries_log2) - 1;
  for (int i = 0; i < n_adds; ++i) {
    unsigned e = mix((tid >> 1) * 2654435761u + i * 40503u + seed) & mask;
    if (PART) e = (e & (mask >> 3)) | (xcc << (entries_log2 - 3));
    T* p = table + 2 * (size_t)e + (tid & 1);
    __hip_atomic_fetch_add(p, (T)1, __ATOMIC_RELAXED, SCOPE);
  }
}

template <typename T, int SCOPE, bool PART>
static int run(const char* name, T* table, unsigned entries_log2, size_t bytes) {
  const int wgs = 2048, threads = 256, n_adds = 64;
  CHECK(hipMemset(table, 0, bytes));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  scatter<T, SCOPE, PART><<<wgs, threads>>>(table, entries_log2, n_adds, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) scatter<T, SCOPE, PART><<<wgs, threads>>>(table, entries_log2, n_adds, 7u + r);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  // check the total: every add must have arrived (sum over the table == adds of the 6 launches)
  std::vector<T> h(bytes / sizeof(T));
  CHECK(hipMemcpy(h.data(), table, bytes, hipMemcpyDeviceToHost));
  double sum = 0; for (T v : h) sum += (double)v;
  const double want = 6.0 * wgs * threads * n_adds;
  printf("%-46s %7.3f ms  %6.1f G lane-atomics/s   sum %s (%.0f of %.0f)\n", name, ms, wgs * threads * (double)n_adds / ms / 1e6,
         sum == want ? "exact" : "LOST ADDS", sum, want);
  return 0;
}

int main() {
  const unsigned entries_log2 = 22;                        // 4 M entry pairs
  const size_t bytes = (size_t)2 * (1u << entries_log2) * 8;
  void* t; CHECK(hipMalloc(&t, bytes));
  run<unsigned long long, __HIP_MEMORY_SCOPE_AGENT, false>("u64 agent scope, any XCD", (unsigned long long*)t, entries_log2, bytes);
  run<unsigned long long, __HIP_MEMORY_SCOPE_AGENT, true>("u64 agent scope, XCD-private slices", (unsigned long long*)t, entries_log2, bytes);
  run<unsigned long long, __HIP_MEMORY_SCOPE_WORKGROUP, true>("u64 workgroup scope, XCD-private slices", (unsigned long long*)t, entries_log2, bytes);
  run<unsigned long long, __HIP_MEMORY_SCOPE_WAVEFRONT, true>("u64 wavefront scope, XCD-private slices", (unsigned long long*)t, entries_log2, bytes);
  run<unsigned, __HIP_MEMORY_SCOPE_AGENT, false>("u32 agent scope, any XCD", (unsigned*)t, entries_log2, bytes / 2);
  run<unsigned, __HIP_MEMORY_SCOPE_WORKGROUP, true>("u32 workgroup scope, XCD-private slices", (unsigned*)t, entries_log2, bytes / 2);
  run<float, __HIP_MEMORY_SCOPE_AGENT, false>("f32 agent scope, any XCD", (float*)t, entries_log2, bytes / 2);
  run<float, __HIP_MEMORY_SCOPE_WORKGROUP, true>("f32 workgroup scope, XCD-private slices", (float*)t, entries_log2, bytes / 2);
  return 0;
}
