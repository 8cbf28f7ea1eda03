// Dynamic pass queue of the persistent ring kernels (round 6), stand-alone: wave 0 of a workgroup takes its next pass with ONE
// returning atomic issued from inline asm (hidden from hipcc's wait counting: a compiler-counted VMEM operation inside a pass makes
// hipcc wait with vmcnt(0), which drains the LDS-DMA ring) into a VGPR primed with a sentinel, and reads it back much later
// (v_readfirstlane; if the sentinel is still there: s_waitcnt vmcnt(0) once).  The last workgroup to finish clears the queue.
// Checks: every pass index is executed exactly once, over many launches through the same queue slot, with uneven pass lengths.
//   hipcc --offload-arch=gfx950 -O3 tools/ticket_probe.hip -o tools/diag/ticket_probe && tools/diag/ticket_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void ticket_issue(unsigned& tv, unsigned* q, bool active) {       // branch-free: `active` is the EXEC mask (lane 0 or none)
  unsigned long long save;
  const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane(active ? 1 : 0);
  asm volatile("v_mov_b32 %0, -1\n\ts_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, %5\n\ts_mov_b32 exec_hi, 0\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
               : "=&v"(tv), "=&s"(save) : "v"(0u), "v"(1u), "s"(q), "s"(mask) : "memory");
}
__device__ __forceinline__ unsigned ticket_take(unsigned& tv) {
  unsigned t;
  asm volatile("v_readfirstlane_b32 %0, %1\n\ts_cmp_lg_u32 %0, -1\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(0)\n\tv_readfirstlane_b32 %0, %1\n1:"
               : "=&s"(t) : "v"(tv) : "memory", "scc");
  return t;
}

__global__ void __launch_bounds__(256) k(unsigned* q, unsigned* hits, int n, int spin) {
  unsigned tv;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int sp = blockIdx.x;
  __shared__ int nxt;
  while (sp < n) {
    ticket_issue(tv, q, wv == 0);
    float v = (float)threadIdx.x;
    const int len = spin * (1 + (blockIdx.x & 7));             // workgroups of different speed
    for (int i = 0; i < len; ++i) v = v * 1.0001f + 0.5f;
    if (threadIdx.x == 0) atomicAdd(&hits[sp], v > 0.0f ? 1u : 2u);
    if (wv == 0) { const unsigned t = ticket_take(tv); if (threadIdx.x == 0) nxt = (int)(gridDim.x + t); }
    __syncthreads();
    sp = nxt;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&q[1], 1u) == gridDim.x - 1) { q[0] = 0u; q[1] = 0u; __threadfence(); }     // last one out clears the slot
  }
}

int main() {
  unsigned *q, *hits;
  const int n = 50000, wgs = 256;
  hipMalloc(&q, 16); hipMemset(q, 0, 16);
  hipMalloc(&hits, n * 4);
  std::vector<unsigned> h(n);
  int bad = 0;
  for (int it = 0; it < 20; ++it) {
    hipMemset(hits, 0, n * 4);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, q, hits, n - it * 777, 20 + it * 13);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch %d failed\n", it); return 1; }
    hipMemcpy(h.data(), hits, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) bad += h[i] != (i < n - it * 777 ? 1u : 0u);
    unsigned qq[4]; hipMemcpy(qq, q, 16, hipMemcpyDeviceToHost);
    bad += qq[0] != 0 || qq[1] != 0;
  }
  printf("ticket probe: %d errors over 20 launches of %d workgroups x ~%d passes\n", bad, wgs, n);
  return bad != 0;
}
