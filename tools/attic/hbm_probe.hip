// HBM streaming ceilings for the access shapes the training kernels use (diagnostic, not part of the library):
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_probe.hip -o tools/diag/hbm_probe && tools/diag/hbm_probe
// 1 KiB per wave-instruction (16 B per lane), fragments of a tile contiguous, tiles strided like the activation
// blocks (167 KiB) -- plain stores, non-temporal stores, plain loads, non-temporal loads.
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ void __launch_bounds__(512) write_k(u32x4* __restrict__ dst, int64_t nfrag, int frags_per_wave) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 8;
  u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  for (int64_t f0 = wave * frags_per_wave; f0 < nfrag; f0 += nwaves * frags_per_wave)
    for (int i = 0; i < frags_per_wave && f0 + i < nfrag; ++i) {
      u32x4* p = dst + (f0 + i) * 64 + lane;
      if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
}
// the fragment stores of the training kernels: lane (r = l & 31, h = l >> 5) writes 16 B at byte 32 r + 16 h
template <int NT>
__global__ void __launch_bounds__(512) write_frag_k(u32x4* __restrict__ dst, int64_t nfrag, int frags_per_wave) {
  const int lane = threadIdx.x & 63, pos = 2 * (lane & 31) + (lane >> 5);
  const int64_t wave = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 8;
  u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  for (int64_t f0 = wave * frags_per_wave; f0 < nfrag; f0 += nwaves * frags_per_wave)
    for (int i = 0; i < frags_per_wave && f0 + i < nfrag; ++i) {
      u32x4* p = dst + (f0 + i) * 64 + pos;
      if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
}
template <int NT>
__global__ void __launch_bounds__(512) read_k(const u32x4* __restrict__ src, int64_t nfrag, int frags_per_wave, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 8;
  u32x4 acc = {0, 0, 0, 0};
  for (int64_t f0 = wave * frags_per_wave; f0 < nfrag; f0 += nwaves * frags_per_wave)
#pragma unroll 4
    for (int i = 0; i < frags_per_wave; ++i) {
      if (f0 + i >= nfrag) break;
      const u32x4* p = src + (f0 + i) * 64 + lane;
      const u32x4 v = NT ? __builtin_nontemporal_load(p) : *p;
      acc ^= v;
    }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}
// the weight-gradient kernel's load skeleton: one 512-thread workgroup per CU (148 KiB of LDS), per 32-sample tile
// two 16 KiB runs (dZ and activation fragments) fetched by LDS-DMA into a 4-stage ring, 3 tiles in flight, one
// counted wait + barrier per tile, nothing computed.  tile_stride / second-run offset in 16-byte units.
extern __shared__ __attribute__((aligned(16))) char smem[];
template <int NT, int STAGES>
__global__ void __launch_bounds__(512) dw_like(const u32x4* __restrict__ src, int ntiles, int64_t stride16, int64_t off2_16) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lo = (int)((int64_t)ntiles * blockIdx.x / gridDim.x), hi = (int)((int64_t)ntiles * (blockIdx.x + 1) / gridDim.x);
  const unsigned lds0 = (unsigned)(uintptr_t)((const __attribute__((address_space(3))) char*)smem);
  auto issue = [&](int tile, int stage) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = wv + 8 * k;
      const u32x4* g = src + (int64_t)tile * stride16 + (i < 16 ? i * 64 : off2_16 + (i - 16) * 64) + lane;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + stage * 36864 + i * 1152);
      unsigned keep;
      if (NT) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                           : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
      else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                        : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    }
  };
  for (int s = 0; s < STAGES - 1; ++s) if (lo + s < hi) issue(lo + s, s);
  for (int t = lo; t < hi; ++t) {
    const int rem = hi - 1 - t < STAGES - 2 ? hi - 1 - t : STAGES - 2;      // younger tiles in flight
    if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + STAGES - 1 < hi) issue(t + STAGES - 1, (t - lo + STAGES - 1) % STAGES);
  }
}

int main() {
  const int64_t bytes = 4ll << 30, nfrag = bytes / 1024;
  u32x4* buf; unsigned* sink;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 4);
  hipMemset(buf, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    for (int grid : {256, 512, 1024, 2048, 4096})
      for (int fpw : {4, 16}) {
        launch(grid, fpw); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch(grid, fpw);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-14s grid %4d frags/wave %2d : %.2f TB/s\n", name, grid, fpw, bytes * 5.0 / (ms * 1e-3) / 1e12);
      }
  };
  time("store", [&](int g, int f) { hipLaunchKernelGGL(write_k<0>, dim3(g), dim3(512), 0, 0, buf, nfrag, f); });
  time("store nt", [&](int g, int f) { hipLaunchKernelGGL(write_k<1>, dim3(g), dim3(512), 0, 0, buf, nfrag, f); });
  time("store frag", [&](int g, int f) { hipLaunchKernelGGL(write_frag_k<0>, dim3(g), dim3(512), 0, 0, buf, nfrag, f); });
  time("store frag nt", [&](int g, int f) { hipLaunchKernelGGL(write_frag_k<1>, dim3(g), dim3(512), 0, 0, buf, nfrag, f); });
  time("load", [&](int g, int f) { hipLaunchKernelGGL(read_k<0>, dim3(g), dim3(512), 0, 0, buf, nfrag, f, sink); });
  time("load nt", [&](int g, int f) { hipLaunchKernelGGL(read_k<1>, dim3(g), dim3(512), 0, 0, buf, nfrag, f, sink); });
  hipFuncSetAttribute(reinterpret_cast<const void*>(dw_like<0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 148480);
  hipFuncSetAttribute(reinterpret_cast<const void*>(dw_like<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 148480);
  hipFuncSetAttribute(reinterpret_cast<const void*>(dw_like<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 74752);
  hipFuncSetAttribute(reinterpret_cast<const void*>(dw_like<1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 111616);
  struct Shape { const char* name; int64_t stride16, off2; };
  // contiguous 32 KiB tiles; the real layout (dZ block 154 KiB per tile, activation block in a second buffer ~2 GB away)
  const Shape shapes[] = {{"contiguous", 2048, 1024}, {"strided 160K + 2 GB", 10240, (2ll << 30) / 16}};
  for (const Shape& sh : shapes) {
    const int ntiles = (int)(((sh.off2 > 4096 ? (2ll << 30) : bytes) / 16 - 2048) / sh.stride16);
    const double moved = (double)ntiles * 32768;
    for (int nt = 0; nt < 2; ++nt)
      for (int grid : {256, 1024, 2560}) {
        auto launch = [&]() {
          if (nt) hipLaunchKernelGGL((dw_like<1, 4>), dim3(grid), dim3(512), 148480, 0, buf, ntiles, sh.stride16, sh.off2);
          else hipLaunchKernelGGL((dw_like<0, 4>), dim3(grid), dim3(512), 148480, 0, buf, ntiles, sh.stride16, sh.off2);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("dw-like %-20s %s grid %4d (%d tiles): %.2f TB/s\n", sh.name, nt ? "nt" : "  ", grid, ntiles, moved * 5.0 / (ms * 1e-3) / 1e12);
      }
  }
  {   // two co-resident workgroups per CU with a 2-stage ring each (74 KiB), and one with 3 stages
    const Shape sh = shapes[1];
    const int ntiles = (int)((((2ll << 30)) / 16 - 2048) / sh.stride16);
    const double moved = (double)ntiles * 32768;
    for (int stages : {2, 3})
      for (int grid : {512, 2560}) {
        auto launch = [&]() {
          if (stages == 2) hipLaunchKernelGGL((dw_like<1, 2>), dim3(grid), dim3(512), 74752, 0, buf, ntiles, sh.stride16, sh.off2);
          else hipLaunchKernelGGL((dw_like<1, 3>), dim3(grid), dim3(512), 111616, 0, buf, ntiles, sh.stride16, sh.off2);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("dw-like strided nt, %d stages (%s), grid %4d: %.2f TB/s\n", stages, stages == 2 ? "2 WGs/CU" : "1 WG/CU", grid, moved * 5.0 / (ms * 1e-3) / 1e12);
      }
  }
  return 0;
}
