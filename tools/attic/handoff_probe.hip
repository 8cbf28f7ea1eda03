// handoff_probe -- go/no-go microbenchmark for the "XCD-level producer/consumer hand-off" training design
// (DESIGN.md 9.1): forward+backward-chain CUs (producers) hand their 1 KiB bf16 fragment blocks (activations / dZ)
// to weight-gradient CUs (consumers) ON THE SAME XCD through that XCD's L2, flag-synchronised, instead of writing them
// to HBM in one kernel and reading them back in the next (today: 21.9 KB per sample).
//
// What it measures, for one persistent launch of one 256-thread workgroup per CU:
//   * the sustained hand-off rate per XCD and for the chip (bytes the consumers received / kernel time),
//   * whether the consumers' loads are served by L2: run it under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
//     (separate passes); FETCH_SIZE ~ 0 means the reads never left the XCD,
//   * correctness of the hand-off under this protocol: every word is checked against the producer's pattern.
//
// Protocol (guide: MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
//   roles by HW_REG_XCC_ID + a per-XCD census, so that a consumer only ever reads producers of its own XCD;
//   producer: plain 16-byte stores of one chunk -> every wave s_waitcnt vmcnt(0) -> workgroup barrier -> lane 0 stores
//             the chunk counter with an agent-scope relaxed atomic (sc1 store);
//   consumer: every wave for itself: lane 0 polls the slot's "produced" word (sc1 load + s_sleep) -> `global_load_dwordx4
//             ... sc1` x 8 in flight (bypass the CU's L1, served by the XCD's L2) -> checks -> lane 0 publishes "consumed".
//   Plain stores KEEP the line in the XCD's L2 (sc1 stores would drop it); there is no agent-scope release because the
//   reader shares the writer's L2 -- this is exactly the property the probe tests (mismatch count must be 0).
//   Every spin has a wall-clock bound: a protocol error ends the kernel with an error code instead of hanging the GPU.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/diag/handoff_probe tools/handoff_probe.hip
//   tools/diag/handoff_probe [ring_KiB_per_producer=128] [chunk_KiB=16] [producers_per_consumer=2] [MiB_per_producer=256] [delay_ns=0] [consumer_waves=16]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Ctl {
  unsigned xcd_count[8];        // census: workgroups per XCD
  unsigned total;               // workgroups registered
  unsigned error;               // nonzero: a spin timed out / roles inconsistent
  unsigned long long mismatches;
  unsigned long long bytes_consumed;
  unsigned consumers, producers;
};

// one per producer: per ring slot a "produced generation" and a "consumed generation" word, each on its own 128-byte line
constexpr int MAX_SLOTS = 64;
struct Chan {
  struct { unsigned v; unsigned pad[31]; } produced[MAX_SLOTS], consumed[MAX_SLOTS];
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t pattern(unsigned prod, unsigned chunk, unsigned word) {
  uint32_t x = prod * 0x9E3779B1u ^ chunk * 0x85EBCA6Bu ^ word * 0xC2B2AE35u;
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
  return x;
}
// wait until *p >= want; false on timeout (2 s of the 100 MHz wall clock)
__device__ __forceinline__ bool spin_ge(const unsigned* p, unsigned want) {
  const unsigned long long t0 = wall_clock64();
  while (ld_agent(p) < want) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > 200000000ull) return false;
  }
  return true;
}
// 8 x 16-byte sc1 loads (L1 bypass) and their wait in ONE asm statement: hipcc does not track vmcnt for asm outputs, so a
// separate s_waitcnt statement may be scheduled after the first use of the registers (cdna_hip_programming.md 5.7)
__device__ __forceinline__ void load8_sc1(const uint4* a, unsigned stride_u4, u32x4 (&v)[8]) {
  const uint4 *a0 = a, *a1 = a + stride_u4, *a2 = a + 2 * stride_u4, *a3 = a + 3 * stride_u4, *a4 = a + 4 * stride_u4,
              *a5 = a + 5 * stride_u4, *a6 = a + 6 * stride_u4, *a7 = a + 7 * stride_u4;
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\t"
      "global_load_dwordx4 %3, %11, off sc1\n\tglobal_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)
      : "memory");
}

// Producers: 256 threads, chunk by chunk behind a workgroup barrier (like a chain kernel that stores a tile's fragments).
// Consumers: every WAVE is an independent reader (like the dW kernel's waves pulling different fragments): wave w takes
// the chunks k = w, w + NW, ... of each of its producers, polls that chunk's "produced" word itself, loads the chunk with
// 8 KiB in flight, and publishes the slot's "consumed" word -- no workgroup barrier on the consumer side.
extern "C" __global__ void __launch_bounds__(1024) handoff_kernel(Ctl* ctl, Chan* chans, uint4* rings, unsigned ring_chunks,
                                                                  unsigned chunk_u4, unsigned nchunks, unsigned ppc,
                                                                  unsigned delay_cycles, unsigned grid_expected, unsigned prod_threads) {
  extern __shared__ char lds_pad[];              // only there to force one workgroup per CU
  __shared__ unsigned s_slot, s_nx, s_ok;
  const unsigned tid = threadIdx.x, nthreads = blockDim.x;
  const unsigned xcc = xcc_id() & 7;
  if (tid == 0) {
    s_slot = atomicAdd(&ctl->xcd_count[xcc], 1u);
    atomicAdd(&ctl->total, 1u);
    s_ok = spin_ge(&ctl->total, grid_expected) ? 1u : 0u;       // census complete: every workgroup is resident
    s_nx = ld_agent(&ctl->xcd_count[xcc]);
    if (!s_ok) atomicOr(&ctl->error, 1u);
  }
  __syncthreads();
  if (!s_ok) return;
  const unsigned slot = s_slot, nx = s_nx;
  // roles inside this XCD: groups of (ppc producers + 1 consumer); leftover workgroups idle
  const unsigned gsz = ppc + 1, ngroups = nx / gsz;
  const unsigned grp = slot / gsz, pos = slot % gsz;
  if (grp >= ngroups) return;
  const bool is_consumer = pos == ppc;
  const unsigned chan0 = (xcc * 32 + grp * ppc);               // channel ids of this group's producers: chan0 .. chan0+ppc-1
  const size_t ring_u4 = (size_t)ring_chunks * chunk_u4;
  if (tid == 0) atomicAdd(is_consumer ? &ctl->consumers : &ctl->producers, 1u);
  if (!is_consumer) {
    if (tid >= prod_threads) return;                            // producers use the first prod_threads threads only
    const unsigned ch = chan0 + pos;
    Chan* c = chans + ch;
    uint4* ring = rings + (size_t)ch * ring_u4;
    for (unsigned k = 0; k < nchunks; ++k) {
      const unsigned rs = k % ring_chunks, gen = k / ring_chunks + 1;
      if (gen > 1) {                                            // the slot we are about to overwrite must have been consumed
        if (tid == 0) s_ok = spin_ge(&c->consumed[rs].v, gen - 1) ? 1u : 0u;
        asm volatile("s_barrier" ::: "memory");
        if (!s_ok) { if (tid == 0) atomicOr(&ctl->error, 2u); return; }
      }
      if (delay_cycles) {                                       // emulate the chain's production rate
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < delay_cycles) __builtin_amdgcn_s_sleep(1);
      }
      uint4* dst = ring + (size_t)rs * chunk_u4;
      for (unsigned i = tid; i < chunk_u4; i += prod_threads) { // 1 KiB per wave instruction, like the fragment stores
        uint4 v;
        v.x = pattern(ch, k, 4 * i); v.y = pattern(ch, k, 4 * i + 1); v.z = pattern(ch, k, 4 * i + 2); v.w = pattern(ch, k, 4 * i + 3);
        dst[i] = v;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have reached L2
      asm volatile("s_barrier" ::: "memory");                   // ... and every other producing wave's
      if (tid == 0) st_agent(&c->produced[rs].v, gen);
    }
    return;
  }
  // consumer waves
  const unsigned lane = tid & 63, wv = tid >> 6, nw = nthreads >> 6;
  unsigned long long bad = 0, bytes = 0;
  for (unsigned k = wv; k < nchunks; k += nw) {
    const unsigned rs = k % ring_chunks, gen = k / ring_chunks + 1;
    for (unsigned p = 0; p < ppc; ++p) {
      const unsigned ch = chan0 + p;
      Chan* c = chans + ch;
      unsigned ok = 1;
      if (lane == 0) ok = spin_ge(&c->produced[rs].v, gen) ? 1u : 0u;
      ok = __builtin_amdgcn_readfirstlane(ok);
      if (!ok) { if (lane == 0) atomicOr(&ctl->error, 4u); return; }
      const uint4* src = rings + (size_t)ch * ring_u4 + (size_t)rs * chunk_u4;
      for (unsigned i0 = lane; i0 < chunk_u4; i0 += 64 * 8) {   // chunk_u4 is a multiple of 512: 8 KiB per wave per group
        u32x4 v[8];
        load8_sc1(src + i0, 64, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned i = i0 + 64 * j;
          bad += (v[j].x != pattern(ch, k, 4 * i)) + (v[j].y != pattern(ch, k, 4 * i + 1)) + (v[j].z != pattern(ch, k, 4 * i + 2)) +
                 (v[j].w != pattern(ch, k, 4 * i + 3));
          bytes += 16;
        }
      }
      if (lane == 0) st_agent(&c->consumed[rs].v, gen);          // this wave's loads have all returned (waited in load8_sc1)
    }
  }
  atomicAdd(&ctl->mismatches, bad);
  atomicAdd(&ctl->bytes_consumed, bytes);
}

int main(int argc, char** argv) {
  const unsigned ring_kib = argc > 1 ? atoi(argv[1]) : 128;
  const unsigned chunk_kib = argc > 2 ? atoi(argv[2]) : 16;
  const unsigned ppc = argc > 3 ? atoi(argv[3]) : 2;
  const unsigned mib_per_prod = argc > 4 ? atoi(argv[4]) : 256;
  const unsigned delay_ns = argc > 5 ? atoi(argv[5]) : 0;
  const unsigned cons_waves = argc > 6 ? atoi(argv[6]) : 16;     // consumer waves per CU (workgroup = 64 * cons_waves threads)
  const unsigned prod_threads = 256;
  if (ring_kib < chunk_kib || ring_kib % chunk_kib || ppc < 1 || ppc > 15 || chunk_kib < 8 || chunk_kib % 8 || ring_kib / chunk_kib > MAX_SLOTS ||
      cons_waves < 4 || cons_waves > 16) { fprintf(stderr, "bad arguments\n"); return 2; }
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const unsigned grid = prop.multiProcessorCount;              // one workgroup per CU (forced by the LDS request below)
  const unsigned ring_chunks = ring_kib / chunk_kib, chunk_u4 = chunk_kib * 64, nchunks = mib_per_prod * 1024 / chunk_kib;
  const size_t nchan = 8 * 32;
  Ctl* ctl; Chan* chans; uint4* rings;
  CHECK(hipMalloc(&ctl, sizeof(Ctl)));
  CHECK(hipMalloc(&chans, nchan * sizeof(Chan)));
  CHECK(hipMalloc(&rings, nchan * (size_t)ring_kib * 1024));
  const int lds = 96 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(handoff_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const unsigned delay_cycles = (unsigned)((double)delay_ns * 0.1);   // wall_clock64: 100 MHz
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("ring_KiB_per_producer,chunk_KiB,producers_per_consumer,MiB_per_producer,delay_ns,consumer_waves,run,producers,consumers,ms,GB_consumed,handoff_GB_per_s_chip,"
         "handoff_GB_per_s_per_XCD,GB_per_s_per_consumer_CU,mismatched_words,error,ring_total_MiB_per_XCD\n");
  for (int run = 0; run < 3; ++run) {
    CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
    CHECK(hipMemset(chans, 0, nchan * sizeof(Chan)));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(handoff_kernel, dim3(grid), dim3(64 * cons_waves), lds, 0, ctl, chans, rings, ring_chunks, chunk_u4, nchunks, ppc,
                       delay_cycles, grid, prod_threads);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    Ctl h;
    CHECK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    const double gb = (double)h.bytes_consumed / 1e9;
    printf("%u,%u,%u,%u,%u,%u,%d,%u,%u,%.3f,%.3f,%.1f,%.1f,%.2f,%llu,%u,%.2f\n", ring_kib, chunk_kib, ppc, mib_per_prod, delay_ns, cons_waves, run,
           h.producers, h.consumers, ms, gb, gb / (ms * 1e-3), gb / (ms * 1e-3) / 8.0, h.consumers ? gb / (ms * 1e-3) / h.consumers : 0.0,
           h.mismatches, h.error, (double)(h.producers / 8) * ring_kib / 1024.0);
    fflush(stdout);
    if (h.error) return 3;
  }
  return 0;
}
