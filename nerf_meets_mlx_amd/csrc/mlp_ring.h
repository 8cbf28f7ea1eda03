// Shared pieces of the fused-MLP kernels (mlp.hip: bf16 operands; mlp22.hip: split-fp16 operands): vector types,
// LDS-DMA helpers, the LDS weight ring (RingW) and the argument block of the forward kernels.
#pragma once
#include "common.h"
#include <atomic>

#ifndef NERF_SPREAD_DMA
#define NERF_SPREAD_DMA 1     // ring refill: one DMA per quarter chunk interval instead of a burst of 4
#endif
#ifndef NERF_ABLATE
#define NERF_ABLATE 0
#endif
// (Round 1 had a switch NERF_EXACT_VMCNT that counted the fragment stores of the last three chunk intervals into the ring's
// vmcnt wait: no gain in the training forward, spills in the backward chain -- and, as the fp32 kernels showed in round 3,
// WRONG in principle: stores complete out of order with respect to loads, so a wait may only count younger LOADS
// (mlp32.hip, frag_wait_n).  Removed.)
// A/B switches (both default on): non-temporal DMA loads in the dW kernel / non-temporal fragment stores
#ifndef NERF_NT_DW_LOADS
#define NERF_NT_DW_LOADS 1
#endif

namespace nerf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PeFreq { float pos[10]; float dir[4]; };

// ------------------------------------------------------------------------------------------
// weight sources: where a wave gets the 1 KiB A-operand fragment `f` of the packed stream from
// ------------------------------------------------------------------------------------------
// one 1 KiB fragment global -> LDS with no VGPR round trip (lane i lands at lds_addr + 16 i).
// Inline asm on purpose: hipcc treats the builtin as a pending LDS write and puts `s_waitcnt vmcnt(0)` in front
// of the next ds_read of the same array, which drains the whole prefetch ring every tile.  Hidden in asm, the
// DMAs are ordered by OUR counted `s_waitcnt vmcnt(N)` + s_barrier (cdna_hip_programming.md 5.7).  M0 carries
// the LDS byte address and is restored because the compiler owns it.
__device__ __forceinline__ void dma_frag(const void* gsrc_lane, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_addr) : "memory");
}
// same with a wave-uniform 64-bit base in SGPRs + a 32-bit per-lane byte offset (no 64-bit VGPR address per fragment,
// which hipcc would otherwise hoist out of persistent loops and spill)
// NERF_DMA_CLOBBER_M0 (per translation unit; on for the one-wave-per-SIMD split kernels, where every scalar instruction of the
// 600 DMAs a wave issues per pass is exposed issue time): M0 is written and left, declared as a clobber, instead of saved and
// restored -- 2 of the 7 scalar instructions per DMA.  Valid while nothing compiler-generated in the kernel reads M0 (these
// kernels have no LDS-DMA builtins, no register-indexed moves, no messages: tools/check_m0.py scans the ISA of the build).
#ifndef NERF_DMA_CLOBBER_M0
#define NERF_DMA_CLOBBER_M0 0
#endif
__device__ __forceinline__ void dma_frag_s(const void* gbase_uniform, unsigned lane_off, unsigned lds_addr) {
#if NERF_DMA_CLOBBER_M0
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(lane_off), "s"(gbase_uniform), "s"(lds_addr) : "memory", "m0");
#else
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_off), "s"(gbase_uniform), "s"(lds_addr) : "memory");
#endif
}
// non-temporal form for bytes that are read exactly once (the dW kernel's dZ / activation stream)
__device__ __forceinline__ void dma_frag_nt(const void* gsrc_lane, unsigned lds_addr) {
  unsigned keep;
#if NERF_NT_DW_LOADS
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_addr) : "memory");
#else
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_addr) : "memory");
#endif
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)((const __attribute__((address_space(3))) char*)p);
}

// (1) straight from global memory / L1: no synchronisation between waves (variants 1 and 2)
struct GlobalW {
  const bf16x8* __restrict__ wlane;    // stream base + lane
  const float* __restrict__ bias;
  __device__ __forceinline__ bf16x8 frag(int f, int) { return wlane[f * 64]; }
  __device__ __forceinline__ void note_stores(int) {}
  __device__ __forceinline__ float4 bias4(int slot) { return *reinterpret_cast<const float4*>(bias + slot); }
};

// (2) LDS ring fed by LDS-DMA and shared by the 8 waves of a workgroup (variant 3).  The stream is consumed
// strictly in order by every wave; chunk = 32 fragments (32 KiB), 4 stages, 3 chunks in flight.  At a chunk
// boundary every wave: waits for ITS share of the chunk (counted vmcnt: the 2 younger chunks stay in flight),
// barrier (everybody's share landed; everybody finished reading the previous chunk), then refills the stage that
// just became free with the chunk 3 ahead (wrapping to the next pass over the weights).
constexpr int RING_CHUNK = 32, RING_STAGES = 4, RING_STAGE_BYTES = RING_CHUNK * 1024;
constexpr int RING_BIAS_OFF = RING_STAGES * RING_STAGE_BYTES;          // fp32 bias slots behind the ring
constexpr int RING_LDS_BYTES = RING_BIAS_OFF + 2560 * 4;

extern __shared__ __attribute__((aligned(16))) char ring_smem[];

// RING_GROUP = fragments per software-pipeline group (one group in use, one in flight); TOTAL = fragments
// consumed per pass (multiple of RING_GROUP)
// CHUNK / STAGES: fragments per ring stage and stages (default 32 x 4 = 128 KiB for one 8-wave workgroup per CU; 16 x 4 =
// 64 KiB lets two independent 4-wave workgroups share a CU, see mlp_fwd_ring_kernel)
// RUN4 (needs NERF_DMA_CLOBBER_M0 and 8 DMAs per wave and chunk): see issue_one.  Opt-in per kernel (the 4-wave rings of the
// split-precision kernels).
// SPREAD2: a ring with RING_GROUP 2 also spreads its refill DMAs over the chunk interval (off for the training forwards, which were tuned
// with the burst; on for the 48-sample split-fp16 forward, which has no registers for groups of 4)
template <int NCHUNK, int TOTAL, int RING_GROUP = 4, int NW = 8, int CHUNK = RING_CHUNK, int STAGES = RING_STAGES, bool RUN4 = false, bool SPREAD2 = false>
struct RingW {
  static constexpr int DPW = CHUNK / NW;               // DMAs per wave per chunk
  static constexpr int STAGE_BYTES = CHUNK * 1024, BIAS_OFF = STAGES * STAGE_BYTES, LDS_BYTES = BIAS_OFF + 2560 * 4;
  static_assert(DPW * NW == CHUNK && (DPW == 4 || DPW == 8), "ring: 4 or 8 DMAs per wave per chunk");
  const char* __restrict__ wsrc;       // global stream base (uniform)
  unsigned lane16;                     // 16 * lane
  unsigned lds0;                       // LDS byte address of ring_smem (M0 values are absolute)
  int wv;                              // wave id in the workgroup (uniform)
  int ring_pos;                        // stage of the chunk the prefetch reads from
  int woff;                            // ring_pos * STAGE + 16 * lane
  bf16x8 cur[RING_GROUP], nxt[RING_GROUP];

  // this wave's k-th (of DPW) share of `chunk`: fragments wv + NW k -- or, with M0-clobbering DMA statements and 8 DMAs per
  // wave and chunk (the 4-wave rings of the split-precision kernels), the CONSECUTIVE fragments DPW wv + k: a run of four then
  // shares one M0 write, one hazard nop and one source base, the other three are a bare DMA with an immediate offset, which moves
  // BOTH the global and the LDS address (1 KiB per fragment; 13-bit signed: up to 3 KiB).  The DMAs of a chunk are issued in order
  // k = 0..7 and nothing else writes M0 in these kernels (tools/check_m0.py)
  const char* run_base;                // source base of the current run of four (NERF_DMA_CLOBBER_M0, DPW == 8)
  __device__ __forceinline__ void issue_one(int chunk, int stage, int k) {
#if NERF_ABLATE == 20         // timing-only build 20: every refill is a BARE LDS-DMA (no M0 write, no hazard nop, stale base): the bookkeeping's share
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(lane16), "s"(wsrc) : "memory");
    return;
#endif
#if NERF_DMA_CLOBBER_M0
    if (RUN4 && DPW == 8) {
      const int i = wv * DPW + k;
      if ((k & 3) == 0) {
        run_base = wsrc + ((int64_t)chunk * CHUNK + i) * 1024;
        dma_frag_s(run_base, lane16, lds0 + stage * STAGE_BYTES + i * 1024);
      } else {
        // ONE asm template with an immediate operand (k is a constant at every call site after unrolling).  A switch over three
        // templated statements compiled to the same DMAs but tipped hipcc's register allocation of the split-bf16 chain kernel
        // into scratch (857 scratch instructions, +3 ms, all tests green): the Makefile scan now fails the build on that.
#ifdef NERF_RING_SWITCH_DMA      // diagnostic form (a run-time k compiles): three literal statements
        switch (k & 3) {
          case 1: asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(lane16), "s"(run_base) : "memory"); break;
          case 2: asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" :: "v"(lane16), "s"(run_base) : "memory"); break;
          default: asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(lane16), "s"(run_base) : "memory"); break;
        }
#else
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" :: "v"(lane16), "s"(run_base), "i"((k & 3) * 1024) : "memory");
#endif
      }
      return;
    }
#endif
    const int i = wv + NW * k;
    dma_frag_s(wsrc + ((int64_t)chunk * CHUNK + i) * 1024, lane16, lds0 + stage * STAGE_BYTES + i * 1024);
  }
  __device__ __forceinline__ void issue(int chunk, int stage) {
#pragma unroll
    for (int k = 0; k < DPW; ++k) issue_one(chunk, stage, k);
  }
  // Whole chunks refill their freed stage one DMA per quarter of the interval (a burst of 4 right behind the barrier
  // stalls both waves of a SIMD on the VMEM issue path at once); the partial last chunk of a pass keeps the burst.
  // (not in the activation-storing training forward, RING_GROUP 2: it is at the VGPR limit and HBM-bound anyway)
  // RING_GROUP 8 (a group = 8 fragments = a quarter chunk: reads issued twice as far ahead of their use): two refill DMAs per group
  static constexpr int DMA_PER_GROUP = RING_GROUP * DPW / CHUNK;        // 1 (group 4, DPW 8) or 2 (group 8, DPW 8)
  static constexpr bool spread(int c) { return NERF_SPREAD_DMA && (RING_GROUP == 4 || (RING_GROUP == 2 && SPREAD2) || (RING_GROUP == 8 && DPW == 8)) && (c + 1) * CHUNK <= TOTAL; }
  __device__ __forceinline__ void boundary(int c, int lane) {
    ring_pos = (ring_pos + 1) & (STAGES - 1);
#if NERF_ABLATE == 1          // timing-only: no workgroup barrier (results are garbage)
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
#else
    if (DPW == 8) {           // 4-wave workgroups: 8 DMAs per wave per chunk, two younger chunks stay in flight
      asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
    // This chunk's DMAs were issued three boundaries ago; the LOADS younger than them are the 8 DMAs of the next two
    // chunks.  Loads (LDS-DMA included) return in issue order among themselves, so vmcnt(8) cannot pass while one of this
    // chunk's DMAs is pending (then all 8 younger ones are too).  Stores share the counter but complete out of order with
    // respect to loads: they may NOT be counted among the operations allowed to stay in flight (a store that completes
    // early would let the wait pass too soon); pending stores simply count against the 8.  Loads the compiler issues
    // itself only make the true count larger, so this never under-waits.
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#endif
#if NERF_ABLATE != 2          // timing-only build 2: no refill DMA (stale weights)
    if (spread(c)) {
      issue_one((c + STAGES - 1) % NCHUNK, (ring_pos + STAGES - 1) & (STAGES - 1), 0);
      if (RING_GROUP == 8) issue_one((c + STAGES - 1) % NCHUNK, (ring_pos + STAGES - 1) & (STAGES - 1), 1);
    } else issue((c + STAGES - 1) % NCHUNK, (ring_pos + STAGES - 1) & (STAGES - 1));
#endif
    woff = ring_pos * STAGE_BYTES + 16 * lane;
  }
  // fetch the group that starts at fragment fn (fn % RING_GROUP == 0) into nxt; crossing into a new chunk first
  // runs the ring protocol for it (the previous chunk's last group is already in registers)
  __device__ __forceinline__ void prefetch(int fn, int lane) {
    if ((fn % CHUNK) == 0) boundary(fn / CHUNK, lane);
#if NERF_ABLATE != 2
    else if (RING_GROUP == 8 && spread(fn / CHUNK)) {
      issue_one((fn / CHUNK + STAGES - 1) % NCHUNK, (ring_pos + STAGES - 1) & (STAGES - 1), (fn % CHUNK) / (CHUNK / DPW));
      issue_one((fn / CHUNK + STAGES - 1) % NCHUNK, (ring_pos + STAGES - 1) & (STAGES - 1), (fn % CHUNK) / (CHUNK / DPW) + 1);
    } else if ((fn % (CHUNK / DPW)) == 0 && spread(fn / CHUNK))
      issue_one((fn / CHUNK + STAGES - 1) % NCHUNK, (ring_pos + STAGES - 1) & (STAGES - 1),
                (fn % CHUNK) / (CHUNK / DPW));
#endif
#if NERF_ABLATE != 41         // timing-only build 41 (with the MFMAs and epilogues of mlp22.hip compiled out): the ring protocol alone
#pragma unroll
    for (int i = 0; i < RING_GROUP; ++i)
      nxt[i] = *reinterpret_cast<const bf16x8*>(ring_smem + woff + ((fn + i) % CHUNK) * 1024);
#endif
  }
  __device__ __forceinline__ void note_stores(int) {}
  __device__ __forceinline__ void new_pass() {
    // the ~1200 chunk/fragment source addresses are loop-invariant; hoisted, they no longer fit the SGPR file and
    // are parked in VGPR lanes (v_writelane / v_readlane per DMA).  Opaque base per pass: two s_add per DMA instead.
    asm volatile("" : "+s"(wsrc));
  }
  __device__ __forceinline__ void start(int lane) {
    ring_pos = STAGES - 1;
    woff = 0;
#pragma unroll
    for (int c = 0; c < STAGES - 1; ++c) issue(c, c);
    prefetch(0, lane);
  }
  __device__ __forceinline__ bf16x8 frag(int f, int lane) {
    if ((f % RING_GROUP) == 0) {
#pragma unroll
      for (int i = 0; i < RING_GROUP; ++i) cur[i] = nxt[i];
      prefetch((f + RING_GROUP) % TOTAL, lane);           // wraps to the next pass over the weights
      __builtin_amdgcn_sched_barrier(0);                  // keep hipcc from hoisting further groups (spills)
    }
    return cur[f % RING_GROUP];
  }
  __device__ __forceinline__ float4 bias4(int slot) {
    return *reinterpret_cast<const float4*>(ring_smem + BIAS_OFF + slot * 4);
  }
  __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// Diagnostic build only (make stamp; tools/probe_launch_profile.py): where a launch of a persistent ring kernel spends its time --
// per workgroup, the 100 MHz wall clock (s_memrealtime: one counter for the chip) at kernel entry, at the start of the pass loop
// (ring primed, biases in LDS), after passes 1, 2, 4, 8, ... 128 and at the end, the shader clock (s_memtime) at four of those
// points, the pass count and the hardware id (XCC, SE, CU).  One lane of wave 0 stores them; no output depends on them; in the
// shipped build none of this exists.
#ifdef NERF_CLOCK_STAMP
static __device__ unsigned long long g_st2[4096][24];
struct Stamp2 {
  unsigned long long passes = 0;
  __device__ __forceinline__ void put(int k, unsigned long long v) const {
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_st2[blockIdx.x][k] = v;
  }
  __device__ __forceinline__ void entry() const {
    put(0, __builtin_amdgcn_s_memrealtime()); put(11, __builtin_amdgcn_s_memtime());
    put(15, ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32)      // HW_REG_XCC_ID
            | __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));                               // HW_REG_HW_ID
  }
  __device__ __forceinline__ void loop_start() const { put(1, __builtin_amdgcn_s_memrealtime()); put(12, __builtin_amdgcn_s_memtime()); }
  __device__ __forceinline__ void pass_done() {
    ++passes;
    if ((passes & (passes - 1)) == 0 && passes <= 128) {
      put(2 + (63 - __builtin_clzll(passes)), __builtin_amdgcn_s_memrealtime());
      if (passes == 1) put(16, __builtin_amdgcn_s_memtime());
      if (passes == 8) put(17, __builtin_amdgcn_s_memtime());
      if (passes == 32) put(18, __builtin_amdgcn_s_memtime());
    }
  }
  __device__ __forceinline__ void end() const {
    put(10, __builtin_amdgcn_s_memrealtime()); put(13, __builtin_amdgcn_s_memtime()); put(14, passes);
  }
};
#define NERF_STAMP2_DECL() Stamp2 st2; st2.entry()
#define NERF_STAMP2_LOOP() st2.loop_start()
#define NERF_STAMP2_PASS() st2.pass_done()
#define NERF_STAMP2_END() st2.end()
#define NERF_STAMP2_EXPORT(name) extern "C" int name(unsigned long long* host_out, int nwg) { \
  if (nwg > 4096) nwg = 4096; \
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nerf::g_st2), (size_t)nwg * 24 * 8) == hipSuccess ? 0 : -4; }
#else
#define NERF_STAMP2_DECL()
#define NERF_STAMP2_LOOP()
#define NERF_STAMP2_PASS()
#define NERF_STAMP2_END()
#define NERF_STAMP2_EXPORT(name)
#endif

template <class WS> struct is_ring { static constexpr bool value = false; };
// wave id of a ring source (the pass queue's wave 0); the L1 / LDS-resident sources never run a queue
template <class WS> __device__ __forceinline__ int ws_wave(const WS&) { return 0; }
template <int N, int T, int G, int W, int C, int S, bool R, bool P> __device__ __forceinline__ int ws_wave(const RingW<N, T, G, W, C, S, R, P>& w) { return w.wv; }
template <int N, int T, int G, int W, int C, int S, bool R, bool P> struct is_ring<RingW<N, T, G, W, C, S, R, P>> { static constexpr bool value = true; };

template <class WS>
__device__ __forceinline__ bf16x8 next_frag(WS& ws, int f, int lane) { return ws.frag(f, lane); }


// ------------------------------------------------------------------------------------------
// Dynamic pass queue of the persistent ring kernels (round 6; tools/probe_launch_profile.py).
// With a static split (pass = blockIdx.x + k gridDim.x) a launch ends when the slowest XCD has finished its share: the eight XCDs of
// an MI355X hold different clocks under load (stamps: all 32 workgroups of an XCD end within 10 us of one another, the XCDs 3-8 %
// apart -- which ones are slow changes with what ran before), i.e. the tail of a launch is 130-330 us of 5-15 ms.  Here a
// workgroup runs pass blockIdx.x first and takes every further pass from ONE device-wide counter.
//   * ask(): lane 0 of wave 0 asks for the workgroup's NEXT pass with a returning atomic at the very head of the current pass, BEFORE
//     the pass's own input loads.  Those loads are waited for at the head anyway (the one vmcnt(0) a pass has always had: hipcc does
//     not see the ring's LDS-DMAs and waits for its own loads with a full drain), and the atomic, being older, is covered by the
//     same wait: no additional drain.  (A first form issued the atomic mid-pass from inline asm into a sentinel-primed VGPR --
//     tools/ticket_probe.hip; as an `if` it split the pass into two basic blocks, hipcc sank the direction encodings across the
//     split and waited for the rays' loads mid-pass: one more drain per pass.  This form needs no asm atomic and no scanner.)
//   * publish(): behind the first use of the inputs, wave 0 writes the next pass index to one of TWO LDS words (pass parity), branch-free
//     (the EXEC mask selects wave 0 or nobody); next(): at the end of the pass every wave reads it.  The >= 37 chunk barriers of the
//     ring between the two order write and read; the parity keeps wave 0's write of the following pass off a word a slower wave has
//     not read yet.  All LDS traffic is inline asm: through a C++ pointer hipcc loses the address space and emits flat_store /
//     flat_load + vmcnt(0).
//   * The counter lives in a slot of a small device array (PASSQ_SLOTS per translation unit and device, taken round-robin by the
//     host: launches in flight at the same time never share one); the last workgroup to leave a launch clears its slot.
//   * nerf_set_option("pass_queue", 0) = static split (queue pointer null): same kernel, same results -- which workgroup computes a
//     pass changes nothing a pass computes.
// ------------------------------------------------------------------------------------------
constexpr int PASSQ_SLOTS = 256;
constexpr int PASSQ_LDS_OFF_FLOAT = 2558;      // two LDS words: the last two floats of the 2560-float bias area (every stream's bias count is <= 2496)
static __device__ unsigned g_passq[PASSQ_SLOTS][4];
struct PassQueue {
  unsigned* q;            // [0] passes handed out, [1] workgroups that have left; nullptr: static split
  unsigned lds_words;     // LDS byte address of the two words
  unsigned parity;        // of the pass being executed (wave-uniform)
  unsigned t;             // lane 0 of wave 0: the returning atomic's destination
  __device__ __forceinline__ void init(unsigned* queue, unsigned lds_bias_addr) { q = queue; lds_words = lds_bias_addr + 4 * PASSQ_LDS_OFF_FLOAT; parity = 0u; t = 0u; }
  __device__ __forceinline__ void ask(int wv, int lane) {
    t = 0u;
    if (q && wv == 0 && lane == 0) t = atomicAdd(q, 1u);
  }
  __device__ __forceinline__ void publish(int wv) {
    const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((q && wv == 0) ? 1 : 0);
    const unsigned v = gridDim.x + (unsigned)__builtin_amdgcn_readfirstlane((int)t);
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, %3\n\ts_mov_b32 exec_hi, 0\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(lds_words + 4u * parity), "v"(v), "s"(mask) : "memory");
  }
  __device__ __forceinline__ int64_t next(int64_t sp) {
    if (!q) return sp + gridDim.x;
    unsigned r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lds_words + 4u * parity) : "memory");
    parity ^= 1u;
    return (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)r);
  }
  // end of the kernel: the last workgroup to leave clears the slot for its next user
  __device__ __forceinline__ void leave() const {
    if (q && threadIdx.x == 0) {
      __threadfence();
      if (atomicAdd(&q[1], 1u) == gridDim.x - 1) { q[0] = 0u; q[1] = 0u; }
    }
  }
};
// host: the queue slot of the next launch of this translation unit on the current device (nullptr: static split)
extern int g_pass_queue;                       // mlp.hip: nerf_set_option("pass_queue")
static inline unsigned* passq_slot() {
  if (!g_pass_queue) return nullptr;
  static unsigned* base[64] = {};
  static DevOnce once;
  static std::atomic<unsigned> next{0};
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
  once.run([&] { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_passq)) == hipSuccess) base[d] = static_cast<unsigned*>(p); });
  return base[d] ? base[d] + 4 * (next.fetch_add(1u) % PASSQ_SLOTS) : nullptr;
}

struct FwdArgs {
  unsigned* queue;       // dynamic pass queue slot of this launch, or nullptr (static split)
  const bf16x8* wf;      // forward fragment stream
  const float* bias;     // bias slots
  const float* x;        // MODE 0: [M,90]
  const float* rays;     // MODE 1: [B,11]
  const float* z;        // MODE 1: [B,n]
  int64_t M;
  int n;
  PeFreq fr;
  float* out;            // [M,4]
  void* acts;            // training store or nullptr
  int64_t astride;       // 16-byte units between sample tiles of the activation store
};

__device__ __forceinline__ void ring_load_bias(const float* __restrict__ bias, int count, int bias_off = RING_BIAS_OFF) {
  for (int i = threadIdx.x; i < count; i += blockDim.x)
    *reinterpret_cast<float*>(ring_smem + bias_off + 4 * i) = bias[i];
}

}  // namespace nerf
