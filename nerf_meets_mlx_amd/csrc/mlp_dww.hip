// Weight gradients of the 256 x 256 jobs (dZ [256] x H [256]: 16 x 16 fragments) with ONE wave per SIMD, for the split-bf16 stores
// (hi + lo blocks, three MFMAs per product: precision 22) and for the bf16 stores (precision 16).
//
// Reference: the adjoint of models/NeRF.py:201-243 under nn.value_and_grad (entrypoints/__test_nerf.py:240-293); same job table,
// pair-block stages (mlp_s16.hip: the operand fragments of a stage as 1 KiB pair blocks, filled by LDS-DMA, read back with
// ds_read_b64_tr_b16), partial-tile slots and reduce kernel as the 16-wave kernels.  What differs is who computes what:
//
//   16 waves (s16_dw_kernel / mlp_dw_kernel, 128 registers per wave): a wave owns 2 x 2 output tiles and cannot hold a second
//   operand set, so every stage is  barrier -> transposed reads -> wait -> MFMAs  with the 4 waves of a SIMD in lock-step behind
//   the stage barrier: measured for the split form (tools/probe_dw22_chain.py, timing-only builds) 0.78 us of MFMA time per stage
//   grow to 1.1 us without any DMA and to 1.46 us with the loads, against 1.25 us for the loads alone.
//
//   here (4 waves, 512 registers each): a wave owns 4 x 4 output tiles (256 accumulator registers), keeps TWO operand sets
//   (2 x 64 registers) and reads the operands of stage j + 1 between the MFMAs of stage j; three stages stay in flight behind the
//   one being read.  Per stage and CU the LDS delivers 64 KiB instead of 128 (a tile is read by 2 waves, not 4), and everything
//   that is not an MFMA (32 transposed reads, 8 DMAs, 16 v_dot2c for the bias sums per wave) is spread over the MFMA issue gaps
//   of the stage in eight slices (sched_barrier between them), where it is nearly free (MI355X_MICROARCH.md, "price of a filler
//   beside MFMAs").  The stage barrier sits behind the first slice: a wave brings its first MFMAs with it into the wait.
//   Measured (split form, 786 k samples): a 256 x 256 job alone 272 against 305 us, the eight of them 2.08 ms = 6.2 TB/s.
//
// A stage is 32 KiB in both forms: SPLIT 16 samples x (dZ hi | dZ lo | H hi | H lo) x 8 pair blocks, 48 MFMAs per wave
// (lo hi + hi lo + hi hi); bf16 32 samples x (dZ first half | dZ second half | H first half | H second half) x 8 pair blocks,
// 32 MFMAs per wave (one per k-step) -- the block a lo part occupies in the one is the second 16 samples in the other.
//
// Bias gradient: row sums of dZ.  Each wave reads its four dZ tiles in an order rotated by its column (local tile i = tile
// 4 wr + ((i + 2 wc) & 3)) and sums the rows of its local tiles 0 and 1: the two waves of a row cover the row's four tiles with
// the SAME instruction stream (no branch, no select).
//
// Jobs that are not 256 x 256 keep the 16-wave kernels (a 4 x 4-tile wave would multiply zeros for them); mlp.hip launches the
// two kernels one after the other, each over its own static split of all CUs.
#include "mlp_s16_dev.h"

#ifndef NERF_DWX
#define NERF_DWX 0            // timing-only builds (bit mask): 1 no DMAs, 2 no transposed reads, 4 no stage barrier, 8 no bias sums, 16 no MFMAs
#endif

namespace nerf {
using s16::mfma32;

constexpr int DWW_STAGES = 4, DWW_STAGE_BYTES = 32 * 1024, DWW_LDS_BYTES = DWW_STAGES * DWW_STAGE_BYTES;
constexpr int DWW_WAVES = 4, DWW_DPW = 8;                  // 32 pair blocks per stage: 8 DMAs per wave
constexpr int DWW_SLICES = 8;                              // the MFMAs of a stage and wave (48 split, 32 bf16) in 8 slices

struct WOps { bf16x8 ah[4], al[4], bh[4], bl[4]; };        // dZ tiles (hi, lo | samples 0-15, 16-31), H tiles likewise: 64 registers

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;
__device__ __forceinline__ bf16x8 trw(unsigned addr) {     // per-lane LDS byte address of the operand's first half
#if NERF_DWX & 2
  bf16x8 r; asm volatile("" : "=v"(r) : "v"(addr)); return r;
#else
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(uintptr_t)addr);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(uintptr_t)(addr + 256));
  union { struct { s16x4 a, b; } s; bf16x8 v; } cvt;
  cvt.s.a = lo; cvt.s.b = hi;
  return cvt.v;
#endif
}

// one 1 KiB block global -> LDS: wave-uniform 64-bit base + per-lane 32-bit offset; M0 written and left (the unit is built with
// NERF_DMA_CLOBBER_M0 and every kernel of it is scanned by check_m0.py)
__device__ __forceinline__ void dma_block_nt(const char* gbase_uniform, unsigned lane_off, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt"
               :: "v"(lane_off), "s"(gbase_uniform), "s"(lds_addr) : "memory", "m0");
}

template <bool B> struct BoolC { static constexpr bool value = B; };

template <bool SPLIT>
__global__ void __launch_bounds__(64 * DWW_WAVES) mlp_dww_kernel(DwArgs a) {
  constexpr int MPS = SPLIT ? 6 : 4;                       // MFMAs per slice
  char* smem = ring_smem;
  int bj = blockIdx.x, job_id = 0;
  while (bj >= a.splits[job_id]) { bj -= a.splits[job_id]; ++job_id; }
  const DwJob jb = a.jobs[job_id];                         // nf == kf == 16 (mlp.hip:launch_dw)
  const int tile_lo = (int)((int64_t)a.ntiles * bj / a.splits[job_id]);
  const int tile_hi = (int)((int64_t)a.ntiles * (bj + 1) / a.splits[job_id]);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wv >> 1, wc = wv & 1, rot = 2 * wc;
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][k][e] = 0.0f;
  float bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};               // [2 tile + part] of local dZ tiles 0, 1 (part 0 = hi block, 1 = lo)

  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  // transposed reads: lane part of an operand address inside its pair block (mlp_s16.hip:tr_pair)
  const int g16 = lane >> 4, i16 = lane & 15, hq = g16 >> 1, fsel = g16 & 1, q = i16 >> 2, p = i16 & 3;
  const unsigned lane_rd = lds0 + 64 * (8 * hq + q) + 32 * fsel + 16 * (p & 1) + 8 * (p >> 1);
  // stage layout: dZ hi [8] | dZ lo [8] | H hi [8] | H lo [8] KiB; local dZ tiles 0, 1 = tiles 4 wr + rot (+1), 2, 3 = 4 wr + (rot ^ 2) (+1)
  const unsigned rdA0 = lane_rd + (4 * wr + rot) * 1024, rdA1 = lane_rd + (4 * wr + (rot ^ 2)) * 1024;
  const unsigned rdB = lane_rd + (16 + 4 * wc) * 1024;
  // DMAs: pair block wv + 4 k of the stage, k < 4 from dZ, k >= 4 from H; per-lane source offset inside the sample tile's store
  const int src_row = lane >> 2, src_sel = (lane >> 1) & 1, src_half = lane & 1;
  unsigned lane_off[DWW_DPW];
#pragma unroll
  for (int k = 0; k < DWW_DPW; ++k) {
    const int j = wv + 4 * (k & 3), lo_part = j >> 3, t = j & 7;
    // blocks 8..15 of an operand: SPLIT the lo blocks (z_lo / a_lo slots further), bf16 samples 16..31 of the same fragments
    const int slot = (k < 4 ? jb.dz_slot + (SPLIT ? lo_part * a.z_lo : 0) : jb.act_slot + (SPLIT ? lo_part * a.a_lo : 0)) + 2 * t + src_sel;
    lane_off[k] = (unsigned)(slot * 1024 + 32 * src_row + 16 * src_half + (SPLIT ? 0 : 512 * lo_part));
  }
  const char* dzb = reinterpret_cast<const char*>(a.dz);
  const char* acb = reinterpret_cast<const char*>(a.acts);
  long long zstride_b = a.zstride * 16, astride_b = a.astride * 16;
  unsigned ldsw = lds0 + wv * 1024;
  asm volatile("" : "+s"(zstride_b), "+s"(astride_b), "+s"(dzb), "+s"(acb), "+s"(ldsw));

  // stages of this workgroup: SPLIT half tiles (two per sample tile), bf16 sample tiles
  const int ht_lo = SPLIT ? 2 * tile_lo : tile_lo, n = SPLIT ? 2 * (tile_hi - tile_lo) : tile_hi - tile_lo;
  auto issue_k = [&](int ht, int slot, int k) {            // DMA k of stage ht (absolute stage index) into ring slot `slot`
#if !(NERF_DWX & 1)
    const long long tile = SPLIT ? ht >> 1 : ht;
    const char* base = (k < 4 ? dzb + tile * zstride_b : acb + tile * astride_b) + (SPLIT ? 512 * (ht & 1) : 0);
    dma_block_nt(base, lane_off[k], ldsw + slot * DWW_STAGE_BYTES + k * 4096);
#endif
  };
  auto read_tile = [&](WOps& o, int r, unsigned so) {      // operand tile r of a stage (so = slot * 32 KiB) -> registers
    if (r < 4) o.al[r] = trw((r < 2 ? rdA0 : rdA1) + so + (r & 1) * 1024 + 8192);
    else if (r < 8) o.bh[r - 4] = trw(rdB + so + (r - 4) * 1024);
    else if (r < 12) o.ah[r - 8] = trw((r - 8 < 2 ? rdA0 : rdA1) + so + (r & 1) * 1024);
    else o.bl[r - 12] = trw(rdB + so + (r - 12) * 1024 + 8192);
  };
  auto mfma_m = [&](const WOps& o, int m) {                // MFMA m of a stage: product type m / 16, tile (i, k) = (m % 16) / 4, m % 4
#if !(NERF_DWX & 16)
    const int pt = m >> 4, i = (m >> 2) & 3, k = m & 3;
    if (SPLIT) {                                           // lo hi + hi lo + hi hi
      if (pt == 0) acc[i][k] = mfma32(o.al[i], o.bh[k], acc[i][k]);
      else if (pt == 1) acc[i][k] = mfma32(o.ah[i], o.bl[k], acc[i][k]);
      else acc[i][k] = mfma32(o.ah[i], o.bh[k], acc[i][k]);
    } else {                                               // samples 0-15, then 16-31
      if (pt == 0) acc[i][k] = mfma32(o.ah[i], o.bh[k], acc[i][k]);
      else acc[i][k] = mfma32(o.al[i], o.bl[k], acc[i][k]);
    }
#else
    asm volatile("" :: "v"(o.al[m & 3]), "v"(o.ah[m & 3]), "v"(o.bh[m & 3]), "v"(o.bl[m & 3]));
#endif
  };
  auto dot_d = [&](const WOps& o, int d) {                 // bias sum piece d of a stage: tile d / 8, part (d / 4) & 1, dword d % 4
#if !(NERF_DWX & 8)
    const int t = d >> 3, part = (d >> 2) & 1, w = d & 3;
    const bf16x8& v = part ? o.al[t] : o.ah[t];
    const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
    const bf16x2 pr = {v[2 * w], v[2 * w + 1]};
    bsum[2 * t + part] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, bsum[2 * t + part], false);
#endif
  };
  // tiles read / DMAs issued in slice s (slice 0 sits in front of the barrier: MFMAs and bias pieces only)
  constexpr int RD0[DWW_SLICES + 1] = {0, 0, 3, 6, 9, 12, 14, 16, 16};
  constexpr int DM0[DWW_SLICES + 1] = {0, 0, 2, 3, 4, 5, 6, 7, 8};

  // step j: MFMAs on `use` (stage j, in registers); stage j + 1 -> `ld`; stage j + 4 -> the slot stage j leaves
  auto step = [&](WOps& use, WOps& ld, int j, int slot, auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;        // FULL: stages j + 1 .. j + 4 all exist (no branch in the step)
    const bool more = FULL || j + 1 < n, iss = FULL || j + 4 < n;
    // the operands of `use` have arrived.  The BUILTIN, so that hipcc's wait-count pass knows it (an asm wait leaves it counting
    // the previous step's reads as pending and waiting for THIS step's reads in front of this step's MFMAs)
    __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0)
#pragma unroll
    for (int m = 0; m < MPS; ++m) mfma_m(use, m);
    dot_d(use, 0); dot_d(use, 1);
    __builtin_amdgcn_sched_barrier(0);
    // stage j + 1 has landed (this wave's DMAs: the stages issued behind it may stay in flight) ...
    if (FULL) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else {
      const int younger = (n - 1 < j + 3 ? n - 1 : j + 3) - (j + 1);
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#if !(NERF_DWX & 4)
    __builtin_amdgcn_s_barrier();                          // ... for every wave, and every wave has read stage j: its slot is free
#endif
    const unsigned so = (unsigned)(((slot + 1) & 3) * DWW_STAGE_BYTES);
#pragma unroll
    for (int s = 1; s < DWW_SLICES; ++s) {
      if (iss) {
#pragma unroll
        for (int k = DM0[s]; k < DM0[s + 1]; ++k) issue_k(ht_lo + j + 4, slot, k);
      }
      if (more) {
#pragma unroll
        for (int r = RD0[s]; r < RD0[s + 1]; ++r) read_tile(ld, r, so);
      }
#pragma unroll
      for (int m = MPS * s; m < MPS * (s + 1); ++m) mfma_m(use, m);
      dot_d(use, 2 * s); dot_d(use, 2 * s + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  WOps x, y;
  if (n > 0) {
#pragma unroll
    for (int s_ = 0; s_ < DWW_STAGES; ++s_)
      if (s_ < n) {
#pragma unroll
        for (int k = 0; k < DWW_DPW; ++k) issue_k(ht_lo + s_, s_, k);
      }
    if (n >= 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // stage 0 landed for every wave
#pragma unroll
    for (int r = 0; r < 16; ++r) read_tile(x, r, 0u);
  }
  int j = 0;
  for (; j + 5 < n; j += 2) {                              // both steps of the pair are FULL: j + 1 + 4 < n
    step(x, y, j, j & 3, BoolC<true>());
    step(y, x, j + 1, (j + 1) & 3, BoolC<true>());
  }
  for (; j < n; j += 2) {                                  // the last stages
    step(x, y, j, j & 3, BoolC<false>());
    if (j + 1 < n) step(y, x, j + 1, (j + 1) & 3, BoolC<false>());
  }

  const int rr = lane & 31, hh = lane >> 5;
  float* slot = a.partial + (size_t)blockIdx.x * DW_SLOT_FLOATS;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int nt = 4 * wr + ((i + rot) & 3);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float* tile = slot + (8 * nt + 4 * wc + k) * 1024;
#pragma unroll
      for (int e = 0; e < 16; ++e) tile[((e & 3) + 8 * (e >> 2) + 4 * hh) * 32 + rr] = acc[i][k][e];      // 128 B per half wave
    }
  }
  if (jb.b_off >= 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float v = bsum[2 * t + 1] + bsum[2 * t];       // lo + hi
      const float tot = v + __shfl_xor(v, 32, 64);         // the two sample halves of the k-step
      if (hh == 0) slot[64 * 1024 + 32 * (4 * wr + rot + t) + rr] = tot;
    }
  }
}

int launch_dw_wide_kernel(const DwArgs& d, int workgroups, bool split_bf16, hipStream_t s) {
  static DevOnce once[2];
  const void* k = split_bf16 ? reinterpret_cast<const void*>(mlp_dww_kernel<true>) : reinterpret_cast<const void*>(mlp_dww_kernel<false>);
  once[split_bf16].run([&] { (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, DWW_LDS_BYTES); });
  if (split_bf16) hipLaunchKernelGGL(mlp_dww_kernel<true>, dim3(workgroups), dim3(64 * DWW_WAVES), DWW_LDS_BYTES, s, d);
  else hipLaunchKernelGGL(mlp_dww_kernel<false>, dim3(workgroups), dim3(64 * DWW_WAVES), DWW_LDS_BYTES, s, d);
  return check_launch(split_bf16 ? "mlp dW (split bf16, 256 x 256 jobs)" : "mlp dW (bf16, 256 x 256 jobs)");
}

}  // namespace nerf
