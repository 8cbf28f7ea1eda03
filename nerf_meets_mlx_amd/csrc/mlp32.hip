// fp32 reference-precision mode of the fused 8 x 256 NeRF MLP (nerf_mlp_arch.precision == 32).
//
// The reference computes in float32 (MLX default dtype; models/NeRF.py:201-243, embedding.py:30-71): this file is the
// same fused chain as mlp.hip with float32 operands on v_mfma_f32_32x32x2_f32 (fp32 matrix peak 157 TFLOP/s, 1/16 of
// the bf16 rate), so that "same results as the reference on the same inputs" can be checked at the reference's own
// arithmetic (<= 1e-4 of the output scale against the fp32 oracle; no rounding-induced ReLU flips in the adjoint).
// bf16 stays the benchmarked mode; this one is the precision witness and is written for clarity first:
//
//  * one wave = one 32-sample tile.  Layers are computed transposed like in mlp.hip (A = W rows, B = activations with
//    the sample on the lane), K = 2 features per MFMA: step i of k-tile kt multiplies features 32 kt + p(i) + 4 h
//    (h = lane half, p(i) = (i & 3) + 8 (i >> 2)), which is exactly the row a lane's accumulator register i holds -- a
//    finished accumulator tile IS the next layer's B operand, register for register.
//  * the n-tile loop is a run-time loop (the fully unrolled chain would be 9 280 MFMAs of straight-line code): each
//    finished 32 x 32 tile goes to a wave-private 32 KiB LDS slab and the whole layer is read back into registers
//    before the next one.  No workgroup barriers: the four waves of a workgroup never share data.
//  * weights stream from a packed float32 image in consumption order, one coalesced 4 KiB fragment (16 floats per
//    lane) per 16 MFMAs, straight from L2 into a ring of S register sets, D = S - 1 fragments ahead (asm loads, counted
//    waits: frag_wait_n).
//  * training stores every layer's activation / dZ as float32 [tile][feature row][32 samples]; the dW kernel reads them
//    as MFMA operands directly (samples are the K axis; a lane reads 16 consecutive samples of its feature row).  A
//    layer's rows are stored by the NEXT layer, one 1 KiB piece per MFMA step (struct Deferred); the backward chain takes
//    its ReLU decisions from sign bits the forward keeps (A_MASK), not from the stored rows.
//  * encodings use sinf / cosf on the float32 product x * f -- the reference's sin(x * f), not the hardware
//    revolution-sine of the bf16 path.
#include "common.h"
#include "mlp_params.h"
#include "mlp_arch2.h"
#include "mlp32.h"
#include <type_traits>
#include <utility>

namespace nerf {
namespace f32 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef NERF_F32_ABLATE
#define NERF_F32_ABLATE 0
#endif

__host__ __device__ constexpr int prow(int i) { return (i & 3) + 8 * (i >> 2); }     // + 4 h

// tail of the packed image (floats)
constexpr int T_B = 0;              // pos biases, 256 l
constexpr int T_BF = 2048, T_BD = 2304, T_BA = 2432, T_BR = 2433, T_WA = 2440, T_WR = 2696;
// image model (no view head): output_linear [out_ch <= 4][256] at T_WO, its bias at T_BO (the view heads' slots are unused there)
constexpr int T_BO = T_BA, T_WO = T_WA;
constexpr int BIAS_FLOATS = T_WO + 1024;  // the forward keeps the whole tail in LDS (biases, alpha and rgb heads | output head)
static_assert(T_WR + 384 <= BIAS_FLOATS && BIAS_FLOATS <= TAIL_FLOATS, "tail too small");
// forward stream fragment bases
constexpr int F_L0 = 0, F_L1 = 16, F_L5 = 272, F_L6 = 352, F_FEAT = 480, F_DIR = 544;
static_assert(F_DIR + 36 == F_FRAGS, "forward stream");
// backward (transposed) stream
constexpr int B_DIR = 0, B_FEAT = 32, B_POS = 96;          // pos7, 6, ..., 1 at B_POS + 64 (7 - l)
static_assert(B_POS + 7 * 64 == B_FRAGS, "backward stream");
// activation / dZ store rows
constexpr int A_PE = 0, A_DPE = 64, A_H0 = 96, A_FEAT = 2144, A_HD = 2400;
constexpr int A_MASK = 2528;        // ReLU sign bits of H0..H7: 8 rows (1 KiB) per layer, 16 bytes per lane (see relu_bits)
constexpr int Z_L0 = 0, Z_F = 2048, Z_D = 2304, Z_A = 2432, Z_RGB = 2464;
static_assert(A_HD + 128 == A_MASK && A_MASK + 64 == A_ROWS && Z_RGB + 32 == Z_ROWS, "store rows");

// ------------------------------------------------------------------------------------------ packing
// img: the image-fitting model (entrypoints/__viser_image_learning.py:203-208: input 40, pos5 [256 x 296], no view head) uses the
// trunk part of the same streams (forward fragments [0, F_FEAT), transposed fragments [B_POS, B_FRAGS)); the rest stays zero
__device__ float fwd_src(const float* __restrict__ p, int f, int n32, int kk, bool img) {
  const int cin = img ? LI::CIN : 63, ld5 = 256 + cin;
  if (f < F_L1) {
    const int nt = f / 2, kt = f % 2, k = 32 * kt + kk;
    return k < cin ? p[(img ? LI::P_W0 : L::P_W0) + (32 * nt + n32) * cin + k] : 0.0f;
  }
  if (f < F_L5) {
    const int q = f - F_L1, l = 1 + q / 64, r = q % 64;
    return p[(img ? LI::pw(l) : L::pw(l)) + (32 * (r / 8) + n32) * 256 + 32 * (r % 8) + kk];
  }
  if (f < F_L6) {
    const int q = f - F_L5, n = 32 * (q / 10) + n32, kt = q % 10, w5 = img ? LI::P_W5 : L::P_W5;
    if (kt < 2) { const int k = 32 * kt + kk; return k < cin ? p[w5 + n * ld5 + k] : 0.0f; }
    return p[w5 + n * ld5 + cin + 32 * (kt - 2) + kk];
  }
  if (f < F_FEAT) {
    const int q = f - F_L6, l = 6 + q / 64, r = q % 64;
    return p[(img ? LI::pw(l) : L::pw(l)) + (32 * (r / 8) + n32) * 256 + 32 * (r % 8) + kk];
  }
  if (img) return 0.0f;
  if (f < F_DIR) {
    const int q = f - F_FEAT;
    return p[L::P_WF + (32 * (q / 8) + n32) * 256 + 32 * (q % 8) + kk];
  }
  const int q = f - F_DIR, n = 32 * (q / 9) + n32, kt = q % 9;
  if (kt < 8) return p[L::P_WD + n * 283 + 32 * kt + kk];
  return kk < 27 ? p[L::P_WD + n * 283 + 256 + kk] : 0.0f;
}
// transposed: A row = input feature 32 kt + k32, K index = output feature nn
__device__ float bwd_src(const float* __restrict__ p, int f, int k32, int kk, bool img) {
  if (img) {
    if (f < B_POS) return 0.0f;
    const int q = f - B_POS, l = 7 - q / 64, r = q % 64, kt = r / 8, ns = r % 8, nn = 32 * ns + kk, row = 32 * kt + k32;
    return l == 5 ? p[LI::P_W5 + nn * 296 + 40 + row] : p[LI::pw(l) + nn * 256 + row];
  }
  if (f < B_FEAT) { const int kt = f / 4, ns = f % 4; return p[L::P_WD + (32 * ns + kk) * 283 + 32 * kt + k32]; }
  if (f < B_POS) { const int q = f - B_FEAT, kt = q / 8, ns = q % 8; return p[L::P_WF + (32 * ns + kk) * 256 + 32 * kt + k32]; }
  const int q = f - B_POS, l = 7 - q / 64, r = q % 64, kt = r / 8, ns = r % 8, nn = 32 * ns + kk, row = 32 * kt + k32;
  return l == 5 ? p[L::P_W5 + nn * 319 + 63 + row] : p[L::pw(l) + nn * 256 + row];
}

__global__ void __launch_bounds__(256) pack32_kernel(const float* __restrict__ p, float* __restrict__ out, int img_out_ch) {
  const bool img = img_out_ch > 0;
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int nfl = (F_FRAGS + B_FRAGS) * 64;
  if (tid < nfl) {
    const int f = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    float* dst = out + (int64_t)tid * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kk = prow(i) + 4 * h;
      dst[i] = f < F_FRAGS ? fwd_src(p, f, r, kk, img) : bwd_src(p, f - F_FRAGS, r, kk, img);
    }
  } else if (tid < nfl + TAIL_FLOATS) {
    const int t = tid - nfl;
    float v = 0.0f;
    if (img) {
      if (t < T_BF) v = p[LI::pb(t >> 8) + (t & 255)];
      else if (t >= T_BO && t < T_BO + img_out_ch) v = p[LI::P_WO + img_out_ch * 256 + (t - T_BO)];
      else if (t >= T_WO && t < T_WO + img_out_ch * 256) v = p[LI::P_WO + (t - T_WO)];
    }
    else if (t < T_BF) v = p[L::pb(t >> 8) + (t & 255)];
    else if (t < T_BD) v = p[L::P_BF + (t - T_BF)];
    else if (t < T_BA) v = p[L::P_BD + (t - T_BD)];
    else if (t == T_BA) v = p[L::P_BA];
    else if (t < T_BR + 3) v = p[L::P_BR + (t - T_BR)];
    else if (t >= T_WA && t < T_WA + 256) v = p[L::P_WA + (t - T_WA)];
    else if (t >= T_WR && t < T_WR + 384) v = p[L::P_WR + (t - T_WR)];
    out[(int64_t)nfl * 16 + t] = v;
  }
}

// ------------------------------------------------------------------------------------------ one layer
extern __shared__ __attribute__((aligned(16))) float slab_smem[];      // 4 waves x 256 rows x 32 floats

// A fragment (16 floats per lane) as four 128-bit registers loaded by INLINE-ASM global loads: the weight stream of the chain
// kernels and the operand tiles of the dW kernel.
// hipcc's scheduler, at 250+ live registers, sinks ordinary loads to just before their first use (load; s_waitcnt vmcnt(0);
// 4 MFMAs; load; ... -- every L2 round trip exposed, one wave per SIMD, nothing else to issue): measured 40-60 % of the
// fp32 matrix peak.  With asm loads the issue point is ours: fragments are requested D steps (of 16 MFMAs = 1024 cycles) ahead
// and the counted wait (frag_wait_n) -- the only place their registers become visible to the compiler ("+v") -- sits behind
// the MFMAs of the step before their use.
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct FragQ { f32x4 q[4]; };
__device__ __forceinline__ void frag_issue(FragQ& f, const float4* __restrict__ lane_base, int frag) {
  const float4* p = lane_base + (int64_t)frag * 256;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f.q[0]) : "v"(p) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(f.q[1]) : "v"(p) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(f.q[2]) : "v"(p) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(f.q[3]) : "v"(p) : "memory");
}
__device__ __forceinline__ void frag_issue_at(FragQ& f, const float* __restrict__ p) {       // 16 consecutive floats per lane
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f.q[0]) : "v"(p) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(f.q[1]) : "v"(p) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(f.q[2]) : "v"(p) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(f.q[3]) : "v"(p) : "memory");
}
// the same loads into ACCUMULATION registers (gfx950: vector-memory loads can target AGPRs and MFMA reads A / B from either
// file).  With more fragments in flight than the 256 VGPRs hold, hipcc parks the surplus in AGPRs -- with v_accvgpr_write
// straight behind the asm load, i.e. BEFORE the data has arrived (it cannot know the load is asynchronous).  Sets that are
// to live in AGPRs are therefore loaded there in the first place.  (tools/check_inflight_regs.py scans the ISA for any
// read of a load's destination ahead of the wait that covers it.)
__device__ __forceinline__ void frag_issue_at_acc(FragQ& f, const float* __restrict__ p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(f.q[0]) : "v"(p) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=a"(f.q[1]) : "v"(p) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=a"(f.q[2]) : "v"(p) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=a"(f.q[3]) : "v"(p) : "memory");
}
__device__ __forceinline__ float* store_row(float* base, int64_t tile, int rows, int row, int col) {
  return base + ((tile * rows + row) * 32 + col);
}

// The slab's [row][32 samples] image of a finished layer IS the layout of its rows in the activation / dZ store
// ([tile][row][32 samples] float32): copy it with 16 bytes per lane (1 KiB per wave instruction) instead of one
// 4-byte store per lane and element -- 4 x fewer store instructions in the training kernels.
__device__ __forceinline__ void slab_to_store(const float* slab, float* dst, int rows) {
  const int lane = threadIdx.x & 63;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this wave's own slab writes
  const float4* s4 = reinterpret_cast<const float4*>(slab);
  float4* d4 = reinterpret_cast<float4*>(dst);
  for (int i = lane; i < rows * 8; i += 64) d4[i] = s4[i];
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// wait until at most N vector-memory operations are outstanding, N = the number of LOADS issued after this fragment's.
// vmcnt counts loads and stores in one counter; loads return in issue order among themselves, stores complete out of order
// with respect to loads.  While one of the fragment's loads is pending so are the N younger ones (count > N): the wait is
// safe whatever the stores do -- counting stores among the "younger operations that may stay in flight" is NOT (measured: a
// store that completes early lets such a wait pass with the fragment still in flight).  Pending stores do count against N,
// so with the younger loads landed (they were issued 1024+ cycles earlier) up to N stores may still be on their way.
template <int N>
__device__ __forceinline__ void frag_wait_n(FragQ& f) {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(f.q[0]), "+v"(f.q[1]), "+v"(f.q[2]), "+v"(f.q[3]) : "n"(N) : "memory");
}
// Deferred stores of the PREVIOUS layer's rows (still intact in the wave's slab) during the next layer's MFMA steps: one
// 1 KiB piece (8 rows) per step.  The piece is read from LDS one step ahead, so the store issues without an LDS wait.
struct Deferred {
  const f32x4* src; f32x4* dst; int n;              // lane-offset pointers; n pieces in all
  f32x4 piece;
  __device__ __forceinline__ void init(const float* slab, float* to, int rows, int lane) {
    src = reinterpret_cast<const f32x4*>(slab) + lane; dst = reinterpret_cast<f32x4*>(to) + lane;
    n = __builtin_amdgcn_readfirstlane(to ? rows / 8 : 0);        // wave-uniform (every lane has the same `to`)
#if NERF_F32_ABLATE == 1     // timing-only build (make ab AB_FLAGS=-DNERF_F32_ABLATE=1): no deferred stores; results are wrong
    n = 0;
#endif
    if (n > 0) piece = *src;
  }
  __device__ __forceinline__ void store_and_fetch(int step) {        // step < n
    asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(piece) : "memory");
    dst += 64; src += 64;
    if (step + 1 < n) piece = *src;
  }
};

// out rows [32 nt + p(i) + 4 h] = act(W[nt-tile] . in + bias) for nt < NT, written to the wave's LDS slab (rows 0..32 NT).
//
// Weight stream: the fragments of a layer are consecutive; S register sets (S divides KT, so the set of a fragment is the
// compile-time kt % S) hold the current fragment and the D = S - 1 requested ahead of it: fragment f + D is requested at
// the top of step f into the set step f - 1 has just consumed, and step f ends with a COUNTED wait for fragment f + 1
// (frag_wait_n<4 (D - 1)>: the loads of the D - 1 fragments behind it may stay in flight).
//
// prev_dst != nullptr: the previous layer's rows (slab rows 0 .. prev_rows: tile nt of THIS layer overwrites rows 32 nt ..
// at its end, and piece p = rows 8 p .. is read at step p - 1 <= 4 nt + 2 < (nt + 1) KT) go out one piece per step.  With
// D = 3 a wait lets 8 operations stay outstanding, i.e. -- the prefetched loads having landed -- the stores of the last 8
// steps; with one burst per layer, and with D = 1 (vmcnt(0) every step), every layer paid its stores' round trips: +33 %
// on the training forward (both measured).
template <int KT, bool RELU, int S>
__device__ __forceinline__ void layer_fwd(const float4* __restrict__ wl, int fbase, const float* bias_lds, int NT,
                                          const f32x16 (&in)[KT], float* slab, int col, int h, float* prev_dst, int prev_rows,
                                          uint4* relu_bits = nullptr) {
  static_assert(KT % S == 0 && S >= 2, "the set of fragment (nt, kt) must be kt % S");
  static_assert(KT >= 4 || S == 2, "");
  constexpr int D = S - 1;
  const int lane = threadIdx.x & 63;
  Deferred df;
  df.init(slab, prev_dst, prev_rows, lane);
  FragQ q[S];
#pragma unroll
  for (int d = 0; d < D; ++d) frag_issue(q[d], wl, fbase + d);
  frag_wait_n<4 * (D - 1)>(q[0]);
  const float4* b4 = reinterpret_cast<const float4*>(bias_lds + 4 * h);
  float4 bn[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bn[j] = b4[2 * j];
  // relu_bits (the 8-tile layers H0..H7 of a training pass): bit 16 (nt & 1) + i of word nt >> 1 = (row 32 nt + p(i) + 4 h of this
  // lane's sample is > 0), one uint4 per lane and layer: what the backward chain needs of H -- 1 KiB per layer and tile
  // instead of reading back the 32 KiB of float32 rows.  Kept as a 128-bit shift register: each tile enters at the top.
  unsigned m0 = 0, m1 = 0, m2 = 0, m3 = 0;
  auto tile = [&](int nt, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[4 * j] = bn[j].x; acc[4 * j + 1] = bn[j].y; acc[4 * j + 2] = bn[j].z; acc[4 * j + 3] = bn[j].w; }
    if (!LAST) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bn[j] = b4[8 * (nt + 1) + 2 * j];          // the next tile's biases
    }
    const int step0 = nt * KT;
    static_for<KT>([&](auto kt_c) {
      constexpr int kt = decltype(kt_c)::value;
      const int step = step0 + kt;
      const bool st = step < df.n;                                              // wave-uniform
      // the step's first MFMA goes ahead of its loads and store: with one wave per SIMD nothing else covers their issue
      // cycles (address arithmetic + five vector-memory instructions), under a 64-cycle MFMA they are free
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[kt % S].q[0][0], in[kt][0], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (!LAST || kt + D < KT) frag_issue(q[(kt + D) % S], wl, fbase + step + D);
      if (st) df.store_and_fetch(step);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 1; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[kt % S].q[i >> 2][i & 3], in[kt][i], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // fragment step + 1: loads younger than it = those of fragments step + 2 .. step + D (fewer at the layer's end)
      constexpr int YL = LAST ? (KT - 2 - kt < D - 1 ? KT - 2 - kt : D - 1) : D - 1;
      if (YL >= 0) frag_wait_n<4 * (YL < 0 ? 0 : YL)>(q[(kt + 1) % S]);
    });
    unsigned bits = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float v = RELU ? __builtin_elementwise_maximum(acc[i], 0.0f) : acc[i];     // NaN-propagating (v_maximum3_f32), like nn.relu
      const int row = 32 * nt + prow(i) + 4 * h;
      slab[row * 32 + col] = v;
      if (RELU) bits |= (v > 0.0f ? 1u : 0u) << i;
    }
    if (RELU && relu_bits) {
      m0 = (m0 >> 16) | (m1 << 16); m1 = (m1 >> 16) | (m2 << 16); m2 = (m2 >> 16) | (m3 << 16); m3 = (m3 >> 16) | (bits << 16);
    }
  };
  for (int nt = 0; nt < NT - 1; ++nt) tile(nt, std::false_type{});
  tile(NT - 1, std::true_type{});
  for (int step = NT * KT; step < df.n; ++step) df.store_and_fetch(step);      // (not taken for the shapes used: NT KT >= prev_rows / 8)
  if (RELU && relu_bits) *relu_bits = make_uint4(m0, m1, m2, m3);              // (NT == 8 for every layer that passes relu_bits)
}
template <int KT>
__device__ __forceinline__ void slab_to_regs(const float* slab, f32x16 (&dst)[KT], int col, int h) {
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[kt][i] = slab[(32 * kt + prow(i) + 4 * h) * 32 + col];
}

struct FwdArgs32 {
  const float4* wf; const float* tail;
  const float* x; const float* rays; const float* z;
  int64_t M; int n;
  float fpos[10], fdir[4];
  float* out; float* acts;
  int out_ch;            // image model: columns of `out` (1..4)
};

// channel c of the embedding [x, sin(f0 x), cos(f0 x), ...] (models/embedding.py:30-71), c >= limit -> 0
struct Chan { int kind, dim, band; };   // kind: 0 identity, 1 sin, 2 cos, 3 zero pad
__host__ __device__ constexpr Chan chan_of(int c, int limit) {
  if (c < 3) return Chan{0, c, 0};
  if (c >= limit) return Chan{3, 0, 0};
  return Chan{((c - 3) % 6) >= 3 ? 2 : 1, (c - 3) % 3, (c - 3) / 6};
}
// register i of a lane holds channel C0 (lane half 0) or C0 + 4 (lane half 1): both descriptions are compile-time, only
// the select is per lane, so the frequency table is never indexed dynamically (no scratch)
template <int C0, int LIMIT, int NB>
__device__ __forceinline__ float embed_pair(const float (&v)[3], const float (&fr)[NB], int h) {
  constexpr Chan a = chan_of(C0, LIMIT), b = chan_of(C0 + 4, LIMIT);
  const float xa = v[a.dim], xb = v[b.dim];
  const float arg = h ? xb * fr[b.band] : xa * fr[a.band];        // x * freq as one float32 product, like the reference
  float sv = 0.0f, cv = 0.0f;
  constexpr bool need_s = a.kind == 1 || b.kind == 1, need_c = a.kind == 2 || b.kind == 2;
  if (need_s && need_c) sincosf(arg, &sv, &cv);                    // one argument reduction for the two lane halves
  else if (need_s) sv = sinf(arg);
  else if (need_c) cv = cosf(arg);
  const float va = a.kind == 0 ? xa : a.kind == 1 ? sv : a.kind == 2 ? cv : 0.0f;
  const float vb = b.kind == 0 ? xb : b.kind == 1 ? sv : b.kind == 2 ? cv : 0.0f;
  return h ? vb : va;
}
template <int KT, int LIMIT, int NB, int... I>
__device__ __forceinline__ f32x16 embed_tile_impl(const float (&v)[3], const float (&fr)[NB], int h) {
  f32x16 r;
  ((r[I] = embed_pair<32 * KT + prow(I), LIMIT, NB>(v, fr, h)), ...);
  return r;
}
template <int KT, int LIMIT, int NB>
__device__ __forceinline__ f32x16 embed_tile(const float (&v)[3], const float (&fr)[NB], int h) {
  return embed_tile_impl<KT, LIMIT, NB, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15>(v, fr, h);
}

template <bool STORE, bool IMG>
__global__ void __launch_bounds__(256) mlp32_fwd_kernel(FwdArgs32 a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int64_t ntiles = (a.M + 31) >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
  // the layer biases (tail floats 0 .. BIAS_FLOATS) behind the slabs: read per n-tile with ds_read (lgkmcnt), so that no
  // compiler-counted global load -- whose wait would be vmcnt(0) -- sits between the counted fragment waits of a layer
  float* bias_lds = slab_smem + 4 * 256 * 32;
  for (int i = threadIdx.x; i < BIAS_FLOATS; i += 256) bias_lds[i] = a.tail[i];
  __syncthreads();
  if (tile >= ntiles) return;
  float* slab = slab_smem + wv * (256 * 32);
  const float4* wl = a.wf + lane * 4;
  int64_t m = tile * 32 + col; if (m >= a.M) m = a.M - 1;
  f32x16 pe[2], dpe[1];
  if (IMG) {                                                           // embedded rows [M,40] only (no fused encoding)
    const float* row = a.x + m * LI::CIN;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) { const int c = 32 * kt + prow(i) + 4 * h; pe[kt][i] = c < LI::CIN ? row[c] : 0.0f; }
#pragma unroll
    for (int i = 0; i < 16; ++i) dpe[0][i] = 0.0f;
    // the row loads are complete before the first store below is issued: hipcc otherwise interleaves them and counts the
    // younger STORES into its vmcnt waits, which is not safe on this chip (frag_wait_n; tools/check_inflight_regs.py flags it)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pe[0]), "+v"(pe[1]) :: "memory");
  } else if (a.x) {
    const float* row = a.x + m * 90;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) { const int c = 32 * kt + prow(i) + 4 * h; pe[kt][i] = c < 63 ? row[c] : 0.0f; }
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = prow(i) + 4 * h; dpe[0][i] = c < 27 ? row[63 + c] : 0.0f; }
  } else {
    const float* rr = a.rays + (m / a.n) * NERF_RAY_STRIDE;
    const float zv = a.z[m];
    float p[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { p[c] = rr[c] + zv * rr[3 + c]; d[c] = rr[8 + c]; }       // render.py:142
    pe[0] = embed_tile<0, 63, 10>(p, a.fpos, h);
    pe[1] = embed_tile<1, 63, 10>(p, a.fpos, h);
    dpe[0] = embed_tile<0, 27, 4>(d, a.fdir, h);
  }
  // rows `row0` ... of this tile in the activation store (training only): each layer is stored by the NEXT one
  auto rows_of = [&](int row0) -> float* { return STORE ? store_row(a.acts, tile, A_ROWS, row0, 0) : nullptr; };
  auto bits_of = [&](int l) -> uint4* { return STORE ? reinterpret_cast<uint4*>(store_row(a.acts, tile, A_ROWS, A_MASK + 8 * l, 0)) + lane : nullptr; };
  if (STORE) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) *store_row(a.acts, tile, A_ROWS, A_PE + 32 * kt + prow(i) + 4 * h, col) = pe[kt][i];
    if (!IMG) {
#pragma unroll
      for (int i = 0; i < 16; ++i) *store_row(a.acts, tile, A_ROWS, A_DPE + prow(i) + 4 * h, col) = dpe[0][i];
    }
  }
  f32x16 hcur[8];
  layer_fwd<2, true, 2>(wl, F_L0, bias_lds + T_B, 8, pe, slab, col, h, nullptr, 0, bits_of(0));
  slab_to_regs<8>(slab, hcur, col, h);
  for (int l = 1; l <= 4; ++l) {                                     // pos1..pos4 (each stores H_{l-1} while it computes)
    layer_fwd<8, true, 4>(wl, F_L1 + (l - 1) * 64, bias_lds + T_B + 256 * l, 8, hcur, slab, col, h, rows_of(A_H0 + 256 * (l - 1)), 256, bits_of(l));
    slab_to_regs<8>(slab, hcur, col, h);
  }
  {                                                                  // pos5 on concat[input_pos, h]  (models/NeRF.py:224-225)
    f32x16 cat[10];
    cat[0] = pe[0]; cat[1] = pe[1];
#pragma unroll
    for (int k = 0; k < 8; ++k) cat[2 + k] = hcur[k];
    layer_fwd<10, true, 5>(wl, F_L5, bias_lds + T_B + 256 * 5, 8, cat, slab, col, h, rows_of(A_H0 + 256 * 4), 256, bits_of(5));
    slab_to_regs<8>(slab, hcur, col, h);
  }
  for (int l = 6; l <= 7; ++l) {
    layer_fwd<8, true, 4>(wl, F_L6 + (l - 6) * 64, bias_lds + T_B + 256 * l, 8, hcur, slab, col, h, rows_of(A_H0 + 256 * (l - 1)), 256, bits_of(l));
    slab_to_regs<8>(slab, hcur, col, h);
  }
  if (IMG) {
    // output_linear(h7) (models/NeRF.py:196-197,241): out_ch <= 4 dot products per sample on the vector ALU
    float o4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < a.out_ch) {                                               // wave-uniform
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) o4[c] += bias_lds[T_WO + c * 256 + 32 * kt + prow(i) + 4 * h] * hcur[kt][i];
        o4[c] += __shfl_xor(o4[c], 32, 64);
        o4[c] += bias_lds[T_BO + c];
      }
    }
    if (STORE) slab_to_store(slab, rows_of(A_H0 + 256 * 7), 256);     // H7's rows: no further layer carries them
    const int64_t mo = tile * 32 + col;
    if (h == 0 && mo < a.M) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < a.out_ch) a.out[mo * a.out_ch + c] = o4[c];
    }
    return;
  }
  // alpha = Linear(256, 1)(h7): a dot product per sample on the vector ALU (models/NeRF.py:230)
  float alpha = 0.0f;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int i = 0; i < 16; ++i) alpha += bias_lds[T_WA + 32 * kt + prow(i) + 4 * h] * hcur[kt][i];
  alpha += __shfl_xor(alpha, 32, 64);
  alpha += bias_lds[T_BA];
  // feature (no activation), then relu(Linear([feature, input_dir])), then rgb   (models/NeRF.py:231-238)
  layer_fwd<8, false, 4>(wl, F_FEAT, bias_lds + T_BF, 8, hcur, slab, col, h, rows_of(A_H0 + 256 * 7), 256);
  {
    f32x16 cat[9];
    slab_to_regs<8>(slab, hcur, col, h);
#pragma unroll
    for (int k = 0; k < 8; ++k) cat[k] = hcur[k];
    cat[8] = dpe[0];
    layer_fwd<9, true, 3>(wl, F_DIR, bias_lds + T_BD, 4, cat, slab, col, h, rows_of(A_FEAT), 256);
  }
  if (STORE) slab_to_store(slab, rows_of(A_HD), 128);               // the last layer's rows: nothing follows to carry them
  f32x16 hd[4];
  slab_to_regs<4>(slab, hd, col, h);
  float rgb[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) rgb[c] += bias_lds[T_WR + c * 128 + 32 * kt + prow(i) + 4 * h] * hd[kt][i];
    rgb[c] += __shfl_xor(rgb[c], 32, 64);
    rgb[c] += bias_lds[T_BR + c];
  }
  const int64_t mo = tile * 32 + col;
  if (h == 0 && mo < a.M) {
    float4 o; o.x = rgb[0]; o.y = rgb[1]; o.z = rgb[2]; o.w = alpha;                       // [rgb, alpha] raw
    *reinterpret_cast<float4*>(a.out + mo * 4) = o;
  }
}

// ------------------------------------------------------------------------------------------ backward chain
struct BwdArgs32 {
  const float4* wb; const float* tail;
  const float* acts; const float* d_raw; int64_t M;
  float* dz;
  int out_ch;            // image model: columns of d_raw (1..4)
};

// dZ rows [32 kt + p(i) + 4 h] = mask(W^T[kt-tile] . in (+ extra)) for kt < KT -> the wave's LDS slab.
// relu_bits != nullptr: ReLU' from the sign bits the forward kept for that layer (layer_fwd; one 16-byte load per lane and
// chain step, issued and waited for in the prologue: no global load sits among the deferred stores, where its wait would
// drain them); extra_w (LDS): rank-1 term w[row] * extra_s.
// Fragment ring and deferred stores of the PREVIOUS chain step's rows as in layer_fwd (S = 4 sets, D = 3 ahead; k-tile kt
// overwrites slab rows 32 kt .. at its end, piece p is read at step p - 1 <= 4 kt + 2 < (kt + 1) NS).
template <int NS>
__device__ __forceinline__ void layer_bwd(const float4* __restrict__ wl, int fbase, int KT, const f32x16 (&in)[NS],
                                          float* slab, int col, int h, const uint4* relu_bits, const float* extra_w,
                                          float extra_s, float* prev_dst, int prev_rows) {
  constexpr int S = 4, D = S - 1;
  static_assert(NS % S == 0, "the set of fragment (kt, ns) must be ns % S");
  const int lane = threadIdx.x & 63;
  uint4 mw = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
  if (relu_bits) mw = *relu_bits;
  Deferred df;
  df.init(slab, prev_dst, prev_rows, lane);
  FragQ q[S];
#pragma unroll
  for (int d = 0; d < D; ++d) frag_issue(q[d], wl, fbase + d);
  frag_wait_n<4 * (D - 1)>(q[0]);
  asm volatile("" : "+v"(mw.x), "+v"(mw.y), "+v"(mw.z), "+v"(mw.w));          // the compiler's wait for the sign bits: here
  unsigned m0 = mw.x, m1 = mw.y, m2 = mw.z, m3 = mw.w;
  auto ktile = [&](int kt, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = extra_w ? extra_w[32 * kt + prow(i) + 4 * h] * extra_s : 0.0f;
    const int step0 = kt * NS;
    static_for<NS>([&](auto ns_c) {
      constexpr int ns = decltype(ns_c)::value;
      const int step = step0 + ns;
      const bool st = step < df.n;                                              // wave-uniform
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[ns % S].q[0][0], in[ns][0], acc, 0, 0, 0);       // see layer_fwd
      __builtin_amdgcn_sched_barrier(0);
      if (!LAST || ns + D < NS) frag_issue(q[(ns + D) % S], wl, fbase + step + D);
      if (st) df.store_and_fetch(step);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 1; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[ns % S].q[i >> 2][i & 3], in[ns][i], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int YL = LAST ? (NS - 2 - ns < D - 1 ? NS - 2 - ns : D - 1) : D - 1;     // see layer_fwd
      if (YL >= 0) frag_wait_n<4 * (YL < 0 ? 0 : YL)>(q[(ns + 1) % S]);
    });
    const unsigned bits = m0;                                                   // this k-tile's 16 decisions: the low half
    m0 = (m0 >> 16) | (m1 << 16); m1 = (m1 >> 16) | (m2 << 16); m2 = (m2 >> 16) | (m3 << 16); m3 >>= 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 32 * kt + prow(i) + 4 * h;
      slab[row * 32 + col] = (bits >> i) & 1u ? acc[i] : 0.0f;
    }
  };
  for (int kt = 0; kt < KT - 1; ++kt) ktile(kt, std::false_type{});
  ktile(KT - 1, std::true_type{});
  for (int step = KT * NS; step < df.n; ++step) df.store_and_fetch(step);       // (not taken: KT NS >= prev_rows / 8)
}

template <bool IMG>
__global__ void __launch_bounds__(256) mlp32_bwd_kernel(BwdArgs32 a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int64_t ntiles = (a.M + 31) >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
  float* wa_lds = slab_smem + 4 * 256 * 32;               // the alpha head's weights (rank-1 term of dZ7): see bias_lds
  float* wr_lds = wa_lds + 256;                           // the rgb head's weights [3][128]
  if (IMG) {                                              // image model: output_linear's weights [4][256] in the same 1024 floats
    for (int i = threadIdx.x; i < 1024; i += 256) wa_lds[i] = a.tail[T_WO + i];
  } else {
    wa_lds[threadIdx.x] = a.tail[T_WA + threadIdx.x];
    for (int i = threadIdx.x; i < 384; i += 256) wr_lds[i] = a.tail[T_WR + i];
  }
  __syncthreads();
  if (tile >= ntiles) return;
  float* slab = slab_smem + wv * (256 * 32);
  const float4* wl = a.wb + lane * 4;
  const int64_t m = tile * 32 + col;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  auto zrows = [&](int row0) -> float* { return store_row(a.dz, tile, Z_ROWS, row0, 0); };
  auto bits_of = [&](int l) -> const uint4* {
    return reinterpret_cast<const uint4*>(store_row(const_cast<float*>(a.acts), tile, A_ROWS, A_MASK + 8 * l, 0)) + lane;
  };
  if (IMG) {
    if (m < a.M) {
      const float* gr = a.d_raw + m * a.out_ch;
      g.x = gr[0];
      if (a.out_ch > 1) g.y = gr[1];
      if (a.out_ch > 2) g.z = gr[2];
      if (a.out_ch > 3) g.w = gr[3];
    }
    // d out (rows 0..3 of the Z_RGB block; the dW job reads out_ch of them)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = prow(i) + 4 * h;
      *store_row(a.dz, tile, Z_ROWS, Z_RGB + row, col) = row == 0 ? g.x : row == 1 ? g.y : row == 2 ? g.z : row == 3 ? g.w : 0.0f;
    }
    // dZ7 = relu'(H7) * (Wo^T d out): <= 4 terms per unit on the vector ALU, ReLU' from the forward's sign bits (layer_fwd)
    const uint4 mw = *bits_of(7);
    const unsigned mws[4] = {mw.x, mw.y, mw.z, mw.w};
    f32x16 z7[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      const unsigned bits = mws[kt >> 1] >> (16 * (kt & 1));
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = 32 * kt + prow(i) + 4 * h;
        const float v = wa_lds[row] * g.x + wa_lds[256 + row] * g.y + wa_lds[512 + row] * g.z + wa_lds[768 + row] * g.w;
        z7[kt][i] = (bits >> i) & 1u ? v : 0.0f;
        slab[row * 32 + col] = z7[kt][i];
      }
    }
    for (int l = 7; l >= 1; --l) {                                 // dZ_{l-1} = relu'(H_{l-1}) * (W_l^T dZ_l)   (stores dZ_l)
      layer_bwd<8>(wl, B_POS + 64 * (7 - l), 8, z7, slab, col, h, bits_of(l - 1), nullptr, 0.0f, zrows(Z_L0 + 256 * l), 256);
      slab_to_regs<8>(slab, z7, col, h);
    }
    slab_to_store(slab, zrows(Z_L0), 256);                          // dZ_0
    return;
  }
  if (m < a.M) g = *reinterpret_cast<const float4*>(a.d_raw + m * 4);
  // d rgb (rows 0..2) and d alpha (row 0) blocks of the dz store; the other rows of those 32-row blocks are zero
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = prow(i) + 4 * h;
    *store_row(a.dz, tile, Z_ROWS, Z_RGB + row, col) = row == 0 ? g.x : row == 1 ? g.y : row == 2 ? g.z : 0.0f;
    *store_row(a.dz, tile, Z_ROWS, Z_A + row, col) = row == 0 ? g.w : 0.0f;
  }
  // d dir0 pre-activation = relu'(HD) * (Wr^T d rgb): 3 terms per unit, vector ALU.  All 64 HD loads first, the stores
  // after them, the weights from LDS: written as one loop (load, use, store per element, weights from global memory) the
  // stores -- which may alias the loads for all hipcc knows -- kept every element's loads behind the previous element's
  // store: 64 serialized L2 round trips, 40-50 us of the 330 a tile takes.
  f32x16 zd[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      zd[kt][i] = *store_row(const_cast<float*>(a.acts), tile, A_ROWS, A_HD + 32 * kt + prow(i) + 4 * h, col);
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 32 * kt + prow(i) + 4 * h;
      const float v = wr_lds[row] * g.x + wr_lds[128 + row] * g.y + wr_lds[256 + row] * g.z;
      zd[kt][i] = zd[kt][i] > 0.0f ? v : 0.0f;
    }
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int i = 0; i < 16; ++i) *store_row(a.dz, tile, Z_ROWS, Z_D + 32 * kt + prow(i) + 4 * h, col) = zd[kt][i];
  f32x16 zc[8];
  // every chain step leaves its rows in the slab; the NEXT step stores them while it computes, the last one at the end
  layer_bwd<4>(wl, B_DIR, 8, zd, slab, col, h, nullptr, nullptr, 0.0f, nullptr, 0);                                         // d feature
  slab_to_regs<8>(slab, zc, col, h);
  layer_bwd<8>(wl, B_FEAT, 8, zc, slab, col, h, bits_of(7), wa_lds, g.w, zrows(Z_F), 256);                                  // dZ7 (stores d feature)
  slab_to_regs<8>(slab, zc, col, h);
  for (int l = 7; l >= 1; --l) {                                   // dZ_{l-1} = relu'(H_{l-1}) * (W_l^T dZ_l)   (stores dZ_l)
    layer_bwd<8>(wl, B_POS + 64 * (7 - l), 8, zc, slab, col, h, bits_of(l - 1), nullptr, 0.0f, zrows(Z_L0 + 256 * l), 256);
    slab_to_regs<8>(slab, zc, col, h);
  }
  slab_to_store(slab, zrows(Z_L0), 256);                            // dZ_0
}

// ------------------------------------------------------------------------------------------ dW / db
struct DwJob32 {
  int zrow0, n_tiles;      // dZ rows (32 per tile)
  int arow0, k_tiles;      // input-activation rows
  int w_off, ldw, col0, n_valid, k_valid, b_off;
};
constexpr int DW32_MAX_JOBS = 16;
struct DwArgs32 {
  DwJob32 jobs[DW32_MAX_JOBS];
  int unit0[DW32_MAX_JOBS + 1];    // first unit of each job; unit = (block, split)
  int blk0[DW32_MAX_JOBS + 1];     // first 64 x 64 block of each job (the reduce kernel's grid index)
  int njobs, splits, ntiles;
  const float* acts; const float* dz; float* grads;
  float* part;                     // per-unit partial blocks: DW32_PART_FLOATS floats each (behind the dZ rows of the workspace)
};

// one wave = one unit: a 64 x 64 block (2 x 2 MFMA tiles) of one job's dW over one slice of the sample tiles.
// ONE wave per SIMD (all 512 registers): the operands of a sample tile (two dZ and two H fragments, 64 registers) come
// through a ring of four register sets, three tiles ahead -- 12 000 MFMA cycles between a request and its use.  (Two waves
// per SIMD with one tile ahead, the first version, left the matrix pipe idle 29 % of the time: an HBM miss takes about as
// long as the 4 096 cycles of one tile's 64 MFMAs.)  Only loads are in flight here, so the counted waits are exact.
__global__ void __launch_bounds__(256) mlp32_dw_kernel(DwArgs32 a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int unit = blockIdx.x * 4 + wv;
  if (unit >= a.unit0[a.njobs]) return;
  int j = 0;
  while (unit >= a.unit0[j + 1]) ++j;
  const DwJob32 jb = a.jobs[j];
  const int local = unit - a.unit0[j];
  // block-fastest order: the four waves of a workgroup (and the neighbouring workgroups) work on DIFFERENT 64 x 64 blocks
  // of the SAME sample slice at the same time, so the dZ / H rows they share (a 256 x 256 job reads every operand row
  // block 4 times) come from L1 / L2 / Infinity Cache instead of from HBM four times at four different moments
  // (split-fastest order: 80 KB of reads per sample against 20 KB stored)
  const int kb = (jb.k_tiles + 1) / 2;
  const int nblocks = ((jb.n_tiles + 1) / 2) * kb;
  const int block = local % nblocks, split = local / nblocks;
  const int nt0 = 2 * (block / kb), kt0 = 2 * (block % kb);
  const int t_lo = (int)((int64_t)a.ntiles * split / a.splits), t_hi = (int)((int64_t)a.ntiles * (split + 1) / a.splits);
  f32x16 acc[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};
  const bool n1 = nt0 + 1 < jb.n_tiles, k1 = kt0 + 1 < jb.k_tiles;
  // lane (row r32, half h) reads samples 16 h .. 16 h + 15 of its feature row: MFMA step s pairs sample s (h = 0) with sample
  // 16 + s (h = 1) in both operands
  const float* zrow[2]; const float* hrow[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) zrow[r] = a.dz + (((int64_t)jb.zrow0 + 32 * (nt0 + ((r == 0 || n1) ? r : 0)) + r32) * 32 + 16 * h);
#pragma unroll
  for (int c = 0; c < 2; ++c) hrow[c] = a.acts + (((int64_t)jb.arow0 + 32 * (kt0 + ((c == 0 || k1) ? c : 0)) + r32) * 32 + 16 * h);
  constexpr int S = 4, D = S - 1;
  FragQ ring[S][4];                                       // [set][dZ r = 0, 1 | H c = 0, 1]
  // sets 0 and 1 live in VGPRs, sets 2 and 3 in AGPRs (with the 64 accumulators: 128 + 192 registers)
  auto issue_tile = [&](int t, FragQ (&f)[4], auto acc_tag) {
    constexpr bool ACC = decltype(acc_tag)::value;
    const float* z0 = zrow[0] + (int64_t)t * (Z_ROWS * 32); const float* z1 = zrow[1] + (int64_t)t * (Z_ROWS * 32);
    const float* h0 = hrow[0] + (int64_t)t * (A_ROWS * 32); const float* h1 = hrow[1] + (int64_t)t * (A_ROWS * 32);
    if (ACC) { frag_issue_at_acc(f[0], z0); frag_issue_at_acc(f[1], z1); frag_issue_at_acc(f[2], h0); frag_issue_at_acc(f[3], h1); }
    else { frag_issue_at(f[0], z0); frag_issue_at(f[1], z1); frag_issue_at(f[2], h0); frag_issue_at(f[3], h1); }
  };
  auto wait_tile = [&](FragQ (&f)[4], int younger_tiles, auto acc_tag) {   // 16 loads per tile; `younger_tiles` is wave-uniform
    constexpr bool ACC = decltype(acc_tag)::value;
    if (younger_tiles >= 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (younger_tiles == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (ACC) asm volatile("" : "+a"(f[k].q[0]), "+a"(f[k].q[1]), "+a"(f[k].q[2]), "+a"(f[k].q[3]));
      else asm volatile("" : "+v"(f[k].q[0]), "+v"(f[k].q[1]), "+v"(f[k].q[2]), "+v"(f[k].q[3]));
    }
  };
  const int nt_mine = t_hi - t_lo;
  static_for<D>([&](auto d_c) {
    constexpr int d = decltype(d_c)::value;
    if (d < nt_mine) issue_tile(t_lo + d, ring[d], std::integral_constant<bool, (d >= 2)>{});
  });
  if (nt_mine > 0) wait_tile(ring[0], (nt_mine - 1 < D - 1 ? nt_mine - 1 : D - 1), std::false_type{});
  for (int t0 = t_lo; t0 < t_hi; t0 += S) {
    static_for<S>([&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      const int t = t0 + i;
      if (t < t_hi) {                                                // wave-uniform
        FragQ (&f)[4] = ring[i];
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[0].q[0][0], f[2].q[0][0], acc[0][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (t + D < t_hi) issue_tile(t + D, ring[(i + D) % S], std::integral_constant<bool, ((i + D) % S >= 2)>{});   // the set tile t - 1 used
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          if (kt0 == 0) {
#pragma unroll
            for (int s = 0; s < 16; ++s) bsum[r] += f[r].q[s >> 2][s & 3];
          }
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s = (r == 0 && c == 0) ? 1 : 0; s < 16; ++s)
              acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[r].q[s >> 2][s & 3], f[2 + c].q[s >> 2][s & 3], acc[r][c], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < t_hi) {
          const int left = t_hi - 2 - t;                                 // tiles behind t + 1 that exist
          wait_tile(ring[(i + 1) % S], left < D - 1 ? left : D - 1, std::integral_constant<bool, ((i + 1) % S >= 2)>{});
        }
      }
    });
  }
  // the unit's partial block goes to its own slot with plain coalesced stores (256 B per instruction); mlp32_dw_reduce_kernel
  // adds the splits of a block in split order: no atomics, no memset, bit-reproducible gradients (as in the bf16 path)
  float* mine = a.part + (int64_t)unit * DW32_PART_FLOATS;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[((r * 2 + c) * 16 + i) * 64 + lane] = acc[r][c][i];
    const float tot = bsum[r] + __shfl_xor(bsum[r], 32, 64);
    if (h == 0) mine[4096 + 32 * r + r32] = tot;
  }
}

// grid (job's 64 x 64 block, slice of its elements): sums the block's `splits` partials in split order and writes the gradient
__global__ void __launch_bounds__(256) mlp32_dw_reduce_kernel(DwArgs32 a) {
  int j = 0;
  while ((int)blockIdx.x >= a.blk0[j + 1]) ++j;
  const DwJob32 jb = a.jobs[j];
  const int kb = (jb.k_tiles + 1) / 2;
  const int nblocks = ((jb.n_tiles + 1) / 2) * kb;
  const int block = blockIdx.x - a.blk0[j];
  const int nt0 = 2 * (block / kb), kt0 = 2 * (block % kb);
  const float* p0 = a.part + ((int64_t)a.unit0[j] + block) * DW32_PART_FLOATS;
  const int64_t stride = (int64_t)nblocks * DW32_PART_FLOATS;                  // unit = unit0 + split * nblocks + block
  for (int e = threadIdx.x + 256 * blockIdx.y; e < DW32_PART_FLOATS; e += 256 * gridDim.y) {
    float sum = 0.0f;
    for (int sp = 0; sp < a.splits; ++sp) sum += p0[sp * stride + e];
    if (e < 4096) {
      const int lane = e & 63, i = (e >> 6) & 15, c = (e >> 10) & 1, r = e >> 11;
      const int r32 = lane & 31, h = lane >> 5;
      if ((r == 1 && nt0 + 1 >= jb.n_tiles) || (c == 1 && kt0 + 1 >= jb.k_tiles)) continue;
      const int colk = 32 * (kt0 + c) + r32, n = 32 * (nt0 + r) + prow(i) + 4 * h;
      if (colk < jb.k_valid && n < jb.n_valid) a.grads[jb.w_off + (int64_t)n * jb.ldw + jb.col0 + colk] = sum;
    } else {
      const int r = (e - 4096) >> 5, r32 = (e - 4096) & 31;
      if (r == 1 && nt0 + 1 >= jb.n_tiles) continue;
      const int n = 32 * (nt0 + r) + r32;
      if (kt0 == 0 && jb.b_off >= 0 && n < jb.n_valid) a.grads[jb.b_off + n] = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------ debug decode
__global__ void __launch_bounds__(256) decode32_kernel(const float* __restrict__ store, int rows, int row0, int width,
                                                       int64_t M, float* __restrict__ out) {
  const int64_t total = M * width;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = t / width; const int c = (int)(t - m * width);
    out[t] = store[(((m >> 5) * rows + row0 + c) * 32) + (m & 31)];
  }
}
static bool debug_slot(int kind, int layer, int* row0, int* width) {
  if (layer < 0 || layer > 11) return false;
  if (kind == 0) {
    if (layer < 8) { *row0 = A_H0 + 256 * layer; *width = 256; }
    else if (layer == 8) { *row0 = A_FEAT; *width = 256; }
    else if (layer == 9) { *row0 = A_HD; *width = 128; }
    else if (layer == 10) { *row0 = A_PE; *width = 64; }
    else { *row0 = A_DPE; *width = 32; }
    return true;
  }
  if (kind == 1) {
    if (layer < 8) { *row0 = Z_L0 + 256 * layer; *width = 256; }
    else if (layer == 8) { *row0 = Z_F; *width = 256; }
    else if (layer == 9) { *row0 = Z_D; *width = 128; }
    else if (layer == 10) { *row0 = Z_A; *width = 16; }
    else { *row0 = Z_RGB; *width = 16; }
    return true;
  }
  return false;
}

// ------------------------------------------------------------------------------------------ host entry points
constexpr int SLAB_BYTES = 4 * 256 * 32 * 4;        // 128 KiB: one 32 KiB slab per wave
constexpr int FWD_LDS_BYTES = SLAB_BYTES + BIAS_FLOATS * 4, BWD_LDS_BYTES = SLAB_BYTES + 1024 * 4;

// opt in to more than 64 KiB of dynamic LDS, once per (kernel, device).  Keyed by the kernel's ADDRESS: the two forward
// instantiations have the same function type, so a per-type flag (the first version) served only whichever ran first.
template <class K>
static void want_lds(K kernel, int bytes = SLAB_BYTES) {
  static const void* seen[8][64] = {};
  const void* fn = reinterpret_cast<const void*>(kernel);
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
  for (int i = 0; i < 8; ++i) {
    if (seen[i][d] == fn) return;
    if (seen[i][d] == nullptr) { seen[i][d] = fn; break; }
  }
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

int pack(const float* params, void* packed32, int img_out_ch, hipStream_t s) {
  const int total = (F_FRAGS + B_FRAGS) * 64 + TAIL_FLOATS;
  hipLaunchKernelGGL(pack32_kernel, dim3((total + 255) / 256), dim3(256), 0, s, params, static_cast<float*>(packed32), img_out_ch);
  return check_launch("nerf_mlp_pack (fp32 streams)");
}

static const float* tail_of(const void* packed32) {
  return reinterpret_cast<const float*>(static_cast<const char*>(packed32) + (size_t)(F_FRAGS + B_FRAGS) * FRAG_BYTES);
}

int forward(const void* packed32, const float* x, const float* rays, const float* z, int64_t M, int n, int freq_mode,
            float* out, void* acts, int img_out_ch, hipStream_t s) {
  FwdArgs32 a;
  a.out_ch = img_out_ch;
  a.wf = static_cast<const float4*>(packed32);
  a.tail = tail_of(packed32);
  a.x = x; a.rays = rays; a.z = z; a.M = M; a.n = n; a.out = out; a.acts = static_cast<float*>(acts);
  for (int k = 0; k < 10; ++k) a.fpos[k] = freq_mode == 0 ? (float)(k * k) : (float)(1 << k);
  for (int k = 0; k < 4; ++k) a.fdir[k] = freq_mode == 0 ? (float)(k * k) : (float)(1 << k);
  const int64_t blocks = (tiles_of(M) + 3) / 4;
  NERF_REQUIRE(blocks < (1ll << 31), NERF_E_SHAPE, "mlp forward (fp32): M too large");
  if (img_out_ch > 0) {
    NERF_REQUIRE(x && img_out_ch <= 4, NERF_E_SHAPE, "mlp forward (fp32, image model): needs embedded rows and out_ch <= 4");
    if (acts) {
      want_lds((mlp32_fwd_kernel<true, true>), FWD_LDS_BYTES);
      hipLaunchKernelGGL((mlp32_fwd_kernel<true, true>), dim3((unsigned)blocks), dim3(256), FWD_LDS_BYTES, s, a);
    } else {
      want_lds((mlp32_fwd_kernel<false, true>), FWD_LDS_BYTES);
      hipLaunchKernelGGL((mlp32_fwd_kernel<false, true>), dim3((unsigned)blocks), dim3(256), FWD_LDS_BYTES, s, a);
    }
  } else if (acts) {
    want_lds((mlp32_fwd_kernel<true, false>), FWD_LDS_BYTES);
    hipLaunchKernelGGL((mlp32_fwd_kernel<true, false>), dim3((unsigned)blocks), dim3(256), FWD_LDS_BYTES, s, a);
  } else {
    want_lds((mlp32_fwd_kernel<false, false>), FWD_LDS_BYTES);
    hipLaunchKernelGGL((mlp32_fwd_kernel<false, false>), dim3((unsigned)blocks), dim3(256), FWD_LDS_BYTES, s, a);
  }
  return check_launch("mlp forward (fp32)");
}

int backward(const void* packed32, const void* acts, const float* d_raw, int64_t M, void* dz, float* grads, int img_out_ch,
             hipStream_t s) {
  BwdArgs32 b;
  b.out_ch = img_out_ch;
  const bool img = img_out_ch > 0;
  b.wb = reinterpret_cast<const float4*>(static_cast<const char*>(packed32) + (size_t)F_FRAGS * FRAG_BYTES);
  b.tail = tail_of(packed32);
  b.acts = static_cast<const float*>(acts); b.d_raw = d_raw; b.M = M; b.dz = static_cast<float*>(dz);
  const int64_t ntiles = tiles_of(M), blocks = (ntiles + 3) / 4;
  NERF_REQUIRE(blocks < (1ll << 31), NERF_E_SHAPE, "mlp backward (fp32): M too large");
  if (img) {
    want_lds(mlp32_bwd_kernel<true>, BWD_LDS_BYTES);
    hipLaunchKernelGGL(mlp32_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), BWD_LDS_BYTES, s, b);
  } else {
    want_lds(mlp32_bwd_kernel<false>, BWD_LDS_BYTES);
    hipLaunchKernelGGL(mlp32_bwd_kernel<false>, dim3((unsigned)blocks), dim3(256), BWD_LDS_BYTES, s, b);
  }
  int rc = check_launch("mlp backward chain (fp32)");
  if (rc) return rc;
  DwArgs32 d;
  int nj = 0;
  auto job = [&](int zrow0, int n_rows, int arow0, int k_rows, int w_off, int ldw, int col0, int nv, int kv, int b_off) {
    d.jobs[nj++] = DwJob32{zrow0, n_rows / 32, arow0, k_rows / 32, w_off, ldw, col0, nv, kv, b_off};
  };
  if (img) {
    job(Z_L0, 256, A_PE, 64, LI::P_W0, 40, 0, 256, 40, LI::P_B0);                                          // pos0
    for (int l = 1; l <= 4; ++l) job(Z_L0 + 256 * l, 256, A_H0 + 256 * (l - 1), 256, LI::pw(l), 256, 0, 256, 256, LI::pb(l));
    job(Z_L0 + 256 * 5, 256, A_H0 + 256 * 4, 256, LI::P_W5, 296, 40, 256, 256, LI::P_B5);                  // pos5 | H4
    job(Z_L0 + 256 * 5, 256, A_PE, 64, LI::P_W5, 296, 0, 256, 40, -1);                                     // pos5 | x
    job(Z_L0 + 256 * 6, 256, A_H0 + 256 * 5, 256, LI::P_W6, 256, 0, 256, 256, LI::P_B6);
    job(Z_L0 + 256 * 7, 256, A_H0 + 256 * 6, 256, LI::P_W7, 256, 0, 256, 256, LI::P_B7);
    job(Z_RGB, 32, A_H0 + 256 * 7, 256, LI::P_WO, 256, 0, img_out_ch, 256, LI::P_WO + img_out_ch * 256);   // output_linear
  } else {
  job(Z_L0, 256, A_PE, 64, L::P_W0, 63, 0, 256, 63, L::P_B0);                                              // pos0
  for (int l = 1; l <= 4; ++l) job(Z_L0 + 256 * l, 256, A_H0 + 256 * (l - 1), 256, L::pw(l), 256, 0, 256, 256, L::pb(l));
  job(Z_L0 + 256 * 5, 256, A_H0 + 256 * 4, 256, L::P_W5, 319, 63, 256, 256, L::P_B5);                      // pos5 | H4
  job(Z_L0 + 256 * 5, 256, A_PE, 64, L::P_W5, 319, 0, 256, 63, -1);                                        // pos5 | PE
  job(Z_L0 + 256 * 6, 256, A_H0 + 256 * 5, 256, L::P_W6, 256, 0, 256, 256, L::P_B6);
  job(Z_L0 + 256 * 7, 256, A_H0 + 256 * 6, 256, L::P_W7, 256, 0, 256, 256, L::P_B7);
  job(Z_F, 256, A_H0 + 256 * 7, 256, L::P_WF, 256, 0, 256, 256, L::P_BF);                                  // feature
  job(Z_A, 32, A_H0 + 256 * 7, 256, L::P_WA, 256, 0, 1, 256, L::P_BA);                                     // alpha
  job(Z_D, 128, A_FEAT, 256, L::P_WD, 283, 0, 128, 256, L::P_BD);                                          // dir0 | feature
  job(Z_D, 128, A_DPE, 32, L::P_WD, 283, 256, 128, 27, -1);                                                // dir0 | dirPE
  job(Z_RGB, 32, A_HD, 128, L::P_WR, 128, 0, 3, 128, L::P_BR);                                             // rgb
  }
  int blocks_total = 0;
  for (int j = 0; j < nj; ++j) blocks_total += ((d.jobs[j].n_tiles + 1) / 2) * ((d.jobs[j].k_tiles + 1) / 2);
  int splits = 2048 / blocks_total;                      // about two waves per SIMD of a 256-CU device
  if (splits < 1) splits = 1;
  if (splits > ntiles) splits = (int)ntiles;
  d.unit0[0] = 0; d.blk0[0] = 0;
  for (int j = 0; j < nj; ++j) {
    const int nb = ((d.jobs[j].n_tiles + 1) / 2) * ((d.jobs[j].k_tiles + 1) / 2);
    d.unit0[j + 1] = d.unit0[j] + nb * splits;
    d.blk0[j + 1] = d.blk0[j] + nb;
  }
  NERF_REQUIRE(d.unit0[nj] <= DW32_MAX_UNITS, NERF_E_SHAPE, "mlp dW (fp32): %d units", d.unit0[nj]);
  d.njobs = nj; d.splits = splits; d.ntiles = (int)ntiles;
  d.acts = b.acts; d.dz = b.dz; d.grads = grads;
  d.part = reinterpret_cast<float*>(static_cast<char*>(dz) + ntiles * Z_ROWS * 128);     // dz_bytes() counts the slots
  hipLaunchKernelGGL(mlp32_dw_kernel, dim3((d.unit0[nj] + 3) / 4), dim3(256), 0, s, d);
  rc = check_launch("mlp dW (fp32)");
  if (rc) return rc;
  hipLaunchKernelGGL(mlp32_dw_reduce_kernel, dim3(d.blk0[nj], 8), dim3(256), 0, s, d);       // every parameter is written: no memset
  return check_launch("mlp dW reduce (fp32)");
}

int debug_width(int kind, int layer) {
  int row0 = 0, width = 0;
  return debug_slot(kind, layer, &row0, &width) ? width : -1;
}
int debug_read(const void* store, int kind, int layer, int64_t M, float* out, hipStream_t s) {
  int row0 = 0, width = 0;
  NERF_REQUIRE(debug_slot(kind, layer, &row0, &width), NERF_E_SHAPE, "nerf_mlp_debug_read: kind must be 0/1 and layer 0..11");
  hipLaunchKernelGGL(decode32_kernel, dim3(grid_for(M * width, 256)), dim3(256), 0, s, static_cast<const float*>(store),
                     kind == 0 ? A_ROWS : Z_ROWS, row0, width, M, out);
  return check_launch("nerf_mlp_debug_read (fp32)");
}

}  // namespace f32
}  // namespace nerf
