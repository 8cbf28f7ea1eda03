// Helpers shared by the fused-MLP training kernels of mlp.hip (bf16 operands) and mlp_s16.hip (split-bf16 operands):
// fragment-block addressing and stores, packed bf16 pair arithmetic (ReLU, sign bits, masks), the epilogue schedule, and the
// job / argument structures of the weight-gradient kernels.
#pragma once
#include "mlp_layout.h"

namespace nerf {

struct Chan { int kind, dim, band; };   // kind: 0 identity, 1 sin, 2 cos, 3 zero pad
__host__ __device__ constexpr Chan chan_of(int c, int limit) {
  if (c < 3) return Chan{0, c, 0};
  if (c >= limit) return Chan{3, 0, 0};
  return Chan{((c - 3) % 6) >= 3 ? 2 : 1, (c - 3) % 3, (c - 3) / 6};
}

// ------------------------------------------------------------------------------------------
// weight packing sources: which master parameter sits in element j of lane (r, h) of fragment f of the 32x32x16 streams
// ------------------------------------------------------------------------------------------
__device__ inline float fwd_src(const float* __restrict__ p, int f, int r, int h, int j) {
  int nt, ks;
  if (f < L::F_L1) {                                   // pos0: K space 64 (63 + pad)
    nt = f / 4; ks = f % 4;
    const int kk = kperm(ks, h, j);
    return kk < 63 ? p[L::P_W0 + (32 * nt + r) * 63 + kk] : 0.0f;
  }
  if (f < L::F_L5) {                                   // pos1..pos4
    const int l = 1 + (f - L::F_L1) / 128, g = (f - L::F_L1) % 128;
    nt = g / 16; ks = g % 16;
    return p[L::pw(l) + (32 * nt + r) * 256 + kperm(ks, h, j)];
  }
  if (f < L::F_L6) {                                   // pos5: [PE(64), H4(256)] vs W5[256][319]
    const int g = f - L::F_L5;
    nt = g / 20; ks = g % 20;
    const int kk = kperm(ks, h, j), n = 32 * nt + r;
    if (kk < 64) return kk < 63 ? p[L::P_W5 + n * 319 + kk] : 0.0f;
    return p[L::P_W5 + n * 319 + 63 + (kk - 64)];
  }
  if (f < L::F_FA) {                                   // pos6, pos7
    const int l = 6 + (f - L::F_L6) / 128, g = (f - L::F_L6) % 128;
    nt = g / 16; ks = g % 16;
    return p[L::pw(l) + (32 * nt + r) * 256 + kperm(ks, h, j)];
  }
  if (f < L::F_DIR) {                                  // feature (8 tiles) + alpha (tile 8, row 0)
    const int g = f - L::F_FA;
    nt = g / 16; ks = g % 16;
    const int kk = kperm(ks, h, j);
    if (nt < 8) return p[L::P_WF + (32 * nt + r) * 256 + kk];
    return r == 0 ? p[L::P_WA + kk] : 0.0f;
  }
  if (f < L::F_RGB) {                                  // dir0: [feature(256), dirPE(27+5 pad)] vs WD[128][283]
    const int g = f - L::F_DIR;
    nt = g / 18; ks = g % 18;
    const int kk = kperm(ks, h, j), n = 32 * nt + r;
    if (kk < 256) return p[L::P_WD + n * 283 + kk];
    return (kk - 256) < 27 ? p[L::P_WD + n * 283 + kk] : 0.0f;
  }
  ks = f - L::F_RGB;                                   // rgb: rows 0..2 of one tile, K = 128
  return r < 3 ? p[L::P_WR + r * 128 + kperm(ks, h, j)] : 0.0f;
}

// transposed stream: A rows = INPUT feature (32 kt + r), k index = OUTPUT feature nn
__device__ inline float bwd_src(const float* __restrict__ p, int f, int r, int h, int j) {
  if (f < L::B_DIR) {                                  // rgb^T: 4 tiles of H_d, one k-step (rows 0..2)
    const int nn = kperm(0, h, j);
    return nn < 3 ? p[L::P_WR + nn * 128 + 32 * f + r] : 0.0f;
  }
  if (f < L::B_FA) {                                   // dir0^T, feature columns only: 8 tiles x 8 k-steps
    const int g = f - L::B_DIR, kt = g / 8, ns = g % 8;
    return p[L::P_WD + kperm(ns, h, j) * 283 + 32 * kt + r];
  }
  if (f < L::B_L7) {                                   // [feature; alpha]^T: 8 tiles x 17 k-steps
    const int g = f - L::B_FA, kt = g / 17, ns = g % 17;
    const int nn = kperm(ns, h, j);
    if (ns < 16) return p[L::P_WF + nn * 256 + 32 * kt + r];
    return nn == 256 ? p[L::P_WA + 32 * kt + r] : 0.0f;
  }
  const int g = f - L::B_L7, li = g / 128, q = g % 128, kt = q / 16, ns = q % 16;   // pos7, 6, 5, 4, 3, 2, 1
  const int l = 7 - li, nn = kperm(ns, h, j), row = 32 * kt + r;
  if (l == 5) return p[L::P_W5 + nn * 319 + 63 + row];
  return p[L::pw(l) + nn * 256 + row];
}

// fragment block address: tile T, slot s, lane (r,h) at byte 32 r + 16 h.  Written as (uniform 64-bit tile base) +
// (constant slot offset) + (32-bit lane offset) so that hipcc keeps the base in SGPRs.
__device__ __forceinline__ bf16x8* frag_ptr(void* base, int64_t tile, int64_t stride16, int slot, int r, int h) {
  char* tb = reinterpret_cast<char*>(base) + tile * stride16 * 16;
  return reinterpret_cast<bf16x8*>(tb + slot * 1024 + (unsigned)(32 * r + 16 * h));
}

#ifndef NERF_NT_STORES
#define NERF_NT_STORES 1
#endif
constexpr bool g_nt_stores = NERF_NT_STORES != 0;

// one fragment of a fragment block.  Written once, read once by a later kernel: non-temporal, so that the 5 KB/sample
// store stream does not push the 2.4 MB weight stream (which every workgroup re-reads through the ring) out of L2
__device__ __forceinline__ void store_frag(void* base, int64_t tile, int64_t stride16, int slot, const bf16x8& v,
                                           int r, int h) {
#if NERF_ABLATE == 7          // timing-only build 7: no fragment stores at all (values kept alive)
  asm volatile("" :: "v"(v));
  return;
#endif
#if NERF_ABLATE == 8          // timing-only build 8: every tile stored over tiles 0-7 (L2-resident: issue cost without HBM)
  tile = tile & 7;
#endif
  if (g_nt_stores) __builtin_nontemporal_store(v, frag_ptr(base, tile, stride16, slot, r, h));
  else *frag_ptr(base, tile, stride16, slot, r, h) = v;
}
template <int COUNT>
__device__ __forceinline__ void store_frags(void* base, int64_t tile, int64_t stride16, int slot0,
                                            const bf16x8 (&frags)[COUNT], int r, int h) {
#pragma unroll
  for (int k = 0; k < COUNT; ++k) store_frag(base, tile, stride16, slot0 + k, frags[k], r, h);
}
// Where a layer's output fragments go besides the next layer: nowhere (inference), or into the fragment block of the
// sample tile as soon as each pair of fragments is final.  A burst of 16 stores per wave behind the layer (128 KiB per
// workgroup, all 8 waves at once) backs up the CU's store path and stalls the waves at issue: measured 0.21 ms of a
// 0.94 ms chain even with the bytes staying in L2 (NERF_ABLATE 8); two stores per n-tile keep the path draining.
struct NoSink {
  __device__ __forceinline__ void put(int, int, const bf16x8&) const {}
};
template <bool ON>
struct FragSink {
  void* base; int64_t tile0, stride16; int slot0, r, h;
  __device__ __forceinline__ void put(int t, int idx, const bf16x8& v) const {
    if (ON) store_frag(base, tile0 + t, stride16, slot0 + idx, v, r, h);
  }
};

template <class WS>
__device__ __forceinline__ void acc_init_bias(f32x16& acc, WS& ws, int slot_tile, int h) {
  // register i <-> row (i&3) + 8 (i>>2) + 4 h : four float4 at rows 8g + 4h
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 b = ws.bias4(slot_tile + 8 * g + 4 * h);
    acc[4 * g + 0] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
  }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x2 pack2(float a, float b) {       // one v_cvt_pk_bf16_f32
  const f32x2 v = {a, b};
  return __builtin_convertvector(v, bf16x2);
}
__device__ __forceinline__ bf16x2 relu_pack(float a, float b) {
  s16x2 q = __builtin_bit_cast(s16x2, pack2(a, b));
  const s16x2 zero = {0, 0};
  q = __builtin_elementwise_max(q, zero);
  return __builtin_bit_cast(bf16x2, q);
}
// Non-finite INPUTS in the bf16 chains (round 6).  Their ReLU is a packed integer max on the bf16 bit patterns (relu_pack) and
// returns 0 for a NaN pre-activation, where the reference's nn.relu = mx.maximum propagates it (models/NeRF.py:222,236: a NaN /
// Inf position makes all four outputs NaN, a NaN / Inf view direction the three colours).  The float-ReLU kernels (precision 22
// and 32) get that from v_maximum3_f32 at no cost; here a NaN-propagating ReLU would be a third VALU op per value pair, so the
// chain instead flags a sample whose INPUT fragments hold a NaN / Inf (exponent field all ones) once, at the head of the pass,
// and the output store writes NaN for it.  nonfinite_bits(f) != 0 <=> some element of the fragment is NaN / Inf.
__device__ __forceinline__ unsigned nonfinite_bits(const bf16x8& f) {
  const u32x4 w = __builtin_bit_cast(u32x4, f);
  unsigned b = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) b |= (w[i] & 0x7F807F80u) + 0x00800080u;       // 0x7F80 + 0x0080 = 0x8000: bit 15 of a half <=> exponent 0xFF
  return b & 0x80008000u;
}
// bit 0: position inputs, bit 1: direction inputs; OR-ed over the lanes that share a sample (XOR_MASKS: lane-id bits that do not
// select the sample -- 32 in the 32x32x16 chains, 16 | 32 in the 16x16x32 chains)
template <int XOR_MASKS>
__device__ __forceinline__ int nonfinite_flags(unsigned pos_bits, unsigned dir_bits) {
  // wave ballots + a per-lane shift: scalar compares and VALU only (a ds_bpermute shuffle here cost the 2 x 64 training forward its
  // layer-0 ReLU sign words -- tests/test_gpu_round5.py::test_weight_gradients_of_the_two_dw_launch_forms_agree[*-small-16])
  const unsigned long long mp = __ballot(pos_bits != 0u), md = __ballot(dir_bits != 0u);
  const int lane = (int)__lane_id();
  int f = 0;
#pragma unroll
  for (int x = 0; x < 64; x += 16)
    if ((x & ~XOR_MASKS) == 0) f |= (int)((mp >> (lane ^ x)) & 1ull) | ((int)((md >> (lane ^ x)) & 1ull) << 1);
  return f;
}
// [rgb, alpha] of a flagged sample: position -> all NaN; direction only -> rgb NaN, alpha as computed (it does not depend on it)
__device__ __forceinline__ float4 poison_raw(float4 o, int flags) {
  const float qnan = __builtin_nanf("");
  if (flags) { o.x = qnan; o.y = qnan; o.z = qnan; }
  if (flags & 1) o.w = qnan;
  return o;
}

// Packed 16-bit integer ops on bf16 bit patterns.  Inline asm on purpose: written as vector arithmetic, hipcc turns
// them back into one float compare + select per element (and v_perm to re-pack), which is what they replace.
// ReLU sign bits of a packed, already ReLU'd bf16 pair: 1 per non-zero half.
__device__ __forceinline__ unsigned nonzero_bits(bf16x2 p) {
  unsigned r;
  asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(__builtin_bit_cast(unsigned, p)));
  return r;
}
// keep a half of the packed pair where bit B of the same half of `w` is set
template <int B>
__device__ __forceinline__ bf16x2 keep_where(bf16x2 p, unsigned w) {
  unsigned sel, r;
  asm("v_pk_lshrrev_b16 %0, %2, %1 op_sel_hi:[0,1]" : "=v"(sel) : "v"(w), "n"(B));      // both halves shift by the constant's low half
  sel &= 0x00010001u;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(__builtin_bit_cast(unsigned, p)), "v"(sel));
  return __builtin_bit_cast(bf16x2, r);
}

// Epilogue of one accumulator tile, in four quarters of 4 registers so that it can be spread between the MFMAs
// of the NEXT n-tile (the two waves of a SIMD run the same stream in lockstep behind the ring barrier; an epilogue
// done in one block would leave the matrix pipe idle in both at once).  ReLU is an integer max on the bit pattern:
// one VALU op, without the canonicalising v_max hipcc puts in front of fmaxf on MFMA results.
// Sign-bit words (MASKOUT): element e = 2 k + odd of n-tile nt -> bit 16 odd + 8 (nt & 1) + k of word nt >> 1, taken
// from the packed ReLU'd pairs (non-zero half = active unit) instead of a compare + select per fp32 value.
template <bool RELU, bool MASKOUT>
__device__ __forceinline__ void finish_quarter(const f32x16& acc, int q, int nt, bf16x8& lo, bf16x8& hi, u32x4& mask) {
  unsigned w = 0;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int i = 4 * q + 2 * p, k = 2 * q + p;
    const bf16x2 pr = RELU ? relu_pack(acc[i], acc[i + 1]) : pack2(acc[i], acc[i + 1]);
    if (MASKOUT) w |= nonzero_bits(pr) << k;
    if (i < 8) { lo[i] = pr[0]; lo[i + 1] = pr[1]; } else { hi[i - 8] = pr[0]; hi[i - 7] = pr[1]; }
  }
  if (MASKOUT) mask[nt >> 1] |= w << (8 * (nt & 1));
}

__host__ __device__ constexpr int quarter_pos(int ks_count, int q) {   // k-step after which quarter q is retired
  return ks_count >= 8 ? (q * ks_count) / 4 + 1 : (q < ks_count ? q : ks_count - 1);
}

// ------------------------------------------------------------------------------------------
// dW: split-K GEMMs  dW[n][k] = sum_m dZ[n][m] H[k][m]  over fragment blocks, samples = MFMA K
// ------------------------------------------------------------------------------------------
struct DwJob {
  int dz_slot, nf;        // dZ fragments (16 features each); nf <= 16
  int act_slot, kf;       // input-activation fragments; kf <= 16
  int w_off, ldw, col0;   // grads[w_off + n*ldw + col0 + k]
  int n_valid, k_valid;
  int b_off;              // bias gradient offset or -1 (only the job with col0 == 0 of a layer owns it)
};
constexpr int DW_MAX_JOBS = 16;
struct DwArgs {
  DwJob jobs[DW_MAX_JOBS];
  int splits[DW_MAX_JOBS];  // workgroups per job; block b works on job j, split b - prefix(j)
  int ntiles;
  int64_t astride, zstride;
  const void* acts;
  const void* dz;
  float* grads;
  float* partial;         // [workgroup][DW_SLOT_FLOATS] split-K partial tiles (deterministic two-stage reduction)
  int ring_cap;           // s16_dw_kernel: most stages its LDS ring may hold (<= 16)
  int private_max_tiles;  // s16_dw_kernel: jobs of at most this many output tiles (<= 4) run as sixteen independent wave pipelines; 0 = none
  int a_lo, z_lo;         // split-bf16 stores (mlp_s16.hip's dW kernel): slot distance from a hi block to its lo block in acts / dz
};
// One slot per dW workgroup: up to 8 x 8 output tiles of 32 x 32 floats (tile (nt, kt) at (8 nt + kt) * 1024, row-major
// inside the tile) + 8 x 32 bias partial sums.  The workgroups of a job write their partial dW here with plain
// coalesced stores and mlp_dw_reduce_kernel adds the splits of every element in a FIXED order: no float atomics (they
// cost 12.6 % of the dW kernel: 157 MB of read-modify-writes at the memory side per launch, whatever the batch size) and
// the gradient is bit-reproducible from run to run -- with atomics the order of the adds, hence the rounding, was not.
constexpr int DW_SLOT_FLOATS = 64 * 1024 + 256;
constexpr int DW_MAX_WGS = 512;
constexpr int64_t DW_PARTIAL_BYTES = (int64_t)DW_MAX_WGS * DW_SLOT_FLOATS * 4;
// dW of jobs of 16 x 16 fragments only, one wave per SIMD (mlp_dww.hip); split_bf16: hi + lo blocks (precision 22), else bf16 blocks
int launch_dw_wide_kernel(const DwArgs& d, int workgroups, bool split_bf16, hipStream_t s);

}  // namespace nerf
