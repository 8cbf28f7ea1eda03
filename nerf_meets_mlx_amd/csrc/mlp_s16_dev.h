// Device building blocks of the split-bf16 kernels (mlp_s16.hip: the 8 x 256 view model; mlp_s16x.hip: the image-fitting and
// the 2 x 64 hash-grid models): the (hi, lo) bf16 split of a float32 value, one linear layer on register-resident (hi, lo)
// activations in both directions, the <= 4-row heads, and the sink that stores finished fragment pairs.  See mlp_s16.hip for the
// arithmetic (a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_32x32x16_bf16, one fp32 accumulator).
#pragma once
#include "mlp_frag.h"

namespace nerf {
namespace s16 {

struct HL { bf16x2 h, l; };
// two float32 values -> packed bf16 pair of their leading 8 bits, packed bf16 pair of the remainders
__device__ __forceinline__ HL split2(float a, float b) {
  HL o;
  o.h = pack2(a, b);                                                       // v_cvt_pk_bf16_f32 (round to nearest even)
  const unsigned hb = __builtin_bit_cast(unsigned, o.h);
  const float h0 = __builtin_bit_cast(float, hb << 16), h1 = __builtin_bit_cast(float, hb & 0xffff0000u);
  o.l = pack2(a - h0, b - h1);                                             // exact differences
  return o;
}
// one v_maximum3_f32 (IEEE-754 maximum: a NaN pre-activation stays NaN like mx.maximum / nn.relu, models/NeRF.py:222,236; the
// integer max on the bit pattern it replaces turned a negative-signed NaN into 0)
__device__ __forceinline__ float relu_bits(float v) { return __builtin_elementwise_maximum(v, 0.0f); }
template <int COUNT>
__device__ __forceinline__ void split_slots(const float (&v)[COUNT], bf16x8* hi, bf16x8* lo) {
#pragma unroll
  for (int j = 0; j < COUNT; j += 2) {
    const HL s = split2(v[j], v[j + 1]);
    hi[j >> 3][j & 7] = s.h[0]; hi[j >> 3][(j & 7) + 1] = s.h[1];
    lo[j >> 3][j & 7] = s.l[0]; lo[j >> 3][(j & 7) + 1] = s.l[1];
  }
}
__device__ __forceinline__ f32x16 mfma32(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// fragment of an already-embedded row x[m][base + c], c < limit else 0   (NeRF.forward(x) entry)
__device__ __forceinline__ void row_frag(const float* __restrict__ row, int ks, int h, int limit, bf16x8& hi, bf16x8& lo) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = kperm(ks, h, j);
    v[j] = c < limit ? row[c] : 0.0f;
  }
  split_slots<8>(v, &hi, &lo);
}

// ------------------------------------------------------------------------------------------
// one linear layer on register-resident (hi, lo) activations
// ------------------------------------------------------------------------------------------
// where a finished pair of output fragments goes besides the next layer: hi and lo fragment blocks of the sample tile
struct PairSink {
  void* base; int64_t tile, stride16; int slot0, lo_off, r, h;
  __device__ __forceinline__ void put(int idx, const bf16x8& vh, const bf16x8& vl) const {
    store_frag(base, tile, stride16, slot0 + idx, vh, r, h);
    store_frag(base, tile, stride16, slot0 + lo_off + idx, vl, r, h);
  }
  __device__ __forceinline__ void put1(int idx, bool lo, const bf16x8& v) const {       // one block: the hi or the lo part of fragment idx
    store_frag(base, tile, stride16, slot0 + (lo ? lo_off : 0) + idx, v, r, h);
  }
};
// The four 1 KiB stores of a finished output tile (fragments 2 t, 2 t + 1, hi and lo) go out ONE per k-step behind the tile's last
// epilogue quarter instead of all four at once: one wave per SIMD, and a wave issuing a store issues nothing else for ~29 cycles, of
// which the MFMA in front of it covers 24 -- a burst of four leaves the matrix pipe idle for three of them (round 5; the ISA of
// round 4's kernels: 50 MFMA gaps with four stores, 16 with three, 16 with one; now 118 with one, 68 with two).  Worth little:
// chain 2.39 -> 2.37 ms per 786 k samples, forward unchanged, the training step -0.4 % (A/B with -DNERF_S16_SPREAD_STORES=0): most of
// what the stores cost (22 % of each chain) is not their issue pattern.  Store s of tile t is due at flattened k-step
// (t + 1) KS + quarter_pos(KS, 3) + s of the layer; what falls behind the layer's last k-step goes out in its tail.
#ifndef NERF_S16_IGLP
#define NERF_S16_IGLP 0       // __builtin_amdgcn_iglp_opt strategy of the layer bodies (as in mlp22.hip; forward 2.51 -> 2.49, chain 2.40 -> 2.39 ms per 786 k samples); -1: none
#endif
#ifndef NERF_S16_SPREAD_STORES
#define NERF_S16_SPREAD_STORES 1
#endif
template <class SINK>
__device__ __forceinline__ void put_due(const SINK& sink, int KS, int g, int tiles, const bf16x8* oh, const bf16x8* ol) {
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const int d = g - quarter_pos(KS, 3) - s4;               // = (t + 1) KS for the tile t whose store s4 is due now
    if (d >= KS && d % KS == 0 && d / KS - 1 < tiles) {
      const int f = 2 * (d / KS - 1) + (s4 >> 1);
      sink.put1(f, (s4 & 1) != 0, (s4 & 1) ? ol[f] : oh[f]);
    }
  }
}

// Epilogue of one accumulator tile in four quarters (issued between the MFMAs of the NEXT n-tile: one wave per SIMD, nothing
// else fills the matrix pipe while a wave does vector work).  f0 / f1: the two 16-feature fragments of the 32-row tile.
// Sign-bit words as in mlp.hip (finish_quarter): bit 16 odd + 8 (nt & 1) + k of word nt >> 1 for element 2 k + odd.
template <bool RELU, bool MASKOUT>
__device__ __forceinline__ void finish_quarter(const f32x16& acc, int q, int nt, bf16x8& f0h, bf16x8& f1h, bf16x8& f0l,
                                               bf16x8& f1l, u32x4& mask) {
  unsigned w = 0;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int i = 4 * q + 2 * p, k = 2 * q + p;
    const float a = RELU ? relu_bits(acc[i]) : acc[i], b = RELU ? relu_bits(acc[i + 1]) : acc[i + 1];
    const HL s = split2(a, b);
    if (MASKOUT) w |= nonzero_bits(s.h) << k;
    if (i < 8) { f0h[i] = s.h[0]; f0h[i + 1] = s.h[1]; f0l[i] = s.l[0]; f0l[i + 1] = s.l[1]; }
    else { f1h[i - 8] = s.h[0]; f1h[i - 7] = s.h[1]; f1l[i - 8] = s.l[0]; f1l[i - 7] = s.l[1]; }
  }
  if (MASKOUT) mask[nt >> 1] |= w << (8 * (nt & 1));
}

// out[2 nt + s] = split( act( W[nt-tile] . in + bias ) );  stream fragments 2 (fbase + nt KS + ks) = hi, + 1 = lo
template <int KS, int NT, bool RELU, bool MASKOUT, class WS, class SINK>
__device__ __forceinline__ void layer_fwd(WS& ws, int fbase, int bias_slot, const bf16x8 (&ih)[KS], const bf16x8 (&il)[KS],
                                          bf16x8 (&oh)[2 * NT], bf16x8 (&ol)[2 * NT], u32x4& mask, int lane,
                                          const SINK& sink) {
  const int h = lane >> 5;
  f32x16 prev;
#if NERF_S16_IGLP >= 0
  __builtin_amdgcn_iglp_opt(NERF_S16_IGLP);
#endif
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    f32x16 acc;
    acc_init_bias(acc, ws, bias_slot + 32 * nt, h);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int fp = fbase + nt * KS + ks;
#if NERF_ABLATE == 11 || NERF_ABLATE == 12      // timing-only builds: no weight fragments from the ring (no LDS reads, no ring protocol)
      const bf16x8 ah = ih[(ks + 1) % KS], al = il[(ks + 1) % KS];
      (void)fp;
#else
      const bf16x8 ah = next_frag(ws, 2 * fp, lane);
      const bf16x8 al = next_frag(ws, 2 * fp + 1, lane);
#endif
      acc = mfma32(ah, il[ks], acc);
      acc = mfma32(al, ih[ks], acc);
      acc = mfma32(ah, ih[ks], acc);
#if NERF_ABLATE == 10 || NERF_ABLATE == 12      // timing-only builds: no epilogue (the next layer computes on this layer's inputs)
      if (ks == 0) { oh[2 * nt] = ih[(2 * nt) % KS]; oh[2 * nt + 1] = ih[(2 * nt + 1) % KS]; ol[2 * nt] = il[(2 * nt) % KS]; ol[2 * nt + 1] = il[(2 * nt + 1) % KS]; }
      if (ks == KS - 1) asm volatile("" :: "v"(acc));
      if (false)
#endif
      if (nt > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (quarter_pos(KS, q) == ks) {
            finish_quarter<RELU, MASKOUT>(prev, q, nt - 1, oh[2 * nt - 2], oh[2 * nt - 1], ol[2 * nt - 2], ol[2 * nt - 1], mask);
            if (q == 3 && !NERF_S16_SPREAD_STORES) { sink.put(2 * nt - 2, oh[2 * nt - 2], ol[2 * nt - 2]); sink.put(2 * nt - 1, oh[2 * nt - 1], ol[2 * nt - 1]); }
          }
      }
      if (NERF_S16_SPREAD_STORES) put_due(sink, KS, nt * KS + ks, NT - 1, oh, ol);
    }
    prev = acc;
  }
#if NERF_ABLATE == 10 || NERF_ABLATE == 12
  return;
#endif
#pragma unroll
  for (int q = 0; q < 4; ++q)
    finish_quarter<RELU, MASKOUT>(prev, q, NT - 1, oh[2 * NT - 2], oh[2 * NT - 1], ol[2 * NT - 2], ol[2 * NT - 1], mask);
  if (NERF_S16_SPREAD_STORES) {                            // stores of tiles < NT - 1 that were due behind the last k-step
#pragma unroll
    for (int g = NT * KS; g < NT * KS + quarter_pos(KS, 3) + 4; ++g) put_due(sink, KS, g, NT - 1, oh, ol);
  }
  sink.put(2 * NT - 2, oh[2 * NT - 2], ol[2 * NT - 2]);
  sink.put(2 * NT - 1, oh[2 * NT - 1], ol[2 * NT - 1]);
}

// a head of <= 4 valid rows (alpha: row 0; rgb: rows 0..2): one 32-row tile, KS k-steps; rows (i & 3) + 8 (i >> 2) + 4 h
template <int KS, class WS>
__device__ __forceinline__ f32x16 head(WS& ws, int fbase, int bias_slot, const bf16x8 (&ih)[KS], const bf16x8 (&il)[KS], int lane) {
  f32x16 acc;
  acc_init_bias(acc, ws, bias_slot, lane >> 5);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const bf16x8 ah = next_frag(ws, 2 * (fbase + ks), lane);
    const bf16x8 al = next_frag(ws, 2 * (fbase + ks) + 1, lane);
    acc = mfma32(ah, il[ks], acc);
    acc = mfma32(al, ih[ks], acc);
    acc = mfma32(ah, ih[ks], acc);
  }
  return acc;
}

// inference: finished fragment pairs go nowhere but the next layer
struct NoPairSink {
  __device__ __forceinline__ void put(int, const bf16x8&, const bf16x8&) const {}
  __device__ __forceinline__ void put1(int, bool, const bf16x8&) const {}
};

// ------------------------------------------------------------------------------------------
// backward chain: dZ_l for every layer (stored as hi and lo fragment blocks for the dW kernel)
// ------------------------------------------------------------------------------------------
template <bool MASK, int Q, int ODD>
__device__ __forceinline__ void finish_quarter_bwd_t(const f32x16& acc, bf16x8& f0h, bf16x8& f1h, bf16x8& f0l, bf16x8& f1l,
                                                     unsigned w) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int i = 4 * Q + 2 * p;
    HL s = split2(acc[i], acc[i + 1]);
    if (MASK) {                                           // ReLU' from the forward's sign bits, on both parts
      if (p == 0) { s.h = keep_where<8 * ODD + 2 * Q>(s.h, w); s.l = keep_where<8 * ODD + 2 * Q>(s.l, w); }
      else { s.h = keep_where<8 * ODD + 2 * Q + 1>(s.h, w); s.l = keep_where<8 * ODD + 2 * Q + 1>(s.l, w); }
    }
    if (i < 8) { f0h[i] = s.h[0]; f0h[i + 1] = s.h[1]; f0l[i] = s.l[0]; f0l[i + 1] = s.l[1]; }
    else { f1h[i - 8] = s.h[0]; f1h[i - 7] = s.h[1]; f1l[i - 8] = s.l[0]; f1l[i - 7] = s.l[1]; }
  }
}
template <bool MASK>
__device__ __forceinline__ void finish_quarter_bwd(const f32x16& acc, int q, int kt, bf16x8& f0h, bf16x8& f1h, bf16x8& f0l,
                                                   bf16x8& f1l, const u32x4& mask) {
  const unsigned w = MASK ? mask[kt >> 1] : 0u;
  switch (2 * q + (kt & 1)) {        // q and kt are compile-time constants at every call site: the switch folds away
    case 0: finish_quarter_bwd_t<MASK, 0, 0>(acc, f0h, f1h, f0l, f1l, w); break;
    case 1: finish_quarter_bwd_t<MASK, 0, 1>(acc, f0h, f1h, f0l, f1l, w); break;
    case 2: finish_quarter_bwd_t<MASK, 1, 0>(acc, f0h, f1h, f0l, f1l, w); break;
    case 3: finish_quarter_bwd_t<MASK, 1, 1>(acc, f0h, f1h, f0l, f1l, w); break;
    case 4: finish_quarter_bwd_t<MASK, 2, 0>(acc, f0h, f1h, f0l, f1l, w); break;
    case 5: finish_quarter_bwd_t<MASK, 2, 1>(acc, f0h, f1h, f0l, f1l, w); break;
    case 6: finish_quarter_bwd_t<MASK, 3, 0>(acc, f0h, f1h, f0l, f1l, w); break;
    default: finish_quarter_bwd_t<MASK, 3, 1>(acc, f0h, f1h, f0l, f1l, w); break;
  }
}

// out[2 kt + s] = split( mask( W^T[kt-tile] . in ) );  mask = ReLU sign bits written by the forward kernel
template <int NS, int KT, bool MASK, class WS, class SINK>
__device__ __forceinline__ void layer_bwd(WS& ws, int fbase, const bf16x8 (&ih)[NS], const bf16x8 (&il)[NS],
                                          bf16x8 (&oh)[2 * KT], bf16x8 (&ol)[2 * KT], const u32x4& mask, int lane,
                                          const SINK& sink) {
  f32x16 prev;
#if NERF_S16_IGLP >= 0
  __builtin_amdgcn_iglp_opt(NERF_S16_IGLP);
#endif
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      const int fp = fbase + kt * NS + ns;
      const bf16x8 ah = next_frag(ws, 2 * fp, lane);
      const bf16x8 al = next_frag(ws, 2 * fp + 1, lane);
      acc = mfma32(ah, il[ns], acc);
      acc = mfma32(al, ih[ns], acc);
      acc = mfma32(ah, ih[ns], acc);
      if (kt > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (quarter_pos(NS, q) == ns) {
            finish_quarter_bwd<MASK>(prev, q, kt - 1, oh[2 * kt - 2], oh[2 * kt - 1], ol[2 * kt - 2], ol[2 * kt - 1], mask);
            if (q == 3 && !NERF_S16_SPREAD_STORES) { sink.put(2 * kt - 2, oh[2 * kt - 2], ol[2 * kt - 2]); sink.put(2 * kt - 1, oh[2 * kt - 1], ol[2 * kt - 1]); }
          }
      }
      if (NERF_S16_SPREAD_STORES) put_due(sink, NS, kt * NS + ns, KT - 1, oh, ol);
    }
    prev = acc;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    finish_quarter_bwd<MASK>(prev, q, KT - 1, oh[2 * KT - 2], oh[2 * KT - 1], ol[2 * KT - 2], ol[2 * KT - 1], mask);
  if (NERF_S16_SPREAD_STORES) {
#pragma unroll
    for (int g = KT * NS; g < KT * NS + quarter_pos(NS, 3) + 4; ++g) put_due(sink, NS, g, KT - 1, oh, ol);
  }
  sink.put(2 * KT - 2, oh[2 * KT - 2], ol[2 * KT - 2]);
  sink.put(2 * KT - 1, oh[2 * KT - 1], ol[2 * KT - 1]);
}

}  // namespace s16
}  // namespace nerf
