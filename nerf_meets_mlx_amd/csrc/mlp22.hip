// Split-fp16 ("precision 22") inference forward of the fused 8 x 256 NeRF MLP for gfx950 (a8-a12 / K3, K4 at the
// reference's float32 tolerance, on the fp16 matrix pipe).
//
// The reference computes NeRF.forward in float32 (models/NeRF.py:201-243).  The fp32 MFMA (v_mfma_f32_32x32x2_f32,
// mlp32.hip) runs at 1/16 of the 16-bit rate.  Here every float32 operand is carried as two fp16 numbers
//     x = hi + lo * 2^-11,   hi = fp16(x),  lo = fp16((x - hi) * 2^11)          (22 significand bits; lo is SCALED so that
// it is a normal fp16 number whenever x is) and a product is evaluated as
//     w x  ~=  w_hi x_hi  +  2^-11 (w_hi x_lo + w_lo x_hi)                       (the dropped w_lo x_lo is 2^-22 relative)
// = three v_mfma_f32_16x16x32_f16 into TWO fp32 accumulators (main, correction) that the layer epilogue combines, adds
// nothing else to (the bias is the main accumulator's initial value), ReLUs, and splits again into the next layer's (hi, lo)
// B fragments -- the chain of mlp.hip's 16x16x32 render kernel (accumulator tile = next layer's B operand, weights streamed
// through the shared LDS ring by LDS-DMA) with a doubled weight stream and a doubled activation register set.
//
// One wave = 32 samples (two 16-sample column tiles), ONE wave per SIMD (4 waves per workgroup): the two register sets
// (input and output layer, hi and lo: 256 VGPRs) do not leave room for a second wave, and with 6 MFMAs per pair of 1 KiB
// fragment reads the LDS traffic per MFMA is 2/3 of the bf16 kernel's.
#include "mlp_layout.h"
#include "mlp22.h"

// 1: OCML sinf / cosf for the positional encodings; 0: two-fma Cody-Waite reduction + v_sin_f32 (tools/f16_probe.hip
// measures both against sin() in double on the encodings' argument range)
#ifndef NERF_F22_SINF
#define NERF_F22_SINF 0
#endif

// A/B switches of the layer epilogue (see layer22)
#ifndef NERF_F22_PIN
#define NERF_F22_PIN 1
#endif
#ifndef NERF_F22_AGPR
#define NERF_F22_AGPR 1
#endif
#ifndef NERF_F22_MIXSUB
#define NERF_F22_MIXSUB 1     // lo part of the activation split: v - hi as one v_fma_mix_f32 (see split2); 0: v_cvt_f32_f16 + v_sub_f32
#endif
#ifndef NERF_F22_RG3
#define NERF_F22_RG3 4        // LDS weight fragments fetched per software-pipeline group in the 48-sample kernel (2: 16 registers less, no spill, 1.3 % slower)
#endif
#ifndef NERF_F22_IGLP
#define NERF_F22_IGLP 0       // __builtin_amdgcn_iglp_opt strategy of the layer body; -1: none
#endif

namespace nerf {
namespace f22 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr float LO_SCALE = 2048.0f, LO_INV = 1.0f / 2048.0f;
constexpr float F16_MIN_NORMAL = 6.103515625e-05f;          // 2^-14
constexpr int CHUNKS = F_PADDED / RING_CHUNK;               // 74
static_assert(CHUNKS * RING_CHUNK == F_PADDED && F_FRAGS % 4 == 0 && F_PAIRS == L::F16_TOTAL, "f22 stream layout");

// two float32 values -> packed fp16 pair of their leading 11 bits, packed fp16 pair of the (scaled) remainders.
// v - hi is exact in float32 (hi is v rounded to 11 bits), so hi + lo / 2^11 == v up to the rounding of lo: 2^-22 |v|.
struct HiLo { unsigned hi, lo; };
__device__ __forceinline__ HiLo split2(float v0, float v1) {
  const f32x2 v = {v0, v1};
  const h2 hh = __builtin_convertvector(v, h2);                                   // v_cvt_pk_f16_f32 (round to nearest even)
  // v - hi as ONE v_fma_mix_f32 (the fp16 half is an operand of the mixed-precision fma: no v_cvt_f32_f16 + v_sub_f32 pair; the
  // difference is exact either way, so the results are bit-identical): 10 instead of 12 vector instructions per value pair
  // (written with the scale inside: fma(hi, -1, v) is folded back into a subtraction; v 2^11 and hi 2^11 are exact)
#if NERF_F22_MIXSUB
  const f32x2 r = {__builtin_fmaf((float)hh[0], -LO_SCALE, v0 * LO_SCALE), __builtin_fmaf((float)hh[1], -LO_SCALE, v1 * LO_SCALE)};
#else
  const f32x2 r = {(v0 - (float)hh[0]) * LO_SCALE, (v1 - (float)hh[1]) * LO_SCALE};
#endif
  return HiLo{__builtin_bit_cast(unsigned, hh), __builtin_bit_cast(unsigned, __builtin_convertvector(r, h2))};
}

// ------------------------------------------------------------------------------------------
// packing: fp32 master parameters -> (hi, lo) fp16 fragment pairs in the 16x16x32 stream order of mlp_layout.h
// ------------------------------------------------------------------------------------------
constexpr int PACK_THREADS = (F_PADDED / 2) * 64 + BIAS_FLOATS;
__global__ void __launch_bounds__(256) pack22_kernel(const float* __restrict__ p, u32x4* __restrict__ wf,
                                                     float* __restrict__ bias) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < (F_PADDED / 2) * 64) {
    const int fp = t >> 6, lane = t & 63;
    u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
    if (fp < F_PAIRS) {
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        float w[2];
        _Float16 wh[2], wl[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          w[e] = fwd_src16(p, fp, lane & 15, lane >> 4, j + e);
          // weights below the fp16 normal range go to the lo part whole (hi = 0): nothing rests on fp16 denormals
          wh[e] = __builtin_fabsf(w[e]) < F16_MIN_NORMAL ? (_Float16)0.0f : (_Float16)w[e];
          wl[e] = (_Float16)((w[e] - (float)wh[e]) * LO_SCALE);
        }
        const h2 a = {wh[0], wh[1]}, b = {wl[0], wl[1]};
        hi[j >> 1] = __builtin_bit_cast(unsigned, a);
        lo[j >> 1] = __builtin_bit_cast(unsigned, b);
      }
    }
    wf[(2 * fp) * 64 + lane] = hi;
    wf[(2 * fp + 1) * 64 + lane] = lo;
  } else if (t < PACK_THREADS) {
    const int s = t - (F_PADDED / 2) * 64;                  // bias slots: numbering of the bf16 image (mlp.hip pack_part)
    float v = 0.0f;
    if (s < 2048) v = p[L::pb(s >> 8) + (s & 255)];
    else if (s < L::BI_ALPHA) v = p[L::P_BF + (s - L::BI_FEAT)];
    else if (s < L::BI_DIR) v = (s == L::BI_ALPHA) ? p[L::P_BA] : 0.0f;
    else if (s < L::BI_RGB) v = p[L::P_BD + (s - L::BI_DIR)];
    else v = (s - L::BI_RGB) < 3 ? p[L::P_BR + (s - L::BI_RGB)] : 0.0f;
    bias[s] = v;
  }
}

// ------------------------------------------------------------------------------------------
// positional encodings (models/embedding.py:30-71), float32-accurate, straight into (hi, lo) B fragments
// ------------------------------------------------------------------------------------------
// sin(x f + 2 pi ph), ph in {0, 1/4} (cos).  x f is the reference's float32 product.
__device__ __forceinline__ float sin_acc(float x, float f, float ph) {
  const float a = x * f;
#if NERF_F22_SINF
  return ph != 0.0f ? cosf(a) : sinf(a);
#else
  // a - k 2 pi with 2 pi = HI + LO in two fused steps (|k| <= 80: exact products), then v_sin_f32 on revolutions in [-3/4, 3/4]
  const float k = __builtin_rintf(a * 0.15915494309189535f);
  float r = __builtin_fmaf(-k, 6.2831854820251465f, a);
  r = __builtin_fmaf(-k, -1.7484556000744487e-07f, r);
  return __builtin_amdgcn_sinf(__builtin_fmaf(r, 0.15915494309189535f, ph));
#endif
}

struct Pe22 { float fa, fb, fc, phc, fd; int g; };          // per-lane-group frequencies of the channel order (pos_chan16)
__device__ __forceinline__ Pe22 pe22_setup(const PeFreq& fr, int g) {
  Pe22 q;
  q.fa = g == 0 ? fr.pos[0] : g == 1 ? fr.pos[2] : g == 2 ? fr.pos[4] : fr.pos[6];
  q.fb = g == 0 ? fr.pos[1] : g == 1 ? fr.pos[3] : g == 2 ? fr.pos[5] : fr.pos[7];
  q.fc = g < 2 ? fr.pos[8] : fr.pos[9];
  q.phc = (g & 1) ? 0.25f : 0.0f;
  q.fd = g == 0 ? fr.dir[0] : g == 1 ? fr.dir[1] : g == 2 ? fr.dir[2] : fr.dir[3];
  q.g = g;
  return q;
}

template <int COUNT>
__device__ __forceinline__ void split_slots(const float (&v)[COUNT], u32x4* hi, u32x4* lo) {
#pragma unroll
  for (int j = 0; j < COUNT; j += 2) {
    const HiLo q = split2(v[j], v[j + 1]);
    hi[j >> 3][(j & 7) >> 1] = q.hi; lo[j >> 3][(j & 7) >> 1] = q.lo;
  }
}

// MODE 1: from a position p and a view direction d
__device__ __forceinline__ void pe22_pos(const Pe22& q, const float (&p)[3], u32x4 (&peh)[2], u32x4 (&pel)[2]) {
  float v[16];
#pragma unroll
  for (int sl = 0; sl < 12; ++sl) v[sl] = sin_acc(p[sl % 3], sl < 6 ? q.fa : q.fb, (sl % 6) >= 3 ? 0.25f : 0.0f);
#pragma unroll
  for (int k = 0; k < 3; ++k) v[12 + k] = sin_acc(p[k], q.fc, q.phc);
  v[15] = q.g == 0 ? p[0] : q.g == 1 ? p[1] : q.g == 2 ? p[2] : 0.0f;
  split_slots<16>(v, peh, pel);
}
__device__ __forceinline__ void pe22_dir(const Pe22& q, const float (&d)[3], u32x4 (&dph)[1], u32x4 (&dpl)[1]) {
  float w[8];
#pragma unroll
  for (int j = 0; j < 6; ++j) w[j] = sin_acc(d[j % 3], q.fd, j >= 3 ? 0.25f : 0.0f);
  w[6] = q.g == 0 ? d[0] : q.g == 1 ? d[1] : q.g == 2 ? d[2] : 0.0f;
  w[7] = 0.0f;
  split_slots<8>(w, dph, dpl);
}
__device__ __forceinline__ void pe22_encode(const Pe22& q, const float (&p)[3], const float (&d)[3], u32x4 (&peh)[2],
                                            u32x4 (&pel)[2], u32x4 (&dph)[1], u32x4 (&dpl)[1]) {
  pe22_pos(q, p, peh, pel);
  pe22_dir(q, d, dph, dpl);
}
// MODE 0: from an already-embedded row x[90] = [embed(pos) 63 | embed(dir) 27]   (NeRF.forward(x) entry)
__device__ __forceinline__ void pe22_row(const float* __restrict__ row, int g, u32x4 (&peh)[2], u32x4 (&pel)[2],
                                         u32x4 (&dph)[1], u32x4 (&dpl)[1]) {
  float v[16], w[8];
#pragma unroll
  for (int sl = 0; sl < 16; ++sl) {
    const int ch = pos_chan16(sl >> 3, g, sl & 7);
    v[sl] = ch >= 0 ? row[ch] : 0.0f;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = dir_chan16(g, j);
    w[j] = ch >= 0 ? row[63 + ch] : 0.0f;
  }
  split_slots<16>(v, peh, pel);
  split_slots<8>(w, dph, dpl);
}

// ------------------------------------------------------------------------------------------
// one linear layer on register-resident (hi, lo) activations
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
#if NERF_ABLATE == 41 || NERF_ABLATE == 42      // timing-only builds: 41 the ring protocol alone (DMAs, waits, barriers); 42 the same + the LDS weight reads
  return c;
#endif
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// out[s][nt>>1] (registers 2 (nt&1), +1) = split( act( W[16-row tile nt] . in[s] + bias ) ), s = 0 .. NS-1 (sample tiles of 16).
// Stream fragments 2 (pbase + nt KS + ks) = hi and + 1 = lo of the weight tile.  NS = 2: the shipped form (32 samples per wave);
// NS = 3: 48 samples per wave -- every weight fragment pair read from the LDS then feeds 9 MFMAs instead of 6 (a third fewer
// LDS reads, ring DMAs and chunk barriers per sample), at 1.5 x the activation registers.
// PARK_LDS != 0 (pos5 of the 48-sample form, the one layer whose inputs + outputs + accumulators + weight groups exceed the 512
// registers by one fragment): the LAST input fragment (lo part of k-step KS - 1, sample tile NS - 1) lives in the LDS for the length of
// the layer -- written once at its head, read back behind the weight fragments of every n-tile (16 ds_read_b128 of 2 400 per pass).
// Left to hipcc the same fragment goes to scratch, and a scratch reload is a compiler-counted VMEM load: s_waitcnt vmcnt(0), a drain
// of the weight ring in the middle of a pass.  park_addr: LDS byte address of this lane's 16 bytes.
template <int NS, int KS, int NT, bool RELU, class WS, bool PARK_LDS = false>
__device__ __forceinline__ void layer22(WS& ws, int pbase, int bias_slot, const u32x4 (&ih)[NS][KS], const u32x4 (&il)[NS][KS],
                                        u32x4 (&oh)[NS][NT / 2], u32x4 (&ol)[NS][NT / 2], int lane, unsigned park_addr = 0u) {
  const int g = lane >> 4;
  f32x4 pm[NS], pc[NS];
  if (PARK_LDS) asm volatile("ds_write_b128 %0, %1" :: "v"(park_addr), "v"(il[NS - 1][KS - 1]) : "memory");
#if NERF_F22_IGLP >= 0
  // LLVM's MFMA-interleaving scheduling strategy for this region (the layer is one basic block): the static gap model goes from 0.770
  // to 0.792 busy (tools/isa_gap_stats.py), measured 15.6-16.1 against 16.1-16.3 ms per fine pass in alternating runs (round 5).
  // (An explicit sched_group_barrier pipeline -- one MFMA, two fillers, per k-step -- reads 0.81 in the same model and measures
  // SLOWER, 15.95 against 15.82 ms: the model has no latencies.)
  __builtin_amdgcn_iglp_opt(NERF_F22_IGLP);
#endif
  // Epilogue of a finished tile in 2 NS pieces (sample tile s = piece >> 1, register pair i = 2 (piece & 1)), issued
  // between the MFMAs of the NEXT tile's k-steps: this kernel runs one wave per SIMD, so an epilogue done in one block
  // (~60 VALU instructions) leaves the matrix pipe idle for its whole length -- there is no second wave to fill it.
  constexpr int NP = 2 * NS;
  auto piece = [&](int nt, int pcs, f32x4 (&m)[NS], f32x4 (&c)[NS]) {
    const int s = pcs >> 1, i = 2 * (pcs & 1);
#if NERF_ABLATE == 41 || NERF_ABLATE == 42
    return;
#endif
#if NERF_F22_PIN
    if (pcs == 0) {
#pragma unroll
      for (int t = 0; t < NS; ++t) asm volatile("" : "+v"(m[t]), "+v"(c[t]));
    }
#endif
    float v0 = __builtin_fmaf(c[s][i], LO_INV, m[s][i]), v1 = __builtin_fmaf(c[s][i + 1], LO_INV, m[s][i + 1]);
    // v_maximum3_f32 (IEEE-754 maximum): a NaN pre-activation stays NaN like mx.maximum / nn.relu (models/NeRF.py:222,236);
    // fmaxf (v_max_f32, maxNum) would return 0 for it
    if (RELU) { v0 = __builtin_elementwise_maximum(v0, 0.0f); v1 = __builtin_elementwise_maximum(v1, 0.0f); }
    const HiLo q = split2(v0, v1);
    oh[s][nt >> 1][2 * (nt & 1) + (i >> 1)] = q.hi; ol[s][nt >> 1][2 * (nt & 1) + (i >> 1)] = q.lo;
#if NERF_F22_AGPR
    // the finished B fragment (both n-tiles of the 32-feature k-step written) goes to the accumulation registers: the MFMA
    // reads B from either file, and the 256 activation registers of the two layers then leave the arch VGPRs to the
    // accumulators, which the epilogue reads with VALU instructions (no v_accvgpr_read per accumulator register)
    if ((nt & 1) && i == 2) asm volatile("" : "+a"(oh[s][nt >> 1]), "+a"(ol[s][nt >> 1]));
#endif
  };
  // k-step of the next tile behind which piece q of the previous tile's epilogue is issued (NS = 2: the tuned round-5 positions)
  auto piece_at = [](int q) { return NS == 2 ? (KS >= 8 ? 1 + 2 * q : (q * KS) / 4) : (KS >= 8 ? 1 + (q * (KS - 1)) / NP : (q * KS) / NP); };
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float4 b = ws.bias4(bias_slot + 16 * nt + 4 * g);
    f32x4 m[NS], c[NS];
    m[0][0] = b.x; m[0][1] = b.y; m[0][2] = b.z; m[0][3] = b.w;
    c[0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 1; t < NS; ++t) { m[t] = m[0]; c[t] = c[0]; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int fp = pbase + nt * KS + ks;
      const u32x4 ah = __builtin_bit_cast(u32x4, next_frag(ws, 2 * fp, lane));
      const u32x4 al = __builtin_bit_cast(u32x4, next_frag(ws, 2 * fp + 1, lane));
      if (NS == 2) {                 // (the order of the shipped kernel)
        m[0] = mfma16(ah, ih[0][ks], m[0]);
        c[0] = mfma16(ah, il[0][ks], c[0]);
        m[1] = mfma16(ah, ih[1][ks], m[1]);
        c[1] = mfma16(ah, il[1][ks], c[1]);
        c[0] = mfma16(al, ih[0][ks], c[0]);
        c[1] = mfma16(al, ih[1][ks], c[1]);
      } else {
#pragma unroll
        for (int t = 0; t < NS; ++t) {
          m[t] = mfma16(ah, ih[t][ks], m[t]);
          if (PARK_LDS && t == NS - 1 && ks == KS - 1) {
            u32x4 b;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(park_addr) : "memory");
            c[t] = mfma16(ah, b, c[t]);
          } else {
            c[t] = mfma16(ah, il[t][ks], c[t]);
          }
        }
#pragma unroll
        for (int t = 0; t < NS; ++t) c[t] = mfma16(al, ih[t][ks], c[t]);
      }
      if (nt > 0) {
#pragma unroll
        for (int q = 0; q < NP; ++q)
          if (piece_at(q) == ks) piece(nt - 1, q, pm, pc);
      }
    }
#pragma unroll
    for (int t = 0; t < NS; ++t) { pm[t] = m[t]; pc[t] = c[t]; }
  }
#pragma unroll
  for (int q = 0; q < NP; ++q) piece(NT - 1, q, pm, pc);
}

// a head of <= 4 rows (alpha: row 0; rgb: rows 0..2): one 16-row tile, KS k-steps, result row r of sample tile s in out[s][r]
template <int NS, int KS, class WS>
__device__ __forceinline__ void head22(WS& ws, int pbase, int bias_slot, const u32x4 (&ih)[NS][KS], const u32x4 (&il)[NS][KS],
                                       f32x4 (&out)[NS], int lane) {
  const int g = lane >> 4;
  const float4 b = ws.bias4(bias_slot + 4 * g);
  f32x4 m[NS], c[NS];
  m[0][0] = b.x; m[0][1] = b.y; m[0][2] = b.z; m[0][3] = b.w;
  c[0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 1; t < NS; ++t) { m[t] = m[0]; c[t] = c[0]; }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const u32x4 ah = __builtin_bit_cast(u32x4, next_frag(ws, 2 * (pbase + ks), lane));
    const u32x4 al = __builtin_bit_cast(u32x4, next_frag(ws, 2 * (pbase + ks) + 1, lane));
    if (NS == 2) {
      m[0] = mfma16(ah, ih[0][ks], m[0]);
      c[0] = mfma16(ah, il[0][ks], c[0]);
      m[1] = mfma16(ah, ih[1][ks], m[1]);
      c[1] = mfma16(ah, il[1][ks], c[1]);
      c[0] = mfma16(al, ih[0][ks], c[0]);
      c[1] = mfma16(al, ih[1][ks], c[1]);
    } else {
#pragma unroll
      for (int t = 0; t < NS; ++t) { m[t] = mfma16(ah, ih[t][ks], m[t]); c[t] = mfma16(ah, il[t][ks], c[t]); }
#pragma unroll
      for (int t = 0; t < NS; ++t) c[t] = mfma16(al, ih[t][ks], c[t]);
    }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) out[s][i] = __builtin_fmaf(c[s][i], LO_INV, m[s][i]);
}

// All 12 layers for the wave's 16 NS samples (sample tiles NS wtile0 .. NS wtile0 + NS - 1 of 16).  Waves past the end compute on
// clamped inputs and store nothing, so every wave of the workgroup runs the same instruction stream (the ring needs that).
// NS = 3 (MODE 1 only): the encodings are not kept through the pass -- 72 registers the 48-sample form does not have -- but
// re-evaluated from the 6 floats of a sample where they are read again (pos5's skip input, the view layer).
// RECOMP parks the 6 floats per sample in the LDS (18 KiB behind the bias slots: 2 x 3 x NS dwords per lane, lane-major): one lane reads
// back what it wrote itself, LDS operations of a wave execute in order -- no barrier.  Inline asm: a plain LDS load would be hoisted to
// the head of the pass (back into the registers this is meant to free); in scratch (what hipcc does with them when left in registers:
// 21 spilled VGPRs) every reload is a compiler-counted VMEM load, i.e. an s_waitcnt vmcnt(0) that drains the weight ring.
constexpr int NW22 = 4, NW22_ = NW22;
constexpr int PARK_OFF = RING_LDS_BYTES;                                  // behind ring + bias slots
template <int NS> constexpr int park_bytes() { return NS > 2 ? (6 * NS + 4) * 64 * NW22_ * 4 : 0; }      // + one 16-byte fragment per lane (layer22, PARK_LDS)
// (the slot offset is an immediate of the instruction: ONE address register for all 18 values)
template <int K> __device__ __forceinline__ void lds_put(unsigned addr, float v) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(v), "i"(1024 * K) : "memory");
}
template <int K> __device__ __forceinline__ float lds_get(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "i"(1024 * K) : "memory");
  return v;
}
template <int S> __device__ __forceinline__ void park_put(unsigned park, const float (&p)[3], const float (&d)[3]) {
  lds_put<6 * S + 0>(park, p[0]); lds_put<6 * S + 1>(park, p[1]); lds_put<6 * S + 2>(park, p[2]);
  lds_put<6 * S + 3>(park, d[0]); lds_put<6 * S + 4>(park, d[1]); lds_put<6 * S + 5>(park, d[2]);
}
template <int S, int OFF> __device__ __forceinline__ void park_get(unsigned park, float (&v)[3]) {
  v[0] = lds_get<6 * S + OFF>(park); v[1] = lds_get<6 * S + OFF + 1>(park); v[2] = lds_get<6 * S + OFF + 2>(park);
}
// s is a constant after unrolling: dispatch to the immediate forms
__device__ __forceinline__ void park_put_s(int s, unsigned park, const float (&p)[3], const float (&d)[3]) {
  if (s == 0) park_put<0>(park, p, d); else if (s == 1) park_put<1>(park, p, d); else park_put<2>(park, p, d);
}
template <int OFF> __device__ __forceinline__ void park_get_s(int s, unsigned park, float (&v)[3]) {
  if (s == 0) park_get<0, OFF>(park, v); else if (s == 1) park_get<1, OFF>(park, v); else park_get<2, OFF>(park, v);
}

template <int NS, int MODE, class WS>
__device__ __forceinline__ void tiles22(const FwdArgs& a, WS& ws, int64_t wtile0, int64_t nwtiles, int lane, PassQueue& pq) {
  constexpr bool RECOMP = NS > 2;
  static_assert(!RECOMP || MODE == 1, "the 48-sample form takes rays + depths");
  const int c = lane & 15, g = lane >> 4;
  u32x4 peh[NS][2], pel[NS][2], dph[NS][1], dpl[NS][1];
  const unsigned park = ws.lds0 + PARK_OFF + 4u * (unsigned)(ws.wv * 64 + lane);      // + 1024 k: value k of this lane
  pq.ask(ws.wv, lane);                    // the workgroup's next pass: asked for before this pass's input loads (mlp_ring.h)
  const Pe22 q = pe22_setup(a.fr, g);
  {
    const int64_t wt = wtile0 < nwtiles ? wtile0 : nwtiles - 1;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      int64_t m = wt * (16 * NS) + 16 * s + c; if (m >= a.M) m = a.M - 1;
      if (MODE == 0) {
        pe22_row(a.x + m * 90, g, peh[s], pel[s], dph[s], dpl[s]);
      } else {
        const int64_t ray = (int64_t)((unsigned)m / (unsigned)a.n);            // M < 2^31 (checked on the host)
        const float* rr = a.rays + ray * NERF_RAY_STRIDE;
        float p[3], d[3];
#if NERF_ABLATE == 3          // timing-only build 3: no input loads at the head of a pass (positions from the lane id)
        (void)rr;
#pragma unroll
        for (int k = 0; k < 3; ++k) { p[k] = 0.001f * (float)(m & 1023) + (float)k; d[k] = 0.5f; }
#else
        const float zv = a.z[m];
#pragma unroll
        for (int k = 0; k < 3; ++k) { p[k] = rr[k] + zv * rr[3 + k]; d[k] = rr[8 + k]; }   // render.py:142
#endif
        if (RECOMP) {
          park_put_s(s, park, p, d);
          pe22_pos(q, p, peh[s], pel[s]);
        } else {
          pe22_encode(q, p, d, peh[s], pel[s], dph[s], dpl[s]);
        }
      }
    }
  }
  pq.publish(ws.wv);
  u32x4 hah[NS][8], hal[NS][8], hbh[NS][8], hbl[NS][8];
  layer22<NS, 2, 16, true>(ws, L16::F_L0, 0, peh, pel, hah, hal, lane);
  layer22<NS, 8, 16, true>(ws, L16::F_L1 + 0 * 128, 256, hah, hal, hbh, hbl, lane);
  layer22<NS, 8, 16, true>(ws, L16::F_L1 + 1 * 128, 512, hbh, hbl, hah, hal, lane);
  layer22<NS, 8, 16, true>(ws, L16::F_L1 + 2 * 128, 768, hah, hal, hbh, hbl, lane);
  layer22<NS, 8, 16, true>(ws, L16::F_L1 + 3 * 128, 1024, hbh, hbl, hah, hal, lane);
  {                                                           // pos5 on concat[input_pos, h]  (models/NeRF.py:224-225)
    u32x4 cth[NS][10], ctl[NS][10];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (RECOMP) {                                                             // re-evaluated HERE, not kept from the head of the pass
        float pp[3];
        park_get_s<0>(s, park, pp);
        pe22_pos(q, pp, peh[s], pel[s]);
      }
      cth[s][0] = peh[s][0]; cth[s][1] = peh[s][1]; ctl[s][0] = pel[s][0]; ctl[s][1] = pel[s][1];
#pragma unroll
      for (int k = 0; k < 8; ++k) { cth[s][2 + k] = hah[s][k]; ctl[s][2 + k] = hal[s][k]; }
    }
    layer22<NS, 10, 16, true, WS, (NS > 2)>(ws, L16::F_L5, 1280, cth, ctl, hbh, hbl, lane,
                                            ws.lds0 + PARK_OFF + 6 * NS * 64 * NW22 * 4 + 16u * (unsigned)(ws.wv * 64 + lane));
  }
  layer22<NS, 8, 16, true>(ws, L16::F_L6, 1536, hbh, hbl, hah, hal, lane);
  layer22<NS, 8, 16, true>(ws, L16::F_L7, 1792, hah, hal, hbh, hbl, lane);
  layer22<NS, 8, 16, false>(ws, L16::F_FA, L::BI_FEAT, hbh, hbl, hah, hal, lane);       // feature: no activation (:231)
  f32x4 alpha[NS];
  head22<NS, 8>(ws, L16::F_FA + 128, L::BI_ALPHA, hbh, hbl, alpha, lane);               // alpha = Linear(256, 1)(h) (:230)
  u32x4 hdh[NS][4], hdl[NS][4];
  {                                                           // relu(Linear(283, 128)([feature, input_dir]))  (:232-236)
    u32x4 cth[NS][9], ctl[NS][9];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { cth[s][k] = hah[s][k]; ctl[s][k] = hal[s][k]; }
      if (RECOMP) {
        float dd[3];
        park_get_s<3>(s, park, dd);
        pe22_dir(q, dd, dph[s], dpl[s]);
      }
      cth[s][8] = dph[s][0]; ctl[s][8] = dpl[s][0];
    }
    layer22<NS, 9, 8, true>(ws, L16::F_DIR, L::BI_DIR, cth, ctl, hdh, hdl, lane);
  }
  f32x4 rgb[NS];
  head22<NS, 4>(ws, L16::F_RGB, L::BI_RGB, hdh, hdl, rgb, lane);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int64_t m = wtile0 * (16 * NS) + 16 * s + c;
    if (g == 0 && wtile0 < nwtiles && m < a.M) {
      float4 o; o.x = rgb[s][0]; o.y = rgb[s][1]; o.z = rgb[s][2]; o.w = alpha[s][0];      // [rgb, alpha] raw (:239)
      *reinterpret_cast<float4*>(a.out + m * 4) = o;
    }
  }
}

template <int MODE, int NS = 2>
__global__ void __launch_bounds__(64 * NW22) mlp22_fwd_kernel(FwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwtiles = (a.M + 16 * NS - 1) / (16 * NS), nsuper = (nwtiles + NW22 - 1) / NW22;
  NERF_STAMP2_DECL();
  // DMA runs of four (mlp_ring.h); 48 samples per wave: fragment groups of 2 (16 registers less than groups of 4), refill DMAs still spread
  RingW<CHUNKS, F_FRAGS, (NS > 2 ? NERF_F22_RG3 : 4), NW22, RING_CHUNK, RING_STAGES, true, true> ws;
  ws.wsrc = reinterpret_cast<const char*>(a.wf);
  ws.lane16 = 16 * lane;
  ws.lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(ring_smem));
  ws.wv = wv;
  ws.start(lane);
  ring_load_bias(a.bias, BIAS_FLOATS);
  __syncthreads();
  NERF_STAMP2_LOOP();
  PassQueue pq;                           // dynamic pass queue (mlp_ring.h); a.queue == nullptr: static split
  pq.init(a.queue, ws.lds0 + RING_BIAS_OFF);
  for (int64_t sp = blockIdx.x; sp < nsuper;) {
    int ln = lane;
    asm volatile("" : "+v"(ln));          // lane-derived values are recomputed per pass, not hoisted and spilled
    ws.new_pass();
    tiles22<NS, MODE>(a, ws, sp * NW22 + wv, nwtiles, ln, pq);
    NERF_STAMP2_PASS();
    sp = pq.next(sp);
  }
  NERF_STAMP2_END();
  ws.drain();                             // the ring always runs 3 chunks ahead
  pq.leave();
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
int pack(const float* params, void* packed22, hipStream_t s) {
  char* base = static_cast<char*>(packed22);
  hipLaunchKernelGGL(pack22_kernel, dim3((PACK_THREADS + 255) / 256), dim3(256), 0, s, params,
                     reinterpret_cast<u32x4*>(base), reinterpret_cast<float*>(base + (size_t)F_PADDED * 1024));
  return check_launch("nerf_mlp_pack (split-fp16 image)");
}

// nerf_set_option("f22_tiles"): 16-sample tiles per wave of the rays + depths launch: 2 (32 samples) | 3 (48 samples) | 0 (default) = 3
// unless the launch is so small that whole passes per workgroup decide: a 48-sample pass takes 1.41 x a 32-sample pass (113 against 80
// us), so 4096 x 64 samples (the training step's re-query of the coarse network) are 8 x 1 against 6 x 1.41 pass times per workgroup
int g_tiles = 0;

int forward(const void* packed22, const float* x, const float* rays, const float* z, int64_t M, int n, int freq_mode,
            float* out, int persistent_wgs, hipStream_t s) {
  FwdArgs a;
  a.queue = passq_slot();
  const char* base = static_cast<const char*>(packed22);
  a.wf = reinterpret_cast<const bf16x8*>(base);
  a.bias = reinterpret_cast<const float*>(base + (size_t)F_PADDED * 1024);
  a.x = x; a.rays = rays; a.z = z; a.M = M; a.n = n; a.out = out; a.acts = nullptr; a.astride = 0;
  for (int k = 0; k < 10; ++k) a.fr.pos[k] = freq_mode == 0 ? (float)(k * k) : (float)(1 << k);
  for (int k = 0; k < 4; ++k) a.fr.dir[k] = freq_mode == 0 ? (float)(k * k) : (float)(1 << k);
  int tiles = g_tiles;
  if (tiles == 0) {
    const int64_t w = persistent_wgs > 0 ? persistent_wgs : 1;
    const int64_t p2 = (((M + 31) / 32 + NW22 - 1) / NW22 + w - 1) / w, p3 = (((M + 47) / 48 + NW22 - 1) / NW22 + w - 1) / w;     // passes of the busiest workgroup
    tiles = 100 * p2 < 141 * p3 ? 2 : 3;
  }
  const int mode = x ? 0 : (tiles == 3 ? 2 : 1);
  const int ns = mode == 2 ? 3 : 2;
  const int64_t nsuper = ((M + 16 * ns - 1) / (16 * ns) + NW22 - 1) / NW22;
  const dim3 g((unsigned)(nsuper < persistent_wgs ? nsuper : persistent_wgs)), b(64 * NW22);
  // dynamic LDS above 64 KiB is an opt-in per kernel AND per device
  static DevOnce once[3];
  once[mode].run([&] {
    const void* k = mode == 0 ? reinterpret_cast<const void*>(mlp22_fwd_kernel<0>) : mode == 1 ? reinterpret_cast<const void*>(mlp22_fwd_kernel<1>)
                                                                                                : reinterpret_cast<const void*>(mlp22_fwd_kernel<1, 3>);
    (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES + park_bytes<3>());
  });
  if (mode == 0) hipLaunchKernelGGL(mlp22_fwd_kernel<0>, g, b, RING_LDS_BYTES, s, a);
  else if (mode == 1) hipLaunchKernelGGL(mlp22_fwd_kernel<1>, g, b, RING_LDS_BYTES, s, a);
  else hipLaunchKernelGGL((mlp22_fwd_kernel<1, 3>), g, b, RING_LDS_BYTES + park_bytes<3>(), s, a);
  return check_launch("mlp forward (split fp16)");
}

}  // namespace f22
}  // namespace nerf
NERF_STAMP2_EXPORT(nerf_debug_stamps2_mlp22)
