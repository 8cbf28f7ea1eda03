// Split-bf16 training kernels of the fused 8 x 256 NeRF MLP for gfx950 (a11 and its adjoint at the reference's float32
// tolerance, on the bf16 matrix pipe): training forward with activation stores, dZ chain, dW / db.
//
// The reference computes NeRF.forward and its gradients in float32 (models/NeRF.py:201-243 under nn.value_and_grad,
// entrypoints/__test_nerf.py:240-293).  The fp32 MFMA (v_mfma_f32_32x32x2_f32, mlp32.hip) runs at 1/16 of the 16-bit rate.
// Here every float32 GEMM operand is carried as two bf16 numbers
//     x = hi + lo,   hi = bf16(x),  lo = bf16(x - hi)            (x - hi is exact in float32; 16 significand bits, float32's
// exponent range: gradients span far more than fp16's 30 binades, so the scaled-fp16 pairs of mlp22.hip do not apply here)
// and a product is evaluated as
//     a b  ~=  a_hi b_hi + a_hi b_lo + a_lo b_hi                 (the dropped a_lo b_lo is 2^-18 relative)
// = three v_mfma_f32_32x32x16_bf16 into ONE fp32 accumulator.  Everything else is the scheme of mlp.hip's training kernels
// (transposed layers, accumulator tile = next layer's B operand, weights streamed through the shared LDS ring by LDS-DMA,
// fragment-block stores, ReLU sign-bit words, split-K dW over fragment blocks with the sample axis as the MFMA K axis) with
// every stream doubled: (hi, lo) weight fragment pairs, hi and lo fragment blocks in the stores.
//
// Forward / chain: one wave = 32 samples, ONE wave per SIMD (4 waves per workgroup): input and output layer, hi and lo, are
// 256 activation registers.  dW: 16 waves per workgroup as in mlp.hip, LDS stages of 16 samples (half a tile) so that four
// stages of hi + lo operands fit the CU's 160 KiB.
#include "mlp_s16_dev.h"
#include "mlp_s16.h"

// 1: the training forward re-reads its own stored encodings before pos5 / dir0 (48 registers less to carry through pos1..4; a
// compiler-counted global load, i.e. a vmcnt(0) that drains the ring's DMAs and the pending stores); 0: keeps them (the 512-register
// file of a one-wave-per-SIMD kernel has the room: they live in AGPRs)
#ifndef NERF_S16_RELOAD_PE
#define NERF_S16_RELOAD_PE 0
#endif

namespace nerf {
namespace s16 {

constexpr int F_CHUNKS = F_FRAGS / RING_CHUNK;          // 74
constexpr int B_CHUNKS = B_PADDED / RING_CHUNK;         // 69
static_assert(F_CHUNKS * RING_CHUNK == F_FRAGS && B_CHUNKS * RING_CHUNK == B_PADDED, "whole ring chunks");
static_assert(F_FRAGS == 2 * L::F_TOTAL && B_FRAGS == 2 * L::B_TOTAL, "pair streams");
static_assert(A_LO == L::A_MASK && A_MASK == 2 * L::A_MASK && A_SLOTS == A_MASK + 9, "activation slots");
static_assert(Z_LO == L::Z_SLOTS && Z_SLOTS == 2 * L::Z_SLOTS, "dZ slots");
constexpr int NW = 4;                                    // waves per workgroup of the forward / chain kernels

typedef RingW<F_CHUNKS, F_FRAGS, 4, NW, RING_CHUNK, RING_STAGES, true> FwdRing;      // DMA runs of four (mlp_ring.h)
typedef RingW<B_CHUNKS, B_FRAGS, 4, NW, RING_CHUNK, RING_STAGES, true> BwdRing;


// ------------------------------------------------------------------------------------------
// packing: fp32 master parameters -> (hi, lo) bf16 fragment pairs in the 32x32x16 stream orders of mlp_frag.h
// ------------------------------------------------------------------------------------------
constexpr int PACK_PAIRS = L::F_TOTAL + B_PADDED / 2;    // 1184 forward + 1104 transposed (4 of them zero padding)
__global__ void __launch_bounds__(256) pack_s16_kernel(const float* __restrict__ p, bf16x8* __restrict__ wf,
                                                       bf16x8* __restrict__ wb) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= PACK_PAIRS * 64) return;
  const int fp = t >> 6, lane = t & 63, r = lane & 31, h = lane >> 5;
  const bool fw = fp < L::F_TOTAL;
  const int f = fw ? fp : fp - L::F_TOTAL;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = fw ? fwd_src(p, f, r, h, j) : (f < L::B_TOTAL ? bwd_src(p, f, r, h, j) : 0.0f);
  bf16x8 hi[1], lo[1];
  split_slots<8>(v, hi, lo);
  bf16x8* dst = fw ? wf : wb;
  dst[(2 * f) * 64 + lane] = hi[0];
  dst[(2 * f + 1) * 64 + lane] = lo[0];
}

// ------------------------------------------------------------------------------------------
// positional encodings (models/embedding.py:30-71), float32-accurate, straight into (hi, lo) B fragments
// ------------------------------------------------------------------------------------------
// sin(x f + 2 pi ph), ph in {0, 1/4} (cos).  x f is the reference's float32 product; a - k 2 pi with 2 pi = HI + LO in two
// fused steps (|k| <= 82: exact products), then v_sin_f32 on revolutions in [-3/4, 3/4]  (same evaluation as mlp22.hip)
__device__ __forceinline__ float sin_acc(float x, float f, float ph) {
  const float a = x * f;
  const float k = __builtin_rintf(a * 0.15915494309189535f);
  float r = __builtin_fmaf(-k, 6.2831854820251465f, a);
  r = __builtin_fmaf(-k, -1.7484556000744487e-07f, r);
  return __builtin_amdgcn_sinf(__builtin_fmaf(r, 0.15915494309189535f, ph));
}
// channel order [x, sin(f0 x), cos(f0 x), ...]: element j of lane half h in k-step KS is channel kperm(KS, h, j)
template <int C0, int LIMIT, int NB>
__device__ __forceinline__ float pe_value_acc(const float (&x)[3], const float (&fr)[NB], int h) {
  constexpr Chan a = chan_of(C0, LIMIT), b = chan_of(C0 + 4, LIMIT);
  const float xa = x[a.dim], xb = x[b.dim];
  const float xv = h ? xb : xa;
  const float fv = h ? fr[b.band] : fr[a.band];
  const float ph = h ? (b.kind == 2 ? 0.25f : 0.0f) : (a.kind == 2 ? 0.25f : 0.0f);
  const float s = sin_acc(xv, fv, ph);
  const float va = a.kind == 0 ? xa : (a.kind == 3 ? 0.0f : s);
  const float vb = b.kind == 0 ? xb : (b.kind == 3 ? 0.0f : s);
  return h ? vb : va;
}
template <int KS, int LIMIT, int NB, int... J>
__device__ __forceinline__ void pe_frag_impl(const float (&x)[3], const float (&fr)[NB], int h, bf16x8& hi, bf16x8& lo) {
  float v[8];
  ((v[J] = pe_value_acc<16 * KS + 8 * (J >> 2) + (J & 3), LIMIT, NB>(x, fr, h)), ...);
  split_slots<8>(v, &hi, &lo);
}
template <int KS, int LIMIT, int NB>
__device__ __forceinline__ void pe_frag(const float (&x)[3], const float (&fr)[NB], int h, bf16x8& hi, bf16x8& lo) {
  pe_frag_impl<KS, LIMIT, NB, 0, 1, 2, 3, 4, 5, 6, 7>(x, fr, h, hi, lo);
}


// All 12 layers for the wave's 32 samples (sample tile `tile0`).  Waves past the end compute on clamped inputs, store into
// the padding tiles of the workspace and write no output, so every wave runs the same instruction stream (the ring needs it).
// MODE 0: embedded rows;  MODE 1: rays + z with fused positional encodings
template <int MODE, class WS>
__device__ __forceinline__ void fwd_tiles(const FwdArgs& a, WS& ws, int64_t tile0, int64_t ntiles, int lane) {
  const int r = lane & 31, h = lane >> 5;
  bf16x8 peh[4], pel[4], dph[2], dpl[2];
  {
    const int64_t tile = tile0 < ntiles ? tile0 : ntiles - 1;
    int64_t m = tile * 32 + r; if (m >= a.M) m = a.M - 1;
    if (MODE == 0) {
      const float* row = a.x + m * 90;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) row_frag(row, ks, h, 63, peh[ks], pel[ks]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) row_frag(row + 63, ks, h, 27, dph[ks], dpl[ks]);
    } else {
      const int64_t ray = (int64_t)((unsigned)m / (unsigned)a.n);     // M < 2^31 (checked on the host): 32-bit divide
      const float* rr = a.rays + ray * NERF_RAY_STRIDE;
      const float zv = a.z[m];
      float p[3], d[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { p[c] = rr[c] + zv * rr[3 + c]; d[c] = rr[8 + c]; }   // render.py:142
      pe_frag<0, 63, 10>(p, a.fr.pos, h, peh[0], pel[0]); pe_frag<1, 63, 10>(p, a.fr.pos, h, peh[1], pel[1]);
      pe_frag<2, 63, 10>(p, a.fr.pos, h, peh[2], pel[2]); pe_frag<3, 63, 10>(p, a.fr.pos, h, peh[3], pel[3]);
      pe_frag<0, 27, 4>(d, a.fr.dir, h, dph[0], dpl[0]); pe_frag<1, 27, 4>(d, a.fr.dir, h, dph[1], dpl[1]);
    }
  }
  store_frags<4>(a.acts, tile0, a.astride, L::A_PE, peh, r, h); store_frags<4>(a.acts, tile0, a.astride, A_LO + L::A_PE, pel, r, h);
  store_frags<2>(a.acts, tile0, a.astride, L::A_DPE, dph, r, h); store_frags<2>(a.acts, tile0, a.astride, A_LO + L::A_DPE, dpl, r, h);

  bf16x8 hah[16], hal[16], hbh[16], hbl[16];
  u32x4 mk;
#define SINK(slot0) PairSink{a.acts, tile0, a.astride, slot0, A_LO, r, h}
#define MASK_BEGIN() mk = u32x4{0u, 0u, 0u, 0u}
#define MASK_STORE(layer) *reinterpret_cast<u32x4*>(frag_ptr(a.acts, tile0, a.astride, A_MASK + (layer), r, h)) = mk
  MASK_BEGIN();
  layer_fwd<4, 8, true, true>(ws, L::F_L0, 0, peh, pel, hah, hal, mk, lane, SINK(L::A_H0));
  MASK_STORE(0);
  MASK_BEGIN();
  layer_fwd<16, 8, true, true>(ws, L::F_L1 + 0 * 128, 256, hah, hal, hbh, hbl, mk, lane, SINK(L::A_H0 + 16));
  MASK_STORE(1);
  MASK_BEGIN();
  layer_fwd<16, 8, true, true>(ws, L::F_L1 + 1 * 128, 512, hbh, hbl, hah, hal, mk, lane, SINK(L::A_H0 + 32));
  MASK_STORE(2);
  MASK_BEGIN();
  layer_fwd<16, 8, true, true>(ws, L::F_L1 + 2 * 128, 768, hah, hal, hbh, hbl, mk, lane, SINK(L::A_H0 + 48));
  MASK_STORE(3);
  MASK_BEGIN();
  layer_fwd<16, 8, true, true>(ws, L::F_L1 + 3 * 128, 1024, hbh, hbl, hah, hal, mk, lane, SINK(L::A_H0 + 64));
  MASK_STORE(4);
  {                                                           // pos5 on concat[input_pos, h]  (models/NeRF.py:224-225)
    bf16x8 cth[20], ctl[20];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#if NERF_S16_RELOAD_PE                                        // register relief: the encoding was just stored
      cth[k] = *frag_ptr(a.acts, tile0, a.astride, L::A_PE + k, r, h);
      ctl[k] = *frag_ptr(a.acts, tile0, a.astride, A_LO + L::A_PE + k, r, h);
#else
      cth[k] = peh[k]; ctl[k] = pel[k];
#endif
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) { cth[4 + k] = hah[k]; ctl[4 + k] = hal[k]; }
    MASK_BEGIN();
    layer_fwd<20, 8, true, true>(ws, L::F_L5, 1280, cth, ctl, hbh, hbl, mk, lane, SINK(L::A_H0 + 80));
    MASK_STORE(5);
  }
  MASK_BEGIN();
  layer_fwd<16, 8, true, true>(ws, L::F_L6, 1536, hbh, hbl, hah, hal, mk, lane, SINK(L::A_H0 + 96));
  MASK_STORE(6);
  MASK_BEGIN();
  layer_fwd<16, 8, true, true>(ws, L::F_L7, 1792, hah, hal, hbh, hbl, mk, lane, SINK(L::A_H0 + 112));
  MASK_STORE(7);
  // feature (no activation) and alpha (row 0 of a ninth tile)   models/NeRF.py:229-231
  layer_fwd<16, 8, false, false>(ws, L::F_FA, L::BI_FEAT, hbh, hbl, hah, hal, mk, lane, SINK(L::A_FEAT));
  const float alpha = head<16>(ws, L::F_FA + 128, L::BI_ALPHA, hbh, hbl, lane)[0];
  // view branch: relu(Linear([feature, input_dir]))  then rgb   models/NeRF.py:232-238
  bf16x8 hdh[8], hdl[8];
  {
    bf16x8 cth[18], ctl[18];
#pragma unroll
    for (int k = 0; k < 16; ++k) { cth[k] = hah[k]; ctl[k] = hal[k]; }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#if NERF_S16_RELOAD_PE
      cth[16 + k] = *frag_ptr(a.acts, tile0, a.astride, L::A_DPE + k, r, h);
      ctl[16 + k] = *frag_ptr(a.acts, tile0, a.astride, A_LO + L::A_DPE + k, r, h);
#else
      cth[16 + k] = dph[k]; ctl[16 + k] = dpl[k];
#endif
    }
    MASK_BEGIN();
    layer_fwd<18, 4, true, true>(ws, L::F_DIR, L::BI_DIR, cth, ctl, hdh, hdl, mk, lane, SINK(L::A_HD));
    MASK_STORE(8);
  }
  const f32x16 rgb = head<8>(ws, L::F_RGB, L::BI_RGB, hdh, hdl, lane);
  const int64_t m = tile0 * 32 + r;
  if (h == 0 && tile0 < ntiles && m < a.M) {
    float4 o; o.x = rgb[0]; o.y = rgb[1]; o.z = rgb[2]; o.w = alpha;     // [rgb, alpha] raw (:239)
    *reinterpret_cast<float4*>(a.out + m * 4) = o;
  }
#undef SINK
#undef MASK_BEGIN
#undef MASK_STORE
}

template <int MODE>
__global__ void __launch_bounds__(64 * NW) s16_fwd_kernel(FwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ntiles = (a.M + 31) >> 5, nsuper = (ntiles + NW - 1) / NW;
  FwdRing ws;
  ws.wsrc = reinterpret_cast<const char*>(a.wf);
  ws.lane16 = 16 * lane;
  ws.lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(ring_smem));
  ws.wv = wv;
  ws.start(lane);
  ring_load_bias(a.bias, L::BI_TOTAL, FwdRing::BIAS_OFF);
  __syncthreads();
  for (int64_t sp = blockIdx.x; sp < nsuper; sp += gridDim.x) {
    int ln = lane;
    asm volatile("" : "+v"(ln));          // lane-derived values are recomputed per pass, not hoisted and spilled
    ws.new_pass();
    fwd_tiles<MODE>(a, ws, sp * NW + wv, ntiles, ln);
  }
  ws.drain();                             // the ring always runs 3 chunks ahead
}


struct BwdArgs {
  const bf16x8* wb;
  const void* acts;
  const float* d_raw;    // [M,4]
  int64_t M;
  void* dz;
  int64_t astride, zstride;
};

template <class WS>
__device__ __forceinline__ void bwd_tiles(const BwdArgs& a, WS& ws, int64_t tile0, int64_t ntiles, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const bool live = tile0 < ntiles;
  const int64_t tile = live ? tile0 : ntiles - 1;
  bf16x8 zrh[1], zrl[1], zah[1], zal[1];
  {
    const int64_t m = tile * 32 + r;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && m < a.M && h == 0) g = *reinterpret_cast<const float4*>(a.d_raw + m * 4);
    float vr[8] = {g.x, g.y, g.z, 0.f, 0.f, 0.f, 0.f, 0.f}, va[8] = {g.w, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    split_slots<8>(vr, zrh, zrl);                         // rows 0..2 (h == 0)
    split_slots<8>(va, zah, zal);                         // row 0
  }
  // every ReLU mask of the pass is fetched here, so the chain itself issues no loads the compiler must wait for
  u32x4 mk[9];
#pragma unroll
  for (int l = 0; l < 9; ++l)
    mk[l] = *reinterpret_cast<const u32x4*>(frag_ptr(const_cast<void*>(a.acts), tile, a.astride, A_MASK + l, r, h));
#define ZSINK(slot0) PairSink{a.dz, tile0, a.zstride, slot0, Z_LO, r, h}
  ZSINK(L::Z_RGB).put(0, zrh[0], zrl[0]);
  ZSINK(L::Z_A).put(0, zah[0], zal[0]);
  bf16x8 zdh[8], zdl[8];
  layer_bwd<1, 4, true>(ws, L::B_RGB, zrh, zrl, zdh, zdl, mk[8], lane, ZSINK(L::Z_D));
  bf16x8 zxh[16], zxl[16], zyh[16], zyl[16];
  layer_bwd<8, 8, false>(ws, L::B_DIR, zdh, zdl, zxh, zxl, mk[8], lane, ZSINK(L::Z_F));          // d feature
  {
    bf16x8 cth[17], ctl[17];
#pragma unroll
    for (int k = 0; k < 16; ++k) { cth[k] = zxh[k]; ctl[k] = zxl[k]; }
    cth[16] = zah[0]; ctl[16] = zal[0];
    layer_bwd<17, 8, true>(ws, L::B_FA, cth, ctl, zyh, zyl, mk[7], lane, ZSINK(L::Z_L0 + 112));   // dZ7
  }
  layer_bwd<16, 8, true>(ws, L::B_L7, zyh, zyl, zxh, zxl, mk[6], lane, ZSINK(L::Z_L0 + 96));     // dZ6
  layer_bwd<16, 8, true>(ws, L::B_L6, zxh, zxl, zyh, zyl, mk[5], lane, ZSINK(L::Z_L0 + 80));     // dZ5
  layer_bwd<16, 8, true>(ws, L::B_L5, zyh, zyl, zxh, zxl, mk[4], lane, ZSINK(L::Z_L0 + 64));     // dZ4
  layer_bwd<16, 8, true>(ws, L::B_L4 + 0 * 128, zxh, zxl, zyh, zyl, mk[3], lane, ZSINK(L::Z_L0 + 48));   // dZ3
  layer_bwd<16, 8, true>(ws, L::B_L4 + 1 * 128, zyh, zyl, zxh, zxl, mk[2], lane, ZSINK(L::Z_L0 + 32));   // dZ2
  layer_bwd<16, 8, true>(ws, L::B_L4 + 2 * 128, zxh, zxl, zyh, zyl, mk[1], lane, ZSINK(L::Z_L0 + 16));   // dZ1
  layer_bwd<16, 8, true>(ws, L::B_L4 + 3 * 128, zyh, zyl, zxh, zxl, mk[0], lane, ZSINK(L::Z_L0 + 0));    // dZ0
#undef ZSINK
}

__global__ void __launch_bounds__(64 * NW) s16_bwd_kernel(BwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ntiles = (a.M + 31) >> 5, nsuper = (ntiles + NW - 1) / NW;
  BwdRing ws;
  ws.wsrc = reinterpret_cast<const char*>(a.wb);
  ws.lane16 = 16 * lane;
  ws.lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(ring_smem));
  ws.wv = wv;
  ws.start(lane);
  for (int64_t sp = blockIdx.x; sp < nsuper; sp += gridDim.x) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    ws.new_pass();
    bwd_tiles(a, ws, sp * NW + wv, ntiles, ln);
    // the pass ends inside the last chunk (2200 is not a multiple of 32): the next pass starts at a chunk boundary again
    // because fragment indices restart at 0
  }
  ws.drain();
}

// ------------------------------------------------------------------------------------------
// dW: split-K GEMMs  dW[n][k] = sum_m dZ[n][m] H[k][m]  over hi / lo fragment blocks, samples = MFMA K
// ------------------------------------------------------------------------------------------
// LDS stage = 16 samples (half a sample tile, one MFMA k-step) of every operand fragment of the job, as "pair blocks" of
// 1 KiB: 16 sample rows x [even fragment 32 B | odd fragment 32 B] -- the two 16-feature fragments of one 32-wide operand
// tile side by side, so that a ds_read_b64_tr_b16 (4 sample rows x 2 fragments x 32 B per 32 lanes) covers 256 contiguous
// bytes: every bank once, no padding.  One LDS-DMA fills one pair block; the per-lane SOURCE address does the shuffle
// (lane i: row i >> 2, fragment (i >> 1) & 1, 16-byte half i & 1).  Blocks of a stage: dZ hi [n_tiles] | dZ lo [n_tiles] |
// H hi [k_tiles] | H lo [k_tiles]  <= 32 KiB; four stages = three half tiles (<= 96 KiB) in flight behind the one in use.
constexpr int DW_STAGE_BYTES = 32 * 1024, DW_STAGES = 4;
constexpr int DW_LDS_BYTES = DW_STAGES * DW_STAGE_BYTES + 1024;   // + 1 KiB sink for padding DMAs
constexpr int DW_WAVES = 16, DW_NPW = 2;                          // 2 x 2 output tiles per wave; 2 DMAs per wave per stage

__device__ __forceinline__ bf16x8 tr_pair(const char* blk, int hq, int fsel, int i16) {
  // A/B operand of v_mfma_f32_32x32x16_bf16 with K = samples 8 hq + (0..7) of the stage and row/col = feature (natural
  // order).  Lane i16 = 4 q + p of a 16-lane group addresses sample row q (and q + 4), feature piece p of fragment fsel.
  const int q = i16 >> 2, p = i16 & 3;
  const char* base = blk + 64 * (8 * hq + q) + 32 * fsel + 16 * (p & 1) + 8 * (p >> 1);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 256));
  union { struct { s16x4 a, b; } s; bf16x8 v; } cvt;
  cvt.s.a = lo; cvt.s.b = hi;
  return cvt.v;
}

// PAIR ("dw22_variant" 2, round 5): one barrier per TWO stages (a whole 32-sample tile): the pair is waited for, the barrier
// frees the previous pair's slots, the next pair is issued into them (64 KiB in flight while 64 KiB are processed), then both
// half tiles are processed back to back -- the fixed cost of a stage boundary (barrier skew, the refill DMAs' blocked issue, the
// LDS latency in front of the first MFMA) is paid once per 24 MFMAs of a wave instead of once per 12.  Measured (tools/probe_dw22.py,
// 786 k samples): 3.42-3.46 ms against 3.72-3.78 (-8 %), gradients bit-identical; the default since round 5.  (The kernel arguments
// the loop needs are copied into SGPRs in front of it: with the stage body a lambda, hipcc re-loaded them inside the loop, and a
// scalar load there makes it wait with lgkmcnt(0) in front of every MFMA group: 4.07 ms.)
template <bool PAIR>
__global__ void __launch_bounds__(64 * DW_WAVES) s16_dw_kernel(DwArgs a) {
  char* smem = ring_smem;
  int bj = blockIdx.x, job_id = 0;
  while (bj >= a.splits[job_id]) { bj -= a.splits[job_id]; ++job_id; }
  const DwJob jb = a.jobs[job_id];
  const int tile_lo = (int)((int64_t)a.ntiles * bj / a.splits[job_id]);
  const int tile_hi = (int)((int64_t)a.ntiles * (bj + 1) / a.splits[job_id]);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_tiles = (jb.nf + 1) >> 1, k_tiles = (jb.kf + 1) >> 1;
  // wave (wr, wc) of the 4 x 4 wave grid; jobs whose work sits in wave column 0 only (k_tiles <= 2) number their waves column-major so
  // that the active ones land on four different SIMDs (a wave runs on SIMD wv % 4; see mlp.hip:mlp_dw_kernel)
  const bool col_major = k_tiles <= 2;
  const int wr = col_major ? (wv & 3) : (wv >> 2), wc = col_major ? (wv >> 2) : (wv & 3);
  const int npairs = 2 * (n_tiles + k_tiles);             // real pair blocks per stage (<= 32)
  const bool active = (wr * DW_NPW < n_tiles) && (wc * 2 < k_tiles);
  f32x16 acc[DW_NPW][2];
#pragma unroll
  for (int i = 0; i < DW_NPW; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][k][e] = 0.0f;
  float bsum[DW_NPW];
#pragma unroll
  for (int i = 0; i < DW_NPW; ++i) bsum[i] = 0.0f;
  const int g16 = lane >> 4, i16 = lane & 15, hq = g16 >> 1, fsel = g16 & 1;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  const unsigned sink = lds0 + DW_STAGES * DW_STAGE_BYTES;
  // per-lane part of a pair block's source address: sample row, fragment of the pair, 16-byte half
  const int src_row = lane >> 2, src_sel = (lane >> 1) & 1, src_half = lane & 1;
  const char* dzb = reinterpret_cast<const char*>(a.dz);
  const char* acb = reinterpret_cast<const char*>(a.acts);
  // every argument the loop needs, in SGPRs before it: a scalar load inside the loop shares lgkmcnt with the transposed reads and
  // returns out of order, which makes hipcc wait with lgkmcnt(0) in front of every MFMA group (round 5)
  long long zstride_b = a.zstride * 16, astride_b = a.astride * 16;
  int z_lo = a.z_lo, a_lo = a.a_lo, dz_slot = jb.dz_slot, act_slot = jb.act_slot, nf = jb.nf, kf = jb.kf;
  asm volatile("" : "+s"(zstride_b), "+s"(astride_b), "+s"(z_lo), "+s"(a_lo), "+s"(dz_slot), "+s"(act_slot), "+s"(nf), "+s"(kf));

  // every wave issues exactly DW_NPW DMAs per stage so that vmcnt arithmetic is uniform: pair blocks wv + DW_WAVES k
  auto issue = [&](int ht, int stage) {
    const unsigned st = lds0 + __builtin_amdgcn_readfirstlane(stage) * DW_STAGE_BYTES;
#if NERF_ABLATE == 32         // timing-only build 32: every stage re-reads the job's first sample tile (L2-resident): the compute path without HBM
    const int64_t tile = tile_lo;
#else
    const int64_t tile = ht >> 1;
#endif
    const unsigned row_off = (unsigned)(32 * (16 * (ht & 1) + src_row) + 16 * src_half);
#pragma unroll
    for (int k = 0; k < DW_NPW; ++k) {
      const int i = wv + DW_WAVES * k;                     // wave-uniform
      if (i < npairs) {
        const bool is_z = i < 2 * n_tiles;
        const int j = is_z ? i : i - 2 * n_tiles;          // index inside the dZ / H half of the stage
        const int nt_ = is_z ? n_tiles : k_tiles;
        const int lo_part = j >= nt_ ? 1 : 0;
        const int t = lo_part ? j - nt_ : j;
        const int nfr = is_z ? nf : kf;
        int fr = 2 * t + src_sel; if (fr >= nfr) fr = nfr - 1;        // odd counts (rgb / alpha: nf = 1): rows past n_valid, never read back
        const int slot = (is_z ? dz_slot + lo_part * z_lo : act_slot + lo_part * a_lo) + fr;
#if NERF_ABLATE == 9          // timing-only build 9: every DMA reads ONE contiguous KiB (wrong operands): what the two 512-byte halves cost
        const char* src = (is_z ? dzb + tile * zstride_b : acb + tile * astride_b) + (int64_t)(slot - src_sel + (ht & 1)) * 1024 + 16 * lane;
        (void)row_off;
#else
        const char* src = (is_z ? dzb + tile * zstride_b : acb + tile * astride_b) + (int64_t)slot * 1024 + row_off;
#endif
        dma_frag_nt(src, st + i * 1024);
      } else {
        dma_frag(dzb + tile * zstride_b + (int64_t)dz_slot * 1024 + 16 * lane, sink);   // padding: L2 hit, result unused
      }
    }
  };
  const int ht_lo = 2 * tile_lo, ht_hi = 2 * tile_hi;
  auto process = [&](const char* st) {
#if NERF_ABLATE == 31         // timing-only build 31: the load skeleton alone (no transposed reads, no MFMAs)
    (void)st;
    if (false) {
#else
    if (active) {
#endif
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int kt = wc * 2 + k;
        const int b = 2 * n_tiles + (kt < k_tiles ? kt : 0);
        bh[k] = tr_pair(st + b * 1024, hq, fsel, i16);
        bl[k] = tr_pair(st + (b + k_tiles) * 1024, hq, fsel, i16);
      }
#pragma unroll
      for (int i = 0; i < DW_NPW; ++i) {
        const int nt = wr * DW_NPW + i;
        const int b = nt < n_tiles ? nt : 0;
        const bf16x8 ah = tr_pair(st + b * 1024, hq, fsel, i16);
        const bf16x8 al = tr_pair(st + (b + n_tiles) * 1024, hq, fsel, i16);
        if (wc == 0) {                         // bias gradient = row sums of dZ = hi + lo: v_dot2c_f32_bf16 against (1, 1)
          const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const bf16x2 ph = {ah[j], ah[j + 1]}, pl = {al[j], al[j + 1]};
            bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(pl, ones, bsum[i], false);
            bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(ph, ones, bsum[i], false);
          }
        }
#if NERF_ABLATE == 30         // timing-only build 30: transposed reads kept (and waited for), no MFMAs
        asm volatile("" :: "v"(ah), "v"(al), "v"(bh[0]), "v"(bl[0]), "v"(bh[1]), "v"(bl[1]));
#else
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          acc[i][k] = mfma32(al, bh[k], acc[i][k]);
          acc[i][k] = mfma32(ah, bl[k], acc[i][k]);
          acc[i][k] = mfma32(ah, bh[k], acc[i][k]);
        }
#endif
      }
    }
  };
  if (PAIR) {
    const int n = ht_hi - ht_lo;                          // even: two half tiles per sample tile
    if (n > 0) { issue(ht_lo, 0); issue(ht_lo + 1, 1); }
    for (int p = 0; p < n; p += 2) {
      // the pair has landed (nothing younger is in flight); lgkmcnt(0): this wave's reads of the previous pair are complete
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // pair p landed for every wave; the slots of pair p - 2 are free
      if (p + 2 < n) { issue(ht_lo + p + 2, (p + 2) % DW_STAGES); issue(ht_lo + p + 3, (p + 3) % DW_STAGES); }   // (one in front of each half: measured slower)
      process(smem + (p % DW_STAGES) * DW_STAGE_BYTES);
      process(smem + ((p + 1) % DW_STAGES) * DW_STAGE_BYTES);
    }
  } else {
#pragma unroll
    for (int s_ = 0; s_ < DW_STAGES - 1; ++s_)
      if (ht_lo + s_ < ht_hi) issue(ht_lo + s_, s_);
    for (int ht = ht_lo; ht < ht_hi; ++ht) {
      const int rem = ht_hi - 1 - ht;                       // stages issued after this one and still in flight (<= 2)
      // lgkmcnt(0): this wave's transposed reads of the previous stage are complete before the barrier lets another wave's
      // DMA refill it
      if (rem >= 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else if (rem == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                         // stage landed for every wave; stage (ht - 1) % 4 is free
      if (ht + DW_STAGES - 1 < ht_hi) issue(ht + DW_STAGES - 1, (ht - ht_lo + DW_STAGES - 1) % DW_STAGES);
      process(smem + ((ht - ht_lo) % DW_STAGES) * DW_STAGE_BYTES);
    }
  }
  if (!active) return;
  const int rr = lane & 31, hh = lane >> 5;
  float* slot = a.partial + (size_t)blockIdx.x * DW_SLOT_FLOATS;
#pragma unroll
  for (int i = 0; i < DW_NPW; ++i) {
    const int nt = wr * DW_NPW + i;
    if (nt >= n_tiles) continue;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int kt = wc * 2 + k;
      if (kt >= k_tiles) continue;
      float* tile = slot + (8 * nt + kt) * 1024;
#pragma unroll
      for (int e = 0; e < 16; ++e) tile[((e & 3) + 8 * (e >> 2) + 4 * hh) * 32 + rr] = acc[i][k][e];      // 128 B per half wave
    }
    if (wc == 0 && jb.b_off >= 0) {
      const float tot = bsum[i] + __shfl_xor(bsum[i], 32, 64);      // the two sample halves of the k-step
      if (hh == 0) slot[64 * 1024 + 32 * nt + rr] = tot;
    }
  }
}

// ------------------------------------------------------------------------------------------
// dW, 8-wave form with two operand register sets (round 5; "dw22_variant" 1; the default is the PAIR form of the 16-wave kernel above).
// ------------------------------------------------------------------------------------------
// What bounds s16_dw_kernel above is NOT HBM (round 4 said so): measured with tools/probe_dw22.py on timing-only builds and
// tools/dw22_probe.hip --
//   * its load skeleton alone (no transposed reads, no MFMAs) moves the 17.5 GB of a 786 k-sample launch in 2.61 ms = 6.7 TB/s;
//   * with the operands L2-resident (every stage re-reads one sample tile: no HBM traffic at all) the kernel still takes 3.60 of
//     its 3.78 ms; reads without MFMAs 2.71 ms; register-only MFMAs beside the load skeleton: no slowdown (6.6 TB/s);
// i.e. the per-stage chain  barrier -> transposed reads -> wait -> MFMAs  of single-buffered operands, run by waves that the stage
// barrier keeps in lock-step: 1.7 us per stage against 0.7 us of MFMA time per SIMD and 1.25 us of load time.
// This form: 8 waves (two per SIMD, 256 registers each) instead of 16; wave (wr, wc) = 2 n-tiles x 4 k-tiles = 8 accumulator
// tiles, 24 MFMAs per stage against 12 operand tiles read (the 16-wave form: 12 against 8: 25 % fewer LDS reads per MFMA); TWO
// operand register sets: the transposed reads of stage j + 1 are issued between the MFMAs of stage j, in halves of 12 (lgkmcnt is
// a 4-bit counter); the ring needs stage j + 1 landed one barrier earlier (still three half tiles in flight); the bias row sums
// split between the two waves that share a dZ tile pair; every kernel argument the loop needs sits in SGPRs before it (a scalar
// load inside the loop shares lgkmcnt with the reads and returns out of order: hipcc then waits with lgkmcnt(0) everywhere).
// Measured: 3.72 against 3.78 ms (-1.5 %; the PAIR form of the 16-wave kernel: 3.42-3.46, -8 %), bit-identical gradients -- hipcc still orders part of the next stage's reads in front
// of the current stage's first MFMAs with waits that count them (lgkmcnt(10), (5), (1), (0) in the ISA), so most of the intended
// overlap is not realised.  Two further forms were measured and not kept: the reads as inline asm with our own wait (the twelve
// 128-bit operands assembled from 64-bit halves cost 48 more registers: spills), and the operands fetched into registers by plain
// global loads + ds_write_b128 instead of LDS-DMA (-2.5 %; at 127 of 128 registers hipcc re-used in-flight load destinations --
// tools/check_inflight_regs.py: 90 hits, wrong gradients).  The floor this kernel is still 40 % above is its 2.6 ms of loads; what
// reaches it is a hand-scheduled inner loop (operand reads of stage j + 1 strictly under the MFMAs of stage j), not another C++
// arrangement of the same statements.
// Same stages, pair blocks, job table, partial-tile slots and reduce kernel as above; gradients bit-identical to the 16-wave form.
constexpr int DW2_WAVES = 8, DW2_NN = 2, DW2_NK = 4, DW2_DPW = 4;          // 4 DMAs per wave per stage

struct DwOps { bf16x8 ah[DW2_NN], al[DW2_NN], bh[DW2_NK], bl[DW2_NK]; };

__global__ void __launch_bounds__(64 * DW2_WAVES) s16_dw2_kernel(DwArgs a) {
  char* smem = ring_smem;
  int bj = blockIdx.x, job_id = 0;
  while (bj >= a.splits[job_id]) { bj -= a.splits[job_id]; ++job_id; }
  const DwJob jb = a.jobs[job_id];
  const int tile_lo = (int)((int64_t)a.ntiles * bj / a.splits[job_id]);
  const int tile_hi = (int)((int64_t)a.ntiles * (bj + 1) / a.splits[job_id]);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wv >> 1, wc = wv & 1;
  const int n_tiles = (jb.nf + 1) >> 1, k_tiles = (jb.kf + 1) >> 1;
  const int npairs = 2 * (n_tiles + k_tiles);             // real pair blocks per stage (<= 32)
  const bool active = (wr * DW2_NN < n_tiles) && (wc * DW2_NK < k_tiles);
  const int kn = k_tiles - wc * DW2_NK;                    // k-tiles of this wave that exist (wave-uniform; <= 0: inactive)
  f32x16 acc[DW2_NN][DW2_NK];
#pragma unroll
  for (int i = 0; i < DW2_NN; ++i)
#pragma unroll
    for (int k = 0; k < DW2_NK; ++k)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][k][e] = 0.0f;
  // bias gradient: the rows of dZ tile 2 wr + i are summed by wave (wr, wc = i) -- or, when the job has one k-column of waves only
  // (k_tiles <= 4: wc == 1 is inactive), both by wave (wr, 0)
  float bsum[DW2_NN] = {0.0f, 0.0f};
  const bool one_col = k_tiles <= DW2_NK;
  const bool sums0 = wc == 0, sums1 = wc == 1 || one_col;  // wave-uniform
  const int g16 = lane >> 4, i16 = lane & 15, hq = g16 >> 1, fsel = g16 & 1;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  const unsigned sink = lds0 + DW_STAGES * DW_STAGE_BYTES;
  const int src_row = lane >> 2, src_sel = (lane >> 1) & 1, src_half = lane & 1;
  const char* dzb = reinterpret_cast<const char*>(a.dz);
  const char* acb = reinterpret_cast<const char*>(a.acts);
  // Everything the loop needs of the argument block is read into SGPRs HERE: a scalar load inside the loop shares lgkmcnt with the
  // transposed reads and returns out of order, which makes hipcc wait with lgkmcnt(0) in front of every MFMA group (no overlap)
  long long zstride_b = a.zstride * 16, astride_b = a.astride * 16;
  int z_lo = a.z_lo, a_lo = a.a_lo, dz_slot = jb.dz_slot, act_slot = jb.act_slot, nf = jb.nf, kf = jb.kf;
  asm volatile("" : "+s"(zstride_b), "+s"(astride_b), "+s"(z_lo), "+s"(a_lo), "+s"(dz_slot), "+s"(act_slot), "+s"(nf), "+s"(kf));

  // every wave issues exactly DW2_DPW DMAs per stage so that vmcnt arithmetic is uniform: pair blocks wv + DW2_WAVES k
  auto issue = [&](int ht, int stage) {
    const unsigned st = lds0 + __builtin_amdgcn_readfirstlane(stage) * DW_STAGE_BYTES;
    const int64_t tile = ht >> 1;
    const unsigned row_off = (unsigned)(32 * (16 * (ht & 1) + src_row) + 16 * src_half);
#pragma unroll
    for (int k = 0; k < DW2_DPW; ++k) {
      const int i = wv + DW2_WAVES * k;                    // wave-uniform
      if (i < npairs) {
        const bool is_z = i < 2 * n_tiles;
        const int j = is_z ? i : i - 2 * n_tiles;
        const int nt_ = is_z ? n_tiles : k_tiles;
        const int lo_part = j >= nt_ ? 1 : 0;
        const int t = lo_part ? j - nt_ : j;
        const int nfr = is_z ? nf : kf;
        int fr = 2 * t + src_sel; if (fr >= nfr) fr = nfr - 1;
        const int slot = (is_z ? dz_slot + lo_part * z_lo : act_slot + lo_part * a_lo) + fr;
        const char* src = (is_z ? dzb + tile * zstride_b : acb + tile * astride_b) + (int64_t)slot * 1024 + row_off;
        dma_frag_nt(src, st + i * 1024);
      } else {
        dma_frag(dzb + tile * zstride_b + (int64_t)dz_slot * 1024 + 16 * lane, sink);   // padding: L2 hit, result unused
      }
    }
  };
  // Operand tiles of one stage -> registers (transposed reads; tiles past the job's edge read a valid block, their products are
  // never stored), in TWO halves of 12 reads: lgkmcnt is a 4-bit counter, so hipcc can express "the older set has arrived, the
  // 12 reads just issued may stay in flight" (lgkmcnt(12)) but not the same with 24 behind it -- issued in one piece, the next
  // stage's reads were followed by lgkmcnt(0) and nothing overlapped (measured: 4.10 ms against the 16-wave kernel's 3.78).
  auto load_z = [&](const char* st, DwOps& o) {            // dZ tiles + the first k-tile: 12 reads
#pragma unroll
    for (int i = 0; i < DW2_NN; ++i) {
      const int nt = wr * DW2_NN + i, b = nt < n_tiles ? nt : 0;
      o.ah[i] = tr_pair(st + b * 1024, hq, fsel, i16);
      o.al[i] = tr_pair(st + (b + n_tiles) * 1024, hq, fsel, i16);
    }
    const int kt = wc * DW2_NK, b = 2 * n_tiles + (kt < k_tiles ? kt : 0);
    o.bh[0] = tr_pair(st + b * 1024, hq, fsel, i16);
    o.bl[0] = tr_pair(st + (b + k_tiles) * 1024, hq, fsel, i16);
  };
  auto load_h = [&](const char* st, DwOps& o) {            // the other three k-tiles: 12 reads
#pragma unroll
    for (int k = 1; k < DW2_NK; ++k) {
      const int kt = wc * DW2_NK + k, b = 2 * n_tiles + (kt < k_tiles ? kt : 0);
      o.bh[k] = tr_pair(st + b * 1024, hq, fsel, i16);
      o.bl[k] = tr_pair(st + (b + k_tiles) * 1024, hq, fsel, i16);
    }
  };
  auto bias_sums = [&](const DwOps& o) {
#pragma unroll
    for (int i = 0; i < DW2_NN; ++i) {                     // row sums of dZ = hi + lo: v_dot2c_f32_bf16 against (1, 1)
      if (i == 0 ? sums0 : sums1) {
        const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const bf16x2 ph = {o.ah[i][j], o.ah[i][j + 1]}, pl = {o.al[i][j], o.al[i][j + 1]};
          bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(pl, ones, bsum[i], false);
          bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(ph, ones, bsum[i], false);
        }
      }
    }
  };
  auto mfmas = [&](const DwOps& o, int k0, int k1) {
#pragma unroll
    for (int k = 0; k < DW2_NK; ++k) {
      if (k >= k0 && k < k1 && k < kn) {                   // wave-uniform
#pragma unroll
        for (int i = 0; i < DW2_NN; ++i) {
          acc[i][k] = mfma32(o.al[i], o.bh[k], acc[i][k]);
          acc[i][k] = mfma32(o.ah[i], o.bl[k], acc[i][k]);
          acc[i][k] = mfma32(o.ah[i], o.bh[k], acc[i][k]);
        }
      }
    }
  };
  const int ht_lo = 2 * tile_lo, n = 2 * (tile_hi - tile_lo);
  // wait until this wave's DMAs of stage j have landed: the stages issued behind it (at most two more exist and are in flight)
  auto wait_stage = [&](int j) {
    const int younger = n - 1 - j;                         // wave-uniform
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
#pragma unroll
  for (int s_ = 0; s_ < DW_STAGES - 1; ++s_)
    if (s_ < n) issue(ht_lo + s_, s_);
  DwOps o0, o1;
  if (n > 0) {
    wait_stage(0);
    __builtin_amdgcn_s_barrier();                          // stage 0 landed for every wave
    if (DW_STAGES - 1 < n) issue(ht_lo + DW_STAGES - 1, DW_STAGES - 1);
    if (active) { load_z(smem, o0); load_h(smem, o0); }
  }
  // step j: `use` holds (or is receiving) the operands of stage j; the reads of stage j + 1 go to `load` between the MFMAs
  auto step = [&](DwOps& use, DwOps& load, int j) {
    const bool more = j + 1 < n;
    const char* nst = smem + ((j + 1) % DW_STAGES) * DW_STAGE_BYTES;
    if (more) {
      // stages j + 2 and j + 3 may stay in flight (j + 3 was issued behind the previous barrier); lgkmcnt(0): this wave's
      // transposed reads of stage j are complete before the barrier lets another wave's DMA refill that slot
      const int younger = n - 2 - j;
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                        // stage j + 1 landed for every wave; slot of stage j is free
      if (j + DW_STAGES < n) issue(ht_lo + j + DW_STAGES, j % DW_STAGES);
      if (active) load_z(nst, load);
      __builtin_amdgcn_sched_barrier(0);                   // the reads stay ahead of the MFMAs below
    }
    if (active) { bias_sums(use); mfmas(use, 0, 2); }
    if (more && active) load_h(nst, load);
    __builtin_amdgcn_sched_barrier(0);
    if (active) mfmas(use, 2, DW2_NK);
  };
  for (int j = 0; j < n; j += 2) {
    step(o0, o1, j);
    if (j + 1 < n) step(o1, o0, j + 1);
  }
  if (!active) return;
  const int rr = lane & 31, hh = lane >> 5;
  float* slot = a.partial + (size_t)blockIdx.x * DW_SLOT_FLOATS;
#pragma unroll
  for (int i = 0; i < DW2_NN; ++i) {
    const int nt = wr * DW2_NN + i;
    if (nt >= n_tiles) continue;
#pragma unroll
    for (int k = 0; k < DW2_NK; ++k) {
      const int kt = wc * DW2_NK + k;
      if (kt >= k_tiles) continue;
      float* tile = slot + (8 * nt + kt) * 1024;
#pragma unroll
      for (int e = 0; e < 16; ++e) tile[((e & 3) + 8 * (e >> 2) + 4 * hh) * 32 + rr] = acc[i][k][e];      // 128 B per half wave
    }
  }
#pragma unroll
  for (int i = 0; i < DW2_NN; ++i) {
    const int nt = wr * DW2_NN + i;
    if (jb.b_off >= 0 && nt < n_tiles && (i == 0 ? sums0 : sums1)) {
      const float tot = bsum[i] + __shfl_xor(bsum[i], 32, 64);      // the two sample halves of the k-step
      if (hh == 0) slot[64 * 1024 + 32 * nt + rr] = tot;
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct DevOnce {
  bool done[64] = {};
  bool first() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
    if (done[d]) return false;
    done[d] = true;
    return true;
  }
};

int pack(const float* params, void* packed_s16, hipStream_t s) {
  char* base = static_cast<char*>(packed_s16);
  hipLaunchKernelGGL(pack_s16_kernel, dim3((PACK_PAIRS * 64 + 255) / 256), dim3(256), 0, s, params,
                     reinterpret_cast<bf16x8*>(base), reinterpret_cast<bf16x8*>(base + (size_t)F_FRAGS * 1024));
  return check_launch("nerf_mlp_pack (split-bf16 image)");
}

int forward(const void* packed_s16, const float* bias_slots, const float* x, const float* rays, const float* z, int64_t M,
            int n, int freq_mode, float* out, void* acts, int64_t astride16, int persistent_wgs, hipStream_t s) {
  FwdArgs a;
  a.wf = reinterpret_cast<const bf16x8*>(packed_s16);
  a.bias = bias_slots;
  a.x = x; a.rays = rays; a.z = z; a.M = M; a.n = n; a.out = out; a.acts = acts; a.astride = astride16;
  for (int k = 0; k < 10; ++k) a.fr.pos[k] = freq_mode == 0 ? (float)(k * k) : (float)(1 << k);
  for (int k = 0; k < 4; ++k) a.fr.dir[k] = freq_mode == 0 ? (float)(k * k) : (float)(1 << k);
  const int64_t nsuper = ((M + 31) / 32 + NW - 1) / NW;
  const dim3 g((unsigned)(nsuper < persistent_wgs ? nsuper : persistent_wgs)), b(64 * NW);
  // dynamic LDS above 64 KiB is an opt-in per kernel AND per device
  static DevOnce once[2];
  const int mode = x ? 0 : 1;
  if (once[mode].first()) {
    const void* k = mode == 0 ? reinterpret_cast<const void*>(s16_fwd_kernel<0>) : reinterpret_cast<const void*>(s16_fwd_kernel<1>);
    (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, FwdRing::LDS_BYTES);
  }
  if (mode == 0) hipLaunchKernelGGL(s16_fwd_kernel<0>, g, b, FwdRing::LDS_BYTES, s, a);
  else hipLaunchKernelGGL(s16_fwd_kernel<1>, g, b, FwdRing::LDS_BYTES, s, a);
  return check_launch("mlp training forward (split bf16)");
}

int backward_chain(const void* packed_s16, const void* acts, const float* d_raw, int64_t M, void* dz, int64_t astride16,
                   int64_t zstride16, int persistent_wgs, hipStream_t s) {
  BwdArgs b;
  b.wb = reinterpret_cast<const bf16x8*>(static_cast<const char*>(packed_s16) + (size_t)F_FRAGS * 1024);
  b.acts = acts; b.d_raw = d_raw; b.M = M; b.dz = dz; b.astride = astride16; b.zstride = zstride16;
  const int64_t nsuper = ((M + 31) / 32 + NW - 1) / NW;
  static DevOnce once;
  if (once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(s16_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BwdRing::LDS_BYTES);
  hipLaunchKernelGGL(s16_bwd_kernel, dim3((unsigned)(nsuper < persistent_wgs ? nsuper : persistent_wgs)), dim3(64 * NW),
                     BwdRing::LDS_BYTES, s, b);
  return check_launch("mlp backward chain (split bf16)");
}

int g_dw_variant = 2;          // nerf_set_option("dw22_variant"): 2 (default) 16 waves, one barrier per sample tile; 1 the 8-wave kernel with two operand sets; 0 round 4's form

int launch_dw_kernel(const DwArgs& d, int workgroups, hipStream_t s) {
  static DevOnce once;
  if (once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(s16_dw_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(s16_dw_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(s16_dw2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS_BYTES);
  }
  if (g_dw_variant == 0) hipLaunchKernelGGL(s16_dw_kernel<false>, dim3(workgroups), dim3(64 * DW_WAVES), DW_LDS_BYTES, s, d);
  else if (g_dw_variant == 2) hipLaunchKernelGGL(s16_dw_kernel<true>, dim3(workgroups), dim3(64 * DW_WAVES), DW_LDS_BYTES, s, d);
  else hipLaunchKernelGGL(s16_dw2_kernel, dim3(workgroups), dim3(64 * DW2_WAVES), DW_LDS_BYTES, s, d);
  return check_launch("mlp dW (split bf16)");
}

}  // namespace s16
}  // namespace nerf
