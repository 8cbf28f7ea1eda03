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
__device__ __forceinline__ void fwd_tiles(const FwdArgs& a, WS& ws, int64_t tile0, int64_t ntiles, int lane, PassQueue& pq) {
  const int r = lane & 31, h = lane >> 5;
  bf16x8 peh[4], pel[4], dph[2], dpl[2];
  pq.ask(ws.wv, lane);                    // dynamic pass queue (mlp_ring.h): before this pass's input loads
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
  pq.publish(ws.wv);
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
  PassQueue pq;                           // dynamic pass queue (mlp_ring.h); a.queue == nullptr: static split
  pq.init(a.queue, ws.lds0 + FwdRing::BIAS_OFF);
  for (int64_t sp = blockIdx.x; sp < nsuper;) {
    int ln = lane;
    asm volatile("" : "+v"(ln));          // lane-derived values are recomputed per pass, not hoisted and spilled
    ws.new_pass();
    fwd_tiles<MODE>(a, ws, sp * NW + wv, ntiles, ln, pq);
    sp = pq.next(sp);
  }
  ws.drain();                             // the ring always runs 3 chunks ahead
  pq.leave();
}


struct BwdArgs {
  unsigned* queue;       // dynamic pass queue slot of this launch, or nullptr (static split): mlp_ring.h
  const bf16x8* wb;
  const void* acts;
  const float* d_raw;    // [M,4]
  int64_t M;
  void* dz;
  int64_t astride, zstride;
};

template <class WS>
__device__ __forceinline__ void bwd_tiles(const BwdArgs& a, WS& ws, int64_t tile0, int64_t ntiles, int lane, PassQueue& pq) {
  const int r = lane & 31, h = lane >> 5;
  const bool live = tile0 < ntiles;
  pq.ask(ws.wv, lane);                    // dynamic pass queue (mlp_ring.h): before this pass's input loads
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
  pq.publish(ws.wv);
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
  PassQueue pq;                           // dynamic pass queue (mlp_ring.h); a.queue == nullptr: static split
  pq.init(a.queue, ws.lds0 + BwdRing::BIAS_OFF);
  for (int64_t sp = blockIdx.x; sp < nsuper;) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    ws.new_pass();
    bwd_tiles(a, ws, sp * NW + wv, ntiles, ln, pq);
    // the pass ends inside the last chunk (2200 is not a multiple of 32): the next pass starts at a chunk boundary again
    // because fragment indices restart at 0
    sp = pq.next(sp);
  }
  ws.drain();
  pq.leave();
}

// ------------------------------------------------------------------------------------------
// dW: split-K GEMMs  dW[n][k] = sum_m dZ[n][m] H[k][m]  over hi / lo fragment blocks, samples = MFMA K
// ------------------------------------------------------------------------------------------
// LDS stage = 16 samples (half a sample tile, one MFMA k-step) of every operand fragment of the job, as "pair blocks" of
// 1 KiB: 16 sample rows x [even fragment 32 B | odd fragment 32 B] -- the two 16-feature fragments of one 32-wide operand
// tile side by side, so that a ds_read_b64_tr_b16 (4 sample rows x 2 fragments x 32 B per 32 lanes) covers 256 contiguous
// bytes: every bank once, no padding.  One LDS-DMA fills one pair block; the per-lane SOURCE address does the shuffle
// (lane i: row i >> 2, fragment (i >> 1) & 1, 16-byte half i & 1).  Blocks of a stage: dZ hi [n_tiles] | dZ lo [n_tiles] |
// H hi [k_tiles] | H lo [k_tiles]  <= 32 KiB.
#ifndef NERF_DWX
#define NERF_DWX 0            // timing-only builds (bit mask): 1 no DMAs, 2 no transposed reads, 4 no stage barrier, 8 no bias sums, 16 no MFMAs
#endif
#if NERF_DWX & 4
#define DW_BARRIER() do {} while (0)
#else
#define DW_BARRIER() __builtin_amdgcn_s_barrier()
#endif
constexpr int DW_WAVES = 16, DW_NPW = 2;                          // 2 x 2 output tiles per wave; 2 DMAs per wave per stage

__device__ __forceinline__ bf16x8 tr_pair(const char* blk, int hq, int fsel, int i16) {
  // A/B operand of v_mfma_f32_32x32x16_bf16 with K = samples 8 hq + (0..7) of the stage and row/col = feature (natural
  // order).  Lane i16 = 4 q + p of a 16-lane group addresses sample row q (and q + 4), feature piece p of fragment fsel.
  const int q = i16 >> 2, p = i16 & 3;
  const char* base = blk + 64 * (8 * hq + q) + 32 * fsel + 16 * (p & 1) + 8 * (p >> 1);
#if NERF_DWX & 2
  bf16x8 r; asm volatile("" : "=v"(r) : "v"(base)); return r;
#endif
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 256));
  union { struct { s16x4 a, b; } s; bf16x8 v; } cvt;
  cvt.s.a = lo; cvt.s.b = hi;
  return cvt.v;
}

// ------------------------------------------------------------------------------------------
// the general dW kernel: 16 waves (4 x 4 waves of 2 x 2 output tiles), any job shape
// ------------------------------------------------------------------------------------------
// Since round 5 the 256 x 256 jobs run on mlp_dww.hip's one-wave-per-SIMD kernel; this one takes the narrow jobs (input layers,
// heads, the 2 x 64 model) and, with nerf_set_option("dw22_variant", 0), every job.
//  * Ring: a stage takes npairs KiB, the ring all 160 KiB of the CU: 5 stages for a 256 x 256 job, up to 8 for the narrow ones ("dw_ring_cap"; 6 / 8 / 16 measure the same in the training step).
//    ONE barrier per sample tile (two stages): the pair is waited for, the barrier frees the slots of the pair before it, every
//    stage whose slot is free is issued (two in steady state; ns - 2 stages stay in flight while a pair is processed), then both
//    half tiles are processed back to back -- the fixed cost of a stage boundary (barrier skew, the refill DMAs' blocked issue,
//    the LDS latency in front of the first MFMA) is paid once per 24 MFMAs of a wave instead of once per 12 (round 4's form: one
//    barrier per stage, 3.72-3.78 ms for the 786 k-sample launch; this form 3.35-3.45).  No padding DMAs: a wave waits with
//    vmcnt(its own DMAs per stage x the stages issued behind the pair).
//  * Jobs with k_tiles <= 2 number their waves column-major: the active waves then sit on four SIMDs instead of one.
//  * Every kernel argument the loop needs is copied into SGPRs in front of it: a scalar load inside the loop shares lgkmcnt with
//    the transposed reads and returns out of order, which makes hipcc wait with lgkmcnt(0) in front of every MFMA group (4.07 ms).
//  * Bias row sums (v_dot2c against (1, 1)) are spread over the waves of a row: (tile i, hi | lo part) c goes to wave column
//    c % (active columns), combined through the LDS once at the end in a fixed order.  The branches around them are kept branches
//    (empty asm): if-converted, every wave ran all sixteen v_dot2c and selected (+7 %).
// What bounds it (tools/probe_dw22_chain.py on the NERF_DWX timing-only builds, 786 k samples, all jobs): MFMAs alone 1.87 ms
// (the 256 x 256 workgroups at 0.78 us per stage, clock-limited), + barrier 0.06, + bias sums 0.3, + transposed reads 0.27 = 2.5 ms
// without a single DMA; the load skeleton alone 2.6-2.8 ms (6.3-6.7 TB/s); together 3.35-3.45: a wave of 128 registers cannot
// hold a second operand set, so reads and MFMAs of a stage are serial per wave and the four waves of a SIMD, in step behind the
// barrier, overlap them only by drifting apart.  An 8-wave form with two operand sets (round 5, removed) measured 3.54-3.72:
// two waves per SIMD in step leave the matrix pipe idle while both run their DMA / address / read instructions.  The 4-wave form
// (mlp_dww.hip) is what hides them.
constexpr int DW_LDS_BYTES = 160 * 1024;                  // the ring holds min(DW_LDS_BYTES / stage bytes, DwArgs.ring_cap) stages

// Jobs whose whole output is at most four tiles (the 2 x 64 model's layers, the view model's dir0 | dirPE and rgb jobs): with the
// 4 x 4 wave grid of 2 x 2 tiles ONE wave (or two) of sixteen has work, and the workgroup's rate is that wave's serial chain
// barrier -> reads -> wait -> MFMAs (measured: the 2 x 64 model's dW at 2.2 TB/s).  Here every wave is a pipeline of its own:
// wave w takes the stages w, w + 16, ... of the workgroup's sample range, moves them through its own 10 KiB of the LDS with its own
// DMAs and its own vmcnt wait -- no barrier in the loop, sixteen stages in flight per CU -- and accumulates its own copy of the
// NT x KT tiles; the sixteen copies are added through the LDS in a fixed binary tree at the end (bit-reproducible).
constexpr int DWP_WAVE_BYTES = 10 * 1024;
template <int NT, int KT, bool SPLIT>
__device__ __forceinline__ void dw_private(const DwArgs& a, const DwJob& jb, int tile_lo, int tile_hi, char* smem, int wv, int lane) {
  constexpr int NP = 2 * (NT + KT);                        // pair blocks per stage: dZ hi [NT] | dZ lo [NT] | H hi [KT] | H lo [KT]
  static_assert(NP * 1024 <= DWP_WAVE_BYTES && NT * KT <= 4, "wave-private dW: one stage per wave, at most four tiles");
  f32x16 acc[NT][KT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][k][e] = 0.0f;
  float bsum[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) bsum[i] = 0.0f;
  const int g16 = lane >> 4, i16 = lane & 15, hq = g16 >> 1, fsel = g16 & 1;
  char* mine = smem + wv * DWP_WAVE_BYTES;
  const unsigned lds_mine = __builtin_amdgcn_readfirstlane(lds_addr_of(mine));
  const int src_row = lane >> 2, src_sel = (lane >> 1) & 1, src_half = lane & 1;
  const char* dzb = reinterpret_cast<const char*>(a.dz);
  const char* acb = reinterpret_cast<const char*>(a.acts);
  long long zstride_b = a.zstride * 16, astride_b = a.astride * 16;
  int z_lo = a.z_lo, a_lo = a.a_lo, dz_slot = jb.dz_slot, act_slot = jb.act_slot, nf = jb.nf, kf = jb.kf;
  asm volatile("" : "+s"(zstride_b), "+s"(astride_b), "+s"(z_lo), "+s"(a_lo), "+s"(dz_slot), "+s"(act_slot), "+s"(nf), "+s"(kf));
  // stages: SPLIT half tiles (two per sample tile: 16 samples of hi + lo blocks), bf16 whole sample tiles (samples 0-15 | 16-31 in
  // the blocks the lo parts occupy in the split form)
  const int ht_hi = SPLIT ? 2 * tile_hi : tile_hi;
  for (int ht = (SPLIT ? 2 * tile_lo : tile_lo) + wv; ht < ht_hi; ht += DW_WAVES) {
    // this wave's transposed reads of its previous stage are complete (every one feeds an MFMA above; the wait makes it explicit)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !(NERF_DWX & 1)
    const int64_t tile = SPLIT ? ht >> 1 : ht;
    const unsigned row_off0 = (unsigned)(32 * ((SPLIT ? 16 * (ht & 1) : 0) + src_row) + 16 * src_half);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const bool is_z = i < 2 * NT;
      const int j = is_z ? i : i - 2 * NT, nt_ = is_z ? NT : KT;
      const int lo_part = j >= nt_ ? 1 : 0, t = lo_part ? j - nt_ : j;
      const int nfr = is_z ? nf : kf;
      int fr = 2 * t + src_sel; if (fr >= nfr) fr = nfr - 1;          // odd counts: rows past n_valid, never read back
      const int slot = (is_z ? dz_slot + (SPLIT ? lo_part * z_lo : 0) : act_slot + (SPLIT ? lo_part * a_lo : 0)) + fr;
      const unsigned row_off = row_off0 + (SPLIT ? 0 : 512 * lo_part);
      dma_frag_nt((is_z ? dzb + tile * zstride_b : acb + tile * astride_b) + (int64_t)slot * 1024 + row_off, lds_mine + i * 1024);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bf16x8 bh[KT], bl[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      bh[k] = tr_pair(mine + (2 * NT + k) * 1024, hq, fsel, i16);
      bl[k] = tr_pair(mine + (2 * NT + KT + k) * 1024, hq, fsel, i16);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const bf16x8 ah = tr_pair(mine + i * 1024, hq, fsel, i16);
      const bf16x8 al = tr_pair(mine + (NT + i) * 1024, hq, fsel, i16);
#if !(NERF_DWX & 8)
      const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const bf16x2 ph = {ah[j], ah[j + 1]}, pl = {al[j], al[j + 1]};
        bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(pl, ones, bsum[i], false);
        bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(ph, ones, bsum[i], false);
      }
#endif
#if NERF_DWX & 16
      asm volatile("" :: "v"(ah), "v"(al), "v"(bh[0]), "v"(bl[0]));
#else
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (SPLIT) {
          acc[i][k] = mfma32(al, bh[k], acc[i][k]);
          acc[i][k] = mfma32(ah, bl[k], acc[i][k]);
          acc[i][k] = mfma32(ah, bh[k], acc[i][k]);
        } else {
          acc[i][k] = mfma32(ah, bh[k], acc[i][k]);
          acc[i][k] = mfma32(al, bl[k], acc[i][k]);
        }
      }
#endif
    }
  }
  // sixteen copies -> one, through the LDS: waves [h, 2h) hand theirs to waves [0, h), h = 8, 4, 2, 1 (fixed order)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float* xl = reinterpret_cast<float*>(smem);              // [giver][tile | bias row][register][lane]: 4 KiB per tile, 256 B per bias row
  constexpr int GIVER_FLOATS = (NT * KT * 16 + NT) * 64;
#pragma unroll 1
  for (int h = DW_WAVES / 2; h >= 1; h >>= 1) {
    __builtin_amdgcn_s_barrier();                          // the loop's reads (first round) / the previous round's reads are done
    if (wv >= h && wv < 2 * h) {
      float* dst = xl + (size_t)(wv - h) * GIVER_FLOATS + lane;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
          for (int e = 0; e < 16; ++e) dst[((i * KT + k) * 16 + e) * 64] = acc[i][k][e];
        dst[(NT * KT * 16 + i) * 64] = bsum[i];
      }
    }
    __builtin_amdgcn_s_barrier();
    if (wv < h) {
      const float* src = xl + (size_t)wv * GIVER_FLOATS + lane;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][k][e] += src[((i * KT + k) * 16 + e) * 64];
        bsum[i] += src[(NT * KT * 16 + i) * 64];
      }
    }
  }
  if (wv != 0) return;
  const int rr = lane & 31, hh = lane >> 5;
  float* slot = a.partial + (size_t)blockIdx.x * DW_SLOT_FLOATS;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      float* tile = slot + (8 * i + k) * 1024;
#pragma unroll
      for (int e = 0; e < 16; ++e) tile[((e & 3) + 8 * (e >> 2) + 4 * hh) * 32 + rr] = acc[i][k][e];
    }
    if (jb.b_off >= 0) {
      const float tot = bsum[i] + __shfl_xor(bsum[i], 32, 64);      // the two sample halves of the k-step
      if (hh == 0) slot[64 * 1024 + 32 * i + rr] = tot;
    }
  }
}

template <bool SPLIT>
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
  const bool col_major = k_tiles <= 2;                    // see s16_dw_kernel
  const int wr = col_major ? (wv & 3) : (wv >> 2), wc = col_major ? (wv >> 2) : (wv & 3);
  const int npairs = 2 * (n_tiles + k_tiles);             // real pair blocks per stage (4 .. 32)
  if (a.private_max_tiles >= n_tiles * k_tiles && npairs * 1024 <= DWP_WAVE_BYTES) {       // tiny jobs: sixteen independent wave pipelines
    switch (8 * n_tiles + k_tiles) {
      case 8 * 1 + 1: dw_private<1, 1, SPLIT>(a, jb, tile_lo, tile_hi, smem, wv, lane); return;
      case 8 * 1 + 2: dw_private<1, 2, SPLIT>(a, jb, tile_lo, tile_hi, smem, wv, lane); return;
      case 8 * 1 + 3: dw_private<1, 3, SPLIT>(a, jb, tile_lo, tile_hi, smem, wv, lane); return;
      case 8 * 1 + 4: dw_private<1, 4, SPLIT>(a, jb, tile_lo, tile_hi, smem, wv, lane); return;
      case 8 * 2 + 1: dw_private<2, 1, SPLIT>(a, jb, tile_lo, tile_hi, smem, wv, lane); return;
      case 8 * 2 + 2: dw_private<2, 2, SPLIT>(a, jb, tile_lo, tile_hi, smem, wv, lane); return;
      case 8 * 3 + 1: dw_private<3, 1, SPLIT>(a, jb, tile_lo, tile_hi, smem, wv, lane); return;
      case 8 * 4 + 1: dw_private<4, 1, SPLIT>(a, jb, tile_lo, tile_hi, smem, wv, lane); return;
      default: break;
    }
  }
  const bool active = (wr * DW_NPW < n_tiles) && (wc * 2 < k_tiles);
  const int kc = (k_tiles + 1) >> 1;                      // active wave columns (1, 2 or 4)
  int stride = npairs * 1024, ns = DW_LDS_BYTES / stride;
  if (ns > a.ring_cap) ns = a.ring_cap;
  const int cnt_w = (wv < npairs ? 1 : 0) + (wv + DW_WAVES < npairs ? 1 : 0);   // this wave's DMAs per stage
  f32x16 acc[DW_NPW][2];
#pragma unroll
  for (int i = 0; i < DW_NPW; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][k][e] = 0.0f;
  float bsum[2 * DW_NPW];                                 // [2 i + part]: part 0 = hi block, 1 = lo block of dZ tile i
  bool own[2 * DW_NPW];                                   // wave-uniform
#pragma unroll
  for (int c = 0; c < 2 * DW_NPW; ++c) { bsum[c] = 0.0f; own[c] = (c % kc) == wc && !(NERF_DWX & 8); }
  const int g16 = lane >> 4, i16 = lane & 15, hq = g16 >> 1, fsel = g16 & 1;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  const int src_row = lane >> 2, src_sel = (lane >> 1) & 1, src_half = lane & 1;
  const char* dzb = reinterpret_cast<const char*>(a.dz);
  const char* acb = reinterpret_cast<const char*>(a.acts);
  long long zstride_b = a.zstride * 16, astride_b = a.astride * 16;
  int z_lo = a.z_lo, a_lo = a.a_lo, dz_slot = jb.dz_slot, act_slot = jb.act_slot, nf = jb.nf, kf = jb.kf;
  asm volatile("" : "+s"(zstride_b), "+s"(astride_b), "+s"(z_lo), "+s"(a_lo), "+s"(dz_slot), "+s"(act_slot), "+s"(nf), "+s"(kf),
               "+s"(stride), "+s"(ns));

  auto issue = [&](int ht, int slot) {
#if NERF_DWX & 1
    return;
#endif
    const unsigned st = lds0 + __builtin_amdgcn_readfirstlane(slot) * stride;
    const int64_t tile = SPLIT ? ht >> 1 : ht;
    const unsigned row_off0 = (unsigned)(32 * ((SPLIT ? 16 * (ht & 1) : 0) + src_row) + 16 * src_half);
#pragma unroll
    for (int k = 0; k < DW_NPW; ++k) {
      const int i = wv + DW_WAVES * k;                     // wave-uniform
      if (i < npairs) {
        const bool is_z = i < 2 * n_tiles;
        const int j = is_z ? i : i - 2 * n_tiles;
        const int nt_ = is_z ? n_tiles : k_tiles;
        const int lo_part = j >= nt_ ? 1 : 0;
        const int t = lo_part ? j - nt_ : j;
        const int nfr = is_z ? nf : kf;
        int fr = 2 * t + src_sel; if (fr >= nfr) fr = nfr - 1;
        // part 1 of an operand: SPLIT the lo blocks (z_lo / a_lo slots further), bf16 samples 16..31 of the same fragments
        const int slot_ = (is_z ? dz_slot + (SPLIT ? lo_part * z_lo : 0) : act_slot + (SPLIT ? lo_part * a_lo : 0)) + fr;
        const unsigned row_off = row_off0 + (SPLIT ? 0 : 512 * lo_part);
        const char* src = (is_z ? dzb + tile * zstride_b : acb + tile * astride_b) + (int64_t)slot_ * 1024 + row_off;
        dma_frag_nt(src, st + i * 1024);
      }
    }
  };
  auto process = [&](const char* st) {
    if (active) {
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
        const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
        // bias gradient = row sums of dZ: v_dot2c_f32_bf16 against (1, 1).  (The empty asm keeps the wave-uniform branch a branch:
        // hipcc otherwise runs all sixteen v_dot2c on every wave and selects with v_cndmask.)
        if (own[2 * i]) {
          asm volatile("" ::: "memory");
#pragma unroll
          for (int j = 0; j < 8; j += 2) { const bf16x2 q = {ah[j], ah[j + 1]}; bsum[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(q, ones, bsum[2 * i], false); }
        }
        if (own[2 * i + 1]) {
          asm volatile("" ::: "memory");
#pragma unroll
          for (int j = 0; j < 8; j += 2) { const bf16x2 q = {al[j], al[j + 1]}; bsum[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(q, ones, bsum[2 * i + 1], false); }
        }
#if NERF_DWX & 16
        asm volatile("" :: "v"(ah), "v"(al), "v"(bh[0]), "v"(bl[0]), "v"(bh[1]), "v"(bl[1]));
#else
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (SPLIT) {                         // lo hi + hi lo + hi hi
            acc[i][k] = mfma32(al, bh[k], acc[i][k]);
            acc[i][k] = mfma32(ah, bl[k], acc[i][k]);
            acc[i][k] = mfma32(ah, bh[k], acc[i][k]);
          } else {                             // samples 0-15, then 16-31
            acc[i][k] = mfma32(ah, bh[k], acc[i][k]);
            acc[i][k] = mfma32(al, bl[k], acc[i][k]);
          }
        }
#endif
      }
    }
  };
  // wait until this wave's DMAs of everything but the `younger` most recently issued stages have landed (wave-uniform, <= 14 stages
  // x <= 2 DMAs), and until its own transposed reads of the previous pair are complete
  auto wait_landed = [&](int younger) {
    switch (younger * cnt_w) {
#define NERF_VM_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory"); break;
      NERF_VM_CASE(0) NERF_VM_CASE(1) NERF_VM_CASE(2) NERF_VM_CASE(3) NERF_VM_CASE(4) NERF_VM_CASE(5) NERF_VM_CASE(6) NERF_VM_CASE(7)
      NERF_VM_CASE(8) NERF_VM_CASE(9) NERF_VM_CASE(10) NERF_VM_CASE(11) NERF_VM_CASE(12) NERF_VM_CASE(13) NERF_VM_CASE(14)
      NERF_VM_CASE(16) NERF_VM_CASE(18) NERF_VM_CASE(20) NERF_VM_CASE(22) NERF_VM_CASE(24) NERF_VM_CASE(26) NERF_VM_CASE(28)
#undef NERF_VM_CASE
      default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
    }
  };
  // stages of this workgroup: SPLIT half tiles (two per sample tile, n even), bf16 sample tiles (n may be odd: the last pair is single)
  const int ht_lo = SPLIT ? 2 * tile_lo : tile_lo, n = SPLIT ? 2 * (tile_hi - tile_lo) : tile_hi - tile_lo;
  int issued = 0, slot_i = 0, slot_p = 0;                  // stages issued so far; the slot the next issue / the next process uses
  for (; issued < ns && issued < n; ++issued) { issue(ht_lo + issued, slot_i); slot_i = slot_i + 1 == ns ? 0 : slot_i + 1; }
  for (int p = 0; p < n; p += 2) {
    wait_landed(issued - (p + 2 < n ? p + 2 : n));         // stages p, p + 1 landed (this wave's part)
    DW_BARRIER();                                          // ... for every wave; every wave is done with the stages before p
#pragma unroll
    for (int k = 0; k < 2; ++k)                            // the slots of stages < p are free: stage s may go once s - ns < p
      if (issued < n && issued - ns < p) { issue(ht_lo + issued, slot_i); slot_i = slot_i + 1 == ns ? 0 : slot_i + 1; ++issued; }
    process(smem + slot_p * stride); slot_p = slot_p + 1 == ns ? 0 : slot_p + 1;
    if (SPLIT || p + 1 < n) { process(smem + slot_p * stride); slot_p = slot_p + 1 == ns ? 0 : slot_p + 1; }
  }
  // bias partial sums of a wave row -> its wc == 0 wave, through the LDS (every wave takes part in the two barriers)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  float* xb = reinterpret_cast<float*>(smem);              // [wr][c][lane]
#pragma unroll
  for (int c = 0; c < 2 * DW_NPW; ++c)
    if (own[c] && active) xb[(wr * 2 * DW_NPW + c) * 64 + lane] = bsum[c];
  __builtin_amdgcn_s_barrier();
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
      const float lo = xb[(wr * 2 * DW_NPW + 2 * i + 1) * 64 + lane], hi = xb[(wr * 2 * DW_NPW + 2 * i) * 64 + lane];
      const float t = lo + hi;
      const float tot = t + __shfl_xor(t, 32, 64);         // the two sample halves of the k-step
      if (hh == 0) slot[64 * 1024 + 32 * nt + rr] = tot;
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

int pack(const float* params, void* packed_s16, hipStream_t s) {
  char* base = static_cast<char*>(packed_s16);
  hipLaunchKernelGGL(pack_s16_kernel, dim3((PACK_PAIRS * 64 + 255) / 256), dim3(256), 0, s, params,
                     reinterpret_cast<bf16x8*>(base), reinterpret_cast<bf16x8*>(base + (size_t)F_FRAGS * 1024));
  return check_launch("nerf_mlp_pack (split-bf16 image)");
}

int forward(const void* packed_s16, const float* bias_slots, const float* x, const float* rays, const float* z, int64_t M,
            int n, int freq_mode, float* out, void* acts, int64_t astride16, int persistent_wgs, hipStream_t s) {
  FwdArgs a;
  a.queue = passq_slot();
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
  once[mode].run([&] { const void* k = mode == 0 ? reinterpret_cast<const void*>(s16_fwd_kernel<0>) : reinterpret_cast<const void*>(s16_fwd_kernel<1>);
    (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, FwdRing::LDS_BYTES); });
  if (mode == 0) hipLaunchKernelGGL(s16_fwd_kernel<0>, g, b, FwdRing::LDS_BYTES, s, a);
  else hipLaunchKernelGGL(s16_fwd_kernel<1>, g, b, FwdRing::LDS_BYTES, s, a);
  return check_launch("mlp training forward (split bf16)");
}

int backward_chain(const void* packed_s16, const void* acts, const float* d_raw, int64_t M, void* dz, int64_t astride16,
                   int64_t zstride16, int persistent_wgs, hipStream_t s) {
  BwdArgs b;
  b.queue = passq_slot();
  b.wb = reinterpret_cast<const bf16x8*>(static_cast<const char*>(packed_s16) + (size_t)F_FRAGS * 1024);
  b.acts = acts; b.d_raw = d_raw; b.M = M; b.dz = dz; b.astride = astride16; b.zstride = zstride16;
  const int64_t nsuper = ((M + 31) / 32 + NW - 1) / NW;
  static DevOnce once;
  once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(s16_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BwdRing::LDS_BYTES); });
  hipLaunchKernelGGL(s16_bwd_kernel, dim3((unsigned)(nsuper < persistent_wgs ? nsuper : persistent_wgs)), dim3(64 * NW),
                     BwdRing::LDS_BYTES, s, b);
  return check_launch("mlp backward chain (split bf16)");
}

int g_dw_variant = 1;          // nerf_set_option("dw22_variant"): 1 (default) 256 x 256 jobs on mlp_dww.hip's kernel, the others here; 0 every job here

int launch_dw_kernel(const DwArgs& d, int workgroups, bool split_bf16, hipStream_t s) {
  static DevOnce once[2];
  once[split_bf16].run([&] { (void)hipFuncSetAttribute(split_bf16 ? reinterpret_cast<const void*>(s16_dw_kernel<true>) : reinterpret_cast<const void*>(s16_dw_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS_BYTES); });
  if (split_bf16) hipLaunchKernelGGL(s16_dw_kernel<true>, dim3(workgroups), dim3(64 * DW_WAVES), DW_LDS_BYTES, s, d);
  else hipLaunchKernelGGL(s16_dw_kernel<false>, dim3(workgroups), dim3(64 * DW_WAVES), DW_LDS_BYTES, s, d);
  return check_launch(split_bf16 ? "mlp dW (split bf16)" : "mlp dW (bf16)");
}

}  // namespace s16
}  // namespace nerf
