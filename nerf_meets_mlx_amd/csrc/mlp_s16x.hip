// Split-bf16 kernels (nerf_mlp_arch.precision == 22) of the image-fitting model and of the 2 x 64 hash-grid model for gfx950.
//
// The reference runs both networks in float32: the 2-D image fit (entrypoints/__viser_image_learning.py:198-236:
// NeRF(channel_input=40, channel_input_views=0, channel_output=3, is_use_view_directions=False) under nn.value_and_grad) and --
// in the wiring BASELINE configs[4] names -- its NeRF class at Instant-NGP size behind encoding/multi_hash.py:79-136 and
// encoding/spherical_harmonics.py:33-94.  Rounds 1-4 had bf16-operand kernels only for these two shapes (mlp.hip); here they
// get the arithmetic of mlp_s16.hip: every float32 operand as (hi, lo) bf16 pair, three v_mfma_f32_32x32x16_bf16 per product,
// fp32 accumulate -- float32-class results (forward <= 1e-5 of the output scale, gradients <= 1e-4 rel-L2 against the fp32
// oracle) at a third of the bf16 matrix rate.  Device building blocks: mlp_s16_dev.h; layouts: mlp_arch2.h (LI / LN) with every
// fragment doubled; the weight-gradient kernel is mlp_s16.hip's (job table and split-K reduce shared with all other modes).
//
// Image model: 8 x 256 trunk as in mlp_s16.hip (one wave = 32 samples, one wave per SIMD, (hi, lo) weight pairs through the
// 4-stage LDS ring by LDS-DMA), `output_linear` as a <= 4-row head (models/NeRF.py:196-197,241).
// 2 x 64 model: the 2 x 31 forward (2 x 27 transposed) fragments are LDS-resident (64 KiB), eight independent waves per
// workgroup; in the fused query the hash features are gathered from the float32 master tables and interpolated in float32
// (encoding/multi_hash.py:112-131) and enter the first layer as (hi, lo) pairs -- nothing is rounded to 8 bits anywhere.
#include "mlp_s16_dev.h"
#include "mlp_arch2.h"
#include "hash_common.h"
#include "mlp_s16x.h"

namespace nerf {
namespace s16x {

using s16::HL; using s16::split2; using s16::split_slots; using s16::mfma32; using s16::PairSink; using s16::NoPairSink;

static_assert(IMG_A_LO == LI::A_MASK && IMG_A_MASK == 2 * LI::A_MASK && IMG_A_SLOTS == IMG_A_MASK + 8, "image activation slots");
static_assert(IMG_Z_LO == LI::Z_SLOTS && IMG_Z_SLOTS == 2 * LI::Z_SLOTS, "image dZ slots");
static_assert(IMG_F_FRAGS == 2 * LI::F_TOTAL && IMG_B_FRAGS == 2 * LI::B_TOTAL && IMG_B_PADDED == 2 * LI::B_PADDED, "image pair streams");
static_assert(SM_A_LO == LN::A_MASK && SM_A_MASK == 2 * LN::A_MASK && SM_A_SLOTS == SM_A_MASK + 3, "2x64 activation slots");
static_assert(SM_Z_LO == LN::Z_SLOTS && SM_Z_SLOTS == 2 * LN::Z_SLOTS, "2x64 dZ slots");
static_assert(SM_F_FRAGS == 2 * LN::F_PADDED && SM_B_FRAGS == 2 * LN::B_PADDED, "2x64 pair streams");


// ==========================================================================================
// image model
// ==========================================================================================
constexpr int NW = 4;                                    // waves per workgroup: one per SIMD (256 activation registers)
constexpr int IMG_F_CHUNKS = IMG_F_FRAGS / RING_CHUNK, IMG_B_CHUNKS = IMG_B_PADDED / RING_CHUNK;      // 60, 58
static_assert(IMG_F_CHUNKS * RING_CHUNK == IMG_F_FRAGS && IMG_B_CHUNKS * RING_CHUNK == IMG_B_PADDED, "whole ring chunks");
#ifndef NERF_S16X_GROUP
#define NERF_S16X_GROUP 4
#endif
typedef RingW<IMG_F_CHUNKS, IMG_F_FRAGS, NERF_S16X_GROUP, NW, RING_CHUNK, RING_STAGES, true> ImgFwdRing;
typedef RingW<IMG_B_CHUNKS, IMG_B_FRAGS, 4, NW, RING_CHUNK, RING_STAGES, true> ImgBwdRing;

constexpr int IMG_PACK_PAIRS = LI::F_TOTAL + LI::B_PADDED;
__global__ void __launch_bounds__(256) pack_s16_img_kernel(const float* __restrict__ p, bf16x8* __restrict__ wf,
                                                           bf16x8* __restrict__ wb, int out_ch) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= IMG_PACK_PAIRS * 64) return;
  const int fp = t >> 6, lane = t & 63, r = lane & 31, h = lane >> 5;
  const bool fw = fp < LI::F_TOTAL;
  const int f = fw ? fp : fp - LI::F_TOTAL;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = fw ? fwd_src_img(p, f, r, h, j, out_ch) : (f < LI::B_TOTAL ? bwd_src_img(p, f, r, h, j, out_ch) : 0.0f);
  bf16x8 hi[1], lo[1];
  split_slots<8>(v, hi, lo);
  bf16x8* dst = fw ? wf : wb;
  dst[(2 * f) * 64 + lane] = hi[0];
  dst[(2 * f + 1) * 64 + lane] = lo[0];
}

struct ImgArgs {
  const bf16x8* wf; const bf16x8* wb; const float* bias;
  const float* x;        // [M,40] embedded rows
  const float* d_out;    // [M,out_ch]
  int64_t M; int out_ch;
  float* out;            // [M,out_ch]
  void* acts; void* dz;
  int64_t astride, zstride;
};

template <bool STORE> struct SinkOf { typedef NoPairSink type; };
template <> struct SinkOf<true> { typedef PairSink type; };
template <bool STORE>
__device__ __forceinline__ typename SinkOf<STORE>::type make_sink(void* base, int64_t tile, int64_t stride16, int slot0, int lo_off, int r, int h);
template <>
__device__ __forceinline__ PairSink make_sink<true>(void* base, int64_t tile, int64_t stride16, int slot0, int lo_off, int r, int h) {
  return PairSink{base, tile, stride16, slot0, lo_off, r, h};
}
template <>
__device__ __forceinline__ NoPairSink make_sink<false>(void*, int64_t, int64_t, int, int, int, int) { return NoPairSink{}; }

// All 9 layers for the wave's 32 samples.  Waves past the end compute on clamped inputs, store into the padding tiles of the
// workspace and write no output, so every wave runs the same instruction stream (the ring needs it).
template <bool STORE, class WS>
__device__ __forceinline__ void img_fwd_tiles(const ImgArgs& a, WS& ws, int64_t tile0, int64_t ntiles, int lane) {
  const int r = lane & 31, h = lane >> 5;
  bf16x8 xh[3], xl[3];
  {
    const int64_t tile = tile0 < ntiles ? tile0 : ntiles - 1;
    int64_t m = tile * 32 + r; if (m >= a.M) m = a.M - 1;
    const float* row = a.x + m * LI::CIN;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) s16::row_frag(row, ks, h, LI::CIN, xh[ks], xl[ks]);
  }
  if (STORE) {
    store_frags<3>(a.acts, tile0, a.astride, LI::A_X, xh, r, h);
    store_frags<3>(a.acts, tile0, a.astride, IMG_A_LO + LI::A_X, xl, r, h);
  }
  bf16x8 hah[16], hal[16], hbh[16], hbl[16];
  u32x4 mk;
#define SINK(slot0) make_sink<STORE>(a.acts, tile0, a.astride, slot0, IMG_A_LO, r, h)
#define MASK_BEGIN() mk = u32x4{0u, 0u, 0u, 0u}
#define MASK_STORE(layer) do { if (STORE) *reinterpret_cast<u32x4*>(frag_ptr(a.acts, tile0, a.astride, IMG_A_MASK + (layer), r, h)) = mk; } while (0)
  MASK_BEGIN();
  s16::layer_fwd<3, 8, true, STORE>(ws, LI::F_L0, 0, xh, xl, hah, hal, mk, lane, SINK(LI::A_H0));
  MASK_STORE(0);
  MASK_BEGIN();
  s16::layer_fwd<16, 8, true, STORE>(ws, LI::F_L1 + 0 * 128, 256, hah, hal, hbh, hbl, mk, lane, SINK(LI::A_H0 + 16));
  MASK_STORE(1);
  MASK_BEGIN();
  s16::layer_fwd<16, 8, true, STORE>(ws, LI::F_L1 + 1 * 128, 512, hbh, hbl, hah, hal, mk, lane, SINK(LI::A_H0 + 32));
  MASK_STORE(2);
  MASK_BEGIN();
  s16::layer_fwd<16, 8, true, STORE>(ws, LI::F_L1 + 2 * 128, 768, hah, hal, hbh, hbl, mk, lane, SINK(LI::A_H0 + 48));
  MASK_STORE(3);
  MASK_BEGIN();
  s16::layer_fwd<16, 8, true, STORE>(ws, LI::F_L1 + 3 * 128, 1024, hbh, hbl, hah, hal, mk, lane, SINK(LI::A_H0 + 64));
  MASK_STORE(4);
  {                                                           // pos5 on concat[input, h]  (models/NeRF.py:224-225)
    bf16x8 cth[19], ctl[19];
#pragma unroll
    for (int k = 0; k < 3; ++k) { cth[k] = xh[k]; ctl[k] = xl[k]; }
#pragma unroll
    for (int k = 0; k < 16; ++k) { cth[3 + k] = hah[k]; ctl[3 + k] = hal[k]; }
    MASK_BEGIN();
    s16::layer_fwd<19, 8, true, STORE>(ws, LI::F_L5, 1280, cth, ctl, hbh, hbl, mk, lane, SINK(LI::A_H0 + 80));
    MASK_STORE(5);
  }
  MASK_BEGIN();
  s16::layer_fwd<16, 8, true, STORE>(ws, LI::F_L6, 1536, hbh, hbl, hah, hal, mk, lane, SINK(LI::A_H0 + 96));
  MASK_STORE(6);
  MASK_BEGIN();
  s16::layer_fwd<16, 8, true, STORE>(ws, LI::F_L7, 1792, hah, hal, hbh, hbl, mk, lane, SINK(LI::A_H0 + 112));
  MASK_STORE(7);
  const f32x16 o = s16::head<16>(ws, LI::F_OUT, LI::BI_OUT, hbh, hbl, lane);      // output_linear  (models/NeRF.py:241)
  const int64_t mo = tile0 * 32 + r;
  if (h == 0 && tile0 < ntiles && mo < a.M) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < a.out_ch) a.out[mo * a.out_ch + c] = o[c];
  }
#undef SINK
#undef MASK_BEGIN
#undef MASK_STORE
}

template <bool STORE>
__global__ void __launch_bounds__(64 * NW) s16_img_fwd_kernel(ImgArgs a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ntiles = (a.M + 31) >> 5, nsuper = (ntiles + NW - 1) / NW;
  ImgFwdRing ws;
  ws.wsrc = reinterpret_cast<const char*>(a.wf);
  ws.lane16 = 16 * lane;
  ws.lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(ring_smem));
  ws.wv = wv;
  ws.start(lane);
  ring_load_bias(a.bias, LI::BI_TOTAL, ImgFwdRing::BIAS_OFF);
  __syncthreads();
  for (int64_t sp = blockIdx.x; sp < nsuper; sp += gridDim.x) {
    int ln = lane;
    asm volatile("" : "+v"(ln));          // lane-derived values are recomputed per pass, not hoisted and spilled
    ws.new_pass();
    img_fwd_tiles<STORE>(a, ws, sp * NW + wv, ntiles, ln);
  }
  ws.drain();
}

template <class WS>
__device__ __forceinline__ void img_bwd_tiles(const ImgArgs& a, WS& ws, int64_t tile0, int64_t ntiles, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const bool live = tile0 < ntiles;
  const int64_t tile = live ? tile0 : ntiles - 1;
  bf16x8 zoh[1], zol[1];
  {
    const int64_t m = tile * 32 + r;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live && m < a.M && h == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < a.out_ch) v[c] = a.d_out[m * a.out_ch + c];                 // rows 0..out_ch-1
    }
    split_slots<8>(v, zoh, zol);
  }
  u32x4 mk[8];
#pragma unroll
  for (int l = 0; l < 8; ++l)
    mk[l] = *reinterpret_cast<const u32x4*>(frag_ptr(const_cast<void*>(a.acts), tile, a.astride, IMG_A_MASK + l, r, h));
#define ZSINK(slot0) PairSink{a.dz, tile0, a.zstride, slot0, IMG_Z_LO, r, h}
  ZSINK(LI::Z_OUT).put(0, zoh[0], zol[0]);
  bf16x8 zxh[16], zxl[16], zyh[16], zyl[16];
  s16::layer_bwd<1, 8, true>(ws, LI::B_OUT, zoh, zol, zyh, zyl, mk[7], lane, ZSINK(LI::Z_L0 + 112));               // dZ7
  s16::layer_bwd<16, 8, true>(ws, LI::B_L7 + 0 * 128, zyh, zyl, zxh, zxl, mk[6], lane, ZSINK(LI::Z_L0 + 96));     // dZ6
  s16::layer_bwd<16, 8, true>(ws, LI::B_L7 + 1 * 128, zxh, zxl, zyh, zyl, mk[5], lane, ZSINK(LI::Z_L0 + 80));     // dZ5
  s16::layer_bwd<16, 8, true>(ws, LI::B_L7 + 2 * 128, zyh, zyl, zxh, zxl, mk[4], lane, ZSINK(LI::Z_L0 + 64));     // dZ4 (pos5's H4 columns)
  s16::layer_bwd<16, 8, true>(ws, LI::B_L7 + 3 * 128, zxh, zxl, zyh, zyl, mk[3], lane, ZSINK(LI::Z_L0 + 48));     // dZ3
  s16::layer_bwd<16, 8, true>(ws, LI::B_L7 + 4 * 128, zyh, zyl, zxh, zxl, mk[2], lane, ZSINK(LI::Z_L0 + 32));     // dZ2
  s16::layer_bwd<16, 8, true>(ws, LI::B_L7 + 5 * 128, zxh, zxl, zyh, zyl, mk[1], lane, ZSINK(LI::Z_L0 + 16));     // dZ1
  s16::layer_bwd<16, 8, true>(ws, LI::B_L7 + 6 * 128, zyh, zyl, zxh, zxl, mk[0], lane, ZSINK(LI::Z_L0 + 0));      // dZ0
#undef ZSINK
}

__global__ void __launch_bounds__(64 * NW) s16_img_bwd_kernel(ImgArgs a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ntiles = (a.M + 31) >> 5, nsuper = (ntiles + NW - 1) / NW;
  ImgBwdRing ws;
  ws.wsrc = reinterpret_cast<const char*>(a.wb);
  ws.lane16 = 16 * lane;
  ws.lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(ring_smem));
  ws.wv = wv;
  ws.start(lane);
  for (int64_t sp = blockIdx.x; sp < nsuper; sp += gridDim.x) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    ws.new_pass();
    img_bwd_tiles(a, ws, sp * NW + wv, ntiles, ln);
  }
  ws.drain();
}

// ==========================================================================================
// 2 x 64 model
// ==========================================================================================
constexpr int SM_STREAM_BYTES = 64 * 1024;               // 64 fragments of 1 KiB: (hi, lo) pairs of one direction's stream
constexpr int SM_LDS_BYTES = SM_STREAM_BYTES + LN::BI_TOTAL * 4;

__global__ void __launch_bounds__(256) pack_s16_small_kernel(const float* __restrict__ p, bf16x8* __restrict__ wf,
                                                             bf16x8* __restrict__ wb) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= (LN::F_PADDED + LN::B_PADDED) * 64) return;
  const int fp = t >> 6, lane = t & 63, r = lane & 31, h = lane >> 5;
  const bool fw = fp < LN::F_PADDED;
  const int f = fw ? fp : fp - LN::F_PADDED;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = fw ? fwd_src_small(p, f, r, h, j) : bwd_src_small(p, f, r, h, j);
  bf16x8 hi[1], lo[1];
  split_slots<8>(v, hi, lo);
  bf16x8* dst = fw ? wf : wb;
  dst[(2 * f) * 64 + lane] = hi[0];
  dst[(2 * f + 1) * 64 + lane] = lo[0];
}

// weight source: the whole pair stream resident in LDS (copied once per workgroup); biases behind it
struct LdsPairW {
  __device__ __forceinline__ bf16x8 frag(int f, int lane) { return *reinterpret_cast<const bf16x8*>(ring_smem + f * 1024 + lane * 16); }
  __device__ __forceinline__ void note_stores(int) {}
  __device__ __forceinline__ float4 bias4(int slot) { return *reinterpret_cast<const float4*>(ring_smem + SM_STREAM_BYTES + slot * 4); }
};
__device__ __forceinline__ void lds_load_pairs(const bf16x8* __restrict__ w, const float* __restrict__ bias) {
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) *reinterpret_cast<bf16x8*>(ring_smem + i * 16) = w[i];
  if (bias)
    for (int i = threadIdx.x; i < LN::BI_TOTAL; i += blockDim.x) *reinterpret_cast<float*>(ring_smem + SM_STREAM_BYTES + 4 * i) = bias[i];
  __syncthreads();
}

struct SmallArgs {
  const bf16x8* wf; const bf16x8* wb; const float* bias;
  const float* x;        // [M,48]: 32 position features | 16 direction features
  const float* d_raw;    // [M,4]
  int64_t M;
  float* out;            // [M,4] raw
  float* d_x;            // [M,32] dL/d(position features) or nullptr
  void* acts; void* dz;
  int64_t astride, zstride;
  const float* rays; const float* z; int n; const float* tables; uint32_t T; ResTab rt; float pos_scale, pos_offset;
  int ray_major; int64_t B;
};

// B fragments of one sample straight from the float32 hash tables and the view direction (lane / channel mapping of
// mlp.hip:ngp_row_frags): lane (r, h) owns channels kperm(ks, h, j) of k-step ks = levels 8 ks + 4 (j >> 2) + 2 h + ((j & 3) >> 1),
// feature j & 1; SH degree 3 = 16 channels = one k-step.  Interpolated values stay float32 until they are split.
__device__ __forceinline__ void ngp_row_pairs(const SmallArgs& a, int64_t m, int h, bf16x8 (&xh)[2], bf16x8 (&xl)[2],
                                              bf16x8 (&dh)[1], bf16x8 (&dl)[1]) {
  const float* rr = a.rays + (int64_t)((uint64_t)m / (unsigned)a.n) * NERF_RAY_STRIDE;
  const float zv = a.z[m];
  // render.py:142, then the scene box -> unit cube map (same two roundings as hash_common.h:point_of)
  const float px = (rr[0] + zv * rr[3]) * a.pos_scale + a.pos_offset, py = (rr[1] + zv * rr[4]) * a.pos_scale + a.pos_offset;
  const float pz = (rr[2] + zv * rr[5]) * a.pos_scale + a.pos_offset;
  const uint32_t mask = a.T - 1;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int l = 8 * ks + 4 * q + 2 * h + e;
        const Corners c = corners_of(px, py, pz, a.rt.res[l], mask);
        const FeatVec<2> fv = hash_level<2>(a.tables + (size_t)l * a.T * 2, c);
        v[4 * q + 2 * e] = fv.v[0];
        v[4 * q + 2 * e + 1] = fv.v[1];
      }
    split_slots<8>(v, &xh[ks], &xl[ks]);
  }
  float sh[16];
  sh_eval(rr[8], rr[9], rr[10], 3, sh);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = h == 0 ? sh[8 * (j >> 2) + (j & 3)] : sh[8 * (j >> 2) + 4 + (j & 3)];
  split_slots<8>(v, dh, dl);
}

template <bool STORE, bool FUSED>
__global__ void __launch_bounds__(512) s16_small_fwd_kernel(SmallArgs a) {
  lds_load_pairs(a.wf, a.bias);
  const int lane0 = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool rmaj = FUSED && !STORE && a.ray_major;
  const int64_t ntiles = rmaj ? ((a.B + 31) >> 5) * a.n : (a.M + 31) >> 5;
  LdsPairW ws;
  for (int64_t tile0 = (int64_t)blockIdx.x * 8 + wv; tile0 < ntiles; tile0 += (int64_t)gridDim.x * 8) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));        // fragment addresses are per-tile values: the LDS reads are not hoisted (spills)
    const int r = lane & 31, h = lane >> 5;
    int64_t m = tile0 * 32 + r;
    bool valid = m < a.M;
    if (rmaj) {                            // tile = (block of 32 rays, depth index): sample m = ray * n + depth
      const int64_t rb = tile0 / a.n;
      const int depth = (int)(tile0 - rb * a.n);
      int64_t ray = rb * 32 + r;
      valid = ray < a.B;
      if (!valid) ray = a.B - 1;
      m = ray * a.n + depth;
    }
    if (m >= a.M) m = a.M - 1;
    bf16x8 xh[2], xl[2], dh[1], dl[1];
    if (FUSED) {
      ngp_row_pairs(a, m, h, xh, xl, dh, dl);
    } else {
      const float* row = a.x + m * LN::CIN;
      s16::row_frag(row, 0, h, LN::CPOS, xh[0], xl[0]); s16::row_frag(row, 1, h, LN::CPOS, xh[1], xl[1]);
      s16::row_frag(row + LN::CPOS, 0, h, LN::CDIR, dh[0], dl[0]);
    }
#define SINK(slot0) make_sink<STORE>(a.acts, tile0, a.astride, slot0, SM_A_LO, r, h)
#define MASK_STORE(layer) do { if (STORE) *reinterpret_cast<u32x4*>(frag_ptr(a.acts, tile0, a.astride, SM_A_MASK + (layer), r, h)) = mk; } while (0)
    if (STORE) {
      store_frags<2>(a.acts, tile0, a.astride, LN::A_X, xh, r, h); store_frags<2>(a.acts, tile0, a.astride, SM_A_LO + LN::A_X, xl, r, h);
      store_frags<1>(a.acts, tile0, a.astride, LN::A_DX, dh, r, h); store_frags<1>(a.acts, tile0, a.astride, SM_A_LO + LN::A_DX, dl, r, h);
    }
    u32x4 mk;
    bf16x8 h0h[4], h0l[4], h1h[4], h1l[4], fth[4], ftl[4];
    mk = u32x4{0u, 0u, 0u, 0u};
    s16::layer_fwd<2, 2, true, STORE>(ws, LN::F_L0, LN::BI_L0, xh, xl, h0h, h0l, mk, lane, SINK(LN::A_H0));
    MASK_STORE(0);
    mk = u32x4{0u, 0u, 0u, 0u};
    s16::layer_fwd<4, 2, true, STORE>(ws, LN::F_L1, LN::BI_L1, h0h, h0l, h1h, h1l, mk, lane, SINK(LN::A_H1));
    MASK_STORE(1);
    s16::layer_fwd<4, 2, false, false>(ws, LN::F_FA, LN::BI_FEAT, h1h, h1l, fth, ftl, mk, lane, SINK(LN::A_FEAT));     // feature: no activation
    const float alpha = s16::head<4>(ws, LN::F_FA + 8, LN::BI_ALPHA, h1h, h1l, lane)[0];
    bf16x8 hdh[2], hdl[2];
    {
      bf16x8 cth[5], ctl[5];
#pragma unroll
      for (int k = 0; k < 4; ++k) { cth[k] = fth[k]; ctl[k] = ftl[k]; }
      cth[4] = dh[0]; ctl[4] = dl[0];
      mk = u32x4{0u, 0u, 0u, 0u};
      s16::layer_fwd<5, 1, true, STORE>(ws, LN::F_DIR, LN::BI_DIR, cth, ctl, hdh, hdl, mk, lane, SINK(LN::A_HD));
      MASK_STORE(2);
    }
    const f32x16 rgb = s16::head<2>(ws, LN::F_RGB, LN::BI_RGB, hdh, hdl, lane);
    if (h == 0 && valid) {
      float4 o; o.x = rgb[0]; o.y = rgb[1]; o.z = rgb[2]; o.w = alpha;
      *reinterpret_cast<float4*>(a.out + m * 4) = o;
    }
#undef SINK
#undef MASK_STORE
  }
}

__global__ void __launch_bounds__(512) s16_small_bwd_kernel(SmallArgs a) {
  lds_load_pairs(a.wb, nullptr);
  const int lane0 = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ntiles = (a.M + 31) >> 5;
  LdsPairW ws;
  for (int64_t tile0 = (int64_t)blockIdx.x * 8 + wv; tile0 < ntiles; tile0 += (int64_t)gridDim.x * 8) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int r = lane & 31, h = lane >> 5;
    const int64_t m = tile0 * 32 + r;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < a.M && h == 0) g = *reinterpret_cast<const float4*>(a.d_raw + m * 4);
    bf16x8 zrh[1], zrl[1], zah[1], zal[1];
    {
      float vr[8] = {g.x, g.y, g.z, 0.f, 0.f, 0.f, 0.f, 0.f}, va[8] = {g.w, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      split_slots<8>(vr, zrh, zrl);
      split_slots<8>(va, zah, zal);
    }
    u32x4 mk[3];
#pragma unroll
    for (int l = 0; l < 3; ++l)
      mk[l] = *reinterpret_cast<const u32x4*>(frag_ptr(const_cast<void*>(a.acts), tile0, a.astride, SM_A_MASK + l, r, h));
#define ZSINK(slot0) PairSink{a.dz, tile0, a.zstride, slot0, SM_Z_LO, r, h}
    ZSINK(LN::Z_RGB).put(0, zrh[0], zrl[0]);
    ZSINK(LN::Z_A).put(0, zah[0], zal[0]);
    bf16x8 zdh[2], zdl[2], zfh[4], zfl[4], z1h[4], z1l[4], z0h[4], z0l[4];
    s16::layer_bwd<1, 1, true>(ws, LN::B_RGB, zrh, zrl, zdh, zdl, mk[2], lane, ZSINK(LN::Z_D));
    s16::layer_bwd<2, 2, false>(ws, LN::B_DIR, zdh, zdl, zfh, zfl, mk[2], lane, ZSINK(LN::Z_F));               // d feature
    {
      bf16x8 cth[5], ctl[5];
#pragma unroll
      for (int k = 0; k < 4; ++k) { cth[k] = zfh[k]; ctl[k] = zfl[k]; }
      cth[4] = zah[0]; ctl[4] = zal[0];
      s16::layer_bwd<5, 2, true>(ws, LN::B_FA, cth, ctl, z1h, z1l, mk[1], lane, ZSINK(LN::Z_L1));              // dZ1
    }
    s16::layer_bwd<4, 2, true>(ws, LN::B_L1, z1h, z1l, z0h, z0l, mk[0], lane, ZSINK(LN::Z_L0));                // dZ0
#undef ZSINK
    if (a.d_x) {                                          // dL/dx = W0^T dZ0, float32: rows (i&3) + 8 (i>>2) + 4 h
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) {
        const bf16x8 ah = ws.frag(2 * (LN::B_L0 + ns), lane), al = ws.frag(2 * (LN::B_L0 + ns) + 1, lane);
        acc = mfma32(ah, z0l[ns], acc);
        acc = mfma32(al, z0h[ns], acc);
        acc = mfma32(ah, z0h[ns], acc);
      }
      if (m < a.M) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 o; o.x = acc[4 * q]; o.y = acc[4 * q + 1]; o.z = acc[4 * q + 2]; o.w = acc[4 * q + 3];
          *reinterpret_cast<float4*>(a.d_x + m * LN::CPOS + 8 * q + 4 * h) = o;
        }
      }
    }
  }
}

// ==========================================================================================
// host side
// ==========================================================================================
int img_pack(const float* params, int out_ch, void* packed, hipStream_t s) {
  char* base = static_cast<char*>(packed);
  hipLaunchKernelGGL(pack_s16_img_kernel, dim3((IMG_PACK_PAIRS * 64 + 255) / 256), dim3(256), 0, s, params,
                     reinterpret_cast<bf16x8*>(base), reinterpret_cast<bf16x8*>(base + (size_t)IMG_F_FRAGS * 1024), out_ch);
  return check_launch("nerf_mlp_pack (image model, split-bf16 image)");
}

static void img_fill(ImgArgs& a, const void* packed, const float* bias_slots, int out_ch, int64_t astride16, int64_t zstride16) {
  const char* base = static_cast<const char*>(packed);
  a.wf = reinterpret_cast<const bf16x8*>(base);
  a.wb = reinterpret_cast<const bf16x8*>(base + (size_t)IMG_F_FRAGS * 1024);
  a.bias = bias_slots; a.out_ch = out_ch; a.astride = astride16; a.zstride = zstride16;
  a.x = nullptr; a.d_out = nullptr; a.out = nullptr; a.acts = nullptr; a.dz = nullptr; a.M = 0;
}

int img_forward(const void* packed, const float* bias_slots, const float* x, int64_t M, int out_ch, float* out, void* acts,
                int64_t astride16, int persistent_wgs, hipStream_t s) {
  ImgArgs a;
  img_fill(a, packed, bias_slots, out_ch, astride16, 0);
  a.x = x; a.out = out; a.acts = acts; a.M = M;
  const int64_t nsuper = ((M + 31) / 32 + NW - 1) / NW;
  const dim3 g((unsigned)(nsuper < persistent_wgs ? nsuper : persistent_wgs)), b(64 * NW);
  static DevOnce once;
  once.run([&] { // dynamic LDS above 64 KiB is an opt-in per kernel AND per device
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(s16_img_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, ImgFwdRing::LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(s16_img_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, ImgFwdRing::LDS_BYTES); });
  if (acts) hipLaunchKernelGGL(s16_img_fwd_kernel<true>, g, b, ImgFwdRing::LDS_BYTES, s, a);
  else hipLaunchKernelGGL(s16_img_fwd_kernel<false>, g, b, ImgFwdRing::LDS_BYTES, s, a);
  return check_launch("mlp forward (image model, split bf16)");
}

int img_backward_chain(const void* packed, const void* acts, const float* d_out, int64_t M, int out_ch, void* dz,
                       int64_t astride16, int64_t zstride16, int persistent_wgs, hipStream_t s) {
  ImgArgs a;
  img_fill(a, packed, nullptr, out_ch, astride16, zstride16);
  a.d_out = d_out; a.acts = const_cast<void*>(acts); a.dz = dz; a.M = M;
  const int64_t nsuper = ((M + 31) / 32 + NW - 1) / NW;
  static DevOnce once;
  once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(s16_img_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ImgBwdRing::LDS_BYTES); });
  hipLaunchKernelGGL(s16_img_bwd_kernel, dim3((unsigned)(nsuper < persistent_wgs ? nsuper : persistent_wgs)), dim3(64 * NW),
                     ImgBwdRing::LDS_BYTES, s, a);
  return check_launch("mlp backward chain (image model, split bf16)");
}

int small_pack(const float* params, void* packed, hipStream_t s) {
  char* base = static_cast<char*>(packed);
  hipLaunchKernelGGL(pack_s16_small_kernel, dim3(((LN::F_PADDED + LN::B_PADDED) * 64 + 255) / 256), dim3(256), 0, s, params,
                     reinterpret_cast<bf16x8*>(base), reinterpret_cast<bf16x8*>(base + (size_t)SM_F_FRAGS * 1024));
  return check_launch("nerf_mlp_pack (2x64 model, split-bf16 image)");
}

static void small_fill(SmallArgs& a, const void* packed, const float* bias_slots, int64_t astride16, int64_t zstride16) {
  const char* base = static_cast<const char*>(packed);
  a.wf = reinterpret_cast<const bf16x8*>(base);
  a.wb = reinterpret_cast<const bf16x8*>(base + (size_t)SM_F_FRAGS * 1024);
  a.bias = bias_slots;
  a.x = nullptr; a.d_raw = nullptr; a.out = nullptr; a.d_x = nullptr; a.acts = nullptr; a.dz = nullptr; a.M = 0;
  a.astride = astride16; a.zstride = zstride16;
  a.rays = nullptr; a.z = nullptr; a.n = 1; a.tables = nullptr; a.T = 0; a.pos_scale = 1.0f; a.pos_offset = 0.0f;
  a.ray_major = 0; a.B = 0;
  for (int l = 0; l < 32; ++l) a.rt.res[l] = 0.0f;
}

template <class K>
static void want_lds(K kernel, DevOnce& once) {
  once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SM_LDS_BYTES); });
}

int small_forward(const void* packed, const float* bias_slots, const float* x, int64_t M, float* out, void* acts,
                  int64_t astride16, const SmallQuery* q, hipStream_t s) {
  SmallArgs a;
  small_fill(a, packed, bias_slots, astride16, 0);
  a.x = x; a.out = out; a.acts = acts; a.M = M;
  int64_t ntiles = (M + 31) / 32;
  if (q && q->rays) {
    a.rays = q->rays; a.z = q->z; a.n = q->n; a.tables = q->tables; a.T = q->T; a.pos_scale = q->pos_scale; a.pos_offset = q->pos_offset;
    for (int l = 0; l < 32; ++l) a.rt.res[l] = q->res[l];
    a.B = q->B;
    a.ray_major = (!acts && q->ray_major && q->B >= 32) ? 1 : 0;
    if (a.ray_major) ntiles = ((q->B + 31) / 32) * (int64_t)q->n;
  }
  const int64_t nwg = (ntiles + 7) / 8;
  const dim3 g((unsigned)(nwg < 2048 ? nwg : 2048)), b(512);
  static DevOnce once[4];
  const bool fused = a.rays != nullptr;
  if (acts && fused) { want_lds(s16_small_fwd_kernel<true, true>, once[0]); hipLaunchKernelGGL((s16_small_fwd_kernel<true, true>), g, b, SM_LDS_BYTES, s, a); }
  else if (acts) { want_lds(s16_small_fwd_kernel<true, false>, once[1]); hipLaunchKernelGGL((s16_small_fwd_kernel<true, false>), g, b, SM_LDS_BYTES, s, a); }
  else if (fused) { want_lds(s16_small_fwd_kernel<false, true>, once[2]); hipLaunchKernelGGL((s16_small_fwd_kernel<false, true>), g, b, SM_LDS_BYTES, s, a); }
  else { want_lds(s16_small_fwd_kernel<false, false>, once[3]); hipLaunchKernelGGL((s16_small_fwd_kernel<false, false>), g, b, SM_LDS_BYTES, s, a); }
  return check_launch("mlp forward (2x64 model, split bf16)");
}

int small_backward_chain(const void* packed, const void* acts, const float* d_raw, int64_t M, void* dz, float* d_x,
                         int64_t astride16, int64_t zstride16, hipStream_t s) {
  SmallArgs a;
  small_fill(a, packed, nullptr, astride16, zstride16);
  a.d_raw = d_raw; a.acts = const_cast<void*>(acts); a.dz = dz; a.M = M; a.d_x = d_x;
  const int64_t nwg = ((M + 31) / 32 + 7) / 8;
  static DevOnce once;
  want_lds(s16_small_bwd_kernel, once);
  hipLaunchKernelGGL(s16_small_bwd_kernel, dim3((unsigned)(nwg < 2048 ? nwg : 2048)), dim3(512), SM_LDS_BYTES, s, a);
  return check_launch("mlp backward chain (2x64 model, split bf16)");
}

}  // namespace s16x
}  // namespace nerf
