// Fully-fused 8x256 NeRF MLP for gfx950 (a11, a12 / K3, K4, K5): positional encoding +
// 12 linear layers in ONE kernel, bf16 MFMA (v_mfma_f32_32x32x16_bf16) with fp32 accumulate.
//
// Orientation.  Every layer is computed TRANSPOSED:  H_out^T[n][m] = sum_k W[n][k] H_in^T[k][m]
// with A = W (rows n, nn.Linear's [out][in] layout gives 8 contiguous k per lane) and
// B = H_in^T (k on the lane half / element, sample m on lane&31).  The accumulator tile then
// has the SAMPLE on the lane and the output FEATURES in the 16 registers, which -- after
// bias + ReLU + cvt to bf16 -- is already the B operand of the next layer (the k order inside
// a 16-step is permuted: element j of lane half h is feature 16s + 8(j>>2) + 4h + (j&3); the
// weight fragments are packed in that order by nerf_mlp_pack).  So a wave keeps its 32 (or
// 64) samples in registers through all layers: no LDS round trip, no HBM traffic for
// activations in inference, and weights stream as fully coalesced 1 KiB fragments.
//
// Training stores the per-layer bf16 activations as 1 KiB "fragment blocks" (32 samples x
// 16 features, lane (r,h) at byte 32r+16h) which the backward chain kernel reloads as ReLU
// masks and the dW kernel consumes through LDS with ds_read_b64_tr_b16 (the sample axis
// becomes the MFMA K axis there).
//
// Map of the file: layouts + packing (namespace L) | weight sources (GlobalW: L1, RingW: LDS ring fed by LDS-DMA,
// LdsW: LDS-resident) | layer_fwd / fwd_tiles and the 32x32x16 forward kernels | the 16x16x32 render forward
// (mlp_fwd_ring16_kernel, the default for inference) | layer_bwd / bwd_tiles (dZ chain) | mlp_dw_kernel (dW / db) |
// second layout: image-fitting model (namespace LI) | third layout: 2 x 64 model of configs[4] (namespace LN) |
// C ABI entry points.
#include "common.h"
#include "mlp_ring.h"
#include "mlp_layout.h"
#include "mlp_frag.h"
#include "mlp_arch2.h"
#include "hash_common.h"
#include "mlp_params.h"
#include "mlp32.h"
#include "mlp22.h"
#include "mlp_s16.h"
#include "mlp_s16x.h"
#include <string.h>
#include <mutex>
#include <unordered_map>


namespace nerf {

extern int g_hash_combine_max_res;        // encode.hip


// static layout (namespace L), kperm: mlp_layout.h

// ------------------------------------------------------------------------------------------
// weight packing: fp32 master parameters -> bf16 MFMA fragments (+ fp32 bias slots)
// ------------------------------------------------------------------------------------------
// fwd_src / bwd_src (fragment element -> master parameter): mlp_frag.h

constexpr int PACK_THREADS = (L::F_TOTAL + L::B_PADDED) * 64 + L::BI_TOTAL;      // one thread per 16-byte lane slot / bias
__device__ __forceinline__ void pack_part(int tid, const float* __restrict__ p, bf16x8* __restrict__ wf,
                                          bf16x8* __restrict__ wb, float* __restrict__ bias) {
  const int nf = L::F_TOTAL * 64, nb = L::B_PADDED * 64;
  if (tid < nf + nb) {
    const bool fw = tid < nf;
    const int t = fw ? tid : tid - nf;
    const int f = t >> 6, lane = t & 63, r = lane & 31, h = lane >> 5;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)(fw ? fwd_src(p, f, r, h, j) : (f < L::B_TOTAL ? bwd_src(p, f, r, h, j) : 0.0f));
    (fw ? wf : wb)[t] = v;
  } else if (tid < nf + nb + L::BI_TOTAL) {
    const int s = tid - nf - nb;
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
// positional encoding straight into B-operand fragments
// ------------------------------------------------------------------------------------------


// models/embedding.py:30-71 channel order [x, sin(f0 x), cos(f0 x), ...]; cos(a) = sin(a + 1/4 rev)
template <int C0, int LIMIT, int NB>
__device__ __forceinline__ float pe_value(const float (&x)[3], const float (&fr)[NB], int h) {
  constexpr Chan a = chan_of(C0, LIMIT), b = chan_of(C0 + 4, LIMIT);
  const float xa = x[a.dim], xb = x[b.dim];
  const float xv = h ? xb : xa;
  const float fv = h ? fr[b.band] : fr[a.band];
  const float ph = h ? (b.kind == 2 ? 0.25f : 0.0f) : (a.kind == 2 ? 0.25f : 0.0f);
  const float arg = xv * fv;                                    // x * freq in fp32, as the reference
  const float t = __builtin_amdgcn_fractf(arg * 0.15915494309189535f + ph);
  const float s = __builtin_amdgcn_sinf(t);                     // sin(2 pi t)
  const float va = a.kind == 0 ? xa : (a.kind == 3 ? 0.0f : s);
  const float vb = b.kind == 0 ? xb : (b.kind == 3 ? 0.0f : s);
  return h ? vb : va;
}

template <int KS, int LIMIT, int NB, int... J>
__device__ __forceinline__ bf16x8 pe_frag_impl(const float (&x)[3], const float (&fr)[NB], int h) {
  bf16x8 v;
  ((v[J] = (__bf16)pe_value<16 * KS + 8 * (J >> 2) + (J & 3), LIMIT, NB>(x, fr, h)), ...);
  return v;
}
template <int KS, int LIMIT, int NB>
__device__ __forceinline__ bf16x8 pe_frag(const float (&x)[3], const float (&fr)[NB], int h) {
  return pe_frag_impl<KS, LIMIT, NB, 0, 1, 2, 3, 4, 5, 6, 7>(x, fr, h);
}

// fragment of an already-embedded row x[m][base + c], c < limit else 0   (NeRF.forward(x) entry)
__device__ __forceinline__ bf16x8 row_frag(const float* __restrict__ row, int ks, int h, int limit) {
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = kperm(ks, h, j);
    v[j] = (__bf16)(c < limit ? row[c] : 0.0f);
  }
  return v;
}

// weight sources (GlobalW: L1, RingW: LDS ring fed by LDS-DMA), FwdArgs, PeFreq: mlp_ring.h
// ------------------------------------------------------------------------------------------
// one linear layer on register-resident activations
// ------------------------------------------------------------------------------------------
// fragment blocks, packed-pair helpers, epilogue schedule, DwJob / DwArgs: mlp_frag.h

// out[t][2 nt + s] = act( W[nt-tile] . in[t] + bias );  fragments fbase + nt*KS + ks of the stream
template <int ST, int KS, int NT, bool RELU, bool MASKOUT, class WS, class SINK = NoSink>
__device__ __forceinline__ void layer_fwd(WS& ws, int fbase, int bias_slot, const bf16x8 (&in)[ST][KS],
                                          bf16x8 (&out)[ST][2 * NT], u32x4 (&mask)[ST], int lane,
                                          const SINK& sink = SINK()) {
  const int h = lane >> 5;
  f32x16 prev[ST];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    f32x16 acc[ST];
    acc_init_bias(acc[0], ws, bias_slot + 32 * nt, h);
#pragma unroll
    for (int t = 1; t < ST; ++t) acc[t] = acc[0];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 a = next_frag(ws, fbase + nt * KS + ks, lane);
#pragma unroll
      for (int t = 0; t < ST; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, in[t][ks], acc[t], 0, 0, 0);
      if (nt > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (quarter_pos(KS, q) == ks) {
#pragma unroll
            for (int t = 0; t < ST; ++t) {
              finish_quarter<RELU, MASKOUT>(prev[t], q, nt - 1, out[t][2 * nt - 2], out[t][2 * nt - 1], mask[t]);
              if (q == 3) { sink.put(t, 2 * nt - 2, out[t][2 * nt - 2]); sink.put(t, 2 * nt - 1, out[t][2 * nt - 1]); }
            }
          }
      }
    }
#pragma unroll
    for (int t = 0; t < ST; ++t) prev[t] = acc[t];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < ST; ++t)
      finish_quarter<RELU, MASKOUT>(prev[t], q, NT - 1, out[t][2 * NT - 2], out[t][2 * NT - 1], mask[t]);
#pragma unroll
  for (int t = 0; t < ST; ++t) { sink.put(t, 2 * NT - 2, out[t][2 * NT - 2]); sink.put(t, 2 * NT - 1, out[t][2 * NT - 1]); }
}


// All 12 layers for ST sample tiles of this wave (tile0 .. tile0+ST-1).  Tiles >= ntiles are computed on clamped
// inputs and never stored, so every wave runs the same instruction stream (the ring needs that).
// MODE 0: embedded rows;  MODE 1: rays + z with fused positional encodings
template <int ST, int MODE, bool STORE, class WS>
__device__ __forceinline__ void fwd_tiles(const FwdArgs& a, WS& ws, int64_t tile0, int64_t ntiles, int lane, PassQueue* pq = nullptr) {
  const int r = lane & 31, h = lane >> 5;
  bf16x8 pe[ST][4], dpe[ST][2];
  if (pq) pq->ask(ws_wave(ws), lane);          // dynamic pass queue (mlp_ring.h): before this pass's input loads
#pragma unroll
  for (int t = 0; t < ST; ++t) {
    int64_t tile = tile0 + t; if (tile >= ntiles) tile = ntiles - 1;
    int64_t m = tile * 32 + r; if (m >= a.M) m = a.M - 1;
    if (MODE == 0) {
      const float* row = a.x + m * 90;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) pe[t][ks] = row_frag(row, ks, h, 63);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) dpe[t][ks] = row_frag(row + 63, ks, h, 27);
    } else {
      const int64_t ray = (int64_t)((unsigned)m / (unsigned)a.n);     // M < 2^31 (checked on the host): 32-bit divide
      const float* rr = a.rays + ray * NERF_RAY_STRIDE;
      const float zv = a.z[m];
      float p[3], d[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { p[c] = rr[c] + zv * rr[3 + c]; d[c] = rr[8 + c]; }   // render.py:142
#if NERF_ABLATE == 3          // timing-only build 3: no positional-encoding arithmetic
      for (int q = 0; q < 4; ++q) for (int j = 0; j < 8; ++j) pe[t][q][j] = (__bf16)p[j % 3];
      for (int q = 0; q < 2; ++q) for (int j = 0; j < 8; ++j) dpe[t][q][j] = (__bf16)d[j % 3];
      if (false)
#endif
      {
      pe[t][0] = pe_frag<0, 63, 10>(p, a.fr.pos, h); pe[t][1] = pe_frag<1, 63, 10>(p, a.fr.pos, h);
      pe[t][2] = pe_frag<2, 63, 10>(p, a.fr.pos, h); pe[t][3] = pe_frag<3, 63, 10>(p, a.fr.pos, h);
      dpe[t][0] = pe_frag<0, 27, 4>(d, a.fr.dir, h); dpe[t][1] = pe_frag<1, 27, 4>(d, a.fr.dir, h);
      }
    }
  }
  if (pq) pq->publish(ws_wave(ws));
  int bad[ST];                      // non-finite inputs of this lane's sample (mlp_frag.h: nonfinite_flags), used at the output store
#pragma unroll
  for (int t = 0; t < ST; ++t)
  {
    bad[t] = nonfinite_flags<32>(nonfinite_bits(pe[t][0]) | nonfinite_bits(pe[t][1]) | nonfinite_bits(pe[t][2]) | nonfinite_bits(pe[t][3]),
                                 nonfinite_bits(dpe[t][0]) | nonfinite_bits(dpe[t][1]));
    // computed HERE, at the head of the pass.  Left to the scheduler, this arithmetic was sunk into the layer epilogues of the 2 x 64
    // training forward, whose layer-0 ReLU sign words then lost bits (tests: test_weight_gradients_of_the_two_dw_launch_forms_agree
    // [*-small-16]; the ISA of the interleaved form showed nothing wrong, the pinned form is bit-identical to the build without flags)
    asm volatile("" : "+v"(bad[t]));
  }
#define store(slot0, t, frags, count) \
  do { if (STORE) { store_frags<count>(a.acts, tile0 + (t), a.astride, slot0, frags, r, h); ws.note_stores(count); } } while (0)
#pragma unroll
  for (int t = 0; t < ST; ++t) { store(L::A_PE, t, pe[t], 4); store(L::A_DPE, t, dpe[t], 2); }

  bf16x8 ha[ST][16], hb[ST][16];
  u32x4 mk[ST];
#define SINK(slot0) FragSink<STORE>{a.acts, tile0, a.astride, slot0, r, h}
#define MASK_BEGIN() do { _Pragma("unroll") for (int t = 0; t < ST; ++t) mk[t] = u32x4{0u, 0u, 0u, 0u}; } while (0)
#define MASK_STORE(layer) do { if (STORE) { _Pragma("unroll") for (int t = 0; t < ST; ++t) \
    *reinterpret_cast<u32x4*>(frag_ptr(a.acts, tile0 + t, a.astride, L::A_MASK + (layer), r, h)) = mk[t]; ws.note_stores(ST); } } while (0)
  MASK_BEGIN();
  layer_fwd<ST, 4, 8, true, STORE>(ws, L::F_L0, 0, pe, ha, mk, lane, SINK(L::A_H0));
  MASK_STORE(0);
  // pos1..pos4 (ping-pong)
  MASK_BEGIN();
  layer_fwd<ST, 16, 8, true, STORE>(ws, L::F_L1 + 0 * 128, 256, ha, hb, mk, lane, SINK(L::A_H0 + 16));
  MASK_STORE(1);
  MASK_BEGIN();
  layer_fwd<ST, 16, 8, true, STORE>(ws, L::F_L1 + 1 * 128, 512, hb, ha, mk, lane, SINK(L::A_H0 + 32));
  MASK_STORE(2);
  MASK_BEGIN();
  layer_fwd<ST, 16, 8, true, STORE>(ws, L::F_L1 + 2 * 128, 768, ha, hb, mk, lane, SINK(L::A_H0 + 48));
  MASK_STORE(3);
  MASK_BEGIN();
  layer_fwd<ST, 16, 8, true, STORE>(ws, L::F_L1 + 3 * 128, 1024, hb, ha, mk, lane, SINK(L::A_H0 + 64));
  MASK_STORE(4);
  // pos5 on concat[input_pos, h]  (models/NeRF.py:224-225)
  {
    bf16x8 cat[ST][20];
#pragma unroll
    for (int t = 0; t < ST; ++t) {
      if (STORE && is_ring<WS>::value) {   // register relief in the training ring kernel: the encoding was just stored
#pragma unroll
        for (int k = 0; k < 4; ++k) pe[t][k] = *frag_ptr(a.acts, tile0 + t, a.astride, L::A_PE + k, r, h);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) cat[t][k] = pe[t][k];
#pragma unroll
      for (int k = 0; k < 16; ++k) cat[t][4 + k] = ha[t][k];
    }
    MASK_BEGIN();
    layer_fwd<ST, 20, 8, true, STORE>(ws, L::F_L5, 1280, cat, hb, mk, lane, SINK(L::A_H0 + 80));
    MASK_STORE(5);
  }
  MASK_BEGIN();
  layer_fwd<ST, 16, 8, true, STORE>(ws, L::F_L6, 1536, hb, ha, mk, lane, SINK(L::A_H0 + 96));
  MASK_STORE(6);
  MASK_BEGIN();
  layer_fwd<ST, 16, 8, true, STORE>(ws, L::F_L7, 1792, ha, hb, mk, lane, SINK(L::A_H0 + 112));
  MASK_STORE(7);
  // feature (no activation) and alpha (row 0 of a ninth tile)   models/NeRF.py:229-231
  layer_fwd<ST, 16, 8, false, false>(ws, L::F_FA, L::BI_FEAT, hb, ha, mk, lane, SINK(L::A_FEAT));
  float alpha[ST];
  {
    f32x16 acc[ST];
    acc_init_bias(acc[0], ws, L::BI_ALPHA, h);
#pragma unroll
    for (int t = 1; t < ST; ++t) acc[t] = acc[0];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const bf16x8 wa = next_frag(ws, L::F_FA + 128 + ks, lane);
#pragma unroll
      for (int t = 0; t < ST; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, hb[t][ks], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < ST; ++t) alpha[t] = acc[t][0];
  }
  // view branch: relu(Linear([feature, input_dir]))  then rgb   models/NeRF.py:232-238
  bf16x8 hd[ST][8];
  {
    bf16x8 cat[ST][18];
#pragma unroll
    for (int t = 0; t < ST; ++t) {
#pragma unroll
      for (int k = 0; k < 16; ++k) cat[t][k] = ha[t][k];
      if (STORE && is_ring<WS>::value) {
        dpe[t][0] = *frag_ptr(a.acts, tile0 + t, a.astride, L::A_DPE, r, h);
        dpe[t][1] = *frag_ptr(a.acts, tile0 + t, a.astride, L::A_DPE + 1, r, h);
      }
      cat[t][16] = dpe[t][0]; cat[t][17] = dpe[t][1];
    }
    MASK_BEGIN();
    layer_fwd<ST, 18, 4, true, STORE>(ws, L::F_DIR, L::BI_DIR, cat, hd, mk, lane, SINK(L::A_HD));
    MASK_STORE(8);
  }
  {
    f32x16 acc[ST];
    acc_init_bias(acc[0], ws, L::BI_RGB, h);
#pragma unroll
    for (int t = 1; t < ST; ++t) acc[t] = acc[0];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 wr = next_frag(ws, L::F_RGB + ks, lane);
#pragma unroll
      for (int t = 0; t < ST; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr, hd[t][ks], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < ST; ++t) {
      const int64_t m = (tile0 + t) * 32 + r;
      if (h == 0 && tile0 + t < ntiles && m < a.M) {
        float4 o; o.x = acc[t][0]; o.y = acc[t][1]; o.z = acc[t][2]; o.w = alpha[t];     // [rgb, alpha] raw
        *reinterpret_cast<float4*>(a.out + m * 4) = poison_raw(o, bad[t]);
      }
    }
  }
#undef store
#undef SINK
#undef MASK_BEGIN
#undef MASK_STORE
}

// variants 1 / 2: 4 independent waves per workgroup, weights through L1
template <int ST, int MODE, bool STORE>
__global__ void __launch_bounds__(256, (ST == 1 ? 2 : 1)) mlp_fwd_kernel(FwdArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wv) * ST;
  const int64_t ntiles = (a.M + 31) >> 5;
  if (tile0 >= ntiles) return;
  GlobalW ws{a.wf + lane, a.bias};
  fwd_tiles<ST, MODE, STORE>(a, ws, tile0, ntiles, lane);
}


// variant 3: persistent workgroups of 8 waves x 32 samples, weights through the shared LDS ring
constexpr int F_CHUNKS = L::F_TOTAL / RING_CHUNK;     // 37
static_assert(F_CHUNKS * RING_CHUNK == L::F_TOTAL, "forward stream must be whole chunks");
constexpr int SPLIT_NW = 4, SPLIT_CHUNK = 16;         // the two-workgroups-per-CU form of the training kernels
constexpr int SPLIT_LDS_BYTES = RING_STAGES * SPLIT_CHUNK * 1024 + 2560 * 4;      // 75 776 B: two fit the 160 KiB of a CU

#ifdef NERF_CLOCK_STAMP
// Diagnostic build only (tools/probe_clock.py): in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
// (MI355X_MICROARCH "DVFS" item 6).  Stamps go to a buffer of their own; no output depends on them.
__device__ unsigned long long g_stamps[4096][4];
#define NERF_STAMP_BEGIN() const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime(); unsigned long long st_passes = 0
#define NERF_STAMP_PASS() ++st_passes
#define NERF_STAMP_END() do { if (threadIdx.x == 0 && blockIdx.x < 4096) { g_stamps[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - st_t0; \
  g_stamps[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - st_r0; g_stamps[blockIdx.x][2] = st_passes; g_stamps[blockIdx.x][3] = 1; } } while (0)
#else
#define NERF_STAMP_BEGIN()
#define NERF_STAMP_PASS()
#define NERF_STAMP_END()
#endif
// NW = 8, CHUNK = 32: ONE persistent workgroup of 8 waves per CU behind one 128 KiB ring (all 8 waves in lockstep).
// NW = 4, CHUNK = 16: TWO independent workgroups of 4 waves per CU, each behind its own 64 KiB ring: the two waves of a
// SIMD then belong to different workgroups and are not tied to one ring barrier, so the fragment stores of one group
// (the CU's store path takes ~29 cycles per KiB, and a wave blocked on a store issues no MFMA) can run under the MFMAs
// of the other; the price is each group streaming the whole weight image for 128 instead of 256 samples.
template <int MODE, bool STORE, int NW = 8, int CHUNK = RING_CHUNK>
__global__ void __launch_bounds__(64 * NW, 2) mlp_fwd_ring_kernel(FwdArgs a) {
  static_assert(L::F_TOTAL % CHUNK == 0, "forward stream must be whole chunks");
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ntiles = (a.M + 31) >> 5;
  const int64_t nsuper = (ntiles + NW - 1) / NW;
  typedef RingW<L::F_TOTAL / CHUNK, L::F_TOTAL, (STORE ? 2 : 4), NW, CHUNK> Ring;
  Ring ws;
  ws.wsrc = reinterpret_cast<const char*>(a.wf);
  ws.lane16 = 16 * lane;
  ws.lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(ring_smem));
  ws.wv = wv;
  ws.start(lane);
  ring_load_bias(a.bias, L::BI_TOTAL, Ring::BIAS_OFF);
  __syncthreads();
  NERF_STAMP_BEGIN();
  PassQueue pq;                                         // dynamic pass queue (mlp_ring.h); a.queue == nullptr: static split
  pq.init(a.queue, ws.lds0 + Ring::BIAS_OFF);
  for (int64_t sp = blockIdx.x; sp < nsuper;) {
    int ln = lane;
    asm volatile("" : "+v"(ln));          // lane-derived values are recomputed per pass, not hoisted and spilled
    ws.new_pass();
    fwd_tiles<1, MODE, STORE>(a, ws, sp * NW + wv, ntiles, ln, &pq);
    NERF_STAMP_PASS();
    sp = pq.next(sp);
  }
  NERF_STAMP_END();
  ws.drain();                                           // the ring always runs 3 chunks ahead
  pq.leave();
}


// ==========================================================================================
// Variant 4 (inference only): the same register-resident chain on v_mfma_f32_16x16x32_bf16.
// The guide measures a higher held clock for this shape (MI355X_MICROARCH "DVFS give-back" item 7).  A wave still
// carries 32 samples, as two 16-sample column tiles that share every A fragment (16 output features x 32 k).
// Accumulator: col = lane&15 (sample), row = 4 (lane>>4) + reg; two stacked 16-feature tiles give the next layer's
// B fragment, element j of lane group g = feature 16 (j>>2) + 4 g + (j&3) of the 32-feature k-step.
// ==========================================================================================
// L16 stream offsets, kperm16, pos_chan16 / dir_chan16, fwd_src16: mlp_layout.h
// the whole bf16 image in ONE launch (was two: at N_rand = 1024 the ~30 five-microsecond launches of an iteration are a tenth
// of it): blocks [0, PACK_BLOCKS) build the 32x32x16 forward / backward streams and the bias slots, the rest the 16x16x32
// forward stream of the render kernels
constexpr int PACK_BLOCKS = (PACK_THREADS + 255) / 256, PACK16_BLOCKS = L::F16_PADDED * 64 / 256;
__global__ void __launch_bounds__(256) pack_kernel(const float* __restrict__ p, bf16x8* __restrict__ wf,
                                                   bf16x8* __restrict__ wb, float* __restrict__ bias,
                                                   bf16x8* __restrict__ w16) {
  if (blockIdx.x < PACK_BLOCKS) {
    pack_part(blockIdx.x * 256 + threadIdx.x, p, wf, wb, bias);
    return;
  }
  const int t = (blockIdx.x - PACK_BLOCKS) * 256 + threadIdx.x;
  if (t >= L::F16_PADDED * 64) return;
  const int f = t >> 6, lane = t & 63;
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (__bf16)fwd_src16(p, f, lane & 15, lane >> 4, j);
  w16[t] = v;
}

constexpr int RING16_LDS_BYTES = RING_LDS_BYTES;
// per-lane frequencies of the channel order above (pass-invariant; recomputed per pass, cheaper than keeping them)
struct Pe16 {
  float fa, fb, fc, phc;      // position: bands 2g, 2g+1, 8 + (g>>1); phase of the third (0 sin, 1/4 rev cos)
  float fd;                   // direction: band g
  int g;
};
__device__ __forceinline__ Pe16 pe16_setup(const PeFreq& fr, int g) {
  Pe16 q;
  q.fa = g == 0 ? fr.pos[0] : g == 1 ? fr.pos[2] : g == 2 ? fr.pos[4] : fr.pos[6];
  q.fb = g == 0 ? fr.pos[1] : g == 1 ? fr.pos[3] : g == 2 ? fr.pos[5] : fr.pos[7];
  q.fc = g < 2 ? fr.pos[8] : fr.pos[9];
  q.phc = (g & 1) ? 0.25f : 0.0f;
  q.fd = g == 0 ? fr.dir[0] : g == 1 ? fr.dir[1] : g == 2 ? fr.dir[2] : fr.dir[3];
  q.g = g;
  return q;
}
// sin(x f + phase [rev] 2 pi): x f is the reference's fp32 product; v_sin_f32 takes revolutions
__device__ __forceinline__ float sin_rev(float x, float f, float ph) {
  return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf((x * f) * 0.15915494309189535f + ph));
}
__device__ __forceinline__ void pe16_pos(const Pe16& q, const float (&x)[3], bf16x8& k0, bf16x8& k1) {
  float v[16];
#pragma unroll
  for (int sl = 0; sl < 12; ++sl) v[sl] = sin_rev(x[sl % 3], sl < 6 ? q.fa : q.fb, (sl % 6) >= 3 ? 0.25f : 0.0f);
#pragma unroll
  for (int d = 0; d < 3; ++d) v[12 + d] = sin_rev(x[d], q.fc, q.phc);
  v[15] = q.g == 0 ? x[0] : q.g == 1 ? x[1] : q.g == 2 ? x[2] : 0.0f;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const bf16x2 a = pack2(v[j], v[j + 1]), b = pack2(v[8 + j], v[8 + j + 1]);
    k0[j] = a[0]; k0[j + 1] = a[1]; k1[j] = b[0]; k1[j + 1] = b[1];
  }
}
__device__ __forceinline__ bf16x8 pe16_dir(const Pe16& q, const float (&x)[3]) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 6; ++j) v[j] = sin_rev(x[j % 3], q.fd, j >= 3 ? 0.25f : 0.0f);
  v[6] = q.g == 0 ? x[0] : q.g == 1 ? x[1] : q.g == 2 ? x[2] : 0.0f;
  v[7] = 0.0f;
  bf16x8 k;
#pragma unroll
  for (int j = 0; j < 8; j += 2) { const bf16x2 a = pack2(v[j], v[j + 1]); k[j] = a[0]; k[j + 1] = a[1]; }
  return k;
}

#ifndef NERF_BF16_IGLP
#define NERF_BF16_IGLP 0      // __builtin_amdgcn_iglp_opt strategy of the bf16 render layers (as in mlp22.hip: 4.79 -> 4.71 ms per fine pass); -1: none
#endif
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
// out[s][nt>>1] (elements 4 (nt&1) + i) = act( W[16-row tile nt] . in[s] + bias ), s = 0..NS-1 sample tiles of 16.
// hook(nt * KS + ks) runs behind the MFMAs of k-step (nt, ks): a place to put independent VALU work (the NEXT pass's
// positional encoding) in the shadow of the matrix pipe; the slot index is a constant at every call site after unrolling.
template <int NS, int KS, int NT, bool RELU, class WS, class HOOK = NoHook>
__device__ __forceinline__ void layer_fwd16(WS& ws, int fbase, int bias_slot, const bf16x8 (&in)[NS][KS],
                                            bf16x8 (&out)[NS][NT / 2], int lane, const HOOK& hook = HOOK()) {
  const int g = lane >> 4;
  f32x4 prev[NS];
#if NERF_BF16_IGLP >= 0
  __builtin_amdgcn_iglp_opt(NERF_BF16_IGLP);
#endif
  // epilogue of a finished tile, run under the next tile's MFMAs.  The empty asm pins it there: without it hipcc
  // reads the accumulators right behind their last MFMA (s_nop 6 in both waves of the SIMD at once).  ReLU after
  // the bf16 rounding, as a packed 16-bit integer max on the bit patterns: one VALU op per two values.
  auto finish = [&](int nt, f32x4 (&a)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(a[s]));
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        const bf16x2 pr = RELU ? relu_pack(a[s][i], a[s][i + 1]) : pack2(a[s][i], a[s][i + 1]);
        out[s][nt >> 1][4 * (nt & 1) + i] = pr[0];
        out[s][nt >> 1][4 * (nt & 1) + i + 1] = pr[1];
      }
  };
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float4 b = ws.bias4(bias_slot + 16 * nt + 4 * g);
    f32x4 acc[NS];
    acc[0][0] = b.x; acc[0][1] = b.y; acc[0][2] = b.z; acc[0][3] = b.w;
#pragma unroll
    for (int s = 1; s < NS; ++s) acc[s] = acc[0];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 a = next_frag(ws, fbase + nt * KS + ks, lane);
      if (nt > 0 && ks == KS / 2) finish(nt - 1, prev);      // previous tile's epilogue, half a tile of MFMAs later
#pragma unroll
      for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, in[s][ks], acc[s], 0, 0, 0);
      hook(nt * KS + ks);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) prev[s] = acc[s];
  }
  finish(NT - 1, prev);
}

// 1 = compute the NEXT pass's encodings pair by pair behind the MFMAs of pos6 (ray loads) / pos7 (arithmetic) instead of
// at the top of the pass, where the arithmetic of all 8 lock-stepped waves leaves the matrix pipe idle (4.3 % of the
// pass's cycles by ablation).  Measured round 2 (profiles/r02_pe_hoist_ab.log, parity tests green with it): 8 waves x 32
// samples 4.66 ms with and without, 4 waves x 64 samples 4.92 vs 4.82 ms (worse) -- the kernel is power-limited, the
// cycles saved come back as a lower clock, and the hoist costs 28 VGPRs.  Kept as a negative result, off.
#ifndef NERF_PE_HOIST
#define NERF_PE_HOIST 0
#endif
#if NERF_ABLATE == 3          // the encoding-free timing build computes its placeholder at the top of the pass
#undef NERF_PE_HOIST
#define NERF_PE_HOIST 0
#endif
// The encodings of one wave's NS x 16 samples, and the raw ray data they are made from
template <int NS> struct PeRegs { bf16x8 pe[NS][2], dpe[NS][1]; };
template <int NS> struct PeRaw { float o[NS][3], d[NS][3], v[NS][3], z[NS]; };
template <int NS>
__device__ __forceinline__ void pe16_load(const FwdArgs& a, int64_t wtile, int c, PeRaw<NS>& r) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    int64_t m = wtile * (16 * NS) + 16 * s + c; if (m >= a.M) m = a.M - 1;
    const int64_t ray = (int64_t)((unsigned)m / (unsigned)a.n);
    const float* rr = a.rays + ray * NERF_RAY_STRIDE;
    r.z[s] = a.z[m];
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.o[s][k] = rr[k]; r.d[s][k] = rr[3 + k]; r.v[s][k] = rr[8 + k]; }
  }
}
// pair j of one sample: j 0..7 = position slots (2 j, 2 j + 1) -> pe[j >> 2] elements 2 (j & 3), +1; j 8..11 = direction
template <int NS>
__device__ __forceinline__ void pe16_pair(const Pe16& q, const float (&p)[3], const float (&d)[3], int j, PeRegs<NS>& out, int s) {
  auto slot = [&](int sl) -> float {              // position slot sl of this lane group (see pos_chan16)
    if (sl < 12) return sin_rev(p[sl % 3], sl < 6 ? q.fa : q.fb, (sl % 6) >= 3 ? 0.25f : 0.0f);
    if (sl < 15) return sin_rev(p[sl - 12], q.fc, q.phc);
    return q.g == 0 ? p[0] : q.g == 1 ? p[1] : q.g == 2 ? p[2] : 0.0f;
  };
  auto dslot = [&](int sl) -> float {             // direction slot (see dir_chan16)
    if (sl < 6) return sin_rev(d[sl % 3], q.fd, sl >= 3 ? 0.25f : 0.0f);
    if (sl == 6) return q.g == 0 ? d[0] : q.g == 1 ? d[1] : q.g == 2 ? d[2] : 0.0f;
    return 0.0f;
  };
  if (j < 8) {
    const bf16x2 pr = pack2(slot(2 * j), slot(2 * j + 1));
    out.pe[s][j >> 2][2 * (j & 3)] = pr[0]; out.pe[s][j >> 2][2 * (j & 3) + 1] = pr[1];
  } else {
    const bf16x2 pr = pack2(dslot(2 * (j - 8)), dslot(2 * (j - 8) + 1));
    out.dpe[s][0][2 * (j - 8)] = pr[0]; out.dpe[s][0][2 * (j - 8) + 1] = pr[1];
  }
}
template <int NS>
__device__ __forceinline__ void pe16_all(const FwdArgs& a, const Pe16& pq, int64_t wtile, int c, PeRegs<NS>& out) {
  PeRaw<NS> r;
  pe16_load<NS>(a, wtile, c, r);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    float p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = r.o[s][k] + r.z[s] * r.d[s][k];          // render.py:142
#if NERF_ABLATE == 3          // timing-only build 3: no positional-encoding arithmetic
    bf16x8 cst;
    for (int j = 0; j < 8; ++j) cst[j] = (__bf16)(p[0] + r.v[s][0]);
    out.pe[s][0] = cst; out.pe[s][1] = cst; out.dpe[s][0] = cst;
#else
#pragma unroll
    for (int j = 0; j < 12; ++j) pe16_pair<NS>(pq, p, r.v[s], j, out, s);
#endif
  }
}

// one wave: NS x 16 samples starting at sample wtile * 16 NS.  `cur` holds this tile's encodings on entry and the NEXT
// tile's (wtile_next) on return: they are produced pair by pair behind the MFMAs of pos6 (ray loads) and pos7
// (arithmetic) -- the position fragments are dead after pos5, so they are overwritten in place; only the direction
// fragment needs a second copy until the view layer has read it.  Done at the top of a pass, the encoding arithmetic of
// all 8 waves (in lockstep behind the ring barrier) left the matrix pipe idle for 4.3 % of the pass.
template <int NS, class WS>
__device__ __forceinline__ void fwd_tiles16(const FwdArgs& a, WS& ws, int64_t wtile0, int64_t wtile_next, int64_t nwtiles,
                                            int lane, PeRegs<NS>& cur) {
  const int c = lane & 15, g = lane >> 4;
  const Pe16 pq = pe16_setup(a.fr, g);
  bf16x8 (&pe)[NS][2] = cur.pe;
  bf16x8 (&dpe)[NS][1] = cur.dpe;
  PeRaw<NS> raw;
  PeRegs<NS> nxt;                 // only nxt.dpe is used (position fragments are written into cur.pe in place)
  float pn[NS][3];
  const int64_t wt_n = wtile_next < nwtiles ? wtile_next : nwtiles - 1;
  auto hook_load = [&](int slot) {
    if (NERF_PE_HOIST && slot == 0) pe16_load<NS>(a, wt_n, c, raw);
  };
  auto hook_math = [&](int slot) {
    if (!NERF_PE_HOIST) return;
    if (slot == 0) {
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int k = 0; k < 3; ++k) pn[s][k] = raw.o[s][k] + raw.z[s] * raw.d[s][k];
    }
    // 12 NS pairs over the remaining slots of the layer (128 k-steps): one pair every 4th slot (NS <= 2) / 2nd slot
    constexpr int EVERY = NS <= 2 ? 4 : 2;
    if (slot >= 4 && (slot % EVERY) == 0) {
      const int idx = (slot - 4) / EVERY;
      if (idx < 12 * NS) {
        const int s = idx / 12, j = idx % 12;
        if (j < 8) pe16_pair<NS>(pq, pn[s], raw.v[s], j, cur, s);      // in place: pos5 was the last reader
        else pe16_pair<NS>(pq, pn[s], raw.v[s], j, nxt, s);
      }
    }
  };
  int bad[NS];                      // non-finite inputs of this lane's samples (mlp_frag.h: nonfinite_flags), used at the output store
#pragma unroll
  for (int s = 0; s < NS; ++s)
  {
    bad[s] = nonfinite_flags<16 | 32>(nonfinite_bits(pe[s][0]) | nonfinite_bits(pe[s][1]), nonfinite_bits(dpe[s][0]));
    asm volatile("" : "+v"(bad[s]));          // at the head of the pass (see fwd_tiles)
  }
  bf16x8 ha[NS][8], hb[NS][8];
  layer_fwd16<NS, 2, 16, true>(ws, L16::F_L0, 0, pe, ha, lane);
  layer_fwd16<NS, 8, 16, true>(ws, L16::F_L1 + 0 * 128, 256, ha, hb, lane);
  layer_fwd16<NS, 8, 16, true>(ws, L16::F_L1 + 1 * 128, 512, hb, ha, lane);
  layer_fwd16<NS, 8, 16, true>(ws, L16::F_L1 + 2 * 128, 768, ha, hb, lane);
  layer_fwd16<NS, 8, 16, true>(ws, L16::F_L1 + 3 * 128, 1024, hb, ha, lane);
  {
    bf16x8 cat[NS][10];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      cat[s][0] = pe[s][0]; cat[s][1] = pe[s][1];
#pragma unroll
      for (int k = 0; k < 8; ++k) cat[s][2 + k] = ha[s][k];
    }
    layer_fwd16<NS, 10, 16, true>(ws, L16::F_L5, 1280, cat, hb, lane);
  }
  layer_fwd16<NS, 8, 16, true>(ws, L16::F_L6, 1536, hb, ha, lane, hook_load);
  layer_fwd16<NS, 8, 16, true>(ws, L16::F_L7, 1792, ha, hb, lane, hook_math);
  layer_fwd16<NS, 8, 16, false>(ws, L16::F_FA, L::BI_FEAT, hb, ha, lane);
  float alpha[NS];
  {
    const float4 b = ws.bias4(L::BI_ALPHA + 4 * g);
    f32x4 acc[NS];
    acc[0][0] = b.x; acc[0][1] = b.y; acc[0][2] = b.z; acc[0][3] = b.w;
#pragma unroll
    for (int s = 1; s < NS; ++s) acc[s] = acc[0];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 wa = next_frag(ws, L16::F_FA + 128 + ks, lane);
#pragma unroll
      for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, hb[s][ks], acc[s], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) alpha[s] = acc[s][0];
  }
  bf16x8 hd[NS][4];
  {
    bf16x8 cat[NS][9];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int k = 0; k < 8; ++k) cat[s][k] = ha[s][k];
      cat[s][8] = dpe[s][0];
    }
    layer_fwd16<NS, 9, 8, true>(ws, L16::F_DIR, L::BI_DIR, cat, hd, lane);
  }
  {
    const float4 b = ws.bias4(L::BI_RGB + 4 * g);
    f32x4 acc[NS];
    acc[0][0] = b.x; acc[0][1] = b.y; acc[0][2] = b.z; acc[0][3] = b.w;
#pragma unroll
    for (int s = 1; s < NS; ++s) acc[s] = acc[0];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 wr = next_frag(ws, L16::F_RGB + ks, lane);
#pragma unroll
      for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr, hd[s][ks], acc[s], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int64_t m = wtile0 * (16 * NS) + 16 * s + c;
      if (g == 0 && wtile0 < nwtiles && m < a.M) {
        float4 o; o.x = acc[s][0]; o.y = acc[s][1]; o.z = acc[s][2]; o.w = alpha[s];
        *reinterpret_cast<float4*>(a.out + m * 4) = poison_raw(o, bad[s]);
      }
    }
  }
  if (NERF_PE_HOIST) {
#pragma unroll
    for (int s = 0; s < NS; ++s) cur.dpe[s][0] = nxt.dpe[s][0];
  }
}

// NW waves x NS sample tiles of 16 = 256 samples per workgroup pass either way: <8, 2> two waves per SIMD at 256
// registers each; <4, 4> one wave per SIMD with the whole 512-register file, half the LDS fragment reads per FLOP.
template <int NW, int NS>
__global__ void __launch_bounds__(64 * NW) mlp_fwd_ring16_kernel(FwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwtiles = (a.M + 16 * NS - 1) / (16 * NS), nsuper = (nwtiles + NW - 1) / NW;
  NERF_STAMP2_DECL();
  RingW<L16::CHUNKS, L::F16_TOTAL, 4, NW> ws;
  ws.wsrc = reinterpret_cast<const char*>(a.wf);          // points at the 16x16x32 stream
  ws.lane16 = 16 * lane;
  ws.lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(ring_smem));
  ws.wv = wv;
  ws.start(lane);
  ring_load_bias(a.bias, L::BI_TOTAL);
  __syncthreads();
  NERF_STAMP_BEGIN();
  NERF_STAMP2_LOOP();
  PeRegs<NS> cur;
  if (NERF_PE_HOIST) {
    const int64_t first = (int64_t)blockIdx.x * NW + wv;
    pe16_all<NS>(a, pe16_setup(a.fr, lane >> 4), first < nwtiles ? first : nwtiles - 1, lane & 15, cur);
  }
  PassQueue pq;                                         // dynamic pass queue (mlp_ring.h); a.queue == nullptr: static split
  pq.init(NERF_PE_HOIST ? nullptr : a.queue, ws.lds0 + RING_BIAS_OFF);      // (the hoisted-encoding build needs the next pass a pass early)
  for (int64_t sp = blockIdx.x; sp < nsuper;) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    ws.new_pass();
    if (!NERF_PE_HOIST) {
      const int64_t wt = sp * NW + wv;
      pq.ask(wv, ln);                                   // before this pass's input loads
      pe16_all<NS>(a, pe16_setup(a.fr, ln >> 4), wt < nwtiles ? wt : nwtiles - 1, ln & 15, cur);
      pq.publish(wv);
    }
    fwd_tiles16<NS>(a, ws, sp * NW + wv, (sp + gridDim.x) * NW + wv, nwtiles, ln, cur);
    NERF_STAMP_PASS();
    NERF_STAMP2_PASS();
    sp = pq.next(sp);
  }
  NERF_STAMP_END();
  NERF_STAMP2_END();
  ws.drain();
  pq.leave();
}

// ------------------------------------------------------------------------------------------
// backward chain: dZ_l for every layer (stored as fragment blocks for the dW kernel)
// ------------------------------------------------------------------------------------------
// backward epilogue quarter: ReLU' from the forward's sign bits, then bf16
template <bool MASK, int Q, int ODD>
__device__ __forceinline__ void finish_quarter_bwd_t(const f32x16& acc, bf16x8& lo, bf16x8& hi, unsigned w) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int i = 4 * Q + 2 * p;
    bf16x2 pr = pack2(acc[i], acc[i + 1]);
    if (MASK) pr = p == 0 ? keep_where<8 * ODD + 2 * Q>(pr, w) : keep_where<8 * ODD + 2 * Q + 1>(pr, w);
    if (i < 8) { lo[i] = pr[0]; lo[i + 1] = pr[1]; } else { hi[i - 8] = pr[0]; hi[i - 7] = pr[1]; }
  }
}
template <bool MASK>
__device__ __forceinline__ void finish_quarter_bwd(const f32x16& acc, int q, int kt, bf16x8& lo, bf16x8& hi,
                                                   const u32x4& mask) {
  const unsigned w = MASK ? mask[kt >> 1] : 0u;
  // q and kt are compile-time constants at every call site (fully unrolled loops): the switch folds away
  switch (2 * q + (kt & 1)) {
    case 0: finish_quarter_bwd_t<MASK, 0, 0>(acc, lo, hi, w); break;
    case 1: finish_quarter_bwd_t<MASK, 0, 1>(acc, lo, hi, w); break;
    case 2: finish_quarter_bwd_t<MASK, 1, 0>(acc, lo, hi, w); break;
    case 3: finish_quarter_bwd_t<MASK, 1, 1>(acc, lo, hi, w); break;
    case 4: finish_quarter_bwd_t<MASK, 2, 0>(acc, lo, hi, w); break;
    case 5: finish_quarter_bwd_t<MASK, 2, 1>(acc, lo, hi, w); break;
    case 6: finish_quarter_bwd_t<MASK, 3, 0>(acc, lo, hi, w); break;
    default: finish_quarter_bwd_t<MASK, 3, 1>(acc, lo, hi, w); break;
  }
}

// out[t][2 kt + s] = mask( W^T[kt-tile] . in[t] );  mask = ReLU sign bits written by the forward kernel
template <int ST, int NS, int KT, bool MASK, class WS, class SINK = NoSink>
__device__ __forceinline__ void layer_bwd(WS& ws, int fbase, const bf16x8 (&in)[ST][NS], bf16x8 (&out)[ST][2 * KT],
                                          const u32x4 (&mask)[ST], int lane, const SINK& sink = SINK()) {
  f32x16 prev[ST];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    f32x16 acc[ST];
#pragma unroll
    for (int t = 0; t < ST; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      const bf16x8 a = next_frag(ws, fbase + kt * NS + ns, lane);
#pragma unroll
      for (int t = 0; t < ST; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, in[t][ns], acc[t], 0, 0, 0);
      if (kt > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (quarter_pos(NS, q) == ns) {
#pragma unroll
            for (int t = 0; t < ST; ++t) {
              finish_quarter_bwd<MASK>(prev[t], q, kt - 1, out[t][2 * kt - 2], out[t][2 * kt - 1], mask[t]);
              if (q == 3) { sink.put(t, 2 * kt - 2, out[t][2 * kt - 2]); sink.put(t, 2 * kt - 1, out[t][2 * kt - 1]); }
            }
          }
      }
    }
#pragma unroll
    for (int t = 0; t < ST; ++t) prev[t] = acc[t];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < ST; ++t)
      finish_quarter_bwd<MASK>(prev[t], q, KT - 1, out[t][2 * KT - 2], out[t][2 * KT - 1], mask[t]);
#pragma unroll
  for (int t = 0; t < ST; ++t) { sink.put(t, 2 * KT - 2, out[t][2 * KT - 2]); sink.put(t, 2 * KT - 1, out[t][2 * KT - 1]); }
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

template <int ST, class WS>
__device__ __forceinline__ void bwd_tiles(const BwdArgs& a, WS& ws, int64_t tile0, int64_t ntiles, int lane, PassQueue* pq = nullptr) {
  const int r = lane & 31, h = lane >> 5;
  if (pq) pq->ask(ws_wave(ws), lane);          // dynamic pass queue (mlp_ring.h): before this pass's input loads
  int64_t tile[ST];
  bool live[ST];
  bf16x8 zrgb[ST][1], zal[ST][1];
#pragma unroll
  for (int t = 0; t < ST; ++t) {
    live[t] = tile0 + t < ntiles;
    tile[t] = live[t] ? tile0 + t : ntiles - 1;
    const int64_t m = tile[t] * 32 + r;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live[t] && m < a.M && h == 0) g = *reinterpret_cast<const float4*>(a.d_raw + m * 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) { zrgb[t][0][j] = (__bf16)0.0f; zal[t][0][j] = (__bf16)0.0f; }
    zrgb[t][0][0] = (__bf16)g.x; zrgb[t][0][1] = (__bf16)g.y; zrgb[t][0][2] = (__bf16)g.z;   // rows 0..2 (h == 0)
    zal[t][0][0] = (__bf16)g.w;                                                              // row 0
  }
  if (pq) pq->publish(ws_wave(ws));
  // every ReLU mask of the pass is fetched here, so the chain itself issues no loads the compiler must wait for
  u32x4 mk[9][ST];
#pragma unroll
  for (int l = 0; l < 9; ++l)
#pragma unroll
    for (int t = 0; t < ST; ++t)
      mk[l][t] = *reinterpret_cast<const u32x4*>(frag_ptr(const_cast<void*>(a.acts), tile[t], a.astride, L::A_MASK + l, r, h));
#define store(slot0, t, frags, count) \
  do { store_frags<count>(a.dz, tile0 + (t), a.zstride, slot0, frags, r, h); ws.note_stores(count); } while (0)
#pragma unroll
  for (int t = 0; t < ST; ++t) { store(L::Z_RGB, t, zrgb[t], 1); store(L::Z_A, t, zal[t], 1); }

#define ZSINK(slot0) FragSink<true>{a.dz, tile0, a.zstride, slot0, r, h}
  bf16x8 zd[ST][8];
  layer_bwd<ST, 1, 4, true>(ws, L::B_RGB, zrgb, zd, mk[8], lane, ZSINK(L::Z_D));
  bf16x8 za[ST][16], zb[ST][16];
  layer_bwd<ST, 8, 8, false>(ws, L::B_DIR, zd, za, mk[8], lane, ZSINK(L::Z_F));            // d feature
  {
    bf16x8 cat[ST][17];
#pragma unroll
    for (int t = 0; t < ST; ++t) {
#pragma unroll
      for (int k = 0; k < 16; ++k) cat[t][k] = za[t][k];
      cat[t][16] = zal[t][0];
    }
    layer_bwd<ST, 17, 8, true>(ws, L::B_FA, cat, zb, mk[7], lane, ZSINK(L::Z_L0 + 112));   // dZ7
  }
  layer_bwd<ST, 16, 8, true>(ws, L::B_L7, zb, za, mk[6], lane, ZSINK(L::Z_L0 + 96));       // dZ6
  layer_bwd<ST, 16, 8, true>(ws, L::B_L6, za, zb, mk[5], lane, ZSINK(L::Z_L0 + 80));       // dZ5
  layer_bwd<ST, 16, 8, true>(ws, L::B_L5, zb, za, mk[4], lane, ZSINK(L::Z_L0 + 64));       // dZ4
  layer_bwd<ST, 16, 8, true>(ws, L::B_L4 + 0 * 128, za, zb, mk[3], lane, ZSINK(L::Z_L0 + 48));   // dZ3
  layer_bwd<ST, 16, 8, true>(ws, L::B_L4 + 1 * 128, zb, za, mk[2], lane, ZSINK(L::Z_L0 + 32));   // dZ2
  layer_bwd<ST, 16, 8, true>(ws, L::B_L4 + 2 * 128, za, zb, mk[1], lane, ZSINK(L::Z_L0 + 16));   // dZ1
  layer_bwd<ST, 16, 8, true>(ws, L::B_L4 + 3 * 128, zb, za, mk[0], lane, ZSINK(L::Z_L0 + 0));    // dZ0
#undef store
#undef ZSINK
}

template <int ST>
__global__ void __launch_bounds__(256, (ST == 1 ? 2 : 1)) mlp_bwd_kernel(BwdArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wv) * ST;
  const int64_t ntiles = (a.M + 31) >> 5;
  if (tile0 >= ntiles) return;
  GlobalW ws{a.wb + lane, nullptr};
  bwd_tiles<ST>(a, ws, tile0, ntiles, lane);
}

// the transposed stream is padded with zero fragments to whole chunks (1100 -> 1120 = 35 x 32 = 70 x 16)
template <int NW = 8, int CHUNK = RING_CHUNK>
__global__ void __launch_bounds__(64 * NW, 2) mlp_bwd_ring_kernel(BwdArgs a) {
  static_assert(L::B_PADDED % CHUNK == 0, "padded backward stream must be whole chunks");
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ntiles = (a.M + 31) >> 5;
  const int64_t nsuper = (ntiles + NW - 1) / NW;
  RingW<(L::B_TOTAL + CHUNK - 1) / CHUNK, L::B_TOTAL, 4, NW, CHUNK> ws;
  ws.wsrc = reinterpret_cast<const char*>(a.wb);
  ws.lane16 = 16 * lane;
  ws.lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(ring_smem));
  ws.wv = wv;
  ws.start(lane);
  PassQueue pq;                                         // dynamic pass queue (mlp_ring.h); a.queue == nullptr: static split
  pq.init(a.queue, ws.lds0 + decltype(ws)::BIAS_OFF);
  for (int64_t sp = blockIdx.x; sp < nsuper;) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    ws.new_pass();
    bwd_tiles<1>(a, ws, sp * NW + wv, ntiles, ln, &pq);
    // the pass ends inside the last chunk (1100 is not a multiple of the chunk): nothing else to do, the next pass starts
    // at a chunk boundary again because fragment indices restart at 0
    sp = pq.next(sp);
  }
  ws.drain();
  pq.leave();
}

// ------------------------------------------------------------------------------------------
// dW: split-K GEMMs  dW[n][k] = sum_m dZ[n][m] H[k][m]  over fragment blocks, samples = MFMA K
// ------------------------------------------------------------------------------------------

constexpr int DW_FRAG_STRIDE = 1152;                 // 1 KiB + 128 B: neighbouring fragments hit disjoint banks
constexpr int DW_STAGE_BYTES = 32 * DW_FRAG_STRIDE;  // 16 dZ + 16 act fragments per 32-sample tile
constexpr int DW_STAGES = 4;                         // LDS-DMA ring: 3 tiles in flight behind the one being consumed
constexpr int DW_LDS_BYTES = DW_STAGES * DW_STAGE_BYTES + 1024;   // + 1 KiB sink for padding DMAs

__device__ __forceinline__ bf16x8 tr_frag(const char* lds_frag, int u, int hq, int i16) {
  // A/B operand of v_mfma_f32_32x32x16_bf16 with K = samples 16u + 8hq + (0..7) and row/col = feature (natural
  // order).  Lane i16 = 4q + p of a 16-lane group addresses sample row q, feature piece p (h_src = p&1, j half = p>>1).
  const int q = i16 >> 2, p = i16 & 3;
  const char* base = lds_frag + 16 * (p & 1) + 8 * (p >> 1) + 32 * (16 * u + 8 * hq + q);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 128));
  union { struct { s16x4 a, b; } s; bf16x8 v; } cvt;
  cvt.s.a = lo; cvt.s.b = hi;
  return cvt.v;
}

#ifndef NERF_DW_WAVES
#define NERF_DW_WAVES 16
#endif
constexpr int DW_WAVES = NERF_DW_WAVES;          // 8: 4 x 2 output tiles per wave; 16: 2 x 2 (half the accumulators, 4 waves per SIMD)
constexpr int DW_NPW = 32 / DW_WAVES;            // n-tiles per wave = DMAs per wave per tile
__global__ void __launch_bounds__(64 * DW_WAVES) mlp_dw_kernel(DwArgs a) {
  char* smem = ring_smem;                             // the one dynamic-LDS array of this file
  int bj = blockIdx.x, job_id = 0;
  while (bj >= a.splits[job_id]) { bj -= a.splits[job_id]; ++job_id; }
  const DwJob jb = a.jobs[job_id];
  const int tile_lo = (int)((int64_t)a.ntiles * bj / a.splits[job_id]);
  const int tile_hi = (int)((int64_t)a.ntiles * (bj + 1) / a.splits[job_id]);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_tiles = (jb.nf + 1) >> 1, k_tiles = (jb.kf + 1) >> 1;
  // wave (wr, wc) of the 4 x 4 wave grid.  A wave runs on SIMD wv % 4: with wr = wv / 4 the waves of one wave COLUMN share a SIMD,
  // and a job with k_tiles <= 2 (pos0, pos5 | PE, dir0 | dirPE: only column 0 has work) puts all its MFMAs on SIMD 0 while three
  // SIMDs idle.  Such jobs use the transposed numbering (round 5): their active waves are wv = 0..3, one per SIMD.
  const bool col_major = DW_WAVES == 16 && k_tiles <= 2;
  const int wr = col_major ? (wv & 3) : (wv >> 2), wc = col_major ? (wv >> 2) : (wv & 3);
  const int nf_pad = n_tiles * 2;
  const int nfk = jb.nf + jb.kf;                        // real fragments per sample tile (<= 32)
  // this wave's output tiles: n-tiles wr*DW_NPW + (0..DW_NPW-1), k-tiles wc*2 + (0..1)
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
  const bf16x8* dzp = reinterpret_cast<const bf16x8*>(a.dz) + lane;
  const bf16x8* acp = reinterpret_cast<const bf16x8*>(a.acts) + lane;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  const unsigned sink = lds0 + DW_STAGES * DW_STAGE_BYTES;

  // odd fragment counts (rgb / alpha jobs: nf = 1): the partner fragment of the tile is never written by DMA
  if (jb.nf & 1) {
    for (int c = tid; c < DW_STAGES * 64; c += 64 * DW_WAVES) {
      bf16x8 zv;
#pragma unroll
      for (int j = 0; j < 8; ++j) zv[j] = (__bf16)0.0f;
      *reinterpret_cast<bf16x8*>(smem + (c >> 6) * DW_STAGE_BYTES + jb.nf * DW_FRAG_STRIDE + (c & 63) * 16) = zv;
    }
  }
  // every wave issues exactly DW_NPW DMAs per tile so that vmcnt arithmetic is uniform: fragment ids wv + DW_WAVES k
  auto issue = [&](int tile, int stage) {
    const unsigned st = lds0 + __builtin_amdgcn_readfirstlane(stage) * DW_STAGE_BYTES;
#pragma unroll
    for (int k = 0; k < DW_NPW; ++k) {
      const int i = wv + DW_WAVES * k;
      if (i < jb.nf) dma_frag_nt(dzp + (int64_t)tile * a.zstride + (jb.dz_slot + i) * 64, st + i * DW_FRAG_STRIDE);
      else if (i < nfk) dma_frag_nt(acp + (int64_t)tile * a.astride + (jb.act_slot + (i - jb.nf)) * 64,
                                 st + (nf_pad + i - jb.nf) * DW_FRAG_STRIDE);
      else dma_frag(dzp + (int64_t)tile * a.zstride + jb.dz_slot * 64, sink);      // padding: L2 hit, result unused
    }
  };
  __syncthreads();                                      // zero fill visible before any stage is consumed
#pragma unroll
  for (int s_ = 0; s_ < DW_STAGES - 1; ++s_)
    if (tile_lo + s_ < tile_hi) issue(tile_lo + s_, s_);

  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int rem = tile_hi - 1 - tile;                 // tiles issued after this one and still in flight (<= 2)
    // lgkmcnt(0): this wave's transposed reads of the previous tile are complete before the barrier lets another
    // wave's DMA refill that stage (hipcc may sink register-only MFMAs, and their waits, below an asm statement)
    if (DW_NPW == 4) {
      if (rem >= 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else if (rem == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
      if (rem >= 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else if (rem == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                       // tile landed for every wave; stage (tile-1)%4 is free
    if (tile + DW_STAGES - 1 < tile_hi) issue(tile + DW_STAGES - 1, (tile - tile_lo + DW_STAGES - 1) % DW_STAGES);
    const char* st = smem + ((tile - tile_lo) % DW_STAGES) * DW_STAGE_BYTES;
#if NERF_ABLATE == 4          // timing-only: loads and synchronisation only
    if (false) {
#else
    if (active) {
#endif
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bf16x8 bfr[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int kt = wc * 2 + k;
          const int f = nf_pad + (kt < k_tiles ? 2 * kt : 0) + fsel;
          bfr[k] = tr_frag(st + f * DW_FRAG_STRIDE, u, hq, i16);
        }
#pragma unroll
        for (int i = 0; i < DW_NPW; ++i) {
          const int nt = wr * DW_NPW + i;
          const int f = (nt < n_tiles ? 2 * nt : 0) + fsel;
          const bf16x8 afr = tr_frag(st + f * DW_FRAG_STRIDE, u, hq, i16);
#if NERF_ABLATE != 5          // timing-only build 5: no bias row sums
          if (wc == 0) {                       // bias gradient = row sums of dZ: v_dot2c_f32_bf16 against (1, 1)
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              const bf16x2_t pr = {afr[j], afr[j + 1]};
              bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, bsum[i], false);
            }
          }
#endif
#pragma unroll
          for (int k = 0; k < 2; ++k) acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr[k], acc[i][k], 0, 0, 0);
        }
      }
    }
  }
  if (!active) return;
#if NERF_ABLATE == 6          // timing-only build 6: no epilogue
  if (a.ntiles > 0) return;
#endif
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

// grads[w_off + n ldw + col0 + k] = sum over the job's splits of its partial tiles, in split order (deterministic);
// blockIdx.y = job, one thread per output element (weights first, then the bias row of jobs that own one).  The S loads
// of an element are independent (issued eight at a time), only the adds are a chain.
__global__ void __launch_bounds__(256) mlp_dw_reduce_kernel(DwArgs a) {
  const int j = blockIdx.y;
  const DwJob jb = a.jobs[j];
  int first = 0;
  for (int q = 0; q < j; ++q) first += a.splits[q];
  const int S = a.splits[j];
  const int nw = jb.n_valid * jb.k_valid, nb = jb.b_off >= 0 ? jb.n_valid : 0;
  const float* base = a.partial + (size_t)first * DW_SLOT_FLOATS;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < nw + nb; t += gridDim.x * 256) {
    const float* src;
    float* dst;
    if (t < nw) {
      const int n = t / jb.k_valid, k = t - n * jb.k_valid;
      src = base + (8 * (n >> 5) + (k >> 5)) * 1024 + (n & 31) * 32 + (k & 31);
      dst = a.grads + jb.w_off + (int64_t)n * jb.ldw + jb.col0 + k;
    } else {
      src = base + 64 * 1024 + (t - nw);
      dst = a.grads + jb.b_off + (t - nw);
    }
    float sum = 0.0f;
    int sidx = 0;
    for (; sidx + 8 <= S; sidx += 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load(src + (size_t)(sidx + q) * DW_SLOT_FLOATS);
#pragma unroll
      for (int q = 0; q < 8; ++q) sum += v[q];
    }
    for (; sidx < S; ++sidx) sum += __builtin_nontemporal_load(src + (size_t)sidx * DW_SLOT_FLOATS);
    *dst = sum;
  }
}


// ==========================================================================================
// Second architecture: the no-view-direction model of the reference's 2-D image fitting
// (entrypoints/__viser_image_learning.py:198-208: NeRF(channel_input=40, channel_input_views=0, channel_output=3,
// is_use_view_directions=False)): pos0 [256x40] pos1..4 [256x256] pos5 [256x296] pos6 pos7 output [out_ch x 256]
// (models/NeRF.py:182-197,241).  Same register-resident transposed scheme, same ring, same dW kernel.
// ==========================================================================================

__global__ void __launch_bounds__(256) pack_img_kernel(const float* __restrict__ p, bf16x8* __restrict__ wf,
                                                       bf16x8* __restrict__ wb, float* __restrict__ bias, int out_ch) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int nf = LI::F_TOTAL * 64, nb = LI::B_PADDED * 64;
  if (tid < nf + nb) {
    const bool fw = tid < nf;
    const int t = fw ? tid : tid - nf;
    const int f = t >> 6, lane = t & 63, r = lane & 31, h = lane >> 5;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = (__bf16)(fw ? fwd_src_img(p, f, r, h, j, out_ch) : (f < LI::B_TOTAL ? bwd_src_img(p, f, r, h, j, out_ch) : 0.0f));
    (fw ? wf : wb)[t] = v;
  } else if (tid < nf + nb + LI::BI_TOTAL) {
    const int s = tid - nf - nb;
    bias[s] = s < 2048 ? p[LI::pb(s >> 8) + (s & 255)] : ((s - LI::BI_OUT) < out_ch ? p[LI::P_WO + out_ch * 256 + (s - LI::BI_OUT)] : 0.0f);
  }
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

template <bool STORE, class WS>
__device__ __forceinline__ void fwd_tiles_img(const ImgArgs& a, WS& ws, int64_t tile0, int64_t ntiles, int lane) {
  constexpr int ST = 1;
  const int r = lane & 31, h = lane >> 5;
  int64_t tile = tile0 < ntiles ? tile0 : ntiles - 1;
  int64_t m = tile * 32 + r; if (m >= a.M) m = a.M - 1;
  bf16x8 xin[ST][3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) xin[0][ks] = row_frag(a.x + m * LI::CIN, ks, h, LI::CIN);
  int bad = nonfinite_flags<32>(nonfinite_bits(xin[0][0]) | nonfinite_bits(xin[0][1]) | nonfinite_bits(xin[0][2]), 0u);
  asm volatile("" : "+v"(bad));               // at the head of the pass (see fwd_tiles)
  if (STORE) { store_frags<3>(a.acts, tile0, a.astride, LI::A_X, xin[0], r, h); ws.note_stores(3); }
  bf16x8 ha[ST][16], hb[ST][16];
  u32x4 mk[ST];
#define IMG_LAYER(KS, FB, BS, IN, OUT, LYR)                                                        \
  do { mk[0] = u32x4{0u, 0u, 0u, 0u};                                                              \
       layer_fwd<ST, KS, 8, true, STORE>(ws, FB, BS, IN, OUT, mk, lane);                           \
       if (STORE) { store_frags<16>(a.acts, tile0, a.astride, LI::A_H0 + 16 * (LYR), OUT[0], r, h); \
                    *reinterpret_cast<u32x4*>(frag_ptr(a.acts, tile0, a.astride, LI::A_MASK + (LYR), r, h)) = mk[0]; ws.note_stores(17); } } while (0)
  IMG_LAYER(3, LI::F_L0, 0, xin, ha, 0);
  IMG_LAYER(16, LI::F_L1 + 0 * 128, 256, ha, hb, 1);
  IMG_LAYER(16, LI::F_L1 + 1 * 128, 512, hb, ha, 2);
  IMG_LAYER(16, LI::F_L1 + 2 * 128, 768, ha, hb, 3);
  IMG_LAYER(16, LI::F_L1 + 3 * 128, 1024, hb, ha, 4);
  {
    bf16x8 cat[ST][19];
    if (STORE) {
#pragma unroll
      for (int k = 0; k < 3; ++k) xin[0][k] = *frag_ptr(a.acts, tile0, a.astride, LI::A_X + k, r, h);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) cat[0][k] = xin[0][k];
#pragma unroll
    for (int k = 0; k < 16; ++k) cat[0][3 + k] = ha[0][k];
    IMG_LAYER(19, LI::F_L5, 1280, cat, hb, 5);           // concat [input_pos, h]  models/NeRF.py:224-225
  }
  IMG_LAYER(16, LI::F_L6, 1536, hb, ha, 6);
  IMG_LAYER(16, LI::F_L7, 1792, ha, hb, 7);
#undef IMG_LAYER
  f32x16 acc;
  acc_init_bias(acc, ws, LI::BI_OUT, h);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const bf16x8 wo = next_frag(ws, LI::F_OUT + ks, lane);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wo, hb[0][ks], acc, 0, 0, 0);
  }
  const int64_t mo = tile0 * 32 + r;
  if (h == 0 && tile0 < ntiles && mo < a.M) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < a.out_ch) a.out[mo * a.out_ch + c] = bad ? __builtin_nanf("") : acc[c];      // output_linear (models/NeRF.py:241); NaN / Inf inputs: mlp_frag.h
  }
}

template <bool STORE>
__global__ void __launch_bounds__(512, 2) mlp_img_fwd_ring_kernel(ImgArgs a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ntiles = (a.M + 31) >> 5, nsuper = (ntiles + 7) >> 3;
  RingW<LI::F_CHUNKS, LI::F_TOTAL, (STORE ? 2 : 4)> ws;
  ws.wsrc = reinterpret_cast<const char*>(a.wf);
  ws.lane16 = 16 * lane;
  ws.lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(ring_smem));
  ws.wv = wv;
  ws.start(lane);
  ring_load_bias(a.bias, LI::BI_TOTAL);
  __syncthreads();
  for (int64_t sp = blockIdx.x; sp < nsuper; sp += gridDim.x) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    ws.new_pass();
    fwd_tiles_img<STORE>(a, ws, sp * 8 + wv, ntiles, ln);
  }
  ws.drain();
}

__global__ void __launch_bounds__(512, 2) mlp_img_bwd_ring_kernel(ImgArgs a) {
  constexpr int ST = 1;
  const int lane0 = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ntiles = (a.M + 31) >> 5, nsuper = (ntiles + 7) >> 3;
  RingW<LI::B_CHUNKS, LI::B_TOTAL> ws;
  ws.wsrc = reinterpret_cast<const char*>(a.wb);
  ws.lane16 = 16 * lane0;
  ws.lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(ring_smem));
  ws.wv = wv;
  ws.start(lane0);
  for (int64_t sp = blockIdx.x; sp < nsuper; sp += gridDim.x) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    ws.new_pass();
    const int r = lane & 31, h = lane >> 5;
    const int64_t tile0 = sp * 8 + wv;
    const int64_t tile = tile0 < ntiles ? tile0 : ntiles - 1;
    const int64_t m = tile * 32 + r;
    bf16x8 zo[ST][1];
#pragma unroll
    for (int j = 0; j < 8; ++j) zo[0][0][j] = (__bf16)0.0f;
    if (tile0 < ntiles && m < a.M && h == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < a.out_ch) zo[0][0][c] = (__bf16)a.d_out[m * a.out_ch + c];      // rows 0..out_ch-1
    }
    u32x4 mk[8][ST];
#pragma unroll
    for (int l = 0; l < 8; ++l)
      mk[l][0] = *reinterpret_cast<const u32x4*>(frag_ptr(a.acts, tile, a.astride, LI::A_MASK + l, r, h));
    store_frags<1>(a.dz, tile0, a.zstride, LI::Z_OUT, zo[0], r, h); ws.note_stores(1);
    bf16x8 za[ST][16], zb[ST][16];
    layer_bwd<ST, 1, 8, true>(ws, LI::B_OUT, zo, zb, mk[7], lane);                                  // dZ7
    store_frags<16>(a.dz, tile0, a.zstride, LI::Z_L0 + 112, zb[0], r, h); ws.note_stores(16);
    layer_bwd<ST, 16, 8, true>(ws, LI::B_L7 + 0 * 128, zb, za, mk[6], lane);                        // dZ6
    store_frags<16>(a.dz, tile0, a.zstride, LI::Z_L0 + 96, za[0], r, h); ws.note_stores(16);
    layer_bwd<ST, 16, 8, true>(ws, LI::B_L7 + 1 * 128, za, zb, mk[5], lane);                        // dZ5
    store_frags<16>(a.dz, tile0, a.zstride, LI::Z_L0 + 80, zb[0], r, h); ws.note_stores(16);
    layer_bwd<ST, 16, 8, true>(ws, LI::B_L7 + 2 * 128, zb, za, mk[4], lane);                        // dZ4
    store_frags<16>(a.dz, tile0, a.zstride, LI::Z_L0 + 64, za[0], r, h); ws.note_stores(16);
    layer_bwd<ST, 16, 8, true>(ws, LI::B_L7 + 3 * 128, za, zb, mk[3], lane);                        // dZ3
    store_frags<16>(a.dz, tile0, a.zstride, LI::Z_L0 + 48, zb[0], r, h); ws.note_stores(16);
    layer_bwd<ST, 16, 8, true>(ws, LI::B_L7 + 4 * 128, zb, za, mk[2], lane);                        // dZ2
    store_frags<16>(a.dz, tile0, a.zstride, LI::Z_L0 + 32, za[0], r, h); ws.note_stores(16);
    layer_bwd<ST, 16, 8, true>(ws, LI::B_L7 + 5 * 128, za, zb, mk[1], lane);                        // dZ1
    store_frags<16>(a.dz, tile0, a.zstride, LI::Z_L0 + 16, zb[0], r, h); ws.note_stores(16);
    layer_bwd<ST, 16, 8, true>(ws, LI::B_L7 + 6 * 128, zb, za, mk[0], lane);                        // dZ0
    store_frags<16>(a.dz, tile0, a.zstride, LI::Z_L0 + 0, za[0], r, h); ws.note_stores(16);
  }
  ws.drain();
}


// ==========================================================================================
// Third architecture: the reference's NeRF class at Instant-NGP size (BASELINE configs[4]: hash-grid features +
// tiny MLP): NeRF(n_layers=2, width_layers=64, channel_input=32 [16 levels x 2 features], channel_input_views=16
// [SH degree 3], list_skip_connection_layers=[], is_use_view_directions=True)  (models/NeRF.py:160-243):
// pos0 [64x32] pos1 [64x64] feature [64x64] alpha [1x64] dir0 [32x80] rgb [3x32] = 13 188 parameters.
// All 31 forward (27 backward) fragments fit LDS, so there is no ring: every workgroup copies its stream once and
// its 8 waves then run independently over sample tiles.  Same register-resident chain, same fragment-block stores,
// same dW kernel (7 jobs); the chain additionally returns dL/d(input features) for the hash-grid backward.
// ==========================================================================================
__global__ void __launch_bounds__(256) pack_small_kernel(const float* __restrict__ p, bf16x8* __restrict__ wf,
                                                         bf16x8* __restrict__ wb, float* __restrict__ bias) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int nf = LN::F_PADDED * 64, nb = LN::B_PADDED * 64;
  if (tid < nf + nb) {
    const bool fw = tid < nf;
    const int t = fw ? tid : tid - nf, f = t >> 6, lane = t & 63, r = lane & 31, h = lane >> 5;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)(fw ? fwd_src_small(p, f, r, h, j) : bwd_src_small(p, f, r, h, j));
    (fw ? wf : wb)[t] = v;
  } else if (tid < nf + nb + LN::BI_TOTAL) {
    const int s = tid - nf - nb;
    float v = 0.0f;
    if (s < LN::BI_L1) v = p[LN::P_B0 + s];
    else if (s < LN::BI_FEAT) v = p[LN::P_B1 + (s - LN::BI_L1)];
    else if (s < LN::BI_ALPHA) v = p[LN::P_BF + (s - LN::BI_FEAT)];
    else if (s < LN::BI_DIR) v = (s == LN::BI_ALPHA) ? p[LN::P_BA] : 0.0f;
    else if (s < LN::BI_RGB) v = p[LN::P_BD + (s - LN::BI_DIR)];
    else v = (s - LN::BI_RGB) < 3 ? p[LN::P_BR + (s - LN::BI_RGB)] : 0.0f;
    bias[s] = v;
  }
}

// weight source (3): the whole stream resident in LDS (copied once per workgroup)
struct LdsW {
  __device__ __forceinline__ bf16x8 frag(int f, int lane) { return *reinterpret_cast<const bf16x8*>(ring_smem + f * 1024 + lane * 16); }
  __device__ __forceinline__ void note_stores(int) {}
  __device__ __forceinline__ float4 bias4(int slot) { return *reinterpret_cast<const float4*>(ring_smem + 32 * 1024 + slot * 4); }
};
__device__ __forceinline__ void lds_load_stream(const bf16x8* __restrict__ w, const float* __restrict__ bias) {
  for (int i = threadIdx.x; i < 32 * 64; i += blockDim.x) *reinterpret_cast<bf16x8*>(ring_smem + i * 16) = w[i];
  if (bias)
    for (int i = threadIdx.x; i < LN::BI_TOTAL; i += blockDim.x) *reinterpret_cast<float*>(ring_smem + 32 * 1024 + 4 * i) = bias[i];
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
  // fused configs[4] query: rows are computed in the kernel from rays / depths / hash tables (16 levels x 2 features)
  const float* rays; const float* z; int n; const float* tables; uint32_t T; ResTab rt; float pos_scale, pos_offset;
  const uint32_t* tables_h;   // fp16 shadow image of the tables (one 4-byte pair per entry) or nullptr: gathered instead of `tables`
  // ray_major (inference of the fused query only): a 32-sample tile = ONE depth index of 32 ADJACENT RAYS instead of 32
  // consecutive depths of one ray.  Neighbouring pixels' samples at the same depth are 0.0012 apart in the unit cube, so
  // the 32 lanes of a gather instruction fall into the same or adjacent cells at every level below N_l ~ 800 (13 of 16),
  // where 32 consecutive depths of one ray (0.021 apart) share cells only below N_l ~ 48.  Same values per sample.
  int ray_major; int64_t B;
};

// B fragments of one sample straight from the hash tables and the view direction: lane (r, h) owns channels
// kperm(ks, h, j) of k-step ks = levels 8 ks + 4 (j >> 2) + 2 h + ((j & 3) >> 1), feature j & 1, i.e. the two lanes of
// a sample split the 16 levels between them; SH degree 3 = 16 channels = one k-step.
template <bool HALF>
__device__ __forceinline__ void ngp_row_frags(const SmallArgs& a, int64_t m, int h, bf16x8 (&xin)[1][2], bf16x8 (&din)[1][1]) {
  const float* rr = a.rays + (int64_t)((uint64_t)m / (unsigned)a.n) * NERF_RAY_STRIDE;
  const float zv = a.z[m];
  // render.py:142, then the scene box -> unit cube map (same two roundings as hash_common.h:point_of)
  const float px = (rr[0] + zv * rr[3]) * a.pos_scale + a.pos_offset, py = (rr[1] + zv * rr[4]) * a.pos_scale + a.pos_offset;
  const float pz = (rr[2] + zv * rr[5]) * a.pos_scale + a.pos_offset;
  const uint32_t mask = a.T - 1;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int l = 8 * ks + 4 * q + 2 * h + e;
        const Corners c = corners_of(px, py, pz, a.rt.res[l], mask);
        const FeatVec<2> fv = HALF ? hash_level_h(a.tables_h + (size_t)l * a.T, c) : hash_level<2>(a.tables + (size_t)l * a.T * 2, c);
        xin[0][ks][4 * q + 2 * e] = (__bf16)fv.v[0];
        xin[0][ks][4 * q + 2 * e + 1] = (__bf16)fv.v[1];
      }
  float sh[16];
  sh_eval(rr[8], rr[9], rr[10], 3, sh);
#pragma unroll
  for (int j = 0; j < 8; ++j) din[0][0][j] = (__bf16)(h == 0 ? sh[8 * (j >> 2) + (j & 3)] : sh[8 * (j >> 2) + 4 + (j & 3)]);
}

template <bool STORE, bool FUSED>
__global__ void __launch_bounds__(512) mlp_small_fwd_kernel(SmallArgs a) {
  constexpr int ST = 1;
  lds_load_stream(a.wf, a.bias);
  const int lane0 = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool rmaj = FUSED && !STORE && a.ray_major;
  const int64_t ntiles = rmaj ? ((a.B + 31) >> 5) * a.n : (a.M + 31) >> 5;
  LdsW ws;
  for (int64_t tile0 = (int64_t)blockIdx.x * 8 + wv; tile0 < ntiles; tile0 += (int64_t)gridDim.x * 8) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));        // fragment addresses are per-tile values: the 31 LDS reads are not hoisted (spills)
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
    bf16x8 xin[ST][2], din[ST][1];
    if (FUSED) {
      if (a.tables_h) ngp_row_frags<true>(a, m, h, xin, din);        // wave-uniform
      else ngp_row_frags<false>(a, m, h, xin, din);
    } else {
      const float* row = a.x + m * LN::CIN;
      xin[0][0] = row_frag(row, 0, h, LN::CPOS); xin[0][1] = row_frag(row, 1, h, LN::CPOS);
      din[0][0] = row_frag(row + LN::CPOS, 0, h, LN::CDIR);
    }
    int bad = nonfinite_flags<32>(nonfinite_bits(xin[0][0]) | nonfinite_bits(xin[0][1]), nonfinite_bits(din[0][0]));
    asm volatile("" : "+v"(bad));         // at the head of the tile (see fwd_tiles)
#define SINK(slot0) FragSink<STORE>{a.acts, tile0, a.astride, slot0, r, h}
#define MASK_STORE(layer) do { if (STORE) *reinterpret_cast<u32x4*>(frag_ptr(a.acts, tile0, a.astride, LN::A_MASK + (layer), r, h)) = mk[0]; } while (0)
    if (STORE) { store_frags<2>(a.acts, tile0, a.astride, LN::A_X, xin[0], r, h); store_frags<1>(a.acts, tile0, a.astride, LN::A_DX, din[0], r, h); }
    u32x4 mk[ST];
    bf16x8 h0[ST][4], h1[ST][4], ft[ST][4];
    mk[0] = u32x4{0u, 0u, 0u, 0u};
    layer_fwd<ST, 2, 2, true, STORE>(ws, LN::F_L0, LN::BI_L0, xin, h0, mk, lane, SINK(LN::A_H0));
    MASK_STORE(0);
    mk[0] = u32x4{0u, 0u, 0u, 0u};
    layer_fwd<ST, 4, 2, true, STORE>(ws, LN::F_L1, LN::BI_L1, h0, h1, mk, lane, SINK(LN::A_H1));
    MASK_STORE(1);
    layer_fwd<ST, 4, 2, false, false>(ws, LN::F_FA, LN::BI_FEAT, h1, ft, mk, lane, SINK(LN::A_FEAT));
    float alpha;
    {
      f32x16 acc;
      acc_init_bias(acc, ws, LN::BI_ALPHA, h);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ws.frag(LN::F_FA + 8 + ks, lane), h1[0][ks], acc, 0, 0, 0);
      alpha = acc[0];
    }
    bf16x8 hd[ST][2];
    {
      bf16x8 cat[ST][5];
#pragma unroll
      for (int k = 0; k < 4; ++k) cat[0][k] = ft[0][k];
      cat[0][4] = din[0][0];
      mk[0] = u32x4{0u, 0u, 0u, 0u};
      layer_fwd<ST, 5, 1, true, STORE>(ws, LN::F_DIR, LN::BI_DIR, cat, hd, mk, lane, SINK(LN::A_HD));
      MASK_STORE(2);
    }
    {
      f32x16 acc;
      acc_init_bias(acc, ws, LN::BI_RGB, h);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ws.frag(LN::F_RGB + ks, lane), hd[0][ks], acc, 0, 0, 0);
      if (h == 0 && valid) {
        float4 o; o.x = acc[0]; o.y = acc[1]; o.z = acc[2]; o.w = alpha;
        *reinterpret_cast<float4*>(a.out + m * 4) = poison_raw(o, bad);
      }
    }
#undef SINK
#undef MASK_STORE
  }
}

__global__ void __launch_bounds__(512) mlp_small_bwd_kernel(SmallArgs a) {
  constexpr int ST = 1;
  lds_load_stream(a.wb, nullptr);
  const int lane0 = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ntiles = (a.M + 31) >> 5;
  LdsW ws;
  for (int64_t tile0 = (int64_t)blockIdx.x * 8 + wv; tile0 < ntiles; tile0 += (int64_t)gridDim.x * 8) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int r = lane & 31, h = lane >> 5;
    const int64_t m = tile0 * 32 + r;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < a.M && h == 0) g = *reinterpret_cast<const float4*>(a.d_raw + m * 4);
    bf16x8 zrgb[ST][1], zal[ST][1];
#pragma unroll
    for (int j = 0; j < 8; ++j) { zrgb[0][0][j] = (__bf16)0.0f; zal[0][0][j] = (__bf16)0.0f; }
    zrgb[0][0][0] = (__bf16)g.x; zrgb[0][0][1] = (__bf16)g.y; zrgb[0][0][2] = (__bf16)g.z;
    zal[0][0][0] = (__bf16)g.w;
    u32x4 mk[3][ST];
#pragma unroll
    for (int l = 0; l < 3; ++l)
      mk[l][0] = *reinterpret_cast<const u32x4*>(frag_ptr(const_cast<void*>(a.acts), tile0, a.astride, LN::A_MASK + l, r, h));
    store_frags<1>(a.dz, tile0, a.zstride, LN::Z_RGB, zrgb[0], r, h);
    store_frags<1>(a.dz, tile0, a.zstride, LN::Z_A, zal[0], r, h);
#define ZSINK(slot0) FragSink<true>{a.dz, tile0, a.zstride, slot0, r, h}
    bf16x8 zd[ST][2], zf[ST][4], z1[ST][4], z0[ST][4];
    layer_bwd<ST, 1, 1, true>(ws, LN::B_RGB, zrgb, zd, mk[2], lane, ZSINK(LN::Z_D));
    layer_bwd<ST, 2, 2, false>(ws, LN::B_DIR, zd, zf, mk[2], lane, ZSINK(LN::Z_F));               // d feature
    {
      bf16x8 cat[ST][5];
#pragma unroll
      for (int k = 0; k < 4; ++k) cat[0][k] = zf[0][k];
      cat[0][4] = zal[0][0];
      layer_bwd<ST, 5, 2, true>(ws, LN::B_FA, cat, z1, mk[1], lane, ZSINK(LN::Z_L1));              // dZ1
    }
    layer_bwd<ST, 4, 2, true>(ws, LN::B_L1, z1, z0, mk[0], lane, ZSINK(LN::Z_L0));                 // dZ0
#undef ZSINK
    if (a.d_x) {                                          // dL/dx = W0^T dZ0, kept in fp32: rows (i&3) + 8 (i>>2) + 4 h
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ws.frag(LN::B_L0 + ns, lane), z0[0][ns], acc, 0, 0, 0);
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

static int g_dw_wgs = 0;       // 0: automatic (see launch_dw)
static int g_dw_bias = -1;     // per-tile fixed cost of a dW job in fragment units (cost model of the static split); -1 = automatic:
                               // 128 for the bf16 kernel (round 2 sweep), 32 for the split-bf16 kernel (round 5: its stage boundary is
                               // paid once per sample tile and narrow jobs use all four SIMDs, so a job's time follows its bytes more
                               // closely: tools/sweep_dw22_bias.py, 3.30-3.34 ms at 16-48 against 3.47 at 128)
static int g_bwd_stage = 0;    // diagnostic: 0 chain + dW, 1 chain only, 2 dW only (on whatever dz holds)
static int g_dw_ring_cap = 8;  // "dw_ring_cap": most stages the 16-wave split-bf16 dW kernel's LDS ring may hold (2 .. 16)
static int g_dw_private = 4;   // "dw_private_tiles": split-bf16 dW jobs of at most this many output tiles run as sixteen wave-private pipelines (0 = off)
static int g_dw16_variant = 1;  // "dw16_variant": bf16 weight gradients: 1 (default) = 256 x 256 jobs on mlp_dww.hip's kernel, tiny-job lists on mlp_s16.hip's, the rest on mlp_dw_kernel; 0 = every job on mlp_dw_kernel; 2 = as 1 without the tiny-job rule; 3 = as 1 with mlp_s16.hip's kernel for every narrow job
static int g_dw_narrow_first = 1;   // "dw_narrow_first": order of the two weight-gradient launches (A/B knob; same gradients either way)
static int g_dw_job_mask = 0;  // diagnostic: nonzero = run only these dW jobs (bit j)
static int g_tile_pad16 = 0;     // extra 16-byte units between sample tiles of the fragment stores
static inline int64_t astride16() { return (int64_t)L::A_SLOTS * 64 + g_tile_pad16; }
static inline int64_t zstride16() { return (int64_t)L::Z_SLOTS * 64 + g_tile_pad16; }
static int g_mlp_variant = 0;   // 0: auto, 1: ST=1 via L1, 2: ST=2 via L1, 3: LDS ring, 8 waves x 32 samples, 32x32x16 MFMA,
                                // 4: ring, 16x16x32 MFMA, 8 waves x 32 samples (inference only), 5: same, 4 waves x 64 samples
static int g_ring_wgs = 0;       // persistent workgroups of the ring kernels; 0 = one per CU of the current device
static int g_ngp_ray_major = 1;   // A/B knob: fused configs[4] inference query walks (32 rays x 1 depth) tiles (1) or (1 ray x 32 depths) tiles (0)
int g_pass_queue = 1;               // "pass_queue": 1 (default) the persistent ring kernels take their passes from a device-wide counter (mlp_ring.h), 0 = static split
static int g_ring_split = 1;     // training ring kernels: 1 = one 8-wave workgroup per CU (128 KiB ring), 2 = two 4-wave workgroups (64 KiB rings)
// precision of a model = nerf_mlp_arch.precision (ABI 3): 16 (or 0) bf16 MFMA operands with fp32 accumulate, 32 the fp32
// reference-precision kernels of mlp32.hip.  Nothing process-wide: two models of different precision can be packed,
// queried and trained side by side on any streams.
// 22 = the reference-tolerance mode on the 16-bit matrix pipe: split-fp16 inference (mlp22.hip: float32-class accuracy at
// 1/3 of the fp16 matrix rate) and split-bf16 training (mlp_s16.hip: training forward, dZ chain, dW at 1/3 of the bf16 rate);
// such a model carries the bf16 image (its fp32 bias slots are shared), the split-fp16 and the split-bf16 streams.
static inline int arch_prec(const nerf_mlp_arch* a) { return a->precision == 32 ? 32 : a->precision == 22 ? 22 : 16; }
static inline bool arch_f32(const nerf_mlp_arch* a) { return a->precision == 32; }                          // trains on mlp32.hip
static inline bool arch_s16(const nerf_mlp_arch* a) { return a->precision == 22; }                          // trains on mlp_s16.hip
// CUs of the current device (256 on an MI355X in SPX mode), asked once: the persistent kernels and the dW split are
// sized to it instead of to a constant
// Per-device state: the CU count and the "dynamic LDS attribute set" flags belong to the CURRENT device (a process may
// move between devices with hipSetDevice; hipFuncSetAttribute applies to the device that is current when it is called).
constexpr int MAX_DEVICES = 64;
static int cur_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES) d = 0;
  return d;
}
// (DevOnce -- thread-safe per-device one-time set-up: common.h)
static int cu_count() {
  static int n[MAX_DEVICES] = {};
  const int dev = cur_device();
  if (n[dev] == 0) {
    int v = 0;
    n[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return n[dev];
}
static inline int ring_wgs() { return g_ring_wgs > 0 ? g_ring_wgs : cu_count(); }

template <class K>
static void ensure_lds(K kernel, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// 0: the NeRF view model (63+27 -> 4), 1: the image-fitting model (40 -> out_ch), 2: the Instant-NGP-sized view
// model (32+16 -> 4, 2 x 64), -1: no HIP kernel
static int arch_kind(const nerf_mlp_arch* a) {
  if (!a) return -1;
  if (a->precision != 0 && a->precision != 16 && a->precision != 32 && a->precision != 22) return -1;
  if (arch_f32(a) && !(a->n_layers == 8 && a->width == 256)) return -1;   // fp32 MFMA kernels: the 8 x 256 models (view head or image)
  if (a->n_layers == 2 && a->width == 64 && a->skip_layer < 0 && a->use_viewdirs == 1 && a->in_pos == 32 && a->in_dir == 16) return 2;
  if (a->n_layers != 8 || a->width != 256 || a->skip_layer != 4) return -1;
  if (a->use_viewdirs == 1 && a->in_pos == 63 && a->in_dir == 27) return 0;
  if (a->use_viewdirs == 0 && a->in_pos == 40 && a->out_ch >= 1 && a->out_ch <= 4) return 1;
  return -1;
}
static bool arch_ok(const nerf_mlp_arch* a) { return arch_kind(a) == 0; }
static inline int64_t img_astride16(bool split = false) { return (int64_t)(split ? s16x::IMG_A_SLOTS : LI::A_SLOTS) * 64 + g_tile_pad16; }
static inline int64_t img_zstride16(bool split = false) { return (int64_t)(split ? s16x::IMG_Z_SLOTS : LI::Z_SLOTS) * 64 + g_tile_pad16; }
static inline int64_t img_params(const nerf_mlp_arch* a) { return LI::P_WO + (int64_t)a->out_ch * 257; }
static inline int64_t small_astride16(bool split = false) { return (int64_t)(split ? s16x::SM_A_SLOTS : LN::A_SLOTS) * 64 + g_tile_pad16; }
static inline int64_t small_zstride16(bool split = false) { return (int64_t)(split ? s16x::SM_Z_SLOTS : LN::Z_SLOTS) * 64 + g_tile_pad16; }
// bf16 images of the image-fitting / 2 x 64 models (a precision-22 model carries its split-bf16 pair streams behind them and
// shares their fp32 bias slots)
constexpr int64_t IMG_BF16_BYTES = (int64_t)(LI::F_TOTAL + LI::B_PADDED) * 1024 + LI::BI_TOTAL * 4;
static inline const float* img_bias_of(const void* packed) {
  return reinterpret_cast<const float*>(static_cast<const char*>(packed) + (size_t)(LI::F_TOTAL + LI::B_PADDED) * 1024);
}
static inline const float* small_bias_of(const void* packed) {
  return reinterpret_cast<const float*>(static_cast<const char*>(packed) + (size_t)(LN::F_PADDED + LN::B_PADDED) * 1024);
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_set_option(const char* key, int value) {
  NERF_REQUIRE(key, NERF_E_NULL, "nerf_set_option: key is NULL");
  if (!strcmp(key, "mlp_variant")) { g_mlp_variant = value; return NERF_OK; }
  if (!strcmp(key, "ring_split")) {
    NERF_REQUIRE(value == 1 || value == 2, NERF_E_UNSUPPORTED, "nerf_set_option: ring_split must be 1 or 2");
    g_ring_split = value;
    return NERF_OK;
  }
  NERF_REQUIRE(strcmp(key, "mlp_precision") != 0, NERF_E_UNSUPPORTED,
               "nerf_set_option: \"mlp_precision\" is gone (ABI 3): set nerf_mlp_arch.precision of the model instead");
  if (!strcmp(key, "ring_workgroups")) { g_ring_wgs = value > 0 ? value : 0; return NERF_OK; }
  if (!strcmp(key, "tile_pad16")) { g_tile_pad16 = value >= 0 ? value : 0; return NERF_OK; }
  if (!strcmp(key, "dw_workgroups")) { g_dw_wgs = value > 0 ? value : 0; return NERF_OK; }
  if (!strcmp(key, "dw_unit_bias")) { g_dw_bias = value >= 0 ? value : -1; return NERF_OK; }
  if (!strcmp(key, "bwd_stage")) { g_bwd_stage = value; return NERF_OK; }
  if (!strcmp(key, "dw_job_mask")) { g_dw_job_mask = value; return NERF_OK; }
  if (!strcmp(key, "hash_combine_max_res")) { g_hash_combine_max_res = value > 0 ? value : 0; return NERF_OK; }
  if (!strcmp(key, "ngp_ray_major")) { g_ngp_ray_major = value ? 1 : 0; return NERF_OK; }
  if (!strcmp(key, "dw22_variant")) { s16::g_dw_variant = value == 0 ? 0 : 1; return NERF_OK; }
  if (!strcmp(key, "dw16_variant")) { g_dw16_variant = value < 0 ? 0 : value > 3 ? 3 : value; return NERF_OK; }
  if (!strcmp(key, "dw_private_tiles")) { g_dw_private = value < 0 ? 0 : value > 4 ? 4 : value; return NERF_OK; }
  if (!strcmp(key, "dw_ring_cap")) { g_dw_ring_cap = value < 2 ? 2 : value > 16 ? 16 : value; return NERF_OK; }
  if (!strcmp(key, "pass_queue")) { g_pass_queue = value ? 1 : 0; return NERF_OK; }
  if (!strcmp(key, "dw_narrow_first")) { g_dw_narrow_first = value ? 1 : 0; return NERF_OK; }
  if (!strcmp(key, "f22_tiles")) {
    NERF_REQUIRE(value == 0 || value == 2 || value == 3, NERF_E_UNSUPPORTED, "nerf_set_option: f22_tiles must be 0 (automatic), 2 or 3");
    f22::g_tiles = value;
    return NERF_OK;
  }
  return fail(NERF_E_UNSUPPORTED, "nerf_set_option: unknown key '%s'", key);
}

extern "C" int nerf_get_option(const char* key) {
  if (!key) return NERF_OPTION_UNKNOWN;
  if (!strcmp(key, "mlp_variant")) return g_mlp_variant;
  if (!strcmp(key, "ring_workgroups")) return g_ring_wgs;
  if (!strcmp(key, "ring_split")) return g_ring_split;
  if (!strcmp(key, "dw_workgroups")) return g_dw_wgs;
  if (!strcmp(key, "hash_combine_max_res")) return g_hash_combine_max_res;
  if (!strcmp(key, "ngp_ray_major")) return g_ngp_ray_major;
  if (!strcmp(key, "dw22_variant")) return s16::g_dw_variant;
  if (!strcmp(key, "dw16_variant")) return g_dw16_variant;
  if (!strcmp(key, "tile_pad16")) return g_tile_pad16;
  if (!strcmp(key, "dw_unit_bias")) return g_dw_bias;            // -1 = automatic (a legitimate value: unknown keys are INT_MIN)
  if (!strcmp(key, "bwd_stage")) return g_bwd_stage;
  if (!strcmp(key, "dw_job_mask")) return g_dw_job_mask;
  if (!strcmp(key, "dw_private_tiles")) return g_dw_private;
  if (!strcmp(key, "dw_ring_cap")) return g_dw_ring_cap;
  if (!strcmp(key, "pass_queue")) return g_pass_queue;
  if (!strcmp(key, "dw_narrow_first")) return g_dw_narrow_first;
  if (!strcmp(key, "f22_tiles")) return f22::g_tiles;
  return NERF_OPTION_UNKNOWN;
}

extern "C" int64_t nerf_mlp_param_count(const nerf_mlp_arch* arch) {
  const int k = arch_kind(arch);
  return k == 0 ? L::P_TOTAL : k == 1 ? img_params(arch) : k == 2 ? LN::P_TOTAL : -1;
}
extern "C" int64_t nerf_mlp_packed_bytes(const nerf_mlp_arch* arch) {
  const int k = arch_kind(arch);
  // an fp32 (precision 32) 8 x 256 view model carries its fp32 streams behind the bf16 image
  // ... and a precision-22 model its split-fp16 stream behind that
  return k == 0 ? L::PACKED_BYTES + (arch_f32(arch) ? f32::PACKED_BYTES : 0) + (arch_s16(arch) ? f22::PACKED_BYTES + s16::PACKED_BYTES : 0)
         : k == 1 ? IMG_BF16_BYTES + (arch_s16(arch) ? s16x::IMG_PACKED_BYTES : 0) + (arch_f32(arch) ? f32::PACKED_BYTES : 0)
         : k == 2 ? LN::PACKED_BYTES + (arch_s16(arch) ? s16x::SM_PACKED_BYTES : 0) : -1;
}
static inline const void* packed32_of(const void* packed) { return static_cast<const char*>(packed) + L::PACKED_BYTES; }
static inline const void* packed22_of(const void* packed) { return static_cast<const char*>(packed) + L::PACKED_BYTES; }
static inline const void* packed_s16_of(const void* packed) { return static_cast<const char*>(packed) + L::PACKED_BYTES + f22::PACKED_BYTES; }
static inline const float* bias_slots_of(const void* packed) {
  return reinterpret_cast<const float*>(static_cast<const char*>(packed) + (size_t)(L::F_TOTAL + L::B_PADDED) * 1024);
}
static inline int64_t s16_astride16() { return (int64_t)s16::A_SLOTS * 64 + g_tile_pad16; }
static inline int64_t s16_zstride16() { return (int64_t)s16::Z_SLOTS * 64 + g_tile_pad16; }
static inline int64_t padded_tiles(int64_t M) { return (((M + 31) / 32) + 7) / 8 * 8; }
extern "C" int64_t nerf_mlp_acts_bytes(const nerf_mlp_arch* arch, int64_t M) {
  const int k = arch_kind(arch);
  if (k < 0 || M < 0) return -1;
  const bool sp = arch_s16(arch);
  const int64_t b16 = padded_tiles(M) * (k == 0 ? astride16() : k == 1 ? img_astride16(sp) : small_astride16(sp)) * 16;
  if (k == 0 && arch_s16(arch)) return padded_tiles(M) * s16_astride16() * 16;
  return (k <= 1 && arch_f32(arch)) ? f32::acts_bytes(M) : b16;
}
extern "C" int64_t nerf_mlp_dz_bytes(const nerf_mlp_arch* arch, int64_t M) {
  const int k = arch_kind(arch);
  if (k < 0 || M < 0) return -1;
  const bool sp = arch_s16(arch);
  const int64_t b16 = padded_tiles(M) * (k == 0 ? zstride16() : k == 1 ? img_zstride16(sp) : small_zstride16(sp)) * 16
                      + DW_PARTIAL_BYTES;                 // + the split-K partial tiles of the weight-gradient kernel
  if (k == 0 && arch_s16(arch)) return padded_tiles(M) * s16_zstride16() * 16 + DW_PARTIAL_BYTES;
  return (k <= 1 && arch_f32(arch)) ? f32::dz_bytes(M) : b16;
}

#define NERF_ARCH_MSG ": HIP kernels exist for (8x256, skip 4) with in=63+27 view head, or in=40 / no view head / out_ch<=4, and for (2x64, no skip) with in=32+16 view head"
#define NERF_ARCH_CHECK(who) NERF_REQUIRE(arch_ok(arch), NERF_E_UNSUPPORTED, who NERF_ARCH_MSG)
#define NERF_ARCH_CHECK_ANY(who) NERF_REQUIRE(arch_kind(arch) >= 0, NERF_E_UNSUPPORTED, who NERF_ARCH_MSG)

extern "C" int nerf_mlp_pack(const nerf_mlp_arch* arch, const float* params, void* packed, void* stream) {
  NERF_ARCH_CHECK_ANY("nerf_mlp_pack");
  NERF_REQUIRE(params && packed, NERF_E_NULL, "nerf_mlp_pack: params/packed is NULL");
  char* base = static_cast<char*>(packed);
  if (arch_kind(arch) == 2) {
    const int tot = (LN::F_PADDED + LN::B_PADDED) * 64 + LN::BI_TOTAL;
    hipLaunchKernelGGL(pack_small_kernel, dim3((tot + 255) / 256), dim3(256), 0, as_stream(stream), params,
                       reinterpret_cast<bf16x8*>(base), reinterpret_cast<bf16x8*>(base + (size_t)LN::F_PADDED * 1024),
                       reinterpret_cast<float*>(base + (size_t)(LN::F_PADDED + LN::B_PADDED) * 1024));
    int rcn = check_launch("nerf_mlp_pack (2x64 model)");
    if (!rcn && arch_s16(arch)) rcn = s16x::small_pack(params, base + LN::PACKED_BYTES, as_stream(stream));
    return rcn;
  }
  if (arch_kind(arch) == 1) {
    bf16x8* wfi = reinterpret_cast<bf16x8*>(base);
    bf16x8* wbi = reinterpret_cast<bf16x8*>(base + (size_t)LI::F_TOTAL * 1024);
    float* bi = reinterpret_cast<float*>(base + (size_t)(LI::F_TOTAL + LI::B_PADDED) * 1024);
    const int tot = (LI::F_TOTAL + LI::B_PADDED) * 64 + LI::BI_TOTAL;
    hipLaunchKernelGGL(pack_img_kernel, dim3((tot + 255) / 256), dim3(256), 0, as_stream(stream), params, wfi, wbi, bi,
                       arch->out_ch);
    int rci = check_launch("nerf_mlp_pack (image model)");
    if (!rci && arch_s16(arch)) rci = s16x::img_pack(params, arch->out_ch, base + IMG_BF16_BYTES, as_stream(stream));
    if (!rci && arch_f32(arch)) rci = f32::pack(params, base + IMG_BF16_BYTES, arch->out_ch, as_stream(stream));
    return rci;
  }
  bf16x8* wf = reinterpret_cast<bf16x8*>(base);
  bf16x8* wb = reinterpret_cast<bf16x8*>(base + (size_t)L::F_TOTAL * 1024);
  float* bias = reinterpret_cast<float*>(base + (size_t)(L::F_TOTAL + L::B_PADDED) * 1024);
  hipLaunchKernelGGL(pack_kernel, dim3(PACK_BLOCKS + PACK16_BLOCKS), dim3(256), 0, as_stream(stream), params, wf, wb, bias,
                     reinterpret_cast<bf16x8*>(base + L::F16_OFFSET));
  int rc = check_launch("nerf_mlp_pack");
  if (rc) return rc;
  if (arch_f32(arch)) rc = f32::pack(params, base + L::PACKED_BYTES, 0, as_stream(stream));
  if (!rc && arch_s16(arch)) rc = f22::pack(params, base + L::PACKED_BYTES, as_stream(stream));
  if (!rc && arch_s16(arch)) rc = s16::pack(params, base + L::PACKED_BYTES + f22::PACKED_BYTES, as_stream(stream));
  return rc;
}

static void fill_freqs(PeFreq& fr, int mode) {
  for (int k = 0; k < 10; ++k) fr.pos[k] = mode == 0 ? (float)(k * k) : (float)(1 << k);
  for (int k = 0; k < 4; ++k) fr.dir[k] = mode == 0 ? (float)(k * k) : (float)(1 << k);
}

template <int MODE>
static int launch_fwd(const void* packed, const float* x, const float* rays, const float* z, int64_t M, int n,
                      int freq_mode, float* out, void* acts, void* stream) {
  FwdArgs a;
  a.queue = nullptr;
  const char* base = static_cast<const char*>(packed);
  a.wf = reinterpret_cast<const bf16x8*>(base);
  a.bias = reinterpret_cast<const float*>(base + (size_t)(L::F_TOTAL + L::B_PADDED) * 1024);
  a.x = x; a.rays = rays; a.z = z; a.M = M; a.n = n; a.out = out; a.acts = acts; a.astride = astride16();
  fill_freqs(a.fr, freq_mode);
  const int64_t ntiles = (M + 31) / 32;
  auto s = as_stream(stream);
  // auto: fused query -> LDS ring; the render path (no activations kept) on the 16x16x32 shape, which holds a higher
  // clock: +8 % back to back, +2.5 % inside bench.py's train+render step
  const int variant = (g_mlp_variant == 0) ? (MODE == 1 ? (acts ? 3 : 4) : 1) : g_mlp_variant;
  if ((variant == 4 || variant == 5) && MODE == 1 && !acts) {
    const int64_t nsuper = (M + 255) / 256;
    const dim3 g((unsigned)(nsuper < ring_wgs() ? nsuper : ring_wgs()));
    static DevOnce once16;
    once16.run([&] { ensure_lds(mlp_fwd_ring16_kernel<8, 2>, RING16_LDS_BYTES); ensure_lds(mlp_fwd_ring16_kernel<4, 4>, RING16_LDS_BYTES); });
    FwdArgs a16 = a;
    a16.queue = passq_slot();
    a16.wf = reinterpret_cast<const bf16x8*>(base + L::F16_OFFSET);
    if (variant == 4) hipLaunchKernelGGL((mlp_fwd_ring16_kernel<8, 2>), g, dim3(512), RING16_LDS_BYTES, s, a16);
    else hipLaunchKernelGGL((mlp_fwd_ring16_kernel<4, 4>), g, dim3(256), RING16_LDS_BYTES, s, a16);
    return check_launch("mlp forward (ring, 16x16x32)");
  }
  if (variant >= 3 && MODE == 1) {
    a.queue = passq_slot();
    if (acts && g_ring_split == 2) {                      // two 4-wave workgroups per CU, each with its own 64 KiB ring
      const int64_t nsuper = (ntiles + SPLIT_NW - 1) / SPLIT_NW;
      const int64_t wgs = 2 * (int64_t)ring_wgs();
      static DevOnce once2;
      once2.run([&] { ensure_lds(mlp_fwd_ring_kernel<1, true, SPLIT_NW, SPLIT_CHUNK>, SPLIT_LDS_BYTES); });
      hipLaunchKernelGGL((mlp_fwd_ring_kernel<1, true, SPLIT_NW, SPLIT_CHUNK>), dim3((unsigned)(nsuper < wgs ? nsuper : wgs)),
                         dim3(64 * SPLIT_NW), SPLIT_LDS_BYTES, s, a);
      return check_launch("mlp forward (ring, 2 workgroups per CU)");
    }
    const int64_t nsuper = (ntiles + 7) / 8;
    const dim3 g((unsigned)(nsuper < ring_wgs() ? nsuper : ring_wgs())), b(512);
    static DevOnce once;
    once.run([&] { ensure_lds(mlp_fwd_ring_kernel<1, true>, RING_LDS_BYTES); ensure_lds(mlp_fwd_ring_kernel<1, false>, RING_LDS_BYTES); });
    if (acts) hipLaunchKernelGGL((mlp_fwd_ring_kernel<1, true>), g, b, RING_LDS_BYTES, s, a);
    else hipLaunchKernelGGL((mlp_fwd_ring_kernel<1, false>), g, b, RING_LDS_BYTES, s, a);
    return check_launch("mlp forward (ring)");
  }
  const int st = variant == 2 ? 2 : 1;
  const int64_t blocks = (ntiles + 4 * st - 1) / (4 * st);
  NERF_REQUIRE(blocks < (1ll << 31), NERF_E_SHAPE, "mlp forward: M too large");
  const dim3 g((unsigned)blocks), b(256);
  if (st == 1) {
    if (acts) hipLaunchKernelGGL((mlp_fwd_kernel<1, MODE, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((mlp_fwd_kernel<1, MODE, false>), g, b, 0, s, a);
  } else {
    if (acts) hipLaunchKernelGGL((mlp_fwd_kernel<2, MODE, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((mlp_fwd_kernel<2, MODE, false>), g, b, 0, s, a);
  }
  return check_launch("mlp forward");
}

static void img_args(ImgArgs& a, const nerf_mlp_arch* arch, const void* packed) {
  const char* base = static_cast<const char*>(packed);
  a.wf = reinterpret_cast<const bf16x8*>(base);
  a.wb = reinterpret_cast<const bf16x8*>(base + (size_t)LI::F_TOTAL * 1024);
  a.bias = reinterpret_cast<const float*>(base + (size_t)(LI::F_TOTAL + LI::B_PADDED) * 1024);
  a.out_ch = arch->out_ch; a.astride = img_astride16(); a.zstride = img_zstride16();
  a.x = nullptr; a.d_out = nullptr; a.out = nullptr; a.acts = nullptr; a.dz = nullptr; a.M = 0;
}

static void small_args(SmallArgs& a, const void* packed) {
  const char* base = static_cast<const char*>(packed);
  a.wf = reinterpret_cast<const bf16x8*>(base);
  a.wb = reinterpret_cast<const bf16x8*>(base + (size_t)LN::F_PADDED * 1024);
  a.bias = reinterpret_cast<const float*>(base + (size_t)(LN::F_PADDED + LN::B_PADDED) * 1024);
  a.x = nullptr; a.d_raw = nullptr; a.out = nullptr; a.d_x = nullptr; a.acts = nullptr; a.dz = nullptr; a.M = 0;
  a.astride = small_astride16(); a.zstride = small_zstride16();
  a.rays = nullptr; a.z = nullptr; a.n = 1; a.tables = nullptr; a.tables_h = nullptr; a.T = 0; a.pos_scale = 1.0f; a.pos_offset = 0.0f;
  a.ray_major = 0; a.B = 0;
}

extern "C" int nerf_mlp_forward_train(const nerf_mlp_arch* arch, const void* packed, const float* x, int64_t M,
                                      float* out, void* acts, void* stream) {
  NERF_ARCH_CHECK_ANY("nerf_mlp_forward");
  if (M <= 0) return NERF_OK;
  NERF_REQUIRE(packed && x && out, NERF_E_NULL, "nerf_mlp_forward: NULL pointer");
  if (arch_kind(arch) == 2) {
    SmallArgs a;
    small_args(a, packed);
    if (arch_s16(arch))                        // split bf16: float32-class products (mlp_s16x.hip)
      return s16x::small_forward(static_cast<const char*>(packed) + LN::PACKED_BYTES, small_bias_of(packed), x, M, out, acts,
                                 small_astride16(true), nullptr, as_stream(stream));
    a.x = x; a.out = out; a.acts = acts; a.M = M;
    const int64_t nwg = ((M + 31) / 32 + 7) / 8;
    const dim3 g((unsigned)(nwg < 2048 ? nwg : 2048)), b(512);
    if (acts) hipLaunchKernelGGL((mlp_small_fwd_kernel<true, false>), g, b, LN::LDS_BYTES, as_stream(stream), a);
    else hipLaunchKernelGGL((mlp_small_fwd_kernel<false, false>), g, b, LN::LDS_BYTES, as_stream(stream), a);
    return check_launch("mlp forward (2x64 model)");
  }
  if (arch_kind(arch) == 1) {
    if (arch_f32(arch))                        // float32 operands on the fp32 MFMA (mlp32.hip)
      return f32::forward(static_cast<const char*>(packed) + IMG_BF16_BYTES, x, nullptr, nullptr, M, 1, 0, out, acts, arch->out_ch,
                          as_stream(stream));
    if (arch_s16(arch))
      return s16x::img_forward(static_cast<const char*>(packed) + IMG_BF16_BYTES, img_bias_of(packed), x, M, arch->out_ch, out, acts,
                               img_astride16(true), ring_wgs(), as_stream(stream));
    ImgArgs a;
    img_args(a, arch, packed);
    a.x = x; a.out = out; a.acts = acts; a.M = M;
    const int64_t nsuper = ((M + 31) / 32 + 7) / 8;
    static DevOnce once;
    once.run([&] { ensure_lds(mlp_img_fwd_ring_kernel<true>, RING_LDS_BYTES); ensure_lds(mlp_img_fwd_ring_kernel<false>, RING_LDS_BYTES); });
    const dim3 g((unsigned)(nsuper < ring_wgs() ? nsuper : ring_wgs())), b(512);
    if (acts) hipLaunchKernelGGL(mlp_img_fwd_ring_kernel<true>, g, b, RING_LDS_BYTES, as_stream(stream), a);
    else hipLaunchKernelGGL(mlp_img_fwd_ring_kernel<false>, g, b, RING_LDS_BYTES, as_stream(stream), a);
    return check_launch("mlp forward (image model)");
  }
  if (arch_s16(arch) && !acts)                 // inference: split fp16 on the 16-bit matrix pipe
    return f22::forward(packed22_of(packed), x, nullptr, nullptr, M, 1, 0, out, ring_wgs(), as_stream(stream));
  if (arch_s16(arch))                          // training forward: split bf16, hi + lo fragment blocks kept
    return s16::forward(packed_s16_of(packed), bias_slots_of(packed), x, nullptr, nullptr, M, 1, 0, out, acts, s16_astride16(),
                        ring_wgs(), as_stream(stream));
  if (arch_f32(arch)) {
    return f32::forward(packed32_of(packed), x, nullptr, nullptr, M, 1, 0, out, acts, 0, as_stream(stream));
  }
  return launch_fwd<0>(packed, x, nullptr, nullptr, M, 1, 0, out, acts, stream);
}

extern "C" int nerf_mlp_forward(const nerf_mlp_arch* arch, const void* packed, const float* x, int64_t M, float* out,
                                void* stream) {
  return nerf_mlp_forward_train(arch, packed, x, M, out, nullptr, stream);
}

extern "C" int nerf_query_fused(const nerf_mlp_arch* arch, const void* packed, const float* rays, const float* z,
                                int64_t B, int n, int freq_mode, float* raw, void* acts, void* stream) {
  NERF_ARCH_CHECK("nerf_query_fused");
  NERF_REQUIRE(n >= 1, NERF_E_SHAPE, "nerf_query_fused: n must be >= 1");
  if (B <= 0) return NERF_OK;
  NERF_REQUIRE(packed && rays && z && raw, NERF_E_NULL, "nerf_query_fused: NULL pointer");
  NERF_REQUIRE(freq_mode == 0 || freq_mode == 1, NERF_E_UNSUPPORTED, "nerf_query_fused: freq_mode must be 0 or 1");
  NERF_REQUIRE(B * (int64_t)n < (1ll << 31), NERF_E_SHAPE, "nerf_query_fused: B*n must be < 2^31 samples per call");
  if (arch_s16(arch) && !acts)
    return f22::forward(packed22_of(packed), nullptr, rays, z, B * n, n, freq_mode, raw, ring_wgs(), as_stream(stream));
  if (arch_s16(arch))
    return s16::forward(packed_s16_of(packed), bias_slots_of(packed), nullptr, rays, z, B * n, n, freq_mode, raw, acts,
                        s16_astride16(), ring_wgs(), as_stream(stream));
  if (arch_f32(arch)) {
    return f32::forward(packed32_of(packed), nullptr, rays, z, B * n, n, freq_mode, raw, acts, 0, as_stream(stream));
  }
  return launch_fwd<1>(packed, nullptr, rays, z, B * n, n, freq_mode, raw, acts, stream);
}

// one dW launch + its reduce over a job list.  kind 0: bf16, 16 waves (mlp_dw_kernel); 1: split bf16, 16 waves (s16_dw_kernel);
// 2 / 3: split bf16 / bf16, 256 x 256 jobs only, one wave per SIMD (mlp_dww_kernel).  slot_base: first partial-tile slot of this
// launch (the two launches of one backward pass use disjoint slots).
static int launch_dw_part(DwArgs& d, int nj, int64_t ntiles, int64_t nparams, const void* acts, void* dz, int64_t astride,
                          int64_t zstride, float* grads, hipStream_t s, int kind, int a_lo, int z_lo, int slot_base, int max_wgs) {
  const bool split_bf16 = kind == 1 || kind == 2;
  // A job's cost per sample tile = its bytes (nf + kf KiB) + a fixed part (barrier, waits, the 4 DMA issues per wave,
  // transposed reads, MFMAs) worth about 128 KiB of streaming: single-job timings fit t = a (nf + kf + c0) with c0 = 24
  // at a full grid, but under load the sweep over c0 keeps improving up to ~128 and is flat beyond (tools/sweep_dw.py).  Split the sample range of every job in proportion.
  // Split-bf16 kernels (round 6): since the 256 x 256 jobs run in a launch of their own (all equal: the bias is moot there), the bias only
  // balances the six narrow jobs among themselves, and their times alone fit t = 9.0 (nf + kf) + 0 ... 14: 32 gave the two tiny jobs
  // (dir0 | dirPE, rgb) 36 workgroups each instead of 26-28 and the launch waited for dir0 | feature; 2 is worth -1.8 % of the
  // training step (tools/ab_train_step.py dw_unit_bias -1 2: 10.66 -> 10.46 ms; 0 / 1 / 3 within noise of it, 8: -0.8 %).
  int64_t units[DW_MAX_JOBS], total_units = 0;
  for (int j = 0; j < nj; ++j) {
    units[j] = d.jobs[j].nf + d.jobs[j].kf + (g_dw_bias >= 0 ? g_dw_bias : split_bf16 ? 2 : 128);
    total_units += units[j];
  }
  // One workgroup per CU and launch (256), shares by largest remainder so that they sum to exactly 256: every
  // workgroup starts at once and, with the cost model above, ends at about the same time -- one prologue and one atomic
  // flush per CU instead of 6-16.  (tools/sweep_dw.py: 1.54 ms against 1.70 for the 192-sample pass, 0.52 against 0.66
  // for the 64-sample pass; with the old byte-only cost model 256 workgroups took 3.1 ms because the small jobs'
  // workgroups ran twice as long as the others.)  "dw_workgroups" overrides the total.
  int target_wgs = g_dw_wgs > 0 ? g_dw_wgs : cu_count();
  if (target_wgs > max_wgs) target_wgs = max_wgs;               // one partial-tile slot per workgroup
  const int64_t max_splits = (ntiles + 3) / 4;                  // >= 4 sample tiles per workgroup
  int nw = 0;
  double frac[DW_MAX_JOBS];
  for (int j = 0; j < DW_MAX_JOBS; ++j) d.splits[j] = 0;
  for (int j = 0; j < nj; ++j) {
    const double share = (double)units[j] * target_wgs / (double)total_units;
    int64_t sp = (int64_t)share;
    frac[j] = share - (double)sp;
    if (sp < 1) { sp = 1; frac[j] = 0.0; }
    if (sp > max_splits) { sp = max_splits; frac[j] = 0.0; }
    d.splits[j] = (int)sp;
    nw += (int)sp;
  }
  while (nw < target_wgs) {                                     // hand out the remainder, largest fraction first
    int best = -1;
    for (int j = 0; j < nj; ++j)
      if (d.splits[j] < max_splits && (best < 0 || frac[j] > frac[best])) best = j;
    if (best < 0 || frac[best] <= 0.0) break;
    d.splits[best] += 1; frac[best] = 0.0; nw += 1;
  }
  while (nw > target_wgs) {                                     // (minimum-of-one bumps) take back from the largest
    int big = 0;
    for (int j = 1; j < nj; ++j) if (d.splits[j] > d.splits[big]) big = j;
    if (d.splits[big] <= 1) break;
    d.splits[big] -= 1; nw -= 1;
  }
  NERF_REQUIRE(slot_base + nw <= DW_MAX_WGS, NERF_E_SHAPE, "nerf_mlp_backward: dw_workgroups must be <= %d", DW_MAX_WGS);
  bool all_tiny = true;
  for (int j = 0; j < nj; ++j) all_tiny = all_tiny && ((d.jobs[j].nf + 1) / 2) * ((d.jobs[j].kf + 1) / 2) <= 4;
  d.ntiles = (int)ntiles; d.astride = astride; d.zstride = zstride;
  d.acts = acts; d.dz = dz; d.grads = grads;
  d.a_lo = a_lo; d.z_lo = z_lo; d.ring_cap = g_dw_ring_cap; d.private_max_tiles = g_dw_private;
  // the partial-tile slots live behind the dZ fragment blocks in the caller's dz workspace (nerf_mlp_dz_bytes counts them)
  d.partial = reinterpret_cast<float*>(static_cast<char*>(dz) + padded_tiles(ntiles * 32) * zstride * 16) + (size_t)slot_base * DW_SLOT_FLOATS;
  int rc;
  if (kind >= 2) {             // 16 x 16-fragment jobs, one wave per SIMD (mlp_dww.hip); same slots and reduce
    rc = launch_dw_wide_kernel(d, nw, kind == 2, s);
  } else if (kind == 1) {      // hi + lo fragment blocks, three MFMAs per product (mlp_s16.hip); same jobs, slots and reduce
    rc = s16::launch_dw_kernel(d, nw, true, s);
  } else if (g_dw16_variant == 3 || (g_dw16_variant == 1 && all_tiny)) {
    // the 16-wave kernel of mlp_s16.hip over the bf16 stores: for job lists of tiny jobs only (the 2 x 64 model: sixteen wave-private
    // pipelines, configs[4] bf16 training 3.10 -> 3.21 M rays/s).  For the view model's narrow jobs it measures like round 2's
    // mlp_dw_kernel below (4.45 against 4.43 ms per training step), which keeps them ("dw16_variant" 3 forces this kernel).
    rc = s16::launch_dw_kernel(d, nw, false, s);
  } else {
    static DevOnce lds_attr_set;
    lds_attr_set.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS_BYTES); });
    hipLaunchKernelGGL(mlp_dw_kernel, dim3(nw), dim3(64 * DW_WAVES), DW_LDS_BYTES, s, d);
    rc = check_launch("mlp dW");
  }
  if (rc) return rc;
  hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3(257, nj), dim3(256), 0, s, d);        // 256 x 256 weights + 256 biases: one element per thread
  return check_launch("mlp dW reduce");
}

// split the (dZ, H) jobs over workgroups and launch the dW kernel(s); grads[0..nparams) is overwritten
static int launch_dw(DwArgs& d, int nj, int64_t ntiles, int64_t nparams, const void* acts, void* dz, int64_t astride,
                     int64_t zstride, float* grads, hipStream_t s, bool split_bf16 = false, int a_lo = 0, int z_lo = 0) {
  if (g_dw_job_mask) {          // diagnostic subset of jobs: the parameters of the jobs left out read as zero
    hipError_t e = hipMemsetAsync(grads, 0, sizeof(float) * nparams, s);
    if (e != hipSuccess) return fail(NERF_E_HIP, "nerf_mlp_backward: memset: %s", hipGetErrorString(e));
  }
  // "dw22_variant" / "dw16_variant" 1 (default): the 256 x 256 jobs on the one-wave-per-SIMD kernel, the others on the 16-wave kernel;
  // two launches, each with its own static split over all CUs and its own half of the partial-tile slots.  0: one 16-wave launch.
  if ((split_bf16 ? s16::g_dw_variant : g_dw16_variant) == 0)
    return launch_dw_part(d, nj, ntiles, nparams, acts, dz, astride, zstride, grads, s, split_bf16 ? 1 : 0, a_lo, z_lo, 0, DW_MAX_WGS);
  DwArgs wide = d, rest = d;
  int nwide = 0, nrest = 0;
  for (int j = 0; j < nj; ++j) {
    if (d.jobs[j].nf == 16 && d.jobs[j].kf == 16) wide.jobs[nwide++] = d.jobs[j];
    else rest.jobs[nrest++] = d.jobs[j];
  }
  const int half = DW_MAX_WGS / 2;
  for (int turn = 0; turn < 2; ++turn) {        // "dw_narrow_first" 1: the load-bound narrow jobs before the MFMA-bound 256 x 256 jobs
    const bool do_wide = (turn == 0) != (g_dw_narrow_first != 0);
    if (do_wide && nwide) {
      const int rc = launch_dw_part(wide, nwide, ntiles, nparams, acts, dz, astride, zstride, grads, s, split_bf16 ? 2 : 3, a_lo, z_lo, half, half);
      if (rc) return rc;
    }
    if (!do_wide && nrest) {
      const int rc = launch_dw_part(rest, nrest, ntiles, nparams, acts, dz, astride, zstride, grads, s, split_bf16 ? 1 : 0, a_lo, z_lo, 0, half);
      if (rc) return rc;
    }
  }
  return NERF_OK;
}

static int mlp_backward_impl(const nerf_mlp_arch* arch, const void* packed, const void* acts, const float* d_raw,
                             int64_t M, void* dz, float* grads, float* d_x, void* stream) {
  NERF_ARCH_CHECK_ANY("nerf_mlp_backward");
  NERF_REQUIRE(!d_x || arch_kind(arch) == 2, NERF_E_UNSUPPORTED,
               "nerf_mlp_backward_inputs: input gradients exist for the (2x64) model only (the 8x256 models take fixed encodings)");
  NERF_REQUIRE(packed && acts && d_raw && dz && grads, NERF_E_NULL, "nerf_mlp_backward: NULL pointer");
  NERF_REQUIRE(M > 0, NERF_E_SHAPE, "nerf_mlp_backward: M must be > 0");
  auto s = as_stream(stream);
  const int64_t ntiles = (M + 31) / 32;
  if (arch_kind(arch) == 2) {
    const bool sp = arch_s16(arch);
    int rcs;
    if (sp) {
      rcs = s16x::small_backward_chain(static_cast<const char*>(packed) + LN::PACKED_BYTES, acts, d_raw, M, dz, d_x,
                                       small_astride16(true), small_zstride16(true), s);
    } else {
      SmallArgs a;
      small_args(a, packed);
      a.d_raw = d_raw; a.acts = const_cast<void*>(acts); a.dz = dz; a.M = M; a.d_x = d_x;
      const int64_t nwg = (ntiles + 7) / 8;
      hipLaunchKernelGGL(mlp_small_bwd_kernel, dim3((unsigned)(nwg < 2048 ? nwg : 2048)), dim3(512), LN::LDS_BYTES, s, a);
      rcs = check_launch("mlp backward chain (2x64 model)");
    }
    if (rcs) return rcs;
    DwArgs ds;
    int njs = 0;
    auto jobs = [&](int dz_slot, int nf, int act_slot, int kf, int w_off, int ldw, int col0, int nv, int kv, int b_off) {
      ds.jobs[njs++] = DwJob{dz_slot, nf, act_slot, kf, w_off, ldw, col0, nv, kv, b_off};
    };
    jobs(LN::Z_L0, 4, LN::A_X, 2, LN::P_W0, 32, 0, 64, 32, LN::P_B0);          // pos0
    jobs(LN::Z_L1, 4, LN::A_H0, 4, LN::P_W1, 64, 0, 64, 64, LN::P_B1);         // pos1
    jobs(LN::Z_F, 4, LN::A_H1, 4, LN::P_WF, 64, 0, 64, 64, LN::P_BF);          // feature
    jobs(LN::Z_A, 1, LN::A_H1, 4, LN::P_WA, 64, 0, 1, 64, LN::P_BA);           // alpha
    jobs(LN::Z_D, 2, LN::A_FEAT, 4, LN::P_WD, 80, 0, 32, 64, LN::P_BD);        // dir0 | feature
    jobs(LN::Z_D, 2, LN::A_DX, 1, LN::P_WD, 80, 64, 32, 16, -1);               // dir0 | direction features
    jobs(LN::Z_RGB, 1, LN::A_HD, 2, LN::P_WR, 32, 0, 3, 32, LN::P_BR);         // rgb
    return launch_dw(ds, njs, ntiles, LN::P_TOTAL, acts, dz, small_astride16(sp), small_zstride16(sp), grads, s, sp, s16x::SM_A_LO,
                     s16x::SM_Z_LO);
  }
  if (arch_kind(arch) == 1) {
    if (arch_f32(arch))
      return f32::backward(static_cast<const char*>(packed) + IMG_BF16_BYTES, acts, d_raw, M, dz, grads, arch->out_ch, s);
    const bool sp = arch_s16(arch);
    int rci;
    if (sp) {
      rci = s16x::img_backward_chain(static_cast<const char*>(packed) + IMG_BF16_BYTES, acts, d_raw, M, arch->out_ch, dz,
                                     img_astride16(true), img_zstride16(true), ring_wgs(), s);
    } else {
      ImgArgs a;
      img_args(a, arch, packed);
      a.d_out = d_raw; a.acts = const_cast<void*>(acts); a.dz = dz; a.M = M;
      const int64_t nsuper = (ntiles + 7) / 8;
      static DevOnce once_i;
      once_i.run([&] { ensure_lds(mlp_img_bwd_ring_kernel, RING_LDS_BYTES); });
      hipLaunchKernelGGL(mlp_img_bwd_ring_kernel, dim3((unsigned)(nsuper < ring_wgs() ? nsuper : ring_wgs())), dim3(512),
                         RING_LDS_BYTES, s, a);
      rci = check_launch("mlp backward chain (image model)");
    }
    if (rci) return rci;
    DwArgs di;
    int nji = 0;
    auto jobi = [&](int dz_slot, int nf, int act_slot, int kf, int w_off, int ldw, int col0, int nv, int kv, int b_off) {
      di.jobs[nji++] = DwJob{dz_slot, nf, act_slot, kf, w_off, ldw, col0, nv, kv, b_off};
    };
    jobi(LI::Z_L0, 16, LI::A_X, 3, LI::P_W0, 40, 0, 256, 40, LI::P_B0);                                    // pos0
    for (int l = 1; l <= 4; ++l)
      jobi(LI::Z_L0 + 16 * l, 16, LI::A_H0 + 16 * (l - 1), 16, LI::pw(l), 256, 0, 256, 256, LI::pb(l));    // pos1..4
    jobi(LI::Z_L0 + 80, 16, LI::A_H0 + 64, 16, LI::P_W5, 296, 40, 256, 256, LI::P_B5);                     // pos5 | H4
    jobi(LI::Z_L0 + 80, 16, LI::A_X, 3, LI::P_W5, 296, 0, 256, 40, -1);                                    // pos5 | x
    jobi(LI::Z_L0 + 96, 16, LI::A_H0 + 80, 16, LI::P_W6, 256, 0, 256, 256, LI::P_B6);                      // pos6
    jobi(LI::Z_L0 + 112, 16, LI::A_H0 + 96, 16, LI::P_W7, 256, 0, 256, 256, LI::P_B7);                     // pos7
    jobi(LI::Z_OUT, 1, LI::A_H0 + 112, 16, LI::P_WO, 256, 0, arch->out_ch, 256, LI::P_WO + arch->out_ch * 256);   // output
    return launch_dw(di, nji, ntiles, img_params(arch), acts, dz, img_astride16(sp), img_zstride16(sp), grads, s, sp, s16x::IMG_A_LO,
                     s16x::IMG_Z_LO);
  }
  if (arch_f32(arch)) {
    return f32::backward(packed32_of(packed), acts, d_raw, M, dz, grads, 0, s);
  }
  const bool split = arch_s16(arch);
  const int64_t astr = split ? s16_astride16() : astride16(), zstr = split ? s16_zstride16() : zstride16();
  if (split && g_bwd_stage != 2) {
    int rcs = s16::backward_chain(packed_s16_of(packed), acts, d_raw, M, dz, astr, zstr, ring_wgs(), s);
    if (rcs) return rcs;
  }
  // ---- 1. dZ chain
  BwdArgs b;
  b.queue = nullptr;
  b.wb = reinterpret_cast<const bf16x8*>(static_cast<const char*>(packed) + (size_t)L::F_TOTAL * 1024);
  b.acts = acts; b.d_raw = d_raw; b.M = M; b.dz = dz; b.astride = astride16(); b.zstride = zstride16();
  const int variant = g_mlp_variant == 0 ? 3 : g_mlp_variant;
  if (split) {
  } else if (variant >= 3) {
    b.queue = passq_slot();
    if (g_bwd_stage != 2 && g_ring_split == 2) {
      const int64_t nsuper = (ntiles + SPLIT_NW - 1) / SPLIT_NW;
      const int64_t wgs = 2 * (int64_t)ring_wgs();
      static DevOnce once2;
      once2.run([&] { ensure_lds(mlp_bwd_ring_kernel<SPLIT_NW, SPLIT_CHUNK>, SPLIT_LDS_BYTES); });
      hipLaunchKernelGGL((mlp_bwd_ring_kernel<SPLIT_NW, SPLIT_CHUNK>), dim3((unsigned)(nsuper < wgs ? nsuper : wgs)),
                         dim3(64 * SPLIT_NW), SPLIT_LDS_BYTES, s, b);
    } else if (g_bwd_stage != 2) {
      const int64_t nsuper = (ntiles + 7) / 8;
      static DevOnce once;
      once.run([&] { ensure_lds(mlp_bwd_ring_kernel<>, RING_LDS_BYTES); });
      hipLaunchKernelGGL(mlp_bwd_ring_kernel<>, dim3((unsigned)(nsuper < ring_wgs() ? nsuper : ring_wgs())), dim3(512),
                         RING_LDS_BYTES, s, b);
    }
  } else {
    const int st = variant == 2 ? 2 : 1;
    const int64_t blocks = (ntiles + 4 * st - 1) / (4 * st);
    if (st == 1) hipLaunchKernelGGL((mlp_bwd_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, b);
    else hipLaunchKernelGGL((mlp_bwd_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, b);
  }
  int rc = check_launch("mlp backward chain");
  if (rc) return rc;
  if (g_bwd_stage == 1) return NERF_OK;
  // ---- 2. dW / db
  DwArgs d;
  int nj = 0, jseq = 0;
  auto job = [&](int dz_slot, int nf, int act_slot, int kf, int w_off, int ldw, int col0, int nv, int kv, int b_off) {
    if (g_dw_job_mask && !((g_dw_job_mask >> jseq++) & 1)) return;
    d.jobs[nj++] = DwJob{dz_slot, nf, act_slot, kf, w_off, ldw, col0, nv, kv, b_off};
  };
  job(L::Z_L0, 16, L::A_PE, 4, L::P_W0, 63, 0, 256, 63, L::P_B0);                                  // pos0
  for (int l = 1; l <= 4; ++l)
    job(L::Z_L0 + 16 * l, 16, L::A_H0 + 16 * (l - 1), 16, L::pw(l), 256, 0, 256, 256, L::pb(l));   // pos1..4
  job(L::Z_L0 + 80, 16, L::A_H0 + 64, 16, L::P_W5, 319, 63, 256, 256, L::P_B5);                    // pos5 | H4
  job(L::Z_L0 + 80, 16, L::A_PE, 4, L::P_W5, 319, 0, 256, 63, -1);                                 // pos5 | PE
  job(L::Z_L0 + 96, 16, L::A_H0 + 80, 16, L::P_W6, 256, 0, 256, 256, L::P_B6);                     // pos6
  job(L::Z_L0 + 112, 16, L::A_H0 + 96, 16, L::P_W7, 256, 0, 256, 256, L::P_B7);                    // pos7
  job(L::Z_F, 16, L::A_H0 + 112, 16, L::P_WF, 256, 0, 256, 256, L::P_BF);                          // feature
  job(L::Z_A, 1, L::A_H0 + 112, 16, L::P_WA, 256, 0, 1, 256, L::P_BA);                             // alpha
  job(L::Z_D, 8, L::A_FEAT, 16, L::P_WD, 283, 0, 128, 256, L::P_BD);                               // dir0 | feature
  job(L::Z_D, 8, L::A_DPE, 2, L::P_WD, 283, 256, 128, 27, -1);                                     // dir0 | dirPE
  job(L::Z_RGB, 1, L::A_HD, 8, L::P_WR, 128, 0, 3, 128, L::P_BR);                                  // rgb
  return launch_dw(d, nj, ntiles, L::P_TOTAL, acts, dz, astr, zstr, grads, s, split, s16::A_LO, s16::Z_LO);
}

extern "C" int nerf_ngp_query_fused_h(const nerf_mlp_arch* arch, const void* packed, const float* rays, const float* z,
                                      int64_t B, int n, const float* tables, const void* tables_half, int L, int log2_T, int F,
                                      const int* resolutions_host, int sh_degree, float pos_scale, float pos_offset,
                                      float* raw, void* acts, void* stream) {
  NERF_REQUIRE(arch_kind(arch) == 2, NERF_E_UNSUPPORTED, "nerf_ngp_query_fused: needs the (2x64, in 32+16) model");
  NERF_REQUIRE(L == 16 && F == 2 && sh_degree == 3, NERF_E_UNSUPPORTED,
               "nerf_ngp_query_fused: fused rows exist for 16 levels x 2 features + SH degree 3 (use nerf_ngp_encode + nerf_mlp_forward otherwise)");
  NERF_REQUIRE(log2_T >= 1 && log2_T <= 30, NERF_E_SHAPE, "nerf_ngp_query_fused: need 1<=log2_T<=30");
  if (B <= 0 || n <= 0) return NERF_OK;
  NERF_REQUIRE(packed && rays && z && tables && resolutions_host && raw, NERF_E_NULL, "nerf_ngp_query_fused: NULL pointer");
  const int64_t M = B * n;
  NERF_REQUIRE(M < (1ll << 31), NERF_E_SHAPE, "nerf_ngp_query_fused: B*n must be < 2^31");
  if (arch_s16(arch)) {         // reference tolerance: float32 gathers from the master tables (the fp16 shadow is a reduced-precision
    s16x::SmallQuery q;         // image: not read in this mode), float32 interpolation, split-bf16 MLP (mlp_s16x.hip)
    q.rays = rays; q.z = z; q.n = n; q.tables = tables; q.T = 1u << log2_T; q.pos_scale = pos_scale; q.pos_offset = pos_offset;
    for (int l = 0; l < 32; ++l) q.res[l] = l < L ? (float)resolutions_host[l] : 0.0f;
    q.B = B; q.ray_major = g_ngp_ray_major;
    return s16x::small_forward(static_cast<const char*>(packed) + LN::PACKED_BYTES, small_bias_of(packed), nullptr, M, raw, acts,
                               small_astride16(true), &q, as_stream(stream));
  }
  SmallArgs a;
  small_args(a, packed);
  a.out = raw; a.acts = acts; a.M = M; a.rays = rays; a.z = z; a.n = n; a.tables = tables; a.T = 1u << log2_T;
  a.tables_h = static_cast<const uint32_t*>(tables_half);
  a.pos_scale = pos_scale; a.pos_offset = pos_offset;
  for (int l = 0; l < 32; ++l) a.rt.res[l] = l < L ? (float)resolutions_host[l] : 0.0f;
  a.B = B;
  a.ray_major = (!acts && g_ngp_ray_major && B >= 32) ? 1 : 0;
  const int64_t nwg = ((a.ray_major ? ((B + 31) / 32) * (int64_t)n : (M + 31) / 32) + 7) / 8;
  const dim3 g((unsigned)(nwg < 2048 ? nwg : 2048)), b(512);
  if (acts) hipLaunchKernelGGL((mlp_small_fwd_kernel<true, true>), g, b, LN::LDS_BYTES, as_stream(stream), a);
  else hipLaunchKernelGGL((mlp_small_fwd_kernel<false, true>), g, b, LN::LDS_BYTES, as_stream(stream), a);
  return check_launch("nerf_ngp_query_fused");
}

extern "C" int nerf_ngp_query_fused(const nerf_mlp_arch* arch, const void* packed, const float* rays, const float* z,
                                    int64_t B, int n, const float* tables, int L, int log2_T, int F,
                                    const int* resolutions_host, int sh_degree, float pos_scale, float pos_offset,
                                    float* raw, void* acts, void* stream) {
  return nerf_ngp_query_fused_h(arch, packed, rays, z, B, n, tables, nullptr, L, log2_T, F, resolutions_host, sh_degree, pos_scale,
                                pos_offset, raw, acts, stream);
}

extern "C" int nerf_mlp_backward(const nerf_mlp_arch* arch, const void* packed, const void* acts, const float* d_raw,
                                 int64_t M, void* dz, float* grads, void* stream) {
  return mlp_backward_impl(arch, packed, acts, d_raw, M, dz, grads, nullptr, stream);
}

extern "C" int nerf_mlp_backward_inputs(const nerf_mlp_arch* arch, const void* packed, const void* acts,
                                        const float* d_raw, int64_t M, void* dz, float* grads, float* d_x,
                                        void* stream) {
  NERF_REQUIRE(d_x, NERF_E_NULL, "nerf_mlp_backward_inputs: d_x is NULL");
  return mlp_backward_impl(arch, packed, acts, d_raw, M, dz, grads, d_x, stream);
}

// ---- test hook: one layer of the training stores (fragment blocks) as row-major fp32 ----------------------------
namespace nerf {
__global__ void __launch_bounds__(256) decode_frags_kernel(const void* base, int64_t stride16, int slot0, int nfrag,
                                                           int64_t M, float* __restrict__ out, int lo_off) {
  const int64_t total = M * nfrag * 2;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int h = (int)(t & 1), ks = (int)((t >> 1) % nfrag);
    const int64_t m = t / (2 * nfrag);
    const bf16x8 v = *frag_ptr(const_cast<void*>(base), m >> 5, stride16, slot0 + ks, (int)(m & 31), h);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[m * (16 * nfrag) + kperm(ks, h, j)] = (float)v[j];
    if (lo_off > 0) {           // split-bf16 stores: value = hi block + lo block
      const bf16x8 w = *frag_ptr(const_cast<void*>(base), m >> 5, stride16, slot0 + lo_off + ks, (int)(m & 31), h);
#pragma unroll
      for (int j = 0; j < 8; ++j) out[m * (16 * nfrag) + kperm(ks, h, j)] += (float)w[j];
    }
  }
}
// (slot, fragments) of `layer` in the activation (kind 0) or dZ (kind 1) store of an architecture (arch_kind 0 / 1 / 2)
static bool debug_slot(int ak, int kind, int layer, int* slot, int* nfrag) {
  if (layer < 0 || layer > 11 || (kind != 0 && kind != 1)) return false;
  if (ak == 1) {                       // image model: pos0..7; 8 = output gradient; 10 = input rows (48 = 40 + pad)
    if (layer < 8) { *slot = (kind == 0 ? LI::A_H0 : LI::Z_L0) + 16 * layer; *nfrag = 16; return true; }
    if (kind == 0 && layer == 10) { *slot = LI::A_X; *nfrag = 3; return true; }
    if (kind == 1 && layer == 8) { *slot = LI::Z_OUT; *nfrag = 1; return true; }
    return false;
  }
  if (ak == 2) {                       // 2 x 64 model: pos0, pos1; 8 = feature; 9 = dir0; 10 / 11 = inputs (acts) or d alpha / d rgb (dz)
    if (kind == 0) {
      if (layer == 0) { *slot = LN::A_H0; *nfrag = 4; } else if (layer == 1) { *slot = LN::A_H1; *nfrag = 4; }
      else if (layer == 8) { *slot = LN::A_FEAT; *nfrag = 4; } else if (layer == 9) { *slot = LN::A_HD; *nfrag = 2; }
      else if (layer == 10) { *slot = LN::A_X; *nfrag = 2; } else if (layer == 11) { *slot = LN::A_DX; *nfrag = 1; }
      else return false;
      return true;
    }
    if (layer == 0) { *slot = LN::Z_L0; *nfrag = 4; } else if (layer == 1) { *slot = LN::Z_L1; *nfrag = 4; }
    else if (layer == 8) { *slot = LN::Z_F; *nfrag = 4; } else if (layer == 9) { *slot = LN::Z_D; *nfrag = 2; }
    else if (layer == 10) { *slot = LN::Z_A; *nfrag = 1; } else if (layer == 11) { *slot = LN::Z_RGB; *nfrag = 1; }
    else return false;
    return true;
  }
  if (kind == 0) {
    if (layer < 8) { *slot = L::A_H0 + 16 * layer; *nfrag = 16; }
    else if (layer == 8) { *slot = L::A_FEAT; *nfrag = 16; }
    else if (layer == 9) { *slot = L::A_HD; *nfrag = 8; }
    else if (layer == 10) { *slot = L::A_PE; *nfrag = 4; }
    else { *slot = L::A_DPE; *nfrag = 2; }
    return true;
  }
  if (layer < 8) { *slot = L::Z_L0 + 16 * layer; *nfrag = 16; }
  else if (layer == 8) { *slot = L::Z_F; *nfrag = 16; }
  else if (layer == 9) { *slot = L::Z_D; *nfrag = 8; }
  else if (layer == 10) { *slot = L::Z_A; *nfrag = 1; }
  else { *slot = L::Z_RGB; *nfrag = 1; }
  return true;
}
}  // namespace nerf

extern "C" int nerf_mlp_debug_width(const nerf_mlp_arch* arch, int kind, int layer) {
  int slot = 0, nfrag = 0;
  const int ak = arch_kind(arch);
  if (ak < 0) return -1;
  if (arch_f32(arch)) return f32::debug_width(kind, layer);
  if (!debug_slot(ak, kind, layer, &slot, &nfrag)) return -1;
  return 16 * nfrag;
}

extern "C" int nerf_mlp_debug_read(const nerf_mlp_arch* arch, const void* store, int kind, int layer, int64_t M,
                                   float* out, void* stream) {
  NERF_ARCH_CHECK_ANY("nerf_mlp_debug_read");
  if (arch_f32(arch)) {
    NERF_REQUIRE(store && out, NERF_E_NULL, "nerf_mlp_debug_read: NULL pointer");
    return M <= 0 ? NERF_OK : f32::debug_read(store, kind, layer, M, out, as_stream(stream));
  }
  int slot = 0, nfrag = 0;
  const int ak = arch_kind(arch);
  NERF_REQUIRE(debug_slot(ak, kind, layer, &slot, &nfrag), NERF_E_SHAPE, "nerf_mlp_debug_read: no such (kind, layer) in this architecture's stores");
  NERF_REQUIRE(store && out, NERF_E_NULL, "nerf_mlp_debug_read: NULL pointer");
  if (M <= 0) return NERF_OK;
  const bool sp = arch_s16(arch);
  const int64_t stride = ak == 0 ? (sp ? (kind == 0 ? s16_astride16() : s16_zstride16()) : (kind == 0 ? astride16() : zstride16()))
                         : ak == 1 ? (kind == 0 ? img_astride16(sp) : img_zstride16(sp)) : (kind == 0 ? small_astride16(sp) : small_zstride16(sp));
  const int lo_off = !sp ? 0 : ak == 0 ? (kind == 0 ? s16::A_LO : s16::Z_LO) : ak == 1 ? (kind == 0 ? s16x::IMG_A_LO : s16x::IMG_Z_LO)
                                                                                     : (kind == 0 ? s16x::SM_A_LO : s16x::SM_Z_LO);
  hipLaunchKernelGGL(decode_frags_kernel, dim3(grid_for(M * nfrag * 2, 256)), dim3(256), 0, as_stream(stream), store, stride, slot, nfrag,
                     M, out, lo_off);
  return check_launch("nerf_mlp_debug_read");
}

NERF_STAMP2_EXPORT(nerf_debug_stamps2_ring16)
#ifdef NERF_CLOCK_STAMP
extern "C" int nerf_debug_stamps(unsigned long long* host_out, int nwg) {
  if (nwg > 4096) nwg = 4096;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nerf::g_stamps), (size_t)nwg * 32) == hipSuccess ? 0 : -4;
}
#endif
