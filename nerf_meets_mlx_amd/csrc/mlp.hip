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
#include "common.h"
#include <string.h>

namespace nerf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------
// static layout of the one supported architecture (8 x 256, skip 4, view head)
// ------------------------------------------------------------------------------------------
namespace L {
// float32 parameter offsets (include/nerf_hip.h "Parameter layout")
constexpr int P_W0 = 0, P_B0 = 16128;
constexpr int P_W1 = 16384;                       // W_l = P_W1 + (l-1)*65792, l = 1..4
constexpr int P_W5 = 279552, P_B5 = 361216;
constexpr int P_W6 = 361472, P_B6 = 427008, P_W7 = 427264, P_B7 = 492800;
constexpr int P_WF = 493056, P_BF = 558592, P_WA = 558848, P_BA = 559104;
constexpr int P_WD = 559105, P_BD = 595329, P_WR = 595457, P_BR = 595841;
constexpr int P_TOTAL = 595844;
__host__ __device__ constexpr int pw(int l) {   // weight offset of pos layer l
  return l == 0 ? P_W0 : l <= 4 ? P_W1 + (l - 1) * 65792 : l == 5 ? P_W5 : l == 6 ? P_W6 : P_W7;
}
__host__ __device__ constexpr int pb(int l) {
  return l == 0 ? P_B0 : l <= 4 ? P_W1 + (l - 1) * 65792 + 65536 : l == 5 ? P_B5 : l == 6 ? P_B6 : P_B7;
}
__host__ __device__ constexpr int pin(int l) { return l == 0 ? 63 : l == 5 ? 319 : 256; }

// forward weight stream, 1 KiB fragments in consumption order
constexpr int F_L0 = 0, F_L1 = 32, F_L5 = 544, F_L6 = 704, F_L7 = 832, F_FA = 960, F_DIR = 1104, F_RGB = 1176;
constexpr int F_TOTAL = 1184;
// backward (transposed) weight stream
constexpr int B_RGB = 0, B_DIR = 4, B_FA = 68, B_L7 = 204, B_L6 = 332, B_L5 = 460, B_L4 = 588;
constexpr int B_TOTAL = 1100;
// fp32 bias slots
constexpr int BI_FEAT = 2048, BI_ALPHA = 2304, BI_DIR = 2336, BI_RGB = 2464, BI_TOTAL = 2496;
constexpr int64_t PACKED_BYTES = (int64_t)(F_TOTAL + B_TOTAL) * 1024 + BI_TOTAL * 4;

// activation store: fragment slots per 32-sample tile
constexpr int A_PE = 0, A_DPE = 4, A_H0 = 6;       // H_l at A_H0 + 16 l, l = 0..7
constexpr int A_FEAT = 134, A_HD = 150, A_SLOTS = 158;
// gradient store
constexpr int Z_L0 = 0;                            // dZ_l at 16 l, l = 0..7
constexpr int Z_F = 128, Z_A = 144, Z_D = 145, Z_RGB = 153, Z_SLOTS = 154;
}  // namespace L

// element j of lane half h in k-step ks  <->  feature index
__host__ __device__ constexpr int kperm(int ks, int h, int j) { return 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3); }

// ------------------------------------------------------------------------------------------
// weight packing: fp32 master parameters -> bf16 MFMA fragments (+ fp32 bias slots)
// ------------------------------------------------------------------------------------------
__device__ float fwd_src(const float* __restrict__ p, int f, int r, int h, int j) {
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
__device__ float bwd_src(const float* __restrict__ p, int f, int r, int h, int j) {
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

__global__ void __launch_bounds__(256) pack_kernel(const float* __restrict__ p, bf16x8* __restrict__ wf,
                                                   bf16x8* __restrict__ wb, float* __restrict__ bias) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int nf = L::F_TOTAL * 64, nb = L::B_TOTAL * 64;
  if (tid < nf + nb) {
    const bool fw = tid < nf;
    const int t = fw ? tid : tid - nf;
    const int f = t >> 6, lane = t & 63, r = lane & 31, h = lane >> 5;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)(fw ? fwd_src(p, f, r, h, j) : bwd_src(p, f, r, h, j));
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
struct PeFreq { float pos[10]; float dir[4]; };

struct Chan { int kind, dim, band; };   // kind: 0 identity, 1 sin, 2 cos, 3 zero pad
__host__ __device__ constexpr Chan chan_of(int c, int limit) {
  if (c < 3) return Chan{0, c, 0};
  if (c >= limit) return Chan{3, 0, 0};
  return Chan{((c - 3) % 6) >= 3 ? 2 : 1, (c - 3) % 3, (c - 3) / 6};
}

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

// ------------------------------------------------------------------------------------------
// one linear layer on register-resident activations
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void acc_init_bias(f32x16& acc, const float* __restrict__ bias_tile, int h) {
  // register i <-> row (i&3) + 8 (i>>2) + 4 h : four float4 at rows 8g + 4h
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 b = *reinterpret_cast<const float4*>(bias_tile + 8 * g + 4 * h);
    acc[4 * g + 0] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
  }
}

template <bool RELU>
__device__ __forceinline__ void acc_to_frags(const f32x16& acc, bf16x8& lo, bf16x8& hi) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float a = acc[j], b = acc[8 + j];
    lo[j] = (__bf16)(RELU ? fmaxf(a, 0.0f) : a);
    hi[j] = (__bf16)(RELU ? fmaxf(b, 0.0f) : b);
  }
}

// out[t][2 nt + s] = act( W[nt-tile] . in[t] + bias )     weights: wlane = stream base + lane
template <int ST, int KS, int NT, bool RELU>
__device__ __forceinline__ void layer_fwd(const bf16x8* __restrict__ wlane, const float* __restrict__ bias,
                                          const bf16x8 (&in)[ST][KS], bf16x8 (&out)[ST][2 * NT], int h) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    f32x16 acc[ST];
    acc_init_bias(acc[0], bias + 32 * nt, h);
#pragma unroll
    for (int t = 1; t < ST; ++t) acc[t] = acc[0];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 a = wlane[(nt * KS + ks) * 64];
#pragma unroll
      for (int t = 0; t < ST; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, in[t][ks], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < ST; ++t) acc_to_frags<RELU>(acc[t], out[t][2 * nt], out[t][2 * nt + 1]);
  }
}

// fragment block address: tile T, slot s, lane (r,h) at byte 32 r + 16 h
__device__ __forceinline__ bf16x8* frag_ptr(void* base, int64_t tile, int slots, int slot, int r, int h) {
  return reinterpret_cast<bf16x8*>(base) + ((tile * slots + slot) * 64 + 2 * r + h);
}

template <int COUNT>
__device__ __forceinline__ void store_frags(void* base, int64_t tile, int slots, int slot0, const bf16x8 (&frags)[COUNT],
                                            int r, int h) {
#pragma unroll
  for (int k = 0; k < COUNT; ++k) *frag_ptr(base, tile, slots, slot0 + k, r, h) = frags[k];
}

struct FwdArgs {
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
};

// MODE 0: embedded rows;  MODE 1: rays + z with fused positional encodings
template <int ST, int MODE, bool STORE>
__global__ void __launch_bounds__(256, (ST == 1 ? 2 : 1)) mlp_fwd_kernel(FwdArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wv) * ST;
  const int64_t ntiles = (a.M + 31) >> 5;
  if (tile0 >= ntiles) return;
  const bf16x8* __restrict__ w = a.wf + lane;

  bf16x8 pe[ST][4], dpe[ST][2];
  int64_t m_idx[ST];
#pragma unroll
  for (int t = 0; t < ST; ++t) {
    int64_t tile = tile0 + t; if (tile >= ntiles) tile = ntiles - 1;       // duplicate work, never stored
    int64_t m = tile * 32 + r; if (m >= a.M) m = a.M - 1;
    m_idx[t] = m;
    if (MODE == 0) {
      const float* row = a.x + m * 90;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) pe[t][ks] = row_frag(row, ks, h, 63);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) dpe[t][ks] = row_frag(row + 63, ks, h, 27);
    } else {
      const int64_t ray = m / a.n;
      const float* rr = a.rays + ray * NERF_RAY_STRIDE;
      const float zv = a.z[m];
      float p[3], d[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { p[c] = rr[c] + zv * rr[3 + c]; d[c] = rr[8 + c]; }   // render.py:142
      pe[t][0] = pe_frag<0, 63, 10>(p, a.fr.pos, h); pe[t][1] = pe_frag<1, 63, 10>(p, a.fr.pos, h);
      pe[t][2] = pe_frag<2, 63, 10>(p, a.fr.pos, h); pe[t][3] = pe_frag<3, 63, 10>(p, a.fr.pos, h);
      dpe[t][0] = pe_frag<0, 27, 4>(d, a.fr.dir, h); dpe[t][1] = pe_frag<1, 27, 4>(d, a.fr.dir, h);
    }
  }
#define store(slot0, t, frags, count) \
  do { if (STORE && tile0 + (t) < ntiles) store_frags<count>(a.acts, tile0 + (t), L::A_SLOTS, slot0, frags, r, h); } while (0)
#pragma unroll
  for (int t = 0; t < ST; ++t) { store(L::A_PE, t, pe[t], 4); store(L::A_DPE, t, dpe[t], 2); }

  bf16x8 ha[ST][16], hb[ST][16];
  layer_fwd<ST, 4, 8, true>(w + L::F_L0 * 64, a.bias + 0, pe, ha, h);
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::A_H0, t, ha[t], 16);
  // pos1..pos4 (ping-pong)
  layer_fwd<ST, 16, 8, true>(w + (L::F_L1 + 0 * 128) * 64, a.bias + 256, ha, hb, h);
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::A_H0 + 16, t, hb[t], 16);
  layer_fwd<ST, 16, 8, true>(w + (L::F_L1 + 1 * 128) * 64, a.bias + 512, hb, ha, h);
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::A_H0 + 32, t, ha[t], 16);
  layer_fwd<ST, 16, 8, true>(w + (L::F_L1 + 2 * 128) * 64, a.bias + 768, ha, hb, h);
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::A_H0 + 48, t, hb[t], 16);
  layer_fwd<ST, 16, 8, true>(w + (L::F_L1 + 3 * 128) * 64, a.bias + 1024, hb, ha, h);
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::A_H0 + 64, t, ha[t], 16);
  // pos5 on concat[input_pos, h]  (models/NeRF.py:224-225)
  {
    bf16x8 cat[ST][20];
#pragma unroll
    for (int t = 0; t < ST; ++t) {
#pragma unroll
      for (int k = 0; k < 4; ++k) cat[t][k] = pe[t][k];
#pragma unroll
      for (int k = 0; k < 16; ++k) cat[t][4 + k] = ha[t][k];
    }
    layer_fwd<ST, 20, 8, true>(w + L::F_L5 * 64, a.bias + 1280, cat, hb, h);
  }
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::A_H0 + 80, t, hb[t], 16);
  layer_fwd<ST, 16, 8, true>(w + L::F_L6 * 64, a.bias + 1536, hb, ha, h);
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::A_H0 + 96, t, ha[t], 16);
  layer_fwd<ST, 16, 8, true>(w + L::F_L7 * 64, a.bias + 1792, ha, hb, h);
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::A_H0 + 112, t, hb[t], 16);
  // feature (no activation) and alpha (row 0 of a ninth tile)   models/NeRF.py:229-231
  layer_fwd<ST, 16, 8, false>(w + L::F_FA * 64, a.bias + L::BI_FEAT, hb, ha, h);
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::A_FEAT, t, ha[t], 16);
  float alpha[ST];
  {
    f32x16 acc[ST];
    acc_init_bias(acc[0], a.bias + L::BI_ALPHA, h);
#pragma unroll
    for (int t = 1; t < ST; ++t) acc[t] = acc[0];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const bf16x8 wa = w[(L::F_FA + 128 + ks) * 64];
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
      cat[t][16] = dpe[t][0]; cat[t][17] = dpe[t][1];
    }
    layer_fwd<ST, 18, 4, true>(w + L::F_DIR * 64, a.bias + L::BI_DIR, cat, hd, h);
  }
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::A_HD, t, hd[t], 8);
  {
    f32x16 acc[ST];
    acc_init_bias(acc[0], a.bias + L::BI_RGB, h);
#pragma unroll
    for (int t = 1; t < ST; ++t) acc[t] = acc[0];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 wr = w[(L::F_RGB + ks) * 64];
#pragma unroll
      for (int t = 0; t < ST; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr, hd[t][ks], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < ST; ++t) {
      const int64_t m = (tile0 + t) * 32 + r;
      if (h == 0 && tile0 + t < ntiles && m < a.M) {
        float4 o; o.x = acc[t][0]; o.y = acc[t][1]; o.z = acc[t][2]; o.w = alpha[t];     // [rgb, alpha] raw
        *reinterpret_cast<float4*>(a.out + m * 4) = o;
      }
    }
  }
  (void)m_idx;
#undef store
}

// ------------------------------------------------------------------------------------------
// backward chain: dZ_l for every layer (stored as fragment blocks for the dW kernel)
// ------------------------------------------------------------------------------------------
// out[t][2 kt + s] = mask( W^T[kt-tile] . in[t] )
template <int ST, int NS, int KT, bool MASK>
__device__ __forceinline__ void layer_bwd(const bf16x8* __restrict__ wlane, const bf16x8 (&in)[ST][NS],
                                          bf16x8 (&out)[ST][2 * KT], const void* acts, const int64_t (&tile)[ST],
                                          int mask_slot, int r, int h) {
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    f32x16 acc[ST];
#pragma unroll
    for (int t = 0; t < ST; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      const bf16x8 a = wlane[(kt * NS + ns) * 64];
#pragma unroll
      for (int t = 0; t < ST; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, in[t][ns], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < ST; ++t) {
      if (MASK) {   // ReLU': the stored activation fragment has the accumulator's (register, lane) layout
        const bf16x8 m0 = *frag_ptr(const_cast<void*>(acts), tile[t], L::A_SLOTS, mask_slot + 2 * kt, r, h);
        const bf16x8 m1 = *frag_ptr(const_cast<void*>(acts), tile[t], L::A_SLOTS, mask_slot + 2 * kt + 1, r, h);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          out[t][2 * kt][j] = (__bf16)((float)m0[j] > 0.0f ? acc[t][j] : 0.0f);
          out[t][2 * kt + 1][j] = (__bf16)((float)m1[j] > 0.0f ? acc[t][8 + j] : 0.0f);
        }
      } else {
        acc_to_frags<false>(acc[t], out[t][2 * kt], out[t][2 * kt + 1]);
      }
    }
  }
}

struct BwdArgs {
  const bf16x8* wb;
  const void* acts;
  const float* d_raw;    // [M,4]
  int64_t M;
  void* dz;
};

template <int ST>
__global__ void __launch_bounds__(256, (ST == 1 ? 2 : 1)) mlp_bwd_kernel(BwdArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wv) * ST;
  const int64_t ntiles = (a.M + 31) >> 5;
  if (tile0 >= ntiles) return;
  const bf16x8* __restrict__ w = a.wb + lane;
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
    zal[t][0][0] = (__bf16)g.w;                                                               // row 0
  }
#define store(slot0, t, frags, count) \
  do { if (live[t]) store_frags<count>(a.dz, tile[t], L::Z_SLOTS, slot0, frags, r, h); } while (0)
#pragma unroll
  for (int t = 0; t < ST; ++t) { store(L::Z_RGB, t, zrgb[t], 1); store(L::Z_A, t, zal[t], 1); }

  bf16x8 zd[ST][8];
  layer_bwd<ST, 1, 4, true>(w + L::B_RGB * 64, zrgb, zd, a.acts, tile, L::A_HD, r, h);
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::Z_D, t, zd[t], 8);
  bf16x8 za[ST][16], zb[ST][16];
  layer_bwd<ST, 8, 8, false>(w + L::B_DIR * 64, zd, za, a.acts, tile, 0, r, h);            // d feature
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::Z_F, t, za[t], 16);
  {
    bf16x8 cat[ST][17];
#pragma unroll
    for (int t = 0; t < ST; ++t) {
#pragma unroll
      for (int k = 0; k < 16; ++k) cat[t][k] = za[t][k];
      cat[t][16] = zal[t][0];
    }
    layer_bwd<ST, 17, 8, true>(w + L::B_FA * 64, cat, zb, a.acts, tile, L::A_H0 + 112, r, h);   // dZ7
  }
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::Z_L0 + 112, t, zb[t], 16);
  layer_bwd<ST, 16, 8, true>(w + L::B_L7 * 64, zb, za, a.acts, tile, L::A_H0 + 96, r, h);       // dZ6
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::Z_L0 + 96, t, za[t], 16);
  layer_bwd<ST, 16, 8, true>(w + L::B_L6 * 64, za, zb, a.acts, tile, L::A_H0 + 80, r, h);       // dZ5
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::Z_L0 + 80, t, zb[t], 16);
  layer_bwd<ST, 16, 8, true>(w + L::B_L5 * 64, zb, za, a.acts, tile, L::A_H0 + 64, r, h);       // dZ4
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::Z_L0 + 64, t, za[t], 16);
  layer_bwd<ST, 16, 8, true>(w + (L::B_L4 + 0 * 128) * 64, za, zb, a.acts, tile, L::A_H0 + 48, r, h);   // dZ3
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::Z_L0 + 48, t, zb[t], 16);
  layer_bwd<ST, 16, 8, true>(w + (L::B_L4 + 1 * 128) * 64, zb, za, a.acts, tile, L::A_H0 + 32, r, h);   // dZ2
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::Z_L0 + 32, t, za[t], 16);
  layer_bwd<ST, 16, 8, true>(w + (L::B_L4 + 2 * 128) * 64, za, zb, a.acts, tile, L::A_H0 + 16, r, h);   // dZ1
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::Z_L0 + 16, t, zb[t], 16);
  layer_bwd<ST, 16, 8, true>(w + (L::B_L4 + 3 * 128) * 64, zb, za, a.acts, tile, L::A_H0 + 0, r, h);    // dZ0
#pragma unroll
  for (int t = 0; t < ST; ++t) store(L::Z_L0 + 0, t, za[t], 16);
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
  const void* acts;
  const void* dz;
  float* grads;
};

constexpr int DW_FRAG_STRIDE = 1152;                 // 1 KiB + 128 B: neighbouring fragments hit disjoint banks
constexpr int DW_BUF_BYTES = 32 * DW_FRAG_STRIDE;    // 16 dZ + 16 act fragments

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

__global__ void __launch_bounds__(512, 2) mlp_dw_kernel(DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bj = blockIdx.x, job_id = 0;
  while (bj >= a.splits[job_id]) { bj -= a.splits[job_id]; ++job_id; }
  const DwJob jb = a.jobs[job_id];
  const int tile_lo = (int)((int64_t)a.ntiles * bj / a.splits[job_id]);
  const int tile_hi = (int)((int64_t)a.ntiles * (bj + 1) / a.splits[job_id]);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wr = wv >> 2, wc = wv & 3;
  const int n_tiles = (jb.nf + 1) >> 1, k_tiles = (jb.kf + 1) >> 1;
  const int nf_pad = n_tiles * 2, kf_pad = k_tiles * 2;
  // this wave's output tiles: n-tiles wr*4 + (0..3), k-tiles wc*2 + (0..1)
  const bool active = (wr * 4 < n_tiles) && (wc * 2 < k_tiles);
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][k][e] = 0.0f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const int g16 = lane >> 4, i16 = lane & 15, hq = g16 >> 1, fsel = g16 & 1;
  const bf16x8* dzp = reinterpret_cast<const bf16x8*>(a.dz);
  const bf16x8* acp = reinterpret_cast<const bf16x8*>(a.acts);
  const int chunks = (nf_pad + kf_pad) * 64;           // 16-byte chunks per sample tile

  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    __syncthreads();                                   // previous tile's reads are done
    for (int c = tid; c < chunks; c += 512) {
      const int f = c >> 6, l = c & 63;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.0f;
      if (f < nf_pad) { if (f < jb.nf) v = dzp[((int64_t)tile * L::Z_SLOTS + jb.dz_slot + f) * 64 + l]; }
      else { const int g = f - nf_pad; if (g < jb.kf) v = acp[((int64_t)tile * L::A_SLOTS + jb.act_slot + g) * 64 + l]; }
      *reinterpret_cast<bf16x8*>(smem + f * DW_FRAG_STRIDE + l * 16) = v;
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bf16x8 bfr[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int kt = wc * 2 + k;
          const int f = nf_pad + (kt < k_tiles ? 2 * kt : 0) + fsel;
          bfr[k] = tr_frag(smem + f * DW_FRAG_STRIDE, u, hq, i16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int nt = wr * 4 + i;
          const int f = (nt < n_tiles ? 2 * nt : 0) + fsel;
          const bf16x8 afr = tr_frag(smem + f * DW_FRAG_STRIDE, u, hq, i16);
          if (wc == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum[i] += (float)afr[j];
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr[k], acc[i][k], 0, 0, 0);
        }
      }
    }
  }
  if (!active) return;
  const int rr = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int nt = wr * 4 + i;
    if (nt >= n_tiles) continue;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int kt = wc * 2 + k;
      if (kt >= k_tiles) continue;
      const int col = 32 * kt + rr;
      if (col >= jb.k_valid) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = 32 * nt + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (n < jb.n_valid) atomicAdd(a.grads + jb.w_off + (int64_t)n * jb.ldw + jb.col0 + col, acc[i][k][e]);
      }
    }
    if (wc == 0 && jb.b_off >= 0) {
      const float tot = bsum[i] + __shfl_xor(bsum[i], 32, 64);      // the two sample halves of the k-step
      const int n = 32 * nt + rr;
      if (hh == 0 && n < jb.n_valid) atomicAdd(a.grads + jb.b_off + n, tot);
    }
  }
}

static int g_mlp_variant = 0;   // 0: auto, 1: ST=1 (2 waves/SIMD), 2: ST=2 (1 wave/SIMD)

static bool arch_ok(const nerf_mlp_arch* a) {
  return a && a->n_layers == 8 && a->width == 256 && a->in_pos == 63 && a->in_dir == 27 && a->skip_layer == 4 &&
         a->use_viewdirs == 1;
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_set_option(const char* key, int value) {
  NERF_REQUIRE(key, NERF_E_NULL, "nerf_set_option: key is NULL");
  if (!strcmp(key, "mlp_variant")) { g_mlp_variant = value; return NERF_OK; }
  return fail(NERF_E_UNSUPPORTED, "nerf_set_option: unknown key '%s'", key);
}

extern "C" int64_t nerf_mlp_param_count(const nerf_mlp_arch* arch) { return arch_ok(arch) ? L::P_TOTAL : -1; }
extern "C" int64_t nerf_mlp_packed_bytes(const nerf_mlp_arch* arch) { return arch_ok(arch) ? L::PACKED_BYTES : -1; }
extern "C" int64_t nerf_mlp_acts_bytes(const nerf_mlp_arch* arch, int64_t M) {
  return arch_ok(arch) && M >= 0 ? ((M + 31) / 32) * (int64_t)L::A_SLOTS * 1024 : -1;
}
extern "C" int64_t nerf_mlp_dz_bytes(const nerf_mlp_arch* arch, int64_t M) {
  return arch_ok(arch) && M >= 0 ? ((M + 31) / 32) * (int64_t)L::Z_SLOTS * 1024 : -1;
}

#define NERF_ARCH_CHECK(who) \
  NERF_REQUIRE(arch_ok(arch), NERF_E_UNSUPPORTED, who ": only n_layers=8,width=256,in_pos=63,in_dir=27,skip=4,use_viewdirs=1 is implemented")

extern "C" int nerf_mlp_pack(const nerf_mlp_arch* arch, const float* params, void* packed, void* stream) {
  NERF_ARCH_CHECK("nerf_mlp_pack");
  NERF_REQUIRE(params && packed, NERF_E_NULL, "nerf_mlp_pack: params/packed is NULL");
  char* base = static_cast<char*>(packed);
  bf16x8* wf = reinterpret_cast<bf16x8*>(base);
  bf16x8* wb = reinterpret_cast<bf16x8*>(base + (size_t)L::F_TOTAL * 1024);
  float* bias = reinterpret_cast<float*>(base + (size_t)(L::F_TOTAL + L::B_TOTAL) * 1024);
  const int total = (L::F_TOTAL + L::B_TOTAL) * 64 + L::BI_TOTAL;
  hipLaunchKernelGGL(pack_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), params, wf, wb, bias);
  return check_launch("nerf_mlp_pack");
}

static void fill_freqs(PeFreq& fr, int mode) {
  for (int k = 0; k < 10; ++k) fr.pos[k] = mode == 0 ? (float)(k * k) : (float)(1 << k);
  for (int k = 0; k < 4; ++k) fr.dir[k] = mode == 0 ? (float)(k * k) : (float)(1 << k);
}

template <int MODE>
static int launch_fwd(const void* packed, const float* x, const float* rays, const float* z, int64_t M, int n,
                      int freq_mode, float* out, void* acts, void* stream) {
  FwdArgs a;
  const char* base = static_cast<const char*>(packed);
  a.wf = reinterpret_cast<const bf16x8*>(base);
  a.bias = reinterpret_cast<const float*>(base + (size_t)(L::F_TOTAL + L::B_TOTAL) * 1024);
  a.x = x; a.rays = rays; a.z = z; a.M = M; a.n = n; a.out = out; a.acts = acts;
  fill_freqs(a.fr, freq_mode);
  const int64_t ntiles = (M + 31) / 32;
  const int st = g_mlp_variant == 1 ? 1 : 2;
  const int64_t blocks = (ntiles + 4 * st - 1) / (4 * st);
  NERF_REQUIRE(blocks < (1ll << 31), NERF_E_SHAPE, "mlp forward: M too large");
  auto s = as_stream(stream);
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

extern "C" int nerf_mlp_forward(const nerf_mlp_arch* arch, const void* packed, const float* x, int64_t M, float* out,
                                void* stream) {
  NERF_ARCH_CHECK("nerf_mlp_forward");
  NERF_REQUIRE(packed && x && out, NERF_E_NULL, "nerf_mlp_forward: NULL pointer");
  if (M <= 0) return NERF_OK;
  return launch_fwd<0>(packed, x, nullptr, nullptr, M, 1, 0, out, nullptr, stream);
}

extern "C" int nerf_query_fused(const nerf_mlp_arch* arch, const void* packed, const float* rays, const float* z,
                                int64_t B, int n, int freq_mode, float* raw, void* acts, void* stream) {
  NERF_ARCH_CHECK("nerf_query_fused");
  NERF_REQUIRE(packed && rays && z && raw, NERF_E_NULL, "nerf_query_fused: NULL pointer");
  NERF_REQUIRE(n >= 1, NERF_E_SHAPE, "nerf_query_fused: n must be >= 1");
  NERF_REQUIRE(freq_mode == 0 || freq_mode == 1, NERF_E_UNSUPPORTED, "nerf_query_fused: freq_mode must be 0 or 1");
  if (B <= 0) return NERF_OK;
  return launch_fwd<1>(packed, nullptr, rays, z, B * n, n, freq_mode, raw, acts, stream);
}

extern "C" int nerf_mlp_backward(const nerf_mlp_arch* arch, const void* packed, const void* acts, const float* d_raw,
                                 int64_t M, void* dz, float* grads, void* stream) {
  NERF_ARCH_CHECK("nerf_mlp_backward");
  NERF_REQUIRE(packed && acts && d_raw && dz && grads, NERF_E_NULL, "nerf_mlp_backward: NULL pointer");
  NERF_REQUIRE(M > 0, NERF_E_SHAPE, "nerf_mlp_backward: M must be > 0");
  auto s = as_stream(stream);
  const int64_t ntiles = (M + 31) / 32;
  // ---- 1. dZ chain
  BwdArgs b;
  b.wb = reinterpret_cast<const bf16x8*>(static_cast<const char*>(packed) + (size_t)L::F_TOTAL * 1024);
  b.acts = acts; b.d_raw = d_raw; b.M = M; b.dz = dz;
  const int st = g_mlp_variant == 1 ? 1 : 2;
  const int64_t blocks = (ntiles + 4 * st - 1) / (4 * st);
  if (st == 1) hipLaunchKernelGGL((mlp_bwd_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, b);
  else hipLaunchKernelGGL((mlp_bwd_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, b);
  int rc = check_launch("mlp backward chain");
  if (rc) return rc;
  // ---- 2. dW / db
  DwArgs d;
  int nj = 0;
  auto job = [&](int dz_slot, int nf, int act_slot, int kf, int w_off, int ldw, int col0, int nv, int kv, int b_off) {
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
  // split the sample tiles of every job over workgroups in proportion to its MFMA work
  int64_t units[DW_MAX_JOBS], total_units = 0;
  for (int j = 0; j < nj; ++j) {
    units[j] = (int64_t)((d.jobs[j].nf + 1) / 2) * ((d.jobs[j].kf + 1) / 2);
    total_units += units[j];
  }
  const int target_wgs = 512;
  int nw = 0;
  for (int j = 0; j < DW_MAX_JOBS; ++j) d.splits[j] = 0;
  for (int j = 0; j < nj; ++j) {
    int64_t splits = (units[j] * target_wgs + total_units - 1) / total_units;
    const int64_t max_splits = (ntiles + 3) / 4;              // >= 4 sample tiles per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    d.splits[j] = (int)splits;
    nw += (int)splits;
  }
  hipError_t e = hipMemsetAsync(grads, 0, sizeof(float) * L::P_TOTAL, s);
  if (e != hipSuccess) return fail(NERF_E_HIP, "nerf_mlp_backward: memset: %s", hipGetErrorString(e));
  d.ntiles = (int)ntiles;
  d.acts = acts; d.dz = dz; d.grads = grads;
  hipLaunchKernelGGL(mlp_dw_kernel, dim3(nw), dim3(512), DW_BUF_BYTES, s, d);
  return check_launch("mlp dW");
}
