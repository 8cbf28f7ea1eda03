// Static layouts of the two other instances of the reference's NeRF class that have kernels (include/nerf_hip.h):
//   LI: the no-view-direction 8 x 256 model of the 2-D image fitting (entrypoints/__viser_image_learning.py:198-208:
//       pos0 [256x40] pos1..4 [256x256] pos5 [256x296] pos6 pos7 output [out_ch x 256]; models/NeRF.py:182-197,241)
//   LN: the Instant-NGP-sized view model (BASELINE configs[4]: pos0 [64x32] pos1 [64x64] feature [64x64] alpha [1x64]
//       dir0 [32x80] rgb [3x32] = 13 188 parameters; models/NeRF.py:160-243)
// and which master parameter sits in element j of lane (r, h) of fragment f of their 32x32x16 streams.  Shared by mlp.hip (bf16
// operands) and mlp_s16x.hip (split-bf16 operands: the same fragments as (hi, lo) pairs).
#pragma once
#include "mlp_layout.h"

namespace nerf {
namespace LI {
constexpr int CIN = 40;
constexpr int P_W0 = 0, P_B0 = 10240, P_W1 = 10496, P_W5 = 273664, P_B5 = 349440;
constexpr int P_W6 = 349696, P_B6 = 415232, P_W7 = 415488, P_B7 = 481024, P_WO = 481280;
__host__ __device__ constexpr int pw(int l) {
  return l == 0 ? P_W0 : l <= 4 ? P_W1 + (l - 1) * 65792 : l == 5 ? P_W5 : l == 6 ? P_W6 : P_W7;
}
__host__ __device__ constexpr int pb(int l) {
  return l == 0 ? P_B0 : l <= 4 ? P_W1 + (l - 1) * 65792 + 65536 : l == 5 ? P_B5 : l == 6 ? P_B6 : P_B7;
}
constexpr int F_L0 = 0, F_L1 = 24, F_L5 = 536, F_L6 = 688, F_L7 = 816, F_OUT = 944, F_TOTAL = 960;
constexpr int B_OUT = 0, B_L7 = 8, B_TOTAL = 904, B_PADDED = 928;       // OUT^T, pos7..pos1 (pos5: H4 columns)
constexpr int BI_OUT = 2048, BI_TOTAL = 2080;
constexpr int A_X = 0, A_H0 = 3, A_MASK = 131, A_SLOTS = 139;
constexpr int Z_L0 = 0, Z_OUT = 128, Z_SLOTS = 129;
constexpr int F_CHUNKS = F_TOTAL / RING_CHUNK, B_CHUNKS = B_PADDED / RING_CHUNK;   // 30, 29
static_assert(F_CHUNKS * RING_CHUNK == F_TOTAL && B_CHUNKS * RING_CHUNK == B_PADDED, "whole chunks");
}  // namespace LI

__device__ inline float fwd_src_img(const float* __restrict__ p, int f, int r, int h, int j, int out_ch) {
  if (f < LI::F_L1) {                                  // pos0: K space 48 (40 + pad)
    const int nt = f / 3, ks = f % 3, kk = kperm(ks, h, j);
    return kk < 40 ? p[LI::P_W0 + (32 * nt + r) * 40 + kk] : 0.0f;
  }
  if (f < LI::F_L5) {
    const int l = 1 + (f - LI::F_L1) / 128, g = (f - LI::F_L1) % 128;
    return p[LI::pw(l) + (32 * (g / 16) + r) * 256 + kperm(g % 16, h, j)];
  }
  if (f < LI::F_L6) {                                  // pos5: [x(48), H4(256)] vs W5[256][296]
    const int g = f - LI::F_L5, nt = g / 19, ks = g % 19, kk = kperm(ks, h, j), n = 32 * nt + r;
    if (kk < 48) return kk < 40 ? p[LI::P_W5 + n * 296 + kk] : 0.0f;
    return p[LI::P_W5 + n * 296 + 40 + (kk - 48)];
  }
  if (f < LI::F_OUT) {
    const int l = 6 + (f - LI::F_L6) / 128, g = (f - LI::F_L6) % 128;
    return p[LI::pw(l) + (32 * (g / 16) + r) * 256 + kperm(g % 16, h, j)];
  }
  return r < out_ch ? p[LI::P_WO + r * 256 + kperm(f - LI::F_OUT, h, j)] : 0.0f;
}

__device__ inline float bwd_src_img(const float* __restrict__ p, int f, int r, int h, int j, int out_ch) {
  if (f < LI::B_L7) {                                  // output^T: 8 tiles of H7, one k-step (rows 0..out_ch-1)
    const int nn = kperm(0, h, j);
    return nn < out_ch ? p[LI::P_WO + nn * 256 + 32 * f + r] : 0.0f;
  }
  const int g = f - LI::B_L7, li = g / 128, q = g % 128, kt = q / 16, ns = q % 16;
  const int l = 7 - li, nn = kperm(ns, h, j), row = 32 * kt + r;
  if (l == 5) return p[LI::P_W5 + nn * 296 + 40 + row];
  return p[LI::pw(l) + nn * 256 + row];
}

namespace LN {
constexpr int CPOS = 32, CDIR = 16, CIN = CPOS + CDIR;
constexpr int P_W0 = 0, P_B0 = 2048, P_W1 = 2112, P_B1 = 6208, P_WF = 6272, P_BF = 10368, P_WA = 10432, P_BA = 10496;
constexpr int P_WD = 10497, P_BD = 13057, P_WR = 13089, P_BR = 13185, P_TOTAL = 13188;
constexpr int F_L0 = 0, F_L1 = 4, F_FA = 12, F_DIR = 24, F_RGB = 29, F_TOTAL = 31, F_PADDED = 32;
constexpr int B_RGB = 0, B_DIR = 1, B_FA = 5, B_L1 = 15, B_L0 = 23, B_TOTAL = 27, B_PADDED = 32;
constexpr int BI_L0 = 0, BI_L1 = 64, BI_FEAT = 128, BI_ALPHA = 192, BI_DIR = 224, BI_RGB = 256, BI_TOTAL = 288;
constexpr int64_t PACKED_BYTES = (int64_t)(F_PADDED + B_PADDED) * 1024 + BI_TOTAL * 4;
constexpr int A_X = 0, A_DX = 2, A_H0 = 3, A_H1 = 7, A_FEAT = 11, A_HD = 15, A_MASK = 17, A_SLOTS = 20;
constexpr int Z_L0 = 0, Z_L1 = 4, Z_F = 8, Z_A = 12, Z_D = 13, Z_RGB = 15, Z_SLOTS = 16;
constexpr int LDS_BYTES = 32 * 1024 + BI_TOTAL * 4;
}  // namespace LN

__device__ inline float fwd_src_small(const float* __restrict__ p, int f, int r, int h, int j) {
  if (f < LN::F_L1) { const int nt = f / 2, ks = f % 2; return p[LN::P_W0 + (32 * nt + r) * 32 + kperm(ks, h, j)]; }
  if (f < LN::F_FA) { const int g = f - LN::F_L1, nt = g / 4, ks = g % 4; return p[LN::P_W1 + (32 * nt + r) * 64 + kperm(ks, h, j)]; }
  if (f < LN::F_DIR) {
    const int g = f - LN::F_FA;
    if (g < 8) return p[LN::P_WF + (32 * (g / 4) + r) * 64 + kperm(g % 4, h, j)];
    return r == 0 ? p[LN::P_WA + kperm(g - 8, h, j)] : 0.0f;
  }
  if (f < LN::F_RGB) { const int ks = f - LN::F_DIR; return p[LN::P_WD + r * 80 + kperm(ks, h, j)]; }      // [feature(64), sh(16)]
  if (f < LN::F_TOTAL) return r < 3 ? p[LN::P_WR + r * 32 + kperm(f - LN::F_RGB, h, j)] : 0.0f;
  return 0.0f;
}
// transposed stream: A rows = INPUT feature (32 kt + r), k index = OUTPUT feature nn
__device__ inline float bwd_src_small(const float* __restrict__ p, int f, int r, int h, int j) {
  if (f < LN::B_DIR) { const int nn = kperm(0, h, j); return nn < 3 ? p[LN::P_WR + nn * 32 + r] : 0.0f; }
  if (f < LN::B_FA) { const int g = f - LN::B_DIR, kt = g / 2, ns = g % 2; return p[LN::P_WD + kperm(ns, h, j) * 80 + 32 * kt + r]; }
  if (f < LN::B_L1) {
    const int g = f - LN::B_FA, kt = g / 5, ns = g % 5, nn = kperm(ns, h, j);
    if (ns < 4) return p[LN::P_WF + nn * 64 + 32 * kt + r];
    return nn == 64 ? p[LN::P_WA + 32 * kt + r] : 0.0f;
  }
  if (f < LN::B_L0) { const int g = f - LN::B_L1, kt = g / 4, ns = g % 4; return p[LN::P_W1 + kperm(ns, h, j) * 64 + 32 * kt + r]; }
  if (f < LN::B_TOTAL) return p[LN::P_W0 + kperm(f - LN::B_L0, h, j) * 32 + r];
  return 0.0f;
}

}  // namespace nerf
