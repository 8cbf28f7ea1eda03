// Static fragment layouts of the 8 x 256 view model shared by mlp.hip (bf16 operands) and mlp22.hip (split-fp16 operands):
// stream offsets, bias slots, and the 16x16x32 forward stream's lane / element <-> weight mapping.
#pragma once
#include "mlp_ring.h"
#include "mlp_params.h"

namespace nerf {

// ------------------------------------------------------------------------------------------
// static layout of the one supported architecture (8 x 256, skip 4, view head)
// ------------------------------------------------------------------------------------------
namespace L {
// float32 parameter offsets P_*: mlp_params.h (shared with the fp32 reference-precision kernels of mlp32.hip)

// forward weight stream, 1 KiB fragments in consumption order
constexpr int F_L0 = 0, F_L1 = 32, F_L5 = 544, F_L6 = 704, F_L7 = 832, F_FA = 960, F_DIR = 1104, F_RGB = 1176;
constexpr int F_TOTAL = 1184;
// backward (transposed) weight stream
constexpr int B_RGB = 0, B_DIR = 4, B_FA = 68, B_L7 = 204, B_L6 = 332, B_L5 = 460, B_L4 = 588;
constexpr int B_TOTAL = 1100, B_PADDED = 1120;    // padded with zero fragments to whole 32-fragment ring chunks
// fp32 bias slots
constexpr int BI_FEAT = 2048, BI_ALPHA = 2304, BI_DIR = 2336, BI_RGB = 2464, BI_TOTAL = 2496;
constexpr int F16_TOTAL = 1172, F16_PADDED = 1184;   // forward stream of the 16x16x32 variant (inference only)
constexpr int64_t F16_OFFSET = (int64_t)(F_TOTAL + B_PADDED) * 1024 + BI_TOTAL * 4;    // appended after the bias slots
constexpr int64_t PACKED_BYTES = F16_OFFSET + (int64_t)F16_PADDED * 1024;

// activation store: fragment slots per 32-sample tile
constexpr int A_PE = 0, A_DPE = 4, A_H0 = 6;       // H_l at A_H0 + 16 l, l = 0..7
constexpr int A_FEAT = 134, A_HD = 150;
// ReLU sign bits of H0..H7 and HD for the backward chain: one 16-byte word per lane and layer (layout: see
// finish_quarter), so the chain reads 9 KiB per tile instead of 150 KiB
constexpr int A_MASK = 158, A_SLOTS = 167;
// gradient store
constexpr int Z_L0 = 0;                            // dZ_l at 16 l, l = 0..7
constexpr int Z_F = 128, Z_A = 144, Z_D = 145, Z_RGB = 153, Z_SLOTS = 154;
}  // namespace L

// element j of lane half h in k-step ks  <->  feature index
__host__ __device__ constexpr int kperm(int ks, int h, int j) { return 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3); }

namespace L16 {
constexpr int F_L0 = 0, F_L1 = 32, F_L5 = 544, F_L6 = 704, F_L7 = 832, F_FA = 960, F_DIR = 1096, F_RGB = 1168;
constexpr int CHUNKS = L::F16_PADDED / RING_CHUNK;     // 37
}
__host__ __device__ constexpr int kperm16(int ks, int g, int j) { return 32 * ks + 16 * (j >> 2) + 4 * g + (j & 3); }
// Which embedding channel sits in element j of lane group g in the encoding k-steps.  The order is ours to choose
// (the packed weights follow it), so it is chosen to make the per-element (sin | cos, x | y | z) pattern the same in
// all four lane groups -- only the frequency differs, three per-lane registers -- instead of a table lookup and four
// selects per element.  Position (63 channels, 2 k-steps = slots 8 ks + j): slots 0-11 = bands 2g, 2g+1 as
// (sin xyz, cos xyz); slots 12-14 = band 8 + (g>>1), sin for even g / cos for odd g, xyz; slot 15 = identity
// channel g (zero pad for g = 3).  Direction (27 channels, 1 k-step): j 0-5 = band g, j 6 = identity g, j 7 = pad.
__host__ __device__ constexpr int pos_chan16(int ks, int g, int j) {
  const int sl = 8 * ks + j;
  if (sl < 12) return 3 + 6 * (2 * g + sl / 6) + (sl % 6);
  if (sl < 15) return 3 + 6 * (8 + (g >> 1)) + 3 * (g & 1) + (sl - 12);
  return g < 3 ? g : -1;
}
__host__ __device__ constexpr int dir_chan16(int g, int j) {
  if (j < 6) return 3 + 6 * g + j;
  return (j == 6 && g < 3) ? g : -1;
}

__device__ inline float fwd_src16(const float* __restrict__ p, int f, int i, int g, int j) {
  if (f < L16::F_L1) {
    const int nt = f / 2, ch = pos_chan16(f % 2, g, j);
    return ch >= 0 ? p[L::P_W0 + (16 * nt + i) * 63 + ch] : 0.0f;
  }
  if (f < L16::F_L5) {
    const int q = f - L16::F_L1, l = 1 + q / 128, r = q % 128;
    return p[L::pw(l) + (16 * (r / 8) + i) * 256 + kperm16(r % 8, g, j)];
  }
  if (f < L16::F_L6) {
    const int q = f - L16::F_L5, n = 16 * (q / 10) + i, ks = q % 10;
    if (ks < 2) { const int ch = pos_chan16(ks, g, j); return ch >= 0 ? p[L::P_W5 + n * 319 + ch] : 0.0f; }
    return p[L::P_W5 + n * 319 + 63 + kperm16(ks - 2, g, j)];
  }
  if (f < L16::F_FA) {
    const int q = f - L16::F_L6, l = 6 + q / 128, r = q % 128;
    return p[L::pw(l) + (16 * (r / 8) + i) * 256 + kperm16(r % 8, g, j)];
  }
  if (f < L16::F_DIR) {
    const int q = f - L16::F_FA;
    if (q < 128) return p[L::P_WF + (16 * (q / 8) + i) * 256 + kperm16(q % 8, g, j)];
    return i == 0 ? p[L::P_WA + kperm16(q - 128, g, j)] : 0.0f;
  }
  if (f < L16::F_RGB) {
    const int q = f - L16::F_DIR, n = 16 * (q / 9) + i, ks = q % 9;
    if (ks < 8) return p[L::P_WD + n * 283 + kperm16(ks, g, j)];
    const int ch = dir_chan16(g, j);
    return ch >= 0 ? p[L::P_WD + n * 283 + 256 + ch] : 0.0f;
  }
  if (f < L::F16_TOTAL) return i < 3 ? p[L::P_WR + i * 128 + kperm16(f - L16::F_RGB, g, j)] : 0.0f;
  return 0.0f;
}


}  // namespace nerf
