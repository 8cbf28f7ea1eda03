// Flat float32 parameter layout of the 8 x 256 view model (include/nerf_hip.h "Parameter layout"): nn.Linear
// weight [out][in] followed by its bias, layers in the order pos0..pos7, feature, alpha, dir0, rgb
// (models/NeRF.py:182-197 of the reference).  Shared by the bf16 kernels (mlp.hip) and the fp32 ones (mlp32.hip).
#pragma once

namespace nerf {
namespace L {
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
}  // namespace L
}  // namespace nerf
