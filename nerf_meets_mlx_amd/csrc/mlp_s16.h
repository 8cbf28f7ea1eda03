// Split-bf16 TRAINING kernels of the fused 8 x 256 chain (mlp_s16.hip): interface used by the C ABI entry points of
// mlp.hip for a model whose nerf_mlp_arch.precision is 22 (training forward with activation stores, dZ chain, dW / db).
//
// Every float32 operand x of the reference's GEMMs (models/NeRF.py:201-243 and their adjoints) is carried as TWO bf16
// numbers, x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 significand bits at float32's exponent range (gradients
// span > 30 binades, which rules the scaled-fp16 pairs of the inference kernel out here).  A product is evaluated as
// a_hi b_hi + a_hi b_lo + a_lo b_hi -- three v_mfma_f32_32x32x16_bf16 into one fp32 accumulator, the dropped a_lo b_lo term
// is 2^-18 relative -- so the matrix pipe runs at 1/3 of its bf16 rate instead of the 1/16 of the fp32 MFMA.  Measured
// against the fp32 oracle: forward <= 1e-5 of the output scale, gradients <= 1e-4 rel-L2 per tensor (tests/test_gpu_round4.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nerf {
struct DwArgs;
namespace s16 {

// training stores, fragment slots per 32-sample tile (slot numbering of the bf16 kernels, mlp_layout.h namespace L):
//   activations: hi blocks [0, 158) | lo blocks [158, 316) | ReLU sign-bit words [316, 325)
//   dZ:          hi blocks [0, 154) | lo blocks [154, 308)
constexpr int A_LO = 158, A_MASK = 316, A_SLOTS = 325;
constexpr int Z_LO = 154, Z_SLOTS = 308;
// packed image: forward stream of (hi, lo) fragment PAIRS in consumption order (2 x 1184 fragments of 1 KiB = 74 ring
// chunks) | transposed stream (2 x 1100, zero-padded to 69 whole chunks).  Biases: the fp32 slots of the bf16 image.
constexpr int F_FRAGS = 2368, B_FRAGS = 2200, B_PADDED = 2208;
constexpr int64_t PACKED_BYTES = (int64_t)(F_FRAGS + B_PADDED) * 1024;

int pack(const float* params, void* packed_s16, hipStream_t s);
// x != nullptr: embedded rows [M,90]; else rays [B,11] + z [B,n] with the encodings evaluated in the kernel.  acts != nullptr.
int forward(const void* packed_s16, const float* bias_slots, const float* x, const float* rays, const float* z, int64_t M,
            int n, int freq_mode, float* out, void* acts, int64_t astride16, int persistent_wgs, hipStream_t s);
int backward_chain(const void* packed_s16, const void* acts, const float* d_raw, int64_t M, void* dz, int64_t astride16,
                   int64_t zstride16, int persistent_wgs, hipStream_t s);
// the weight-gradient kernel proper (job table, split and partial slots prepared by mlp.hip's launch_dw)
int launch_dw_kernel(const DwArgs& d, int workgroups, bool split_bf16, hipStream_t s);     // split_bf16: hi + lo blocks; else the bf16 stores
extern int g_dw_variant;      // A/B knob ("dw22_variant"): 1 = 256 x 256 jobs on the one-wave-per-SIMD kernel (mlp_dww.hip), the others on the 16-wave kernel (default); 0 = every job on the 16-wave kernel

}  // namespace s16
}  // namespace nerf
