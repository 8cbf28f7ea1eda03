// Split-fp16 ("precision 22") inference forward of the fused 8 x 256 chain (mlp22.hip): interface used by the C ABI
// entry points of mlp.hip for a model whose nerf_mlp_arch.precision is 22.
//
// Every float32 operand x of the reference's GEMMs (models/NeRF.py:201-243) is carried as TWO fp16 numbers,
// x = hi + lo * 2^-11 with hi = fp16(x), lo = fp16((x - hi) * 2^11): 22 significand bits.  A product w x is evaluated as
// w_hi x_hi + 2^-11 (w_hi x_lo + w_lo x_hi) -- three v_mfma_f32_16x16x32_f16 per float32 product, fp32 accumulate, the
// dropped w_lo x_lo term is 2^-22 relative -- so the matrix pipe runs at 1/3 of its fp16 rate instead of the 1/16 of the
// fp32 MFMA, at float32-class accuracy (measured <= 2e-6 of the output scale against the fp32 oracle).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nerf {
namespace f22 {

// packed image: forward stream of (hi, lo) fragment PAIRS in consumption order (2 x 1172 fragments of 1 KiB, padded to whole
// 32-fragment ring chunks) | fp32 bias slots (same slot numbering as the bf16 image)
constexpr int F_PAIRS = 1172, F_FRAGS = 2 * F_PAIRS, F_PADDED = 2368, BIAS_FLOATS = 2496;
constexpr int64_t PACKED_BYTES = (int64_t)F_PADDED * 1024 + (int64_t)BIAS_FLOATS * 4;

extern int g_tiles;    // "f22_tiles" (mlp22.hip)
int pack(const float* params, void* packed22, hipStream_t s);
// x != nullptr: embedded rows [M,90]; else rays [B,11] + z [B,n] with the encodings evaluated in the kernel.
// persistent_wgs: workgroups of the persistent launch (one per CU)
int forward(const void* packed22, const float* x, const float* rays, const float* z, int64_t M, int n, int freq_mode,
            float* out, int persistent_wgs, hipStream_t s);

}  // namespace f22
}  // namespace nerf
