// Split-bf16 ("precision 22") kernels of the two other instances of the reference's NeRF class (mlp_s16x.hip): interface used
// by the C ABI entry points of mlp.hip.
//
//   image model  (entrypoints/__viser_image_learning.py:198-208; models/NeRF.py:196-197,241): 8 x 256, in 40, no view head
//   2 x 64 model (BASELINE configs[4]: hash-grid features + SH -> NeRF(n_layers=2, width_layers=64, ...), models/NeRF.py:160-243)
//
// The reference computes both in float32.  As in mlp_s16.hip every float32 GEMM operand x is carried as hi = bf16(x),
// lo = bf16(x - hi) (16 significand bits at float32's exponent range) and a product is a_hi b_hi + a_hi b_lo + a_lo b_hi on
// v_mfma_f32_32x32x16_bf16 into one fp32 accumulator; inference and training run the same arithmetic (the inference launch
// keeps no activations).  Stores: hi blocks | lo blocks | ReLU sign-bit words per 32-sample tile, slot numbering of the bf16
// kernels (mlp_arch2.h, namespaces LI / LN); biases: the fp32 slots of the bf16 image.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nerf {
namespace s16x {

// ---- image model: activations hi [0,131) lo [131,262) masks [262,270); dZ hi [0,129) lo [129,258)
constexpr int IMG_A_LO = 131, IMG_A_MASK = 262, IMG_A_SLOTS = 270;
constexpr int IMG_Z_LO = 129, IMG_Z_SLOTS = 258;
// forward stream 2 x 960 fragments (60 ring chunks) | transposed stream 2 x 904, zero-padded to 58 whole chunks
constexpr int IMG_F_FRAGS = 1920, IMG_B_FRAGS = 1808, IMG_B_PADDED = 1856;
constexpr int64_t IMG_PACKED_BYTES = (int64_t)(IMG_F_FRAGS + IMG_B_PADDED) * 1024;

int img_pack(const float* params, int out_ch, void* packed, hipStream_t s);
// x [M,40] embedded rows -> out [M,out_ch]; acts == nullptr: inference (nothing kept)
int img_forward(const void* packed, const float* bias_slots, const float* x, int64_t M, int out_ch, float* out, void* acts,
                int64_t astride16, int persistent_wgs, hipStream_t s);
int img_backward_chain(const void* packed, const void* acts, const float* d_out, int64_t M, int out_ch, void* dz,
                       int64_t astride16, int64_t zstride16, int persistent_wgs, hipStream_t s);

// ---- 2 x 64 model: activations hi [0,17) lo [17,34) masks [34,37); dZ hi [0,16) lo [16,32)
constexpr int SM_A_LO = 17, SM_A_MASK = 34, SM_A_SLOTS = 37;
constexpr int SM_Z_LO = 16, SM_Z_SLOTS = 32;
// forward stream 2 x 32 fragments | transposed stream 2 x 32: both LDS-resident (64 KiB per kernel), no ring
constexpr int SM_F_FRAGS = 64, SM_B_FRAGS = 64;
constexpr int64_t SM_PACKED_BYTES = (int64_t)(SM_F_FRAGS + SM_B_FRAGS) * 1024;

// the fused configs[4] query: rows computed in the kernel from rays / depths / the FLOAT32 hash tables (16 levels x 2 features,
// SH degree 3); rays == nullptr: rows from x [M,48]
struct SmallQuery {
  const float* rays; const float* z; int n; const float* tables; uint32_t T; float res[32]; float pos_scale, pos_offset;
  int ray_major; int64_t B;
};
int small_pack(const float* params, void* packed, hipStream_t s);
int small_forward(const void* packed, const float* bias_slots, const float* x, int64_t M, float* out, void* acts,
                  int64_t astride16, const SmallQuery* q, hipStream_t s);
// d_x (optional): dL/d(position features) [M,32], float32
int small_backward_chain(const void* packed, const void* acts, const float* d_raw, int64_t M, void* dz, float* d_x,
                         int64_t astride16, int64_t zstride16, hipStream_t s);

}  // namespace s16x
}  // namespace nerf
