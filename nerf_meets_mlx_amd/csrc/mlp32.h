// fp32 reference-precision instantiation of the fused 8 x 256 chain (mlp32.hip): interface used by the C ABI
// entry points of mlp.hip for a model whose nerf_mlp_arch.precision is 32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nerf {
namespace f32 {

// packed fp32 image: forward stream | transposed (backward) stream | tail (biases, alpha and rgb weights)
constexpr int F_FRAGS = 580, B_FRAGS = 544, FRAG_BYTES = 4096, TAIL_FLOATS = 3584;
constexpr int64_t PACKED_BYTES = (int64_t)(F_FRAGS + B_FRAGS) * FRAG_BYTES + (int64_t)TAIL_FLOATS * 4;

// training stores: per 32-sample tile, `rows` feature rows of 32 floats (feature-major inside the tile)
constexpr int A_ROWS = 2592, Z_ROWS = 2496;
inline int64_t tiles_of(int64_t M) { return (M + 31) / 32; }
inline int64_t acts_bytes(int64_t M) { return tiles_of(M) * A_ROWS * 128; }
// dW: split-K partial blocks (64 x 64 floats + 64 bias sums per unit, at most DW32_MAX_UNITS units) behind the dZ rows
constexpr int DW32_PART_FLOATS = 4096 + 64, DW32_MAX_UNITS = 2048;
inline int64_t dz_bytes(int64_t M) { return tiles_of(M) * Z_ROWS * 128 + (int64_t)DW32_MAX_UNITS * DW32_PART_FLOATS * 4; }

// img_out_ch > 0: the image-fitting model (in 40, no view head, out_ch = img_out_ch <= 4); 0: the view model
int pack(const float* params, void* packed32, int img_out_ch, hipStream_t s);
// x != nullptr: embedded rows [M,90]; else rays [B,11] + z [B,n] with the encodings evaluated in the kernel
int forward(const void* packed32, const float* x, const float* rays, const float* z, int64_t M, int n, int freq_mode,
            float* out, void* acts, int img_out_ch, hipStream_t s);
int backward(const void* packed32, const void* acts, const float* d_raw, int64_t M, void* dz, float* grads, int img_out_ch,
             hipStream_t s);
// test hook: one stored layer as row-major [M, width]; returns width or -1
int debug_width(int kind, int layer);
int debug_read(const void* store, int kind, int layer, int64_t M, float* out, hipStream_t s);

}  // namespace f32
}  // namespace nerf
