// Fused multi-tensor Adam over one flat float32 parameter buffer (a21 / K9).
// HBM-bound: 28 B per parameter (read p,g,m,v; write p,m,v) = 16.7 MB per network.
#include "common.h"
#include "hash_common.h"

namespace nerf {

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t count,
                                                   float lr, float b1, float b2, float eps, float c1, float c2,
                                                   float gscale) {
  // mlx.optimizers.Adam (0.7.0): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
  // p = p - lr * m / (sqrt(v) + eps).  c1, c2 = 1 unless bias correction is requested.
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - lr * (mi * c1) / (sqrtf(vi * c2) + eps);
  }
}

// The same update with (a) the gradient taken from float32 values OR from the int64 fixed-point accumulators of the
// deterministic hash-grid scatter (hash_common.h: 2^-52 units), and (b) the gradient buffer zeroed in the same pass
// (read g, write 0): the next step's scatter needs no memset launch.
// (c) SHADOW: the updated parameter is also written as fp16 into a second buffer -- the 4-byte-per-entry image of the hash
// tables that the forward gathers read (half the bytes of the float32 master pairs; the master copy stays float32).
template <bool FIXED, bool ZERO, bool SHADOW>
__global__ void __launch_bounds__(256) adam_ex_kernel(float* __restrict__ p, void* __restrict__ gv, float* __restrict__ m,
                                                      float* __restrict__ v, int64_t count, float lr, float b1, float b2,
                                                      float eps, float c1, float c2, float gscale, _Float16* __restrict__ ph) {
  float* gf = static_cast<float*>(gv);
  long long* gi64 = static_cast<long long*>(gv);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    float gi;
    if (FIXED) {
      const long long a = gi64[i];
      gi = (float)((double)a * (1.0 / NERF_HASH_FIX_SCALE)) * gscale;
      if (nerf_fixed_is_poisoned(a)) gi = __builtin_nanf("");           // saturated / non-finite addend or overflowing sum (hash_common.h)
      if (ZERO) gi64[i] = 0;
    }
    else { gi = gf[i] * gscale; if (ZERO) gf[i] = 0.0f; }
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float pn = p[i] - lr * (mi * c1) / (sqrtf(vi * c2) + eps);
    p[i] = pn;
    if (SHADOW) ph[i] = (_Float16)pn;
  }
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_adam_step_shadow(float* params, void* grads, float* m, float* v, int64_t count, float lr, float beta1,
                                     float beta2, float eps, int bias_correction, int step, float grad_scale, int grads_fixed_point,
                                     int zero_grads, void* params_half, void* stream) {
  NERF_REQUIRE(params && grads && m && v, NERF_E_NULL, "nerf_adam_step_ex: NULL pointer");
  NERF_REQUIRE(count > 0, NERF_E_SHAPE, "nerf_adam_step_ex: count must be > 0");
  float c1 = 1.0f, c2 = 1.0f;
  if (bias_correction) {
    NERF_REQUIRE(step >= 1, NERF_E_SHAPE, "nerf_adam_step_ex: bias correction needs step >= 1");
    c1 = 1.0f / (1.0f - powf(beta1, (float)step));
    c2 = 1.0f / (1.0f - powf(beta2, (float)step));
  }
  const dim3 g(grid_for(count, 256)), b(256);
  auto s = as_stream(stream);
  _Float16* ph = static_cast<_Float16*>(params_half);
#define AX(FX, ZR, SH) hipLaunchKernelGGL((adam_ex_kernel<FX, ZR, SH>), g, b, 0, s, params, grads, m, v, count, lr, beta1, beta2, eps, c1, c2, grad_scale, ph)
#define AY(FX, ZR) do { if (ph) AX(FX, ZR, true); else AX(FX, ZR, false); } while (0)
  if (grads_fixed_point) { if (zero_grads) AY(true, true); else AY(true, false); }
  else { if (zero_grads) AY(false, true); else AY(false, false); }
#undef AY
#undef AX
  return check_launch("nerf_adam_step_ex");
}

extern "C" int nerf_adam_step_ex(float* params, void* grads, float* m, float* v, int64_t count, float lr, float beta1,
                                 float beta2, float eps, int bias_correction, int step, float grad_scale, int grads_fixed_point,
                                 int zero_grads, void* stream) {
  return nerf_adam_step_shadow(params, grads, m, v, count, lr, beta1, beta2, eps, bias_correction, step, grad_scale,
                               grads_fixed_point, zero_grads, nullptr, stream);
}

extern "C" int nerf_adam_step(float* params, const float* grads, float* m, float* v, int64_t count, float lr,
                              float beta1, float beta2, float eps, int bias_correction, int step, float grad_scale,
                              void* stream) {
  NERF_REQUIRE(params && grads && m && v, NERF_E_NULL, "nerf_adam_step: NULL pointer");
  NERF_REQUIRE(count > 0, NERF_E_SHAPE, "nerf_adam_step: count must be > 0");
  float c1 = 1.0f, c2 = 1.0f;
  if (bias_correction) {
    NERF_REQUIRE(step >= 1, NERF_E_SHAPE, "nerf_adam_step: bias correction needs step >= 1");
    c1 = 1.0f / (1.0f - powf(beta1, (float)step));
    c2 = 1.0f / (1.0f - powf(beta2, (float)step));
  }
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(count, 256)), dim3(256), 0, as_stream(stream), params, grads, m, v,
                     count, lr, beta1, beta2, eps, c1, c2, grad_scale);
  return check_launch("nerf_adam_step");
}
