// Alpha compositing forward / backward (a13 / K6, K7) and the MSE loss (a20).
//
// HBM-bound: one wavefront per ray, CH = ceil(n/64) consecutive samples per lane, a
// lane-serial prefix inside the lane plus one wave64 inclusive scan for the transmittance
// exponent.  Algorithmic traffic per ray: read 20n+12 B, write 4n+24 B.
// Everything stays float32 (delta = 1e10 on the last interval, exp of the scan).
#include "common.h"

namespace nerf {

struct RayQ {        // per-lane forward quantities for one sample
  float r, g, b, z, delta, x, alpha, T, w;
};

// computes the forward quantities of the lane's CH samples; returns the lane-local arrays
template <int CH>
__device__ __forceinline__ void composite_lane(const float* __restrict__ raw, const float* __restrict__ zr,
                                               const float* __restrict__ noise, float raw_noise_std, float dnorm,
                                               int n, int lane, RayQ (&q)[CH]) {
  float run = 0.0f;
  float pre[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int k = lane * CH + c;
    RayQ& s = q[c];
    if (k < n) {
      const float4 rv = *reinterpret_cast<const float4*>(raw + 4 * (size_t)k);
      float sigma = rv.w;
      if (raw_noise_std > 0.0f) sigma = sigma + noise[k] * raw_noise_std;        // render.py:41-43
      s.r = rv.x; s.g = rv.y; s.b = rv.z; s.z = zr[k];
      const float dz = (k < n - 1) ? (zr[k + 1] - s.z) : 1e10f;                  // render.py:46-57
      s.delta = dz * dnorm;                                                      // render.py:60
      s.x = s.delta * sigma;                                                     // render.py:67
      s.alpha = 1.0f - expf(-fmaxf(s.x, 0.0f));                                  // render.py:69
    } else {
      s.r = s.g = s.b = s.z = s.delta = s.x = s.alpha = 0.0f;
    }
    pre[c] = run;                         // exclusive prefix inside the lane
    run += (k < n - 1) ? s.x : 0.0f;      // cumsum runs over x[:-1] (render.py:71), x NOT ReLU'd (Q10)
  }
  const float incl = wave_scan_incl(run, lane);
  const float base = incl - run;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int k = lane * CH + c;
    q[c].T = expf(-(base + pre[c]));                                             // render.py:72-80
    q[c].w = (k < n) ? q[c].alpha * q[c].T : 0.0f;                               // render.py:81
  }
}

template <int CH>
__global__ void __launch_bounds__(256) composite_fwd_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                            const float* __restrict__ rays, int64_t B, int n,
                                                            float raw_noise_std, const float* __restrict__ noise,
                                                            int white, float* __restrict__ rgb,
                                                            float* __restrict__ disp, float* __restrict__ acc,
                                                            float* __restrict__ weights, float* __restrict__ depth) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t ray = blockIdx.x * 4 + wv; ray < B; ray += (int64_t)gridDim.x * 4) {
    const float* rr = rays + ray * NERF_RAY_STRIDE;
    const float dnorm = sqrtf(rr[3] * rr[3] + rr[4] * rr[4] + rr[5] * rr[5]);
    RayQ q[CH];
    composite_lane<CH>(raw + ray * n * 4, z + ray * n, noise ? noise + ray * n : nullptr, raw_noise_std, dnorm, n,
                       lane, q);
    float sr = 0, sg = 0, sb = 0, sd = 0, sa = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      sr += q[c].w * q[c].r; sg += q[c].w * q[c].g; sb += q[c].w * q[c].b;
      sd += q[c].w * q[c].z; sa += q[c].w;
      const int k = lane * CH + c;
      if (weights && k < n) weights[ray * n + k] = q[c].w;
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sd = wave_sum(sd); sa = wave_sum(sa);
    if (lane == 0) {
      if (white) { sr = sr + (1.0f - sa); sg = sg + (1.0f - sa); sb = sb + (1.0f - sa); }   // render.py:91-92
      rgb[ray * 3 + 0] = sr; rgb[ray * 3 + 1] = sg; rgb[ray * 3 + 2] = sb;
      if (disp) {                                     // render.py:85-88; maximum() propagates NaN: acc == 0 -> NaN (Q11)
        const float q = sd / sa;
        disp[ray] = 1.0f / ((q != q) ? q : fmaxf(1e-10f, q));
      }
      if (acc) acc[ray] = sa;
      if (depth) depth[ray] = sd;
    }
  }
}

template <int CH>
__global__ void __launch_bounds__(256) composite_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                            const float* __restrict__ rays, int64_t B, int n,
                                                            float raw_noise_std, const float* __restrict__ noise,
                                                            int white, const float* __restrict__ d_rgb,
                                                            const float* __restrict__ d_acc,
                                                            const float* __restrict__ d_depth,
                                                            float* __restrict__ d_raw) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t ray = blockIdx.x * 4 + wv; ray < B; ray += (int64_t)gridDim.x * 4) {
    const float* rr = rays + ray * NERF_RAY_STRIDE;
    const float dnorm = sqrtf(rr[3] * rr[3] + rr[4] * rr[4] + rr[5] * rr[5]);
    RayQ q[CH];
    composite_lane<CH>(raw + ray * n * 4, z + ray * n, noise ? noise + ray * n : nullptr, raw_noise_std, dnorm, n,
                       lane, q);
    const float gr = d_rgb[ray * 3], gg = d_rgb[ray * 3 + 1], gb = d_rgb[ray * 3 + 2];
    // d rgb / d acc = -1 per channel under the white background; acc = sum w
    const float gacc = (d_acc ? d_acc[ray] : 0.0f) - (white ? (gr + gg + gb) : 0.0f);
    const float gdep = d_depth ? d_depth[ray] : 0.0f;
    float G[CH], gw[CH];
    float run = 0.0f;
#pragma unroll
    for (int c = CH - 1; c >= 0; --c) {            // lane-local exclusive suffix of G*w
      G[c] = gr * q[c].r + gg * q[c].g + gb * q[c].b + gacc + gdep * q[c].z;
      gw[c] = run;
      run += G[c] * q[c].w;
    }
    const float incl = wave_rscan_incl(run, lane);
    const float after = incl - run;                // sum over lanes > lane
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k = lane * CH + c;
      if (k >= n) continue;
      // dL/dx_k = G_k T_k exp(-relu(x_k)) [x_k > 0]  -  sum_{k' > k} G_k' w_k'   (x_k enters S_k' for k' > k)
      const float suffix = (k < n - 1) ? (after + gw[c]) : 0.0f;
      const float da = (q[c].x > 0.0f) ? G[c] * q[c].T * expf(-q[c].x) : 0.0f;
      const float dx = da - suffix;
      float4 o;
      o.x = q[c].w * gr; o.y = q[c].w * gg; o.z = q[c].w * gb; o.w = q[c].delta * dx;
      *reinterpret_cast<float4*>(d_raw + (ray * n + k) * 4) = o;
    }
  }
}

// Training form: raw2outputs -> MSE against the target pixel -> d loss / d raw in ONE pass per ray (what
// nn.value_and_grad differentiates in entrypoints/__test_nerf.py:47-126).  The backward kernel recomputes the forward
// quantities of the ray anyway, so the colour, its residual and the upstream gradient 2 (rgb - y) / (3 B) cost one
// wave reduction more; two launches and one pass over raw fewer than forward + mse + backward.  rgb and d_raw are
// bit-identical to the staged kernels (same operations in the same order); the loss is summed in another order.
template <int CH>
__global__ void __launch_bounds__(256) composite_train_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                              const float* __restrict__ rays, int64_t B, int n, int white,
                                                              const float* __restrict__ target, float grad_scale,
                                                              float* __restrict__ loss, float* __restrict__ rgb_out,
                                                              float* __restrict__ d_raw) {
  __shared__ float part[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float inv = 1.0f / (float)(B * 3);
  float sq = 0.0f;
  for (int64_t ray = blockIdx.x * 4 + wv; ray < B; ray += (int64_t)gridDim.x * 4) {
    const float* rr = rays + ray * NERF_RAY_STRIDE;
    const float dnorm = sqrtf(rr[3] * rr[3] + rr[4] * rr[4] + rr[5] * rr[5]);
    RayQ q[CH];
    composite_lane<CH>(raw + ray * n * 4, z + ray * n, nullptr, 0.0f, dnorm, n, lane, q);
    float sr = 0, sg = 0, sb = 0, sa = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) { sr += q[c].w * q[c].r; sg += q[c].w * q[c].g; sb += q[c].w * q[c].b; sa += q[c].w; }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sa = wave_sum(sa);
    if (white) { sr = sr + (1.0f - sa); sg = sg + (1.0f - sa); sb = sb + (1.0f - sa); }          // render.py:91-92
    const float er = sr - target[ray * 3], eg = sg - target[ray * 3 + 1], eb = sb - target[ray * 3 + 2];
    if (lane == 0) {
      sq += er * er; sq += eg * eg; sq += eb * eb;
      if (rgb_out) { rgb_out[ray * 3] = sr; rgb_out[ray * 3 + 1] = sg; rgb_out[ray * 3 + 2] = sb; }
    }
    const float gr = grad_scale * 2.0f * er * inv, gg = grad_scale * 2.0f * eg * inv, gb = grad_scale * 2.0f * eb * inv;
    const float gacc = 0.0f - (white ? (gr + gg + gb) : 0.0f);
    float G[CH], gw[CH];
    float run = 0.0f;
#pragma unroll
    for (int c = CH - 1; c >= 0; --c) {
      G[c] = gr * q[c].r + gg * q[c].g + gb * q[c].b + gacc + 0.0f * q[c].z;
      gw[c] = run;
      run += G[c] * q[c].w;
    }
    const float incl = wave_rscan_incl(run, lane);
    const float after = incl - run;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k = lane * CH + c;
      if (k >= n) continue;
      const float suffix = (k < n - 1) ? (after + gw[c]) : 0.0f;
      const float da = (q[c].x > 0.0f) ? G[c] * q[c].T * expf(-q[c].x) : 0.0f;
      const float dx = da - suffix;
      float4 o;
      o.x = q[c].w * gr; o.y = q[c].w * gg; o.z = q[c].w * gb; o.w = q[c].delta * dx;
      *reinterpret_cast<float4*>(d_raw + (ray * n + k) * 4) = o;
    }
  }
  if (lane == 0) part[wv] = sq;
  __syncthreads();
  if (threadIdx.x == 0 && loss) atomicAdd(loss, (part[0] + part[1] + part[2] + part[3]) * inv);
}

__global__ void __launch_bounds__(256) mse_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                  int64_t count, float grad_scale, float* __restrict__ loss,
                                                  float* __restrict__ d_pred) {
  __shared__ float part[4];
  const float inv = 1.0f / (float)count;
  float s = 0.0f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = p[i] - t[i];
    s += d * d;
    if (d_pred) d_pred[i] = grad_scale * 2.0f * d * inv;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0 && loss) atomicAdd(loss, (part[0] + part[1] + part[2] + part[3]) * inv);
}

}  // namespace nerf

using namespace nerf;

#define DISPATCH_CH(n, CALL)                                  \
  do {                                                        \
    const int ch_ = ((n) + 63) / 64;                          \
    if (ch_ <= 1) { CALL(1); } else if (ch_ == 2) { CALL(2); } else if (ch_ == 3) { CALL(3); } \
    else if (ch_ == 4) { CALL(4); } else if (ch_ <= 8) { CALL(8); } else { CALL(16); }       \
  } while (0)

extern "C" int nerf_composite_forward(const float* raw, const float* z, const float* rays, int64_t B, int n,
                                      float raw_noise_std, const float* noise, int white_bkgd, float* rgb, float* disp,
                                      float* acc, float* weights, float* depth, void* stream) {
  NERF_REQUIRE(n >= 1 && n <= 1024, NERF_E_SHAPE, "nerf_composite_forward: need 1 <= n <= 1024 (n=%d)", n);
  if (B <= 0) return NERF_OK;
  NERF_REQUIRE(raw && z && rays && rgb, NERF_E_NULL, "nerf_composite_forward: raw/z/rays/rgb is NULL");
  NERF_REQUIRE(raw_noise_std <= 0.0f || noise, NERF_E_NULL, "nerf_composite_forward: raw_noise_std > 0 needs noise");
  if (B <= 0) return NERF_OK;
  const dim3 g((unsigned)((B + 3) / 4 > 8192 ? 8192 : (B + 3) / 4)), b(256);
  auto st = as_stream(stream);
#define CALL(C) hipLaunchKernelGGL(composite_fwd_kernel<C>, g, b, 0, st, raw, z, rays, B, n, raw_noise_std, noise, white_bkgd, rgb, disp, acc, weights, depth)
  DISPATCH_CH(n, CALL);
#undef CALL
  return check_launch("nerf_composite_forward");
}

extern "C" int nerf_composite_backward(const float* raw, const float* z, const float* rays, int64_t B, int n,
                                       float raw_noise_std, const float* noise, int white_bkgd, const float* d_rgb,
                                       const float* d_acc, const float* d_depth, float* d_raw, void* stream) {
  NERF_REQUIRE(n >= 1 && n <= 1024, NERF_E_SHAPE, "nerf_composite_backward: need 1 <= n <= 1024 (n=%d)", n);
  if (B <= 0) return NERF_OK;
  NERF_REQUIRE(raw && z && rays && d_rgb && d_raw, NERF_E_NULL, "nerf_composite_backward: NULL pointer");
  NERF_REQUIRE(raw_noise_std <= 0.0f || noise, NERF_E_NULL, "nerf_composite_backward: raw_noise_std > 0 needs noise");
  if (B <= 0) return NERF_OK;
  const dim3 g((unsigned)((B + 3) / 4 > 8192 ? 8192 : (B + 3) / 4)), b(256);
  auto st = as_stream(stream);
#define CALL(C) hipLaunchKernelGGL(composite_bwd_kernel<C>, g, b, 0, st, raw, z, rays, B, n, raw_noise_std, noise, white_bkgd, d_rgb, d_acc, d_depth, d_raw)
  DISPATCH_CH(n, CALL);
#undef CALL
  return check_launch("nerf_composite_backward");
}

extern "C" int nerf_composite_mse_backward(const float* raw, const float* z, const float* rays, int64_t B, int n,
                                          int white_bkgd, const float* target, float grad_scale, float* loss_out,
                                          float* rgb, float* d_raw, void* stream) {
  NERF_REQUIRE(n >= 1 && n <= 1024, NERF_E_SHAPE, "nerf_composite_mse_backward: need 1 <= n <= 1024 (n=%d)", n);
  if (B <= 0) return NERF_OK;
  NERF_REQUIRE(raw && z && rays && target && d_raw, NERF_E_NULL, "nerf_composite_mse_backward: NULL pointer");
  const dim3 g((unsigned)((B + 3) / 4 > 8192 ? 8192 : (B + 3) / 4)), b(256);
  auto st = as_stream(stream);
#define CALL(C) hipLaunchKernelGGL(composite_train_kernel<C>, g, b, 0, st, raw, z, rays, B, n, white_bkgd, target, grad_scale, loss_out, rgb, d_raw)
  DISPATCH_CH(n, CALL);
#undef CALL
  return check_launch("nerf_composite_mse_backward");
}

extern "C" int nerf_mse_loss_grad(const float* pred, const float* target, int64_t count, float grad_scale,
                                  float* loss_out, float* d_pred, void* stream) {
  NERF_REQUIRE(pred && target, NERF_E_NULL, "nerf_mse_loss_grad: pred/target is NULL");
  NERF_REQUIRE(count > 0, NERF_E_SHAPE, "nerf_mse_loss_grad: count must be > 0");
  hipLaunchKernelGGL(mse_kernel, dim3(grid_for(count, 256, 64)), dim3(256), 0, as_stream(stream), pred, target, count,
                     grad_scale, loss_out, d_pred);
  return check_launch("nerf_mse_loss_grad");
}
