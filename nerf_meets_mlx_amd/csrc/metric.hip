// SSIM (ops/metric.py:20-64 of the reference, finished: the upstream body stops at "# TODO" after the five windowed
// moments).  Depthwise VALID convolution with a separable w x w window, then
//   ssim = (2 mu_p mu_g + c1)(2 s_pg + c2) / ((mu_p^2 + mu_g^2 + c1)(s_p^2 + s_g^2 + c2)),   cs = (2 s_pg + c2)/(s_p^2 + s_g^2 + c2)
// summed per image into sums[n][0..1] (double, so the mean does not depend on the atomic order at fp32 precision).
//
// HBM-bound by design: every input pixel is read once per workgroup tile (+ halo), the five moments never leave the
// CU.  A 32 x 8 output tile stages (32 + w - 1) x (8 + w - 1) pixels of both images in LDS, runs the horizontal pass
// into LDS (5 moment planes) and the vertical pass out of it: 2 w MACs x 5 per output instead of w^2 x 5.
// Up to 64 persistent workgroups per image plane walk its tiles and keep their sums in registers; one pair of double
// atomics per workgroup at the end (one pair per wave and tile, as first written, was 237 k same-address atomics for four
// 800 x 800 images: 2.9 ms of serialised read-modify-writes around 20 us of arithmetic).
#include "common.h"

namespace nerf {

constexpr int SSIM_TX = 32, SSIM_TY = 8, SSIM_MAXW = 33;

struct SsimWindow { float w[SSIM_MAXW]; };

__global__ void __launch_bounds__(SSIM_TX * SSIM_TY)
ssim_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int H, int W, int ws, SsimWindow win,
            float c1, float c2, double* __restrict__ sums, int planes_per_image) {
  extern __shared__ float sm[];
  const int IW = SSIM_TX + ws - 1, IH = SSIM_TY + ws - 1;
  float* sp = sm;                       // [IH][IW] pred tile
  float* sg = sp + IH * IW;             // [IH][IW] gt tile
  float* hz = sg + IH * IW;             // [5][IH][SSIM_TX] horizontal pass: p, g, pp, gg, pg
  __shared__ double wsum[2][SSIM_TX * SSIM_TY / WAVE];
  const int plane = blockIdx.z;         // n * C + c
  const int OW = W - ws + 1, OH = H - ws + 1;
  const int tiles_x = (OW + SSIM_TX - 1) / SSIM_TX, tiles_y = (OH + SSIM_TY - 1) / SSIM_TY;
  const float* pp = pred + (int64_t)plane * H * W;
  const float* gp = gt + (int64_t)plane * H * W;
  const int tid = threadIdx.y * SSIM_TX + threadIdx.x, nthr = SSIM_TX * SSIM_TY;
  const int x = threadIdx.x, y = threadIdx.y;
  double s_ssim = 0.0, s_cs = 0.0;
  for (int tile = blockIdx.x; tile < tiles_x * tiles_y; tile += gridDim.x) {
    const int ox0 = (tile % tiles_x) * SSIM_TX, oy0 = (tile / tiles_x) * SSIM_TY;
    __syncthreads();                    // the previous tile's vertical pass is done with hz
    for (int i = tid; i < IH * IW; i += nthr) {
      const int yy = oy0 + i / IW, xx = ox0 + i % IW;
      const bool in = yy < H && xx < W;
      sp[i] = in ? pp[(int64_t)yy * W + xx] : 0.0f;
      sg[i] = in ? gp[(int64_t)yy * W + xx] : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < IH * SSIM_TX; i += nthr) {
      const int yy = i / SSIM_TX, xx = i % SSIM_TX;
      float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
      for (int k = 0; k < ws; ++k) {
        const float wk = win.w[k], p = sp[yy * IW + xx + k], g = sg[yy * IW + xx + k];
        a += wk * p; b += wk * g; aa += wk * (p * p); bb += wk * (g * g); ab += wk * (p * g);
      }
      hz[(0 * IH + yy) * SSIM_TX + xx] = a;  hz[(1 * IH + yy) * SSIM_TX + xx] = b;
      hz[(2 * IH + yy) * SSIM_TX + xx] = aa; hz[(3 * IH + yy) * SSIM_TX + xx] = bb;
      hz[(4 * IH + yy) * SSIM_TX + xx] = ab;
    }
    __syncthreads();
    if (ox0 + x < OW && oy0 + y < OH) {
      float mp = 0.f, mg = 0.f, epp = 0.f, egg = 0.f, epg = 0.f;
      for (int k = 0; k < ws; ++k) {
        const float wk = win.w[k];
        mp += wk * hz[(0 * IH + y + k) * SSIM_TX + x];  mg += wk * hz[(1 * IH + y + k) * SSIM_TX + x];
        epp += wk * hz[(2 * IH + y + k) * SSIM_TX + x]; egg += wk * hz[(3 * IH + y + k) * SSIM_TX + x];
        epg += wk * hz[(4 * IH + y + k) * SSIM_TX + x];
      }
      const float mpp = mp * mp, mgg = mg * mg, mpg = mp * mg;
      const float vp = epp - mpp, vg = egg - mgg, cov = epg - mpg;
      const float v1 = 2.0f * cov + c2, v2 = vp + vg + c2;
      s_cs += (double)(v1 / v2);
      s_ssim += (double)(((2.0f * mpg + c1) * v1) / ((mpp + mgg + c1) * v2));
    }
  }
  s_ssim = wave_sum(s_ssim); s_cs = wave_sum(s_cs);
  if ((tid & (WAVE - 1)) == 0) { wsum[0][tid / WAVE] = s_ssim; wsum[1][tid / WAVE] = s_cs; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < nthr / WAVE; ++w) { a += wsum[0][w]; b += wsum[1][w]; }
    const int n = plane / planes_per_image;
    atomicAdd(sums + 2 * n, a);
    atomicAdd(sums + 2 * n + 1, b);
  }
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_ssim_sums(const float* pred, const float* gt, int N, int C, int H, int W, const float* window_host,
                              int w_size, float c1, float c2, double* sums, void* stream) {
  NERF_REQUIRE(pred && gt && window_host && sums, NERF_E_NULL, "nerf_ssim_sums: NULL pointer");
  NERF_REQUIRE(N >= 1 && C >= 1 && w_size >= 1 && w_size <= SSIM_MAXW, NERF_E_SHAPE,
               "nerf_ssim_sums: need N, C >= 1 and 1 <= w_size <= %d", SSIM_MAXW);
  NERF_REQUIRE(H >= w_size && W >= w_size, NERF_E_SHAPE, "nerf_ssim_sums: image (%d x %d) smaller than the window (%d)",
               H, W, w_size);
  NERF_REQUIRE((int64_t)N * C <= 65535, NERF_E_SHAPE, "nerf_ssim_sums: N*C must be <= 65535 planes per call");
  SsimWindow win;
  for (int k = 0; k < SSIM_MAXW; ++k) win.w[k] = k < w_size ? window_host[k] : 0.0f;
  const int OW = W - w_size + 1, OH = H - w_size + 1;
  const int IW = SSIM_TX + w_size - 1, IH = SSIM_TY + w_size - 1;
  const size_t lds = sizeof(float) * ((size_t)2 * IH * IW + (size_t)5 * IH * SSIM_TX);
  hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 2 * N, as_stream(stream));
  if (e != hipSuccess) return fail(NERF_E_HIP, "nerf_ssim_sums: memset: %s", hipGetErrorString(e));
  const int ntile = ((OW + SSIM_TX - 1) / SSIM_TX) * ((OH + SSIM_TY - 1) / SSIM_TY);
  const dim3 grid(ntile < 64 ? ntile : 64, 1, N * C), block(SSIM_TX, SSIM_TY);
  hipLaunchKernelGGL(ssim_kernel, grid, block, lds, as_stream(stream), pred, gt, H, W, w_size, win, c1, c2, sums, C);
  return check_launch("nerf_ssim_sums");
}
