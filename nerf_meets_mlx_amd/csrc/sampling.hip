// Depth sampling (a5-a7) and inverse-CDF importance sampling + merge (a15, a17 / K2, K8).
// HBM-bound: one wavefront per ray, CDF / depths staged in LDS, per-lane binary search,
// LDS rank-merge of the n sorted + N new depths.
#include "common.h"

namespace nerf {

__global__ void sample_coarse_kernel(const float* __restrict__ rays, int64_t B, int n, int lindisp, float perturb,
                                     const float* __restrict__ t_rand, float step, float* __restrict__ z) {
  const int64_t total = B * n;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = t / n; const int k = (int)(t - b * n);
    const float near = rays[b * NERF_RAY_STRIDE + 6], far = rays[b * NERF_RAY_STRIDE + 7];
    auto zk = [&](int kk) -> float {
      const float tv = (float)kk * step + 0.0f;                   // mx.linspace(0,1,n)
      if (!lindisp) return near * (1.0f - tv) + far * tv;         // sampling/uniform.py:14-16
      return 1.0f / (1.0f / (near * (1.0f - tv)) + 1.0f / (far * tv));  // linear_disparity.py:15-17 (literal)
    };
    float v = zk(k);
    if (perturb > 0.0f) {                                         // sampling/__init__.py:17-29
      const float lo = (k == 0) ? v : 0.5f * (zk(k - 1) + v);
      const float hi = (k == n - 1) ? v : 0.5f * (v + zk(k + 1));
      v = lo + (hi - lo) * (t_rand[t] * perturb);
    }
    z[t] = v;
  }
}

// stand-alone stratified jitter on caller-supplied depths (sampling/__init__.py:10-31, intended semantics: SURVEY Q6)
__global__ void add_noise_z_kernel(const float* __restrict__ z_in, const float* __restrict__ t_rand, int64_t B, int n,
                                   float strength, float* __restrict__ z_out) {
  const int64_t total = B * n;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(t % n);
    const float v = z_in[t];
    const float lo = (k == 0) ? v : 0.5f * (z_in[t - 1] + v);               // lower = [z_first, mids]
    const float hi = (k == n - 1) ? v : 0.5f * (v + z_in[t + 1]);           // upper = [mids, z_last]
    z_out[t] = lo + (hi - lo) * (t_rand[t] * strength);
  }
}

// ---- importance sampling -------------------------------------------------------------------
// LDS per wave: cdf[n+1], zmid[n+1], merged[n+N] (floats).
template <int CH>   // CH = ceil(n/64): consecutive bins handled by one lane in the scan
__global__ void __launch_bounds__(256) importance_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                         const float* __restrict__ u, int64_t B, int n, int N,
                                                         int P, float eps, float* __restrict__ z_new,
                                                         float* __restrict__ z_merged, float* __restrict__ cdf_out,
                                                         int64_t* __restrict__ inds_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int per_wave = (n + 1) * 2 + ((n + P + 3) & ~3);
  float* s_cdf = smem + wv * per_wave;
  float* s_zmid = s_cdf + (n + 1);
  float* s_all = s_zmid + (n + 1);
  for (int64_t ray = blockIdx.x * 4 + wv; ray < B; ray += (int64_t)gridDim.x * 4) {
    const float* zr = z + ray * n;
    const float* wr = w + ray * n;
    // weights + 0.01, sum (float64 tree: at least as accurate as torch.sum), padding
    float wl[CH];
    double part = 0.0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k = lane * CH + c;
      wl[c] = (k < n) ? wr[k] + 0.01f : 0.0f;
      part += (double)wl[c];
    }
    float s = (float)wave_sum(part);
    const float pad = fmaxf(eps - s, 0.0f);
    const float padw = pad / (float)n;
    s = s + pad;
    // pdf, inclusive cumsum accumulated in float64 and rounded per element (torch CPU cumsum)
    double run = 0.0;
    double loc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k = lane * CH + c;
      const float pdf = (k < n) ? (wl[c] + padw) / s : 0.0f;
      run += (double)pdf;
      loc[c] = run;
    }
    const double incl = wave_scan_incl(run, lane);
    const double excl = incl - run;
    if (lane == 0) s_cdf[0] = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k = lane * CH + c;
      if (k < n) s_cdf[k + 1] = fminf(1.0f, (float)(excl + loc[c]));
    }
    // z_mid padded to n+1 by duplicating first and last (sampling/__init__.py:149-159)
    for (int i = lane; i <= n; i += WAVE) {
      int a = i - 1; a = a < 0 ? 0 : (a > n - 2 ? n - 2 : a);
      s_zmid[i] = (zr[a + 1] + zr[a]) / 2.0f;
    }
    for (int i = lane; i < n; i += WAVE) s_all[i] = zr[i];
    __builtin_amdgcn_s_waitcnt(0);          // wave-private LDS: order own writes before reads
    __builtin_amdgcn_wave_barrier();
    if (cdf_out) for (int i = lane; i <= n; i += WAVE) cdf_out[ray * (n + 1) + i] = s_cdf[i];
    // per-lane binary search: inds = #{i : cdf[i] <= u}  (searchsorted side="right")
    for (int j = lane; j < N; j += WAVE) {
      const float uj = u[ray * N + j];
      const bool unan = uj != uj;                                  // torch.searchsorted orders NaN after everything: inds = n + 1
      int lo = 0, hi = n + 1;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (unan || s_cdf[mid] <= uj) lo = mid + 1; else hi = mid; }
      const int inds = lo;
      int below = inds - 1; below = below < 0 ? 0 : (below > n ? n : below);
      int above = inds;     above = above > n ? n : above;
      const float cf = s_cdf[below], ct = s_cdf[above];
      const float zf = s_zmid[below], zt = s_zmid[above];
      float den = ct - cf;
      den = (den < eps) ? 1.0f : den;
      float t = (uj - cf) / den;
      if (t != t) t = 0.0f;                                        // nan_to_num(.., 0)
      t = fminf(fmaxf(t, 0.0f), 1.0f);                             // (+-inf -> +-FLT_MAX -> clipped)
      const float zn = zf + t * (zt - zf);
      if (z_new) z_new[ray * N + j] = zn;
      if (inds_out) inds_out[ray * N + j] = inds;
      s_all[n + j] = zn;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (z_merged) {
      const int tot = n + N;
      float* s_new = s_all + n;
      // is the coarse list ascending (it always is on the render path)?  Then: bitonic sort of the N new depths in
      // LDS + two binary-search rank passes = O((n+N) log) instead of the O((n+N)^2) rank sort below.
      bool asc = true;
      for (int i = lane; i + 1 < n; i += WAVE) asc = asc && (s_all[i] <= s_all[i + 1]);
      if (__all(asc)) {
        for (int i = N + lane; i < P; i += WAVE) s_new[i] = __builtin_inff();          // pad to a power of two
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        for (int k = 2; k <= P; k <<= 1) {
          for (int j = k >> 1; j > 0; j >>= 1) {
            for (int pi = lane; pi < (P >> 1); pi += WAVE) {
              const int i = ((pi & ~(j - 1)) << 1) | (pi & (j - 1));
              const int q = i | j;
              const float a = s_new[i], b = s_new[q];
              const bool up = (i & k) == 0;
              if ((a > b) == up) { s_new[i] = b; s_new[q] = a; }
            }
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
          }
        }
        // ties: coarse depths first.  rank(z_i) = i + #{new < z_i};  rank(new_j) = j + #{z <= new_j}
        for (int i = lane; i < n; i += WAVE) {
          const float v = s_all[i];
          int lo = 0, hi = N;
          while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_new[mid] < v) lo = mid + 1; else hi = mid; }
          z_merged[ray * tot + i + lo] = v;
        }
        for (int j = lane; j < N; j += WAVE) {
          const float v = s_new[j];
          int lo = 0, hi = n;
          while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_all[mid] <= v) lo = mid + 1; else hi = mid; }
          z_merged[ray * tot + j + lo] = v;
        }
      } else {
        // general input: stable rank sort of the n+N depths (ties broken by position), NaN last like torch.sort
        for (int i = lane; i < tot; i += WAVE) {
          const float v = s_all[i];
          int rank = 0;
          for (int k = 0; k < tot; ++k) {
            const float o = s_all[k];
            const bool less = (o < v) || (v != v && o == o) || ((o == v || (o != o && v != v)) && k < i);
            rank += less ? 1 : 0;
          }
          z_merged[ray * tot + rank] = v;
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_sample_coarse(const float* rays, int64_t B, int n, int lindisp, float perturb,
                                  const float* t_rand, float* z, void* stream) {
  NERF_REQUIRE(B >= 0 && n >= 2, NERF_E_SHAPE, "nerf_sample_coarse: need n >= 2 (n=%d)", n);
  if (B == 0) return NERF_OK;
  NERF_REQUIRE(rays && z, NERF_E_NULL, "nerf_sample_coarse: rays/z is NULL");
  NERF_REQUIRE(perturb <= 0.0f || t_rand, NERF_E_NULL, "nerf_sample_coarse: perturb > 0 needs t_rand");
  if (B == 0) return NERF_OK;
  const float step = (float)((1.0 - 0.0) / (double)(n - 1));
  hipLaunchKernelGGL(sample_coarse_kernel, dim3(grid_for(B * n, 256)), dim3(256), 0, as_stream(stream), rays, B, n,
                     lindisp, perturb, t_rand, step, z);
  return check_launch("nerf_sample_coarse");
}

extern "C" int nerf_add_noise_z(const float* z_in, const float* t_rand, int64_t B, int n, float strength, float* z_out,
                               void* stream) {
  NERF_REQUIRE(B >= 0 && n >= 1, NERF_E_SHAPE, "nerf_add_noise_z: bad B/n");
  if (B == 0) return NERF_OK;
  NERF_REQUIRE(z_in && t_rand && z_out, NERF_E_NULL, "nerf_add_noise_z: NULL pointer");
  NERF_REQUIRE(z_in != z_out, NERF_E_SHAPE, "nerf_add_noise_z: in-place use is not supported (neighbours are read)");
  hipLaunchKernelGGL(add_noise_z_kernel, dim3(grid_for(B * n, 256)), dim3(256), 0, as_stream(stream), z_in, t_rand, B, n,
                     strength, z_out);
  return check_launch("nerf_add_noise_z");
}

extern "C" int nerf_importance_sample(const float* z, const float* weights, const float* u, int64_t B, int n, int N,
                                      float eps, float* z_new, float* z_merged, float* cdf, int64_t* inds,
                                      void* stream) {
  NERF_REQUIRE(n >= 2 && n <= 256 && N >= 1 && N <= 512 && n + N <= 768, NERF_E_SHAPE,
               "nerf_importance_sample: unsupported n=%d N=%d (2<=n<=256, 1<=N<=512)", n, N);
  if (B <= 0) return NERF_OK;
  NERF_REQUIRE(z && weights && u, NERF_E_NULL, "nerf_importance_sample: z/weights/u is NULL");
  int P = 2;
  while (P < N) P <<= 1;                                  // bitonic sort width of the new depths
  const int per_wave = (n + 1) * 2 + ((n + P + 3) & ~3);
  const size_t lds = (size_t)per_wave * 4 * sizeof(float);
  const int grid = (int)((B + 3) / 4 > 256 * 8 ? 256 * 8 : (B + 3) / 4);
  const int ch = (n + 63) / 64;
  auto st = as_stream(stream);
#define LAUNCH(C) hipLaunchKernelGGL(importance_kernel<C>, dim3(grid), dim3(256), lds, st, z, weights, u, B, n, N, P, eps, z_new, z_merged, cdf, inds)
  if (ch == 1) LAUNCH(1); else if (ch == 2) LAUNCH(2); else if (ch == 3) LAUNCH(3); else LAUNCH(4);
#undef LAUNCH
  return check_launch("nerf_importance_sample");
}
