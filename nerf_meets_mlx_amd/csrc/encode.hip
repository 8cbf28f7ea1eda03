// Standalone encoders (a9, a10, a22, a23 / K3, K10, K11).  The NeRF render path uses the
// positional encoding fused into the MLP kernel (mlp.hip); these entry points exist for the
// reference's `Encoding` classes and for parity tests, and are HBM/L2-gather bound.
#include "common.h"

namespace nerf {

// [x, sin(f0 x), cos(f0 x), ...]  models/embedding.py:30-71
__global__ void encode_freq_kernel(const float* __restrict__ x, int64_t M, int D, int L, int mode,
                                   float* __restrict__ out) {
  const int C = D * (1 + 2 * L);
  const int64_t total = M * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = t / C; const int c = (int)(t - m * C);
    float v;
    if (c < D) {
      v = x[m * D + c];
    } else {
      const int q = c - D, band = q / (2 * D), w = q - band * 2 * D;
      const int dim = (w >= D) ? w - D : w;
      const float f = mode == 0 ? (float)(band * band) : (float)(1u << band);   // k^2 (Q4) or 2^k
      const float a = x[m * D + dim] * f;
      v = (w >= D) ? cosf(a) : sinf(a);
    }
    out[t] = v;
  }
}

struct FreqTab { float f[32]; };

// sin(concat[s, s + pi/2]) (+ raw input at the end)  encoding/sinusoidal.py:52-64
__global__ void encode_sinusoidal_kernel(const float* __restrict__ x, int64_t M, int D, int L, FreqTab ft,
                                         int include_input, float* __restrict__ out) {
  const int DL = D * L;
  const int C = 2 * DL + (include_input ? D : 0);
  const int64_t total = M * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = t / C; const int c = (int)(t - m * C);
    float v;
    if (c >= 2 * DL) {
      v = x[m * D + (c - 2 * DL)];
    } else {
      const int cc = c >= DL ? c - DL : c;
      const int dim = cc / L, k = cc - dim * L;
      float s = x[m * D + dim] * ft.f[k];
      if (c >= DL) s = s + 1.57079637050628662109375f;   // float32(pi/2)
      v = sinf(s);
    }
    out[t] = v;
  }
}

// real SH basis  encoding/spherical_harmonics.py:62-93 (same operation order)
__global__ void sh_kernel(const float* __restrict__ d, int64_t M, int deg, float* __restrict__ out) {
  const int C = (deg + 1) * (deg + 1);
  for (int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    const float x = d[3 * m], y = d[3 * m + 1], z = d[3 * m + 2];
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    float* o = out + m * C;
    o[0] = 0.28209479177387814f;
    if (deg >= 1) { o[1] = 0.4886025119029199f * y; o[2] = 0.4886025119029199f * z; o[3] = 0.4886025119029199f * x; }
    if (deg >= 2) {
      o[4] = 1.0925484305920792f * xy; o[5] = 1.0925484305920792f * yz;
      o[6] = 0.9461746957575601f * zz - 0.31539156525251999f;
      o[7] = 1.0925484305920792f * xz; o[8] = 0.5462742152960396f * (xx - yy);
    }
    if (deg >= 3) {
      o[9] = 0.5900435899266435f * y * (3 * xx - yy); o[10] = 2.890611442640554f * xy * z;
      o[11] = 0.4570457994644658f * y * (5 * zz - 1); o[12] = 0.3731763325901154f * z * (5 * zz - 3);
      o[13] = 0.4570457994644658f * x * (5 * zz - 1); o[14] = 1.445305721320277f * z * (xx - yy);
      o[15] = 0.5900435899266435f * x * (xx - 3 * yy);
    }
    if (deg >= 4) {
      o[16] = 2.5033429417967046f * xy * (xx - yy); o[17] = 1.7701307697799304f * yz * (3 * xx - yy);
      o[18] = 0.9461746957575601f * xy * (7 * zz - 1); o[19] = 0.6690465435572892f * yz * (7 * zz - 3);
      o[20] = 0.10578554691520431f * (35 * zz * zz - 30 * zz + 3);
      o[21] = 0.6690465435572892f * xz * (7 * zz - 3); o[22] = 0.47308734787878004f * (xx - yy) * (7 * zz - 1);
      o[23] = 1.7701307697799304f * xz * (xx - 3 * yy);
      o[24] = 0.6258357354491761f * (xx * (xx - 3 * yy) - yy * (3 * xx - yy));
    }
  }
}

// ---- multires hash grid  encoding/multi_hash.py:61-136 (intended semantics) ------------------
struct ResTab { float res[32]; };

__device__ __forceinline__ uint32_t hash3(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t mask) {
  return ((cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u)) & mask;     // uint32 wrap-around, mod T = & (T-1)
}

// One thread per sample, blockIdx.y = a group of LG consecutive levels: at any time a workgroup's 256 neighbouring
// samples gather from ONE level table (the coarse ones are a few KiB and stay in L1/L2; with one thread per
// (sample, level) every lane of a wave hit a different 4 MiB table), and a thread's LG x F outputs / incoming gradients
// are contiguous in the row (32 B for LG = 4, F = 2) instead of 8-byte pieces at a 128-byte stride.
template <int F, bool BWD, int LG>
__global__ void __launch_bounds__(256) hashgrid_kernel(const float* __restrict__ x, int64_t M,
                                                       const float* __restrict__ tables, float* __restrict__ d_tables,
                                                       const float* __restrict__ d_out, int L, uint32_t T, ResTab rt,
                                                       float* __restrict__ out) {
  const int l0 = blockIdx.y * LG;
  const uint32_t mask = T - 1;
  for (int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    const float px = x[3 * m], py = x[3 * m + 1], pz = x[3 * m + 2];
    float vals[LG * F];
    if (BWD) {
#pragma unroll
      for (int i = 0; i < LG * F; ++i) vals[i] = (l0 + i / F < L) ? d_out[(m * L + l0) * F + i] : 0.0f;
    }
#pragma unroll
    for (int li = 0; li < LG; ++li) {
      const int l = l0 + li;
      if (l >= L) break;
      const float r = rt.res[l];
      const float p[3] = {px, py, pz};
      float off[3]; uint32_t cf[3], cc[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float xs = p[a] * r;
        const float fl = floorf(xs);
        off[a] = xs - fl;
        cf[a] = (uint32_t)(int32_t)fl; cc[a] = (uint32_t)(int32_t)ceilf(xs);
      }
      // reference corner numbering: 0=(c,c,c) 1=(c,f,c) 2=(f,f,c) 3=(f,c,c) 4=(c,c,f) 5=(c,f,f) 6=(f,f,f) 7=(f,c,f)
      const uint32_t i0 = hash3(cc[0], cc[1], cc[2], mask), i1 = hash3(cc[0], cf[1], cc[2], mask);
      const uint32_t i2 = hash3(cf[0], cf[1], cc[2], mask), i3 = hash3(cf[0], cc[1], cc[2], mask);
      const uint32_t i4 = hash3(cc[0], cc[1], cf[2], mask), i5 = hash3(cc[0], cf[1], cf[2], mask);
      const uint32_t i6 = hash3(cf[0], cf[1], cf[2], mask), i7 = hash3(cf[0], cc[1], cf[2], mask);
      const float ox = off[0], oy = off[1], oz = off[2];
      const size_t base = (size_t)l * T * F;
      if (!BWD) {
        const float* tb = tables + base;
#pragma unroll
        for (int f = 0; f < F; ++f) {
          const float h03 = tb[(size_t)i0 * F + f] * ox + tb[(size_t)i3 * F + f] * (1 - ox);
          const float h12 = tb[(size_t)i1 * F + f] * ox + tb[(size_t)i2 * F + f] * (1 - ox);
          const float h56 = tb[(size_t)i5 * F + f] * ox + tb[(size_t)i6 * F + f] * (1 - ox);
          const float h47 = tb[(size_t)i4 * F + f] * ox + tb[(size_t)i7 * F + f] * (1 - ox);
          const float h0312 = h03 * oy + h12 * (1 - oy);
          const float h4756 = h47 * oy + h56 * (1 - oy);
          vals[li * F + f] = h0312 * oz + h4756 * (1 - oz);
        }
      } else {
        float* tb = d_tables + base;
        const float wx1 = ox, wx0 = 1 - ox, wy1 = oy, wy0 = 1 - oy, wz1 = oz, wz0 = 1 - oz;
#pragma unroll
        for (int f = 0; f < F; ++f) {
          const float g = vals[li * F + f];
          atomicAdd(tb + (size_t)i0 * F + f, g * wz1 * wy1 * wx1);
          atomicAdd(tb + (size_t)i3 * F + f, g * wz1 * wy1 * wx0);
          atomicAdd(tb + (size_t)i1 * F + f, g * wz1 * wy0 * wx1);
          atomicAdd(tb + (size_t)i2 * F + f, g * wz1 * wy0 * wx0);
          atomicAdd(tb + (size_t)i4 * F + f, g * wz0 * wy1 * wx1);
          atomicAdd(tb + (size_t)i7 * F + f, g * wz0 * wy1 * wx0);
          atomicAdd(tb + (size_t)i5 * F + f, g * wz0 * wy0 * wx1);
          atomicAdd(tb + (size_t)i6 * F + f, g * wz0 * wy0 * wx0);
        }
      }
    }
    if (!BWD) {
#pragma unroll
      for (int i = 0; i < LG * F; ++i)
        if (l0 + i / F < L) out[(m * L + l0) * F + i] = vals[i];
    }
  }
}

template <bool BWD>
static int launch_hashgrid(const float* x, int64_t M, const float* tables, float* d_tables, const float* d_out, int L,
                           int log2_T, int F, const int* res, float* out, void* stream, const char* who) {
  NERF_REQUIRE(x && res, NERF_E_NULL, "%s: NULL pointer", who);
  NERF_REQUIRE(L >= 1 && L <= 32 && log2_T >= 1 && log2_T <= 30, NERF_E_SHAPE, "%s: need 1<=L<=32, 1<=log2_T<=30", who);
  NERF_REQUIRE(F == 1 || F == 2 || F == 4 || F == 8, NERF_E_UNSUPPORTED, "%s: F must be 1, 2, 4 or 8", who);
  if (M <= 0) return NERF_OK;
  ResTab rt;
  for (int l = 0; l < L; ++l) rt.res[l] = (float)res[l];
  const uint32_t T = 1u << log2_T;
  constexpr int LG = 4;                                   // levels per thread: LG x F contiguous floats per sample
  const dim3 g(grid_for(M, 256), (unsigned)((L + LG - 1) / LG)), b(256);
  auto st = as_stream(stream);
#define HG(FF) hipLaunchKernelGGL((hashgrid_kernel<FF, BWD, LG>), g, b, 0, st, x, M, tables, d_tables, d_out, L, T, rt, out)
  switch (F) { case 1: HG(1); break; case 2: HG(2); break; case 4: HG(4); break; default: HG(8); }
#undef HG
  return check_launch(who);
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_encode_freq(const float* x, int64_t M, int D, int n_freqs, int freq_mode, float* out,
                                void* stream) {
  if (M <= 0) return NERF_OK;
  NERF_REQUIRE(x && out, NERF_E_NULL, "nerf_encode_freq: x/out is NULL");
  NERF_REQUIRE(D >= 1 && D <= 8 && n_freqs >= 0 && n_freqs <= 16, NERF_E_SHAPE, "nerf_encode_freq: bad D/n_freqs");
  NERF_REQUIRE(freq_mode == 0 || freq_mode == 1, NERF_E_UNSUPPORTED, "nerf_encode_freq: freq_mode must be 0 or 1");
  if (M <= 0) return NERF_OK;
  hipLaunchKernelGGL(encode_freq_kernel, dim3(grid_for(M * D * (1 + 2 * n_freqs), 256)), dim3(256), 0,
                     as_stream(stream), x, M, D, n_freqs, freq_mode, out);
  return check_launch("nerf_encode_freq");
}

extern "C" int nerf_encode_sinusoidal(const float* x, int64_t M, int D, int n_freqs, const float* freqs_host,
                                      int include_input, float* out, void* stream) {
  if (M <= 0) return NERF_OK;
  NERF_REQUIRE(x && out && freqs_host, NERF_E_NULL, "nerf_encode_sinusoidal: NULL pointer");
  NERF_REQUIRE(D >= 1 && n_freqs >= 1 && n_freqs <= 32, NERF_E_SHAPE, "nerf_encode_sinusoidal: need 1<=n_freqs<=32");
  if (M <= 0) return NERF_OK;
  FreqTab ft;
  for (int k = 0; k < 32; ++k) ft.f[k] = k < n_freqs ? freqs_host[k] : 0.0f;
  const int C = 2 * D * n_freqs + (include_input ? D : 0);
  hipLaunchKernelGGL(encode_sinusoidal_kernel, dim3(grid_for(M * C, 256)), dim3(256), 0, as_stream(stream), x, M, D,
                     n_freqs, ft, include_input, out);
  return check_launch("nerf_encode_sinusoidal");
}

extern "C" int nerf_sh_encode(const float* dirs, int64_t M, int degree, float* out, void* stream) {
  NERF_REQUIRE(degree >= 0 && degree <= 4, NERF_E_SHAPE, "nerf_sh_encode: n_degrees=%d must be in range [0, 4]", degree);
  if (M <= 0) return NERF_OK;
  NERF_REQUIRE(dirs && out, NERF_E_NULL, "nerf_sh_encode: NULL pointer");
  hipLaunchKernelGGL(sh_kernel, dim3(grid_for(M, 256)), dim3(256), 0, as_stream(stream), dirs, M, degree, out);
  return check_launch("nerf_sh_encode");
}

extern "C" int nerf_hashgrid_forward(const float* x, int64_t M, const float* tables, int L, int log2_T, int F,
                                     const int* resolutions_host, float* out, void* stream) {
  NERF_REQUIRE(tables && out, NERF_E_NULL, "nerf_hashgrid_forward: tables/out is NULL");
  return launch_hashgrid<false>(x, M, tables, nullptr, nullptr, L, log2_T, F, resolutions_host, out, stream,
                                "nerf_hashgrid_forward");
}

extern "C" int nerf_hashgrid_backward(const float* x, int64_t M, const float* d_out, int L, int log2_T, int F,
                                      const int* resolutions_host, float* d_tables, void* stream) {
  NERF_REQUIRE(d_out && d_tables, NERF_E_NULL, "nerf_hashgrid_backward: d_out/d_tables is NULL");
  return launch_hashgrid<true>(x, M, nullptr, d_tables, d_out, L, log2_T, F, resolutions_host, nullptr, stream,
                               "nerf_hashgrid_backward");
}
