// Probe for the split-fp16 ("precision 22") MLP design: three hardware facts it rests on.
//   1. does v_mfma_f32_16x16x32_f16 honour fp16 DENORMAL inputs, or flush them?
//   2. how accurate is v_sin_f32 behind a Cody-Waite (two-fma) range reduction, against sin() in double?
//   3. is v_cvt_pk_f16_f32 round-to-nearest-even?
// Build: hipcc -O3 --offload-arch=gfx950 tools/f16_probe.hip -o tools/diag/f16_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void mfma_denorm(float a_val, float b_val, float* out) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}

__global__ void cvt_probe(const float* x, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f32x2 v = {x[i], x[i]};
  h2 h = __builtin_convertvector(v, h2);
  out[i] = (float)h[0];
}

// mode 0: the bf16 kernels' reduction  fract(arg * (1/2pi) + ph)          -> v_sin_f32
// mode 1: Cody-Waite  k = rint(arg / 2pi); r = fma(-k, 2pi_hi, arg); r = fma(-k, 2pi_lo, r); t = r / 2pi (+ ph) -> v_sin_f32
// mode 2: sinf / cosf (OCML)
__global__ void sin_probe(const float* x, const float* f, float* s_out, float* c_out, int n, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float arg = x[i] * f[i];
  float s, c;
  if (mode == 0) {
    s = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(arg * 0.15915494309189535f));
    c = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(arg * 0.15915494309189535f + 0.25f));
  } else if (mode == 1) {
    const float k = __builtin_rintf(arg * 0.15915494309189535f);
    float r = __builtin_fmaf(-k, 6.2831854820251465f, arg);          // 2 pi rounded to float
    r = __builtin_fmaf(-k, -1.7484556000744487e-07f, r);             // 2 pi - float(2 pi)
    const float t = r * 0.15915494309189535f;
    s = __builtin_amdgcn_sinf(t);
    c = __builtin_amdgcn_sinf(t + 0.25f);
  } else {
    s = sinf(arg);
    c = cosf(arg);
  }
  s_out[i] = s; c_out[i] = c;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  float* d_out; CK(hipMalloc(&d_out, 64));
  float h;
  const float sub = ldexpf(1.0f, -20);                 // fp16 subnormal (16 x 2^-24)
  struct { float a, b; const char* what; } cases[] = {
    {sub, 1.0f, "A subnormal 2^-20 x B 1.0 (x32 = 3.0518e-05 if honoured)"},
    {1.0f, sub, "A 1.0 x B subnormal 2^-20"},
    {ldexpf(1.0f, -14), ldexpf(1.0f, -14), "A, B = min normal 2^-14 (product 2^-28, x32 = 1.19e-07)"},
  };
  for (auto& c : cases) {
    hipLaunchKernelGGL(mfma_denorm, dim3(1), dim3(64), 0, 0, c.a, c.b, d_out);
    CK(hipMemcpy(&h, d_out, 4, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"mfma_f16_denorm\", \"case\": \"%s\", \"result\": %.6e}\n", c.what, h);
  }
  // cvt rounding
  {
    float xs[4] = {1.0f + ldexpf(1.0f, -11) + ldexpf(1.0f, -20), 1.0f + ldexpf(1.0f, -11), 1.0f + 3 * ldexpf(1.0f, -11), 70000.0f};
    float *dx, *dy; CK(hipMalloc(&dx, 16)); CK(hipMalloc(&dy, 16));
    CK(hipMemcpy(dx, xs, 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dx, dy, 4);
    float ys[4]; CK(hipMemcpy(ys, dy, 16, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"cvt_pk_f16_f32\", \"just_above_half_ulp\": %.8f, \"tie_even_down\": %.8f, \"tie_even_up\": %.8f, \"overflow_70000\": %f, "
           "\"rne_expected\": [1.00097656, 1.0, 1.00195312, \"inf\"]}\n", ys[0], ys[1], ys[2], ys[3]);
  }
  // sin accuracy: x in [-6, 6], f in {k^2, k = 0..9}
  const int n = 1 << 20;
  std::vector<float> x(n), f(n), s(n), c(n);
  srand(1);
  for (int i = 0; i < n; ++i) { x[i] = 12.0f * (rand() / (float)RAND_MAX) - 6.0f; const int k = rand() % 10; f[i] = (float)(k * k); }
  float *dx, *df, *ds, *dc;
  CK(hipMalloc(&dx, 4 * n)); CK(hipMalloc(&df, 4 * n)); CK(hipMalloc(&ds, 4 * n)); CK(hipMalloc(&dc, 4 * n));
  CK(hipMemcpy(dx, x.data(), 4 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(df, f.data(), 4 * n, hipMemcpyHostToDevice));
  const char* names[3] = {"fract(arg/2pi) + v_sin_f32 (bf16 kernels)", "Cody-Waite 2 fma + v_sin_f32", "sinf / cosf (OCML)"};
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(sin_probe, dim3(n / 256), dim3(256), 0, 0, dx, df, ds, dc, n, mode);
    CK(hipMemcpy(s.data(), ds, 4 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), dc, 4 * n, hipMemcpyDeviceToHost));
    double es = 0, ec = 0, rs = 0;
    for (int i = 0; i < n; ++i) {
      const double arg = (double)(x[i] * f[i]);                  // the reference's float32 product
      const double e1 = fabs((double)s[i] - sin(arg)), e2 = fabs((double)c[i] - cos(arg));
      es = e1 > es ? e1 : es; ec = e2 > ec ? e2 : ec; rs += e1 * e1;
    }
    printf("{\"probe\": \"sin\", \"method\": \"%s\", \"max_abs_err_sin\": %.3e, \"max_abs_err_cos\": %.3e, \"rms_err_sin\": %.3e}\n",
           names[mode], es, ec, sqrt(rs / n));
  }
  return 0;
}
