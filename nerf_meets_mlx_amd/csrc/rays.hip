// Ray generation, pixel selection, row gather, NDC warp  (SURVEY 8a rows a1-a4, K1/K12).
// HBM-bound, one thread per ray; 44 B written per ray, nothing read but the index.
#include "common.h"

namespace nerf {

static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NERF_E_HIP, "%s: %s", what, hipGetErrorString(e));
  return NERF_OK;
}

// ---- keyed bijection of [0, domain): balanced Feistel on 2*half bits + cycle walking.
// Host mirror: nerf_meets_mlx_amd/ops/index.py (must stay bit-identical).
struct PermKey { uint32_t k[4]; int half_bits; };

__host__ __device__ inline uint32_t mix32(uint32_t x) {   // murmur3 finaliser
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__host__ __device__ inline uint64_t feistel(uint64_t v, const PermKey& key) {
  const uint32_t mask = (key.half_bits >= 32) ? 0xFFFFFFFFu : ((1u << key.half_bits) - 1u);
  uint32_t L = (uint32_t)(v >> key.half_bits) & mask, R = (uint32_t)v & mask;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    uint32_t f = mix32(R * 0x9E3779B1u + key.k[r]) & mask;
    uint32_t nl = R; R = L ^ f; L = nl;
  }
  return ((uint64_t)L << key.half_bits) | R;
}
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void perm_kernel(int64_t* out, int64_t n, uint64_t domain, uint64_t offset, PermKey key) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t v = offset + (uint64_t)i;
    do { v = feistel(v, key); } while (v >= domain);      // every cycle of P re-enters [0,domain)
    out[i] = (int64_t)v;
  }
}

struct Cam { double fx, fy, cx, cy; float r[9]; float t[3]; };

__device__ __forceinline__ void write_ray(int64_t i, int64_t p, int W, const Cam& cam, float near, float far,
                                          float* __restrict__ rays, int64_t* __restrict__ coords) {
  const int64_t row = p / W, col = p - row * W;
  // dirs = [(i-cx)/fx, -(j-cy)/fy, -1], i = column, j = row, integer pixel centres
  const double dx = ((double)col - cam.cx) / cam.fx;
  const double dy = -((double)row - cam.cy) / cam.fy;
  const double dz = -1.0;
  double d[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
    d[a] = dx * (double)cam.r[3 * a + 0] + dy * (double)cam.r[3 * a + 1] + dz * (double)cam.r[3 * a + 2];
  const float fx_ = (float)d[0], fy_ = (float)d[1], fz_ = (float)d[2];
  // viewdirs = d / |d| on the float32 d (entrypoints/__test_nerf.py:62-64 works on the f32 array)
  const float inv = 1.0f / sqrtf(fx_ * fx_ + fy_ * fy_ + fz_ * fz_);
  float* o = rays + i * NERF_RAY_STRIDE;
  o[0] = cam.t[0]; o[1] = cam.t[1]; o[2] = cam.t[2];
  o[3] = fx_; o[4] = fy_; o[5] = fz_;
  o[6] = near; o[7] = far;
  o[8] = fx_ * inv; o[9] = fy_ * inv; o[10] = fz_ * inv;
  if (coords) { coords[2 * i] = row; coords[2 * i + 1] = col; }
}

__global__ void ray_gen_kernel(const int64_t* __restrict__ idx, int64_t n, int W, Cam cam, float near, float far,
                               float* __restrict__ rays, int64_t* __restrict__ coords) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    write_ray(i, idx ? idx[i] : i, W, cam, near, far, rays, coords);
}

// One training batch in one launch (entrypoints/__test_nerf.py:213-236 + 60-82): the i-th of n distinct pixels of the
// keyed permutation, its ray, and its target colour from the resident image -- perm_kernel, ray_gen_kernel and
// gather_rows_kernel back to back, with the same arithmetic (bit-identical outputs), as ONE kernel: at N_rand = 1024 an
// iteration is ~1.3 ms and each five-microsecond launch shows.
__global__ void sample_batch_kernel(int64_t n, uint64_t domain, uint64_t offset, PermKey key, int W, Cam cam, float near,
                                    float far, const float* __restrict__ image, float* __restrict__ rays,
                                    float* __restrict__ target, int64_t* __restrict__ idx_out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t v = offset + (uint64_t)i;
    do { v = feistel(v, key); } while (v >= domain);
    write_ray(i, (int64_t)v, W, cam, near, far, rays, nullptr);
    target[3 * i] = image[3 * v]; target[3 * i + 1] = image[3 * v + 1]; target[3 * i + 2] = image[3 * v + 2];
    if (idx_out) idx_out[i] = (int64_t)v;
  }
}

// An index outside [0, n_src) never reads: its output row is NaN (the launch is asynchronous, so it cannot fail)
__global__ void gather_rows_kernel(const float* __restrict__ src, int64_t n_src, const int64_t* __restrict__ idx,
                                   int64_t n, int C, float* __restrict__ out) {
  const int64_t total = n * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / C; const int c = (int)(t - i * C);
    const int64_t row = idx[i];
    out[t] = (row >= 0 && row < n_src) ? src[row * C + c] : __builtin_nanf("");
  }
}

__global__ void ndc_kernel(float* rays, int64_t n, float sx, float sy, float near) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float* r = rays + i * NERF_RAY_STRIDE;
    float ox = r[0], oy = r[1], oz = r[2], dx = r[3], dy = r[4], dz = r[5];
    const float tn = -(near + oz) / dz;
    ox += tn * dx; oy += tn * dy; oz += tn * dz;
    r[0] = sx * (ox / oz); r[1] = sy * (oy / oz); r[2] = 1.0f + 2.0f * near / oz;
    r[3] = sx * (dx / dz - ox / oz); r[4] = sy * (dy / dz - oy / oz); r[5] = -2.0f * near * (1.0f / oz);
  }
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_abi_version(void) { return NERF_ABI_VERSION; }
extern "C" const char* nerf_last_error(void) { return err_buf(); }

extern "C" int nerf_pixel_permutation(int64_t* out_idx, int64_t n, int64_t domain, uint64_t seed, uint64_t offset,
                                      void* stream) {
  NERF_REQUIRE(out_idx, NERF_E_NULL, "nerf_pixel_permutation: out_idx is NULL");
  NERF_REQUIRE(n >= 0 && domain >= 1 && (int64_t)offset + n <= domain, NERF_E_SHAPE,
               "nerf_pixel_permutation: need offset+n <= domain (n=%lld domain=%lld)", (long long)n, (long long)domain);
  if (n == 0) return NERF_OK;
  int bits = 2;
  while (bits < 64 && (1ull << bits) < (uint64_t)domain) ++bits;
  if (bits & 1) ++bits;
  PermKey key;
  key.half_bits = bits / 2;
  for (int r = 0; r < 4; ++r) key.k[r] = (uint32_t)splitmix64(seed + (uint64_t)r);
  hipLaunchKernelGGL(perm_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), out_idx, n,
                     (uint64_t)domain, offset, key);
  return check_launch("nerf_pixel_permutation");
}

extern "C" int nerf_ray_gen(const int64_t* pixel_idx, int64_t n, int H, int W, const double* K, const float* c2w,
                            float near, float far, float* rays, int64_t* coords, void* stream) {
  NERF_REQUIRE(H > 0 && W > 0 && n >= 0, NERF_E_SHAPE, "nerf_ray_gen: bad H/W/n");
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(K && c2w && rays, NERF_E_NULL, "nerf_ray_gen: K/c2w/rays is NULL");
  NERF_REQUIRE(pixel_idx || n == (int64_t)H * W, NERF_E_SHAPE, "nerf_ray_gen: pixel_idx NULL requires n == H*W");
  if (n == 0) return NERF_OK;
  Cam cam;
  cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) cam.r[3 * a + b] = c2w[4 * a + b];
    cam.t[a] = c2w[4 * a + 3];
  }
  hipLaunchKernelGGL(ray_gen_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), pixel_idx, n, W, cam,
                     near, far, rays, coords);
  return check_launch("nerf_ray_gen");
}

static PermKey make_key(uint64_t seed, uint64_t domain) {
  int bits = 2;
  while (bits < 64 && (1ull << bits) < domain) ++bits;
  if (bits & 1) ++bits;
  PermKey key;
  key.half_bits = bits / 2;
  for (int r = 0; r < 4; ++r) key.k[r] = (uint32_t)splitmix64(seed + (uint64_t)r);
  return key;
}
static Cam make_cam(const double* K, const float* c2w) {
  Cam cam;
  cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) cam.r[3 * a + b] = c2w[4 * a + b];
    cam.t[a] = c2w[4 * a + 3];
  }
  return cam;
}

extern "C" int nerf_sample_batch(int64_t n, int H, int W, uint64_t seed, uint64_t offset, const double* K,
                                 const float* c2w, float near, float far, const float* image, float* rays,
                                 float* target, int64_t* pixel_idx, void* stream) {
  NERF_REQUIRE(H > 0 && W > 0 && n >= 0, NERF_E_SHAPE, "nerf_sample_batch: bad H/W/n");
  NERF_REQUIRE((int64_t)offset + n <= (int64_t)H * W, NERF_E_SHAPE,
               "nerf_sample_batch: need offset+n <= H*W (n=%lld H*W=%lld)", (long long)n, (long long)H * W);
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(K && c2w && image && rays && target, NERF_E_NULL, "nerf_sample_batch: NULL pointer");
  const uint64_t domain = (uint64_t)H * (uint64_t)W;
  hipLaunchKernelGGL(sample_batch_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), n, domain, offset,
                     make_key(seed, domain), W, make_cam(K, c2w), near, far, image, rays, target, pixel_idx);
  return check_launch("nerf_sample_batch");
}

extern "C" int nerf_gather_rows(const float* src, int64_t n_src, const int64_t* idx, int64_t n, int channels,
                                float* out, void* stream) {
  NERF_REQUIRE(n >= 0 && channels > 0, NERF_E_SHAPE, "nerf_gather_rows: bad sizes");
  if (n == 0) return NERF_OK;
  NERF_REQUIRE(src && idx && out, NERF_E_NULL, "nerf_gather_rows: NULL pointer");
  NERF_REQUIRE(n_src > 0, NERF_E_SHAPE, "nerf_gather_rows: empty source");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n * channels, 256)), dim3(256), 0, as_stream(stream), src, n_src,
                     idx, n, channels, out);
  return check_launch("nerf_gather_rows");
}

extern "C" int nerf_ndc_rays(float* rays, int64_t n, int H, int W, float focal, float near, void* stream) {
  if (n <= 0) return NERF_OK;
  NERF_REQUIRE(rays, NERF_E_NULL, "nerf_ndc_rays: rays is NULL");
  hipLaunchKernelGGL(ndc_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), rays, n,
                     -focal / (0.5f * W), -focal / (0.5f * H), near);
  return check_launch("nerf_ndc_rays");
}
