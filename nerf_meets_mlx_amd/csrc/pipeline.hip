// End-to-end render_rays_eval in one C call (a14 / a18), and the RCCL gradient all-reduce (C1).
#include "common.h"
#include <dlfcn.h>
#include <string.h>

using namespace nerf;

// ---- fused renderer: rendering/render.py:164-241 as a fixed launch sequence on the caller's stream -------------
extern "C" int64_t nerf_render_workspace_bytes(int64_t B, int n, int N) {
  if (B < 0 || n < 1 || N < 0) return -1;
  const int64_t nf = n + N;
  return B * (int64_t)(n /*z*/ + 4 * n /*raw*/ + n /*w*/ + nf /*z_fine*/ + 4 * nf /*raw_fine*/) * (int64_t)sizeof(float);
}

extern "C" int nerf_render_rays_fused(const nerf_mlp_arch* arch, const void* packed_coarse, const void* packed_fine,
                                      const float* rays, int64_t B, int n, int N, const float* u, int freq_mode,
                                      int white_bkgd, void* workspace, float* rgb, float* disp, float* acc,
                                      float* rgb_coarse, float* disp_coarse, float* acc_coarse, float* z_vals,
                                      float* weights, void* stream) {
  NERF_REQUIRE(B >= 0 && n >= 2 && N >= 0, NERF_E_SHAPE, "nerf_render_rays_fused: bad B/n/N");
  if (B == 0) return NERF_OK;
  NERF_REQUIRE(arch && packed_coarse && rays && workspace && rgb, NERF_E_NULL, "nerf_render_rays_fused: NULL pointer");
  NERF_REQUIRE(N == 0 || u, NERF_E_NULL, "nerf_render_rays_fused: N > 0 needs the uniforms u [B,N]");
  float* ws = static_cast<float*>(workspace);
  float* z = ws;                       // [B,n]
  float* raw = z + B * n;              // [B,n,4]
  float* w = raw + B * n * 4;          // [B,n]
  float* zf = w + B * n;               // [B,n+N]
  float* rawf = zf + B * (n + N);      // [B,n+N,4]
  int rc;
  if ((rc = nerf_sample_coarse(rays, B, n, 0, 0.0f, nullptr, z, stream))) return rc;
  if ((rc = nerf_query_fused(arch, packed_coarse, rays, z, B, n, freq_mode, raw, nullptr, stream))) return rc;
  float* rgb_c = N > 0 ? (rgb_coarse ? rgb_coarse : rgb) : rgb;      // with N == 0 the coarse result IS the result
  if ((rc = nerf_composite_forward(raw, z, rays, B, n, 0.0f, nullptr, white_bkgd, rgb_c, N > 0 ? disp_coarse : disp,
                                   N > 0 ? acc_coarse : acc, w, nullptr, stream))) return rc;
  hipStream_t s = as_stream(stream);
  if (z_vals && hipMemcpyAsync(z_vals, z, sizeof(float) * B * n, hipMemcpyDeviceToDevice, s) != hipSuccess)
    return fail(NERF_E_HIP, "nerf_render_rays_fused: copy of z_vals failed");
  if (weights && hipMemcpyAsync(weights, w, sizeof(float) * B * n, hipMemcpyDeviceToDevice, s) != hipSuccess)
    return fail(NERF_E_HIP, "nerf_render_rays_fused: copy of weights failed");
  if (N == 0) {
    if (rgb_coarse && rgb_coarse != rgb &&
        hipMemcpyAsync(rgb_coarse, rgb, sizeof(float) * B * 3, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return fail(NERF_E_HIP, "nerf_render_rays_fused: copy of rgb failed");
    return NERF_OK;
  }
  if ((rc = nerf_importance_sample(z, w, u, B, n, N, 1e-5f, nullptr, zf, nullptr, nullptr, stream))) return rc;
  if ((rc = nerf_query_fused(arch, packed_fine ? packed_fine : packed_coarse, rays, zf, B, n + N, freq_mode, rawf,
                             nullptr, stream))) return rc;
  return nerf_composite_forward(rawf, zf, rays, B, n + N, 0.0f, nullptr, white_bkgd, rgb, disp, acc, nullptr, nullptr,
                                stream);
}

// ---- RCCL over xGMI: one sum all-reduce of the flat gradient buffer per network step ---------------
// librccl is resolved lazily (dlopen) so that the library loads on hosts without RCCL; in a torch process this
// binds to the same librccl.so.1 torch already loaded.
namespace {
typedef struct { char internal[128]; } rccl_unique_id;
typedef int (*fn_get_id)(rccl_unique_id*);
typedef int (*fn_init_rank)(void**, int, rccl_unique_id, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);
struct Rccl {
  void* h = nullptr; fn_get_id get_id = nullptr; fn_init_rank init_rank = nullptr; fn_allreduce allreduce = nullptr;
  fn_destroy destroy = nullptr; fn_errstr errstr = nullptr;
};
Rccl g_rccl;
int rccl_load() {
  if (g_rccl.h) return NERF_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* nm : names) if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) return fail(NERF_E_RCCL, "RCCL not available: %s", dlerror());
  g_rccl.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
  g_rccl.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
  g_rccl.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
  g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
  g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
  if (!g_rccl.get_id || !g_rccl.init_rank || !g_rccl.allreduce || !g_rccl.destroy)
    return fail(NERF_E_RCCL, "RCCL symbols missing in librccl");
  g_rccl.h = h;
  return NERF_OK;
}
int rccl_fail(const char* what, int code) {
  return fail(NERF_E_RCCL, "%s: %s", what, g_rccl.errstr ? g_rccl.errstr(code) : "rccl error");
}
}  // namespace

extern "C" int nerf_comm_unique_id(char* id_out) {
  NERF_REQUIRE(id_out, NERF_E_NULL, "nerf_comm_unique_id: id_out is NULL");
  int rc = rccl_load();
  if (rc) return rc;
  rccl_unique_id id;
  const int e = g_rccl.get_id(&id);
  if (e) return rccl_fail("ncclGetUniqueId", e);
  memcpy(id_out, id.internal, 128);
  return NERF_OK;
}

extern "C" int nerf_comm_init(void** comm_out, int nranks, int rank, const char* id) {
  NERF_REQUIRE(comm_out && id, NERF_E_NULL, "nerf_comm_init: NULL pointer");
  NERF_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, NERF_E_SHAPE, "nerf_comm_init: bad rank %d of %d", rank, nranks);
  int rc = rccl_load();
  if (rc) return rc;
  rccl_unique_id uid;
  memcpy(uid.internal, id, 128);
  const int e = g_rccl.init_rank(comm_out, nranks, uid, rank);
  return e ? rccl_fail("ncclCommInitRank", e) : NERF_OK;
}

extern "C" int nerf_allreduce_grads(void* comm, float* grads, int64_t count, void* stream) {
  NERF_REQUIRE(comm && grads, NERF_E_NULL, "nerf_allreduce_grads: NULL pointer");
  NERF_REQUIRE(count > 0, NERF_E_SHAPE, "nerf_allreduce_grads: count must be > 0");
  int rc = rccl_load();
  if (rc) return rc;
  const int e = g_rccl.allreduce(grads, grads, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, as_stream(stream));
  return e ? rccl_fail("ncclAllReduce", e) : NERF_OK;
}

extern "C" int nerf_comm_destroy(void* comm) {
  if (!comm) return NERF_OK;
  int rc = rccl_load();
  if (rc) return rc;
  const int e = g_rccl.destroy(comm);
  return e ? rccl_fail("ncclCommDestroy", e) : NERF_OK;
}
