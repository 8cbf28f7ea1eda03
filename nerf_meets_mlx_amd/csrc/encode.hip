// Standalone encoders (a9, a10, a22, a23 / K3, K10, K11).  The NeRF render path uses the
// positional encoding fused into the MLP kernel (mlp.hip); these entry points exist for the
// reference's `Encoding` classes and for parity tests, and are HBM/L2-gather bound.
#include "common.h"
#include "hash_common.h"
#include <type_traits>

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

__global__ void sh_kernel(const float* __restrict__ d, int64_t M, int deg, float* __restrict__ out) {
  const int C = (deg + 1) * (deg + 1);
  for (int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x)
    sh_eval(d[3 * m], d[3 * m + 1], d[3 * m + 2], deg, out + m * C);
}

// ---- multires hash grid  encoding/multi_hash.py:61-136 (intended semantics) ------------------
template <int F, int LG>
__global__ void __launch_bounds__(256) hashgrid_fwd_kernel(PointSrc ps, int64_t M, const float* __restrict__ tables,
                                                           int L, uint32_t T, ResTab rt, float* __restrict__ out,
                                                           int64_t out_stride) {
  const int l0 = blockIdx.y * LG;
  const uint32_t mask = T - 1;
  for (int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    float px, py, pz;
    point_of(ps, m, px, py, pz);
    float vals[LG * F];
#pragma unroll
    for (int li = 0; li < LG; ++li) {
      const int l = l0 + li;
      if (l >= L) break;
      const Corners c = corners_of(px, py, pz, rt.res[l], mask);
      const float* tb = tables + (size_t)l * T * F;
      const FeatVec<F> fv = hash_level<F>(tb, c);
#pragma unroll
      for (int f = 0; f < F; ++f) vals[li * F + f] = fv.v[f];
    }
#pragma unroll
    for (int i = 0; i < LG * F; ++i)
      if (l0 + i / F < L) out[m * out_stride + l0 * F + i] = vals[i];
  }
}

// the rest of a configs[4] input row: SH (degree `deg`) of the ray's view direction behind the hash features, and the
// sample position for the table-gradient pass
__global__ void __launch_bounds__(256) ngp_dir_rows_kernel(const float* __restrict__ rays, const float* __restrict__ z,
                                                           int n, int64_t M, int deg, float* __restrict__ x_out,
                                                           int64_t out_stride, int col0, float* __restrict__ pts_out,
                                                           float pos_scale, float pos_offset) {
  const PointSrc ps{nullptr, rays, z, n, pos_scale, pos_offset};
  for (int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    const float* rr = rays + (int64_t)((uint64_t)m / (unsigned)n) * NERF_RAY_STRIDE;
    float o[25];
    sh_eval(rr[8], rr[9], rr[10], deg, o);
    const int C = (deg + 1) * (deg + 1);
    for (int c = 0; c < C; ++c) x_out[m * out_stride + col0 + c] = o[c];
    if (pts_out) {
      float px, py, pz;
      point_of(ps, m, px, py, pz);
      pts_out[3 * m] = px; pts_out[3 * m + 1] = py; pts_out[3 * m + 2] = pz;
    }
  }
}

// Table-gradient scatter.  The scatter is bound by atomic REQUESTS at the memory side (one per cache line touched by a
// wave instruction, ~20 G/s), not by bytes, so the lane mapping is chosen to make the lanes of one atomic instruction
// fall into as few lines as possible: 2 F consecutive lanes serve one sample -- lane (f, dx) adds feature f into the
// corner with x = floor (dx = 0) or x = ceil (dx = 1).  The hash is c_x ^ (c_y P1) ^ (c_z P2): for fixed (c_y, c_z) the
// two x-neighbours differ only in the low bits of the index (same 128-byte line unless a carry crosses bit 3 for
// F = 2: 15 of 16 cases), so one instruction touches ~16 lines for 64 lanes where the per-(sample, feature) mapping
// touched 32 and the per-sample mapping 64.  Each lane issues the 4 (y, z) corner combinations of its x plane.
// FIXED (deterministic mode): every addend is converted to a 64-bit fixed-point number (2^-NERF_HASH_FIX_SHIFT units) and
// added with an INTEGER atomic: integer addition is associative, so the accumulated table gradient does not depend on
// the order in which the memory side serves the requests -- bit-reproducible run to run, unlike float atomics -- and it
// is also the exactly rounded sum of the quantised addends (|addend| < 2^11, resolution 2^-52: finer than float32 for
// every addend above 3e-9, and Adam's eps = 1e-8 hides what is below; out-of-range and non-finite addends: nerf_to_fixed,
// hash_common.h).  nerf_adam_step_ex reads the accumulators.
template <int F, int LG, bool FIXED>
__global__ void __launch_bounds__(256) hashgrid_bwd_kernel(PointSrc ps, int64_t M,
                                                           void* __restrict__ d_tables_v, const float* __restrict__ d_out,
                                                           int L, uint32_t T, ResTab rt, int level_lo, int level_hi) {
  const int l0 = level_lo + blockIdx.y * LG;
  const uint32_t mask = T - 1;
  const int64_t total = M * (2 * F);
  float* d_tables = static_cast<float*>(d_tables_v);
  unsigned long long* d_fixed = static_cast<unsigned long long*>(d_tables_v);
  auto add = [&](size_t idx, float v) {
#ifdef NERF_SCATTER_PAIR_PROBE
    if (F == 2 && !FIXED) { atomicAdd(reinterpret_cast<unsigned long long*>(d_tables + idx), (unsigned long long)__float_as_uint(v)); return; }
#endif
    if (FIXED) atomicAdd(d_fixed + idx, (unsigned long long)nerf_to_fixed(v));
    else atomicAdd(d_tables + idx, v);
  };
#ifdef NERF_SCATTER_PAIR_PROBE
  // TIMING-ONLY build (tools/ab_one.sh ... -DNERF_SCATTER_PAIR_PROBE; wrong gradients): ONE 8-byte atomic per (sample, corner) instead
  // of one per (sample, corner, feature) -- what packing the F = 2 features of a table entry into a single request would buy.
  // Float mode: a 64-bit integer add on the entry's (f0, f1) pair; fixed mode: feature 0's accumulator only.  Half the lanes idle.
  const bool pair_probe = F == 2;
#else
  const bool pair_probe = false;
#endif
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = t / (2 * F);
    const int q = (int)(t - m * (2 * F)), f = q % F, dx = q / F;
    if (pair_probe && f != 0) continue;
    float px, py, pz;
    point_of(ps, m, px, py, pz);
#pragma unroll
    for (int li = 0; li < LG; ++li) {
      const int l = l0 + li;
      if (l >= level_hi) break;
      const float r = rt.res[l];
      // same roundings as corners_of (hash_common.h): xs = p * r, floor / ceil, offset = xs - floor
      const float xs = px * r, ys = py * r, zs = pz * r;
      const float fx = floorf(xs), fy = floorf(ys), fz = floorf(zs);
      const float ox = xs - fx, oy = ys - fy, oz = zs - fz;
      const uint32_t cx = (uint32_t)(int32_t)(dx ? ceilf(xs) : fx);
      const uint32_t yf = (uint32_t)(int32_t)fy * 2654435761u, yc = (uint32_t)(int32_t)ceilf(ys) * 2654435761u;
      const uint32_t zf = (uint32_t)(int32_t)fz * 805459861u, zc = (uint32_t)(int32_t)ceilf(zs) * 805459861u;
      const size_t tb = (size_t)l * T * F + f;
      const float g = d_out[(m * L + l) * F + f];
      // products in the order of the per-corner form (g * wz * wy * wx), so every addend is bit-identical to it
      const float wx = dx ? ox : 1 - ox;
      add(tb + (size_t)((cx ^ yc ^ zc) & mask) * F, g * oz * oy * wx);
      add(tb + (size_t)((cx ^ yf ^ zc) & mask) * F, g * oz * (1 - oy) * wx);
      add(tb + (size_t)((cx ^ yc ^ zf) & mask) * F, g * (1 - oz) * oy * wx);
      add(tb + (size_t)((cx ^ yf ^ zf) & mask) * F, g * (1 - oz) * (1 - oy) * wx);
    }
  }
}

// Coarse levels: write-combining in LDS.  A workgroup iteration covers 64 consecutive samples (one ray at n = 64); at a
// coarse level they fall into a handful of cells (3 samples per cell at resolution 16, ~1 at 58), so their 64 x 8 x F
// addends hit few distinct table entries, and every workgroup of the launch hammers the same few thousand entries of
// that level (measured: level 0 runs at 28 G atomics/s against 70 G for the fine levels).  Here the addends are first
// accumulated in a small open-addressing table in LDS (keys: entry index; LDS atomics), then each distinct (entry,
// feature) is added to the global table ONCE.  The thread whose compare-and-swap claims an empty slot appends the slot
// to a list, and the flush walks only that list and leaves the slots it visits empty again: the cost per iteration is
// proportional to the number of DISTINCT entries, not to the table size.  A key that finds no slot within HC_PROBES steps
// goes to the global table directly (adds commute).  In FIXED mode the LDS accumulators are the same int64 fixed-point
// numbers as the global ones: integer addition is associative, so the result is bit-identical to the direct kernel's.
int g_hash_combine_max_res = 64;    // A/B knob (nerf_set_option "hash_combine_max_res"): levels with N_l <= this go through LDS; 0 = off
constexpr int HC_CAP = 1024;                       // slots; at most 64 samples x 8 corners = 512 distinct keys per iteration
constexpr int HC_PROBES = 32;
constexpr uint32_t HC_EMPTY = 0xFFFFFFFFu;
template <int F, bool FIXED>
__global__ void __launch_bounds__(256) hashgrid_bwd_combine_kernel(PointSrc ps, int64_t M, void* __restrict__ d_tables_v,
                                                                   const float* __restrict__ d_out, int L, uint32_t T,
                                                                   ResTab rt, int level_lo) {
  typedef typename std::conditional<FIXED, unsigned long long, float>::type AccT;
  __shared__ uint32_t keys[HC_CAP];
  __shared__ uint32_t list[HC_CAP];
  __shared__ AccT vals[HC_CAP * F];
  __shared__ uint32_t count[2];
  const int l = level_lo + blockIdx.y;
  const uint32_t mask = T - 1;
  const float r = rt.res[l];
  float* d_tables = static_cast<float*>(d_tables_v);
  unsigned long long* d_fixed = static_cast<unsigned long long*>(d_tables_v);
  const size_t tb = (size_t)l * T * F;
  const int tid = threadIdx.x;
  constexpr int SPW = 256 / (2 * F);               // samples per workgroup iteration (64 for F = 2)
  const int64_t nchunks = (M + SPW - 1) / SPW;
  for (int i = tid; i < HC_CAP; i += 256) {
    keys[i] = HC_EMPTY;
#pragma unroll
    for (int f = 0; f < F; ++f) vals[i * F + f] = (AccT)0;
  }
  if (tid < 2) count[tid] = 0;
  __syncthreads();
  int par = 0;
  for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x, par ^= 1) {
    const int64_t m = chunk * SPW + tid / (2 * F);
    if (m < M) {
      const int q = tid % (2 * F), f = q % F, dx = q / F;
      float px, py, pz;
      point_of(ps, m, px, py, pz);
      const float xs = px * r, ys = py * r, zs = pz * r;
      const float fx = floorf(xs), fy = floorf(ys), fz = floorf(zs);
      const float ox = xs - fx, oy = ys - fy, oz = zs - fz;
      const uint32_t cx = (uint32_t)(int32_t)(dx ? ceilf(xs) : fx);
      const uint32_t yf = (uint32_t)(int32_t)fy * 2654435761u, yc = (uint32_t)(int32_t)ceilf(ys) * 2654435761u;
      const uint32_t zf = (uint32_t)(int32_t)fz * 805459861u, zc = (uint32_t)(int32_t)ceilf(zs) * 805459861u;
      const float g = d_out[(m * L + l) * F + f];
      const float wx = dx ? ox : 1 - ox;
      const uint32_t idx[4] = {(cx ^ yc ^ zc) & mask, (cx ^ yf ^ zc) & mask, (cx ^ yc ^ zf) & mask, (cx ^ yf ^ zf) & mask};
      const float val[4] = {g * oz * oy * wx, g * oz * (1 - oy) * wx, g * (1 - oz) * oy * wx, g * (1 - oz) * (1 - oy) * wx};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        uint32_t slot = (idx[c] * 2654435761u) >> 22;              // top 10 bits: HC_CAP = 1024
        bool found = false;
        for (int probe = 0; probe < HC_PROBES; ++probe) {
          const uint32_t prev = atomicCAS(&keys[slot], HC_EMPTY, idx[c]);
          if (prev == HC_EMPTY) { list[atomicAdd(&count[par], 1u)] = slot; found = true; break; }     // this thread claimed the slot
          if (prev == idx[c]) { found = true; break; }
          slot = (slot + 1) & (HC_CAP - 1);
        }
        if (FIXED) {
          const unsigned long long q64 = (unsigned long long)nerf_to_fixed(val[c]);
          if (found) atomicAdd(reinterpret_cast<unsigned long long*>(&vals[slot * F + f]), q64);
          else atomicAdd(d_fixed + tb + (size_t)idx[c] * F + f, q64);
        } else {
          if (found) atomicAdd(reinterpret_cast<float*>(&vals[slot * F + f]), val[c]);
          else atomicAdd(d_tables + tb + (size_t)idx[c] * F + f, val[c]);
        }
      }
    }
    __syncthreads();
    const int n = (int)count[par] * F;
    if (tid == 0) count[par ^ 1] = 0;                               // last read before the previous barrier
    for (int i = tid; i < n; i += 256) {                            // F consecutive lanes = the F features of one entry
      const uint32_t slot = list[i / F];
      const int f = i % F;
      const uint32_t k = keys[slot];
      const AccT v = vals[slot * F + f];
      vals[slot * F + f] = (AccT)0;
      if (f == F - 1) keys[slot] = HC_EMPTY;                        // the F lanes of a slot sit in one wave: all have read k
      if (FIXED) atomicAdd(d_fixed + tb + (size_t)k * F + f, (unsigned long long)v);
      else atomicAdd(d_tables + tb + (size_t)k * F + f, (float)v);
    }
    __syncthreads();
  }
}

template <bool BWD>
static int launch_hashgrid(const float* x, int64_t M, const float* tables, float* d_tables, const float* d_out, int L,
                           int log2_T, int F, const int* res, float* out, void* stream, const char* who,
                           const float* rays = nullptr, const float* z = nullptr, int n = 1, int64_t out_stride = 0,
                           float pos_scale = 1.0f, float pos_offset = 0.0f, int level_lo = 0, int level_hi = -1,
                           bool fixed = false) {
  NERF_REQUIRE((x || (rays && z)) && res, NERF_E_NULL, "%s: NULL pointer", who);
  NERF_REQUIRE(L >= 1 && L <= 32 && log2_T >= 1 && log2_T <= 30, NERF_E_SHAPE, "%s: need 1<=L<=32, 1<=log2_T<=30", who);
  NERF_REQUIRE(F == 1 || F == 2 || F == 4 || F == 8, NERF_E_UNSUPPORTED, "%s: F must be 1, 2, 4 or 8", who);
  if (M <= 0) return NERF_OK;
  ResTab rt;
  for (int l = 0; l < L; ++l) rt.res[l] = (float)res[l];
  const uint32_t T = 1u << log2_T;
  constexpr int LG = 4;                                   // levels per thread: LG x F contiguous floats per sample
  if (level_hi < 0) level_hi = L;
  NERF_REQUIRE(0 <= level_lo && level_lo <= level_hi && level_hi <= L, NERF_E_SHAPE, "%s: need 0 <= level_lo <= level_hi <= L", who);
  if (level_lo == level_hi) return NERF_OK;
  if (BWD && (F == 2 || F == 4) && g_hash_combine_max_res > 0) {
    // leading levels of the range whose resolution is small enough for LDS write-combining to pay (see the kernel)
    int nc = 0;
    while (level_lo + nc < level_hi && res[level_lo + nc] <= g_hash_combine_max_res) ++nc;
    if (nc > 0) {
      const dim3 gc(grid_for((M + 63) / 64, 1, 256 * 8), (unsigned)nc), bc(256);
      auto stc = as_stream(stream);
      const PointSrc psc{x, rays, z, n, pos_scale, pos_offset};
#define HC(FF) do { if (fixed) hipLaunchKernelGGL((hashgrid_bwd_combine_kernel<FF, true>), gc, bc, 0, stc, psc, M, (void*)d_tables, d_out, L, T, rt, level_lo); \
                    else hipLaunchKernelGGL((hashgrid_bwd_combine_kernel<FF, false>), gc, bc, 0, stc, psc, M, (void*)d_tables, d_out, L, T, rt, level_lo); } while (0)
      if (F == 2) HC(2); else HC(4);
#undef HC
      const int rc = check_launch(who);
      if (rc) return rc;
      level_lo += nc;
      if (level_lo == level_hi) return NERF_OK;
    }
  }
  const int nlev = BWD ? level_hi - level_lo : L;
  const dim3 g(grid_for(BWD ? M * F * 2 : M, 256, 256 * 32), (unsigned)((nlev + LG - 1) / LG)), b(256);
  auto st = as_stream(stream);
#define HG(FF) do { if (BWD && fixed) hipLaunchKernelGGL((hashgrid_bwd_kernel<FF, LG, true>), g, b, 0, st, PointSrc{x, rays, z, n, pos_scale, pos_offset}, M, (void*)d_tables, d_out, L, T, rt, level_lo, level_hi); \
                    else if (BWD) hipLaunchKernelGGL((hashgrid_bwd_kernel<FF, LG, false>), g, b, 0, st, PointSrc{x, rays, z, n, pos_scale, pos_offset}, M, (void*)d_tables, d_out, L, T, rt, level_lo, level_hi); \
                    else hipLaunchKernelGGL((hashgrid_fwd_kernel<FF, LG>), g, b, 0, st, PointSrc{x, rays, z, n, pos_scale, pos_offset}, M, tables, L, T, rt, out, \
                                            out_stride > 0 ? out_stride : (int64_t)L * FF); } while (0)
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

extern "C" int nerf_ngp_encode(const float* rays, const float* z, int64_t B, int n, const float* tables, int L,
                               int log2_T, int F, const int* resolutions_host, int sh_degree, float pos_scale,
                               float pos_offset, float* x_out, float* pts_out, void* stream) {
  if (B <= 0 || n <= 0) return NERF_OK;
  NERF_REQUIRE(rays && z && tables && x_out, NERF_E_NULL, "nerf_ngp_encode: NULL pointer");
  NERF_REQUIRE(sh_degree >= 0 && sh_degree <= 4, NERF_E_SHAPE, "nerf_ngp_encode: sh_degree=%d must be in range [0, 4]", sh_degree);
  const int64_t M = B * n;
  const int64_t stride = (int64_t)L * F + (sh_degree + 1) * (sh_degree + 1);
  const int rc = launch_hashgrid<false>(nullptr, M, tables, nullptr, nullptr, L, log2_T, F, resolutions_host, x_out, stream,
                                        "nerf_ngp_encode", rays, z, n, stride, pos_scale, pos_offset);
  if (rc) return rc;
  hipLaunchKernelGGL(ngp_dir_rows_kernel, dim3(grid_for(M, 256)), dim3(256), 0, as_stream(stream), rays, z, n, M, sh_degree,
                     x_out, stride, L * F, pts_out, pos_scale, pos_offset);
  return check_launch("nerf_ngp_encode");
}

extern "C" int nerf_hashgrid_backward_ex(const float* x, int64_t M, const float* d_out, int L, int log2_T, int F,
                                         const int* resolutions_host, int level_lo, int level_hi, int fixed_point,
                                         void* d_tables, void* stream) {
  NERF_REQUIRE(d_out && d_tables, NERF_E_NULL, "nerf_hashgrid_backward_ex: d_out/d_tables is NULL");
  NERF_REQUIRE(fixed_point == 0 || fixed_point == 1, NERF_E_UNSUPPORTED, "nerf_hashgrid_backward_ex: fixed_point must be 0 or 1");
  return launch_hashgrid<true>(x, M, nullptr, static_cast<float*>(d_tables), d_out, L, log2_T, F, resolutions_host, nullptr,
                               stream, "nerf_hashgrid_backward_ex", nullptr, nullptr, 1, 0, 1.0f, 0.0f, level_lo, level_hi,
                               fixed_point != 0);
}

extern "C" int nerf_hashgrid_backward_rays_ex(const float* rays, const float* z, int64_t B, int n, const float* d_out, int L,
                                              int log2_T, int F, const int* resolutions_host, float pos_scale,
                                              float pos_offset, int level_lo, int level_hi, int fixed_point,
                                              void* d_tables, void* stream) {
  if (B <= 0 || n <= 0) return NERF_OK;
  NERF_REQUIRE(rays && z && d_out && d_tables, NERF_E_NULL, "nerf_hashgrid_backward_rays_ex: NULL pointer");
  NERF_REQUIRE(fixed_point == 0 || fixed_point == 1, NERF_E_UNSUPPORTED, "nerf_hashgrid_backward_rays_ex: fixed_point must be 0 or 1");
  return launch_hashgrid<true>(nullptr, B * n, nullptr, static_cast<float*>(d_tables), d_out, L, log2_T, F, resolutions_host,
                               nullptr, stream, "nerf_hashgrid_backward_rays_ex", rays, z, n, 0, pos_scale, pos_offset,
                               level_lo, level_hi, fixed_point != 0);
}

extern "C" int nerf_hashgrid_backward_rays(const float* rays, const float* z, int64_t B, int n, const float* d_out, int L,
                                           int log2_T, int F, const int* resolutions_host, float pos_scale,
                                           float pos_offset, float* d_tables, void* stream) {
  if (B <= 0 || n <= 0) return NERF_OK;
  NERF_REQUIRE(rays && z && d_out && d_tables, NERF_E_NULL, "nerf_hashgrid_backward_rays: NULL pointer");
  return launch_hashgrid<true>(nullptr, B * n, nullptr, d_tables, d_out, L, log2_T, F, resolutions_host, nullptr, stream,
                               "nerf_hashgrid_backward_rays", rays, z, n, 0, pos_scale, pos_offset);
}
