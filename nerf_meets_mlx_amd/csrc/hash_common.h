// Device helpers shared by encode.hip (stand-alone encoders) and mlp.hip (the fused configs[4] forward):
// real spherical harmonics, hash-grid corner indices / weights, table gathers, sample positions.
#pragma once
#include "common.h"

namespace nerf {

// real SH basis  encoding/spherical_harmonics.py:62-93 (same operation order)
__device__ __forceinline__ void sh_eval(float x, float y, float z, int deg, float* __restrict__ o) {
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
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
struct ResTab { float res[32]; };

// fixed-point unit of the deterministic table-gradient accumulators (int64): 2^-52
#define NERF_HASH_FIX_SHIFT 52
#define NERF_HASH_FIX_SCALE 4503599627370496.0
// Representable range and failure behaviour (include/nerf_hip.h, nerf_hashgrid_backward_rays_ex): a finite addend with
// |v| > 256 saturates to +-(2^60 + 2^59) units -- half a window width OUTSIDE the window, so that the ordinary addends of the same
// entry (up to +-128 in sum; a coarse-level entry collects hundreds) cannot pull the sum back inside it (round 5 saturated to
// +-2^60 exactly: -2^60 plus one small positive addend read as a finite -256) --, a NaN / Inf addend adds 2^61 units;
// nerf_adam_step_ex turns every
// accumulator outside (-2^60, 2^60) -- a saturated addend, a poisoned one, or a per-entry sum that large, also after the
// cross-rank all-reduce -- back into a NaN gradient, so that a diverged run surfaces as NaN parameters exactly as it does
// with float atomics instead of continuing on wrapped integers.  (k poisoned addends on one entry sum to k 2^61 mod 2^64,
// which is inside the window only for k = 0 mod 8: a diverged batch poisons thousands of entries, 7 of 8 of them stay NaN.)
#define NERF_HASH_FIX_LIMIT 256.0f
#define NERF_HASH_FIX_SATURATED ((1ll << 60) + (1ll << 59))
__device__ __forceinline__ long long nerf_to_fixed(float v) {
  if (!(__builtin_fabsf(v) <= NERF_HASH_FIX_LIMIT))
    return (v != v || __builtin_isinf(v)) ? (1ll << 61) : (v > 0.0f ? NERF_HASH_FIX_SATURATED : -NERF_HASH_FIX_SATURATED);
  return __double2ll_rn((double)v * NERF_HASH_FIX_SCALE);
}
__device__ __forceinline__ bool nerf_fixed_is_poisoned(long long a) {
  // a outside the OPEN interval (-2^60, 2^60), symmetric; saturated addends sit at +-1.5 x 2^60 (above)
  return a <= -(1ll << 60) || a >= (1ll << 60);
}

__device__ __forceinline__ uint32_t hash3(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t mask) {
  return ((cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u)) & mask;     // uint32 wrap-around, mod T = & (T-1)
}

// Forward: one thread per sample, blockIdx.y = a group of LG consecutive levels: at any time a workgroup's 256
// neighbouring samples gather from ONE level table (the coarse ones are a few KiB and stay in L1/L2; with one thread
// per (sample, level) every lane of a wave hit a different 4 MiB table), a corner's F features come in ONE load of
// 4 F bytes, and a thread's LG x F outputs are contiguous in the row (32 B for LG = 4, F = 2).
// Backward: one thread per (sample, feature): the F lanes of a sample add into adjacent words of the same table entry,
// so one atomic wave-instruction touches 64 / F cache lines instead of 64 (the scatter is bound by atomic requests
// at the memory side, not by bytes).
template <int F> struct FeatVec { float v[F]; };
template <int F>
__device__ __forceinline__ FeatVec<F> load_entry(const float* __restrict__ tb, uint32_t i) {
  FeatVec<F> r;
  if (F == 2) { const float2 t = reinterpret_cast<const float2*>(tb)[i]; r.v[0] = t.x; r.v[1] = t.y; }
  else if (F == 4) { const float4 t = reinterpret_cast<const float4*>(tb)[i]; r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
  else {
#pragma unroll
    for (int f = 0; f < F; ++f) r.v[f] = tb[(size_t)i * F + f];
  }
  return r;
}

// the same entry from the fp16 shadow image of an F = 2 table (one 4-byte load per corner instead of 8 bytes)
__device__ __forceinline__ FeatVec<2> load_entry_h(const uint32_t* __restrict__ tb, uint32_t i) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  const h2_t t = __builtin_bit_cast(h2_t, tb[i]);
  FeatVec<2> r;
  r.v[0] = (float)t[0]; r.v[1] = (float)t[1];
  return r;
}

struct Corners { uint32_t i[8]; float ox, oy, oz; };
// reference corner numbering: 0=(c,c,c) 1=(c,f,c) 2=(f,f,c) 3=(f,c,c) 4=(c,c,f) 5=(c,f,f) 6=(f,f,f) 7=(f,c,f)
__device__ __forceinline__ Corners corners_of(float px, float py, float pz, float r, uint32_t mask) {
  const float p[3] = {px, py, pz};
  float off[3]; uint32_t cf[3], cc[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float xs = p[a] * r;
    const float fl = floorf(xs);
    off[a] = xs - fl;
    cf[a] = (uint32_t)(int32_t)fl; cc[a] = (uint32_t)(int32_t)ceilf(xs);
  }
  Corners c;
  c.i[0] = hash3(cc[0], cc[1], cc[2], mask); c.i[1] = hash3(cc[0], cf[1], cc[2], mask);
  c.i[2] = hash3(cf[0], cf[1], cc[2], mask); c.i[3] = hash3(cf[0], cc[1], cc[2], mask);
  c.i[4] = hash3(cc[0], cc[1], cf[2], mask); c.i[5] = hash3(cc[0], cf[1], cf[2], mask);
  c.i[6] = hash3(cf[0], cf[1], cf[2], mask); c.i[7] = hash3(cf[0], cc[1], cf[2], mask);
  c.ox = off[0]; c.oy = off[1]; c.oz = off[2];
  return c;
}

// sample position: row m of x [M,3], or o + z d of ray m / n (rendering/render.py:142: one multiply, one add)
struct PointSrc { const float* x; const float* rays; const float* z; int n; float scale, offset; };
__device__ __forceinline__ void point_of(const PointSrc& ps, int64_t m, float& px, float& py, float& pz) {
  if (ps.rays) {
    const float* rr = ps.rays + (int64_t)((uint64_t)m / (unsigned)ps.n) * NERF_RAY_STRIDE;
    const float zv = ps.z[m];
    // world position, then the affine map of the scene box onto the grid's unit cube (scale 1, offset 0 = none)
    px = (rr[0] + zv * rr[3]) * ps.scale + ps.offset;
    py = (rr[1] + zv * rr[4]) * ps.scale + ps.offset;
    pz = (rr[2] + zv * rr[5]) * ps.scale + ps.offset;
  } else {
    px = ps.x[3 * m]; py = ps.x[3 * m + 1]; pz = ps.x[3 * m + 2];
  }
}

template <int F>
__device__ __forceinline__ FeatVec<F> trilerp(const FeatVec<F> (&e)[8], const Corners& c);

template <int F>
__device__ __forceinline__ FeatVec<F> hash_level(const float* __restrict__ tb, const Corners& c) {
  FeatVec<F> e[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = load_entry<F>(tb, c.i[k]);
  return trilerp<F>(e, c);
}
// F = 2, entries from the fp16 shadow image; interpolation in float32 as above
__device__ __forceinline__ FeatVec<2> hash_level_h(const uint32_t* __restrict__ tb, const Corners& c) {
  FeatVec<2> e[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = load_entry_h(tb, c.i[k]);
  return trilerp<2>(e, c);
}

template <int F>
__device__ __forceinline__ FeatVec<F> trilerp(const FeatVec<F> (&e)[8], const Corners& c) {
  const FeatVec<F>&e0 = e[0], &e1 = e[1], &e2 = e[2], &e3 = e[3], &e4 = e[4], &e5 = e[5], &e6 = e[6], &e7 = e[7];
  const float ox = c.ox, oy = c.oy, oz = c.oz;
  FeatVec<F> r;
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const float h03 = e0.v[f] * ox + e3.v[f] * (1 - ox);
    const float h12 = e1.v[f] * ox + e2.v[f] * (1 - ox);
    const float h56 = e5.v[f] * ox + e6.v[f] * (1 - ox);
    const float h47 = e4.v[f] * ox + e7.v[f] * (1 - ox);
    const float h0312 = h03 * oy + h12 * (1 - oy);
    const float h4756 = h47 * oy + h56 * (1 - oy);
    r.v[f] = h0312 * oz + h4756 * (1 - oz);
  }
  return r;
}

}  // namespace nerf
