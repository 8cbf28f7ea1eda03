"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Plain torch-CPU / numpy restatement of the arithmetic of the reference hot path
(piljoong-jeong/nerf_meets_mlx, `mlx_nerf/{sampling,encoding,models,rendering,ops}`),
one function per row of SURVEY.md section 8(a).  Every function cites the reference
file:line it follows (paths relative to the reference checkout).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module, and only as the checker / reported baseline.  The product package
`nerf_meets_mlx_amd` never imports it.

Pinning status (see DESIGN.md "Oracle"):
  * `get_rays` and `sample_from_inverse_cdf` are pinned against the reference's own
    functions executed in the build container (tests/golden/make_golden.py ->
    tests/golden/ref_*.npz).
  * everything whose arithmetic lives inside `mlx==0.7.0` (nn.Linear, Adam, cumsum,
    sort, linspace) is a restatement of the published MLX semantics and of the
    reference call sites: PARITY UNPINNED at the MLX boundary (mlx is not installable
    here; the reference has no tests / golden vectors of its own).

All functions take/return torch tensors, CPU by default.  `dtype` defaults to float32 (the
reference's dtype); float64 gives the "exact" value used to bound float tolerances.
The functions are device-agnostic (they create tensors on the device of their inputs), so
the long PSNR-parity runs (tools/psnr_parity.py) can run this restatement in fp32 on the
torch-ROCm device -- still as the checker, never as the product; the CPU is what
`cpu_baseline` times and what the golden fixtures were checked on.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# a1 / a2  rays
# --------------------------------------------------------------------------------------

def get_rays(H: int, W: int, K, c2w, dtype=torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
    """rendering/ray.py:7-35.  i = column, j = row ("xy" meshgrid), pixel centres at
    integer coordinates (no +0.5); dirs = [(i-cx)/fx, -(j-cy)/fy, -1];
    rays_d = sum_k dirs[k] * c2w[:3, k]; rays_o = c2w[:3, 3] broadcast."""
    K = torch.as_tensor(np.asarray(K), dtype=dtype)
    c2w = torch.as_tensor(np.asarray(c2w), dtype=dtype)
    j, i = torch.meshgrid(torch.arange(H, dtype=dtype), torch.arange(W, dtype=dtype), indexing="ij")
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], dim=-1)  # [H,W,3]
    rays_d = (dirs[..., None, :] * c2w[:3, :3]).sum(-1)
    rays_o = c2w[:3, 3].expand(rays_d.shape).clone()
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """rendering/ray.py:39-70 (NeRF appendix C eq. 25/26)."""
    t_n = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t_n[..., None] * rays_d
    ox, oy, oz = rays_o[..., 0], rays_o[..., 1], rays_o[..., 2]
    dx, dy, dz = rays_d[..., 0], rays_d[..., 1], rays_d[..., 2]
    o0 = (-focal / (0.5 * W)) * (ox / oz)
    o1 = (-focal / (0.5 * H)) * (oy / oz)
    o2 = 1.0 + 2.0 * near / oz
    d0 = (-focal / (0.5 * W)) * (dx / dz - ox / oz)
    d1 = (-focal / (0.5 * H)) * (dy / dz - oy / oz)
    d2 = -2.0 * near * (1.0 / oz)
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


def select_coords(flat_idx: torch.Tensor, W: int) -> torch.Tensor:
    """entrypoints/__test_nerf.py:213-230: coords = row-major (row, col) list of H*W;
    coords[choice] == (idx // W, idx % W).  Integer, bit-exact."""
    flat_idx = flat_idx.to(torch.int64)
    return torch.stack([flat_idx // W, flat_idx % W], dim=-1)


def pack_rays(rays_o, rays_d, near: float, far: float):
    """entrypoints/__test_nerf.py:60-82, rendering/render.py:296-328:
    viewdirs = d/|d|; rays_linear = [o(3), d(3), near, far, viewdirs(3)]  -> [B, 11]."""
    viewdirs = rays_d / torch.linalg.norm(rays_d, dim=-1, keepdim=True)
    nr = near * torch.ones_like(rays_d[..., :1])
    fr = far * torch.ones_like(rays_d[..., :1])
    return torch.cat([rays_o, rays_d, nr, fr, viewdirs], dim=-1)


def decompose_ray_batch(rays):
    """rendering/render.py:98-110 (near, far come out as [B,1])."""
    return rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8], rays[:, -3:]


# --------------------------------------------------------------------------------------
# a5 / a6 / a7  depth sampling
# --------------------------------------------------------------------------------------

def mlx_linspace(start: float, stop: float, num: int, dtype=torch.float32, device=None):
    """mx.linspace as published for mlx 0.7.0 (mlx/ops.cpp, not in /root/reference):
    arange(0,num,float32) * float32((stop-start)/(num-1)) + start.  Differs from
    torch.linspace (symmetric formula) by <= 1 ulp; UNPINNED assumption."""
    step = np.float32((stop - start) / (num - 1)) if dtype == torch.float32 else (stop - start) / (num - 1)
    seq = torch.arange(0, num, dtype=dtype, device=device)
    return seq * torch.tensor(step, dtype=dtype, device=device) + torch.tensor(start, dtype=dtype, device=device)


def sample_z_uniform(near, far, n: int):
    """sampling/uniform.py:7-18: t = linspace(0,1,n); z = near*(1-t) + far*t."""
    t = mlx_linspace(0.0, 1.0, n, dtype=near.dtype, device=near.device)
    return near * (1.0 - t) + far * t


def sample_z_lindisp(near, far, n: int):
    """sampling/linear_disparity.py:8-19, restated literally (both
    end points evaluate to 1/(x + inf) = 0; this is NOT the standard formula)."""
    t = mlx_linspace(0.0, 1.0, n, dtype=near.dtype, device=near.device)
    return 1.0 / (1.0 / (near * (1.0 - t)) + 1.0 / (far * t))


def add_noise_z(z, strength: float, t_rand: Optional[torch.Tensor] = None):
    """sampling/__init__.py:10-31, intended semantics (the committed concat is a rank
    mismatch): mids; upper=[mids, z_last]; lower=[z_first, mids];
    z = lower + (upper-lower) * (U[0,1) * strength).  `t_rand` is the U[0,1) tensor."""
    if strength <= 0.0:
        return z
    t = t_rand * strength
    mids = 0.5 * (z[..., :-1] + z[..., 1:])
    upper = torch.cat([mids, z[..., -1:]], -1)
    lower = torch.cat([z[..., :1], mids], -1)
    return lower + (upper - lower) * t


# --------------------------------------------------------------------------------------
# a9 / a10  positional encodings
# --------------------------------------------------------------------------------------

def embedder_freqs(n_freqs: int, ref_quirks: bool = True, dtype=torch.float32, device=None):
    """models/embedding.py:46-49: linspace(0, L-1, L) ** 2  (k^2, NOT 2^k: SURVEY Q4).
    ref_quirks=False gives the intended 2 ** linspace(0, L-1, L)."""
    lin = torch.linspace(0.0, float(n_freqs - 1), n_freqs, dtype=dtype)       # on the host: identical values on any device
    return (lin ** 2.0 if ref_quirks else 2.0 ** lin).to(device)


def embedder(x, n_freqs: int, ref_quirks: bool = True):
    """models/embedding.py:30-71: [x, sin(f0 x), cos(f0 x), sin(f1 x), cos(f1 x), ...]
    each block in_dim wide; include_input truthy for 3-d inputs (:79)."""
    outs = [x]
    for f in embedder_freqs(n_freqs, ref_quirks, x.dtype, x.device):
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, dim=-1)


def embed(pos, dirs, L_pos: int = 10, L_dir: int = 4, ref_quirks: bool = True):
    """models/embedding.py:4-21: flatten pos; repeat dirs to every sample; concat."""
    B, n = pos.shape[0], pos.shape[1]
    e_pos = embedder(pos.reshape(-1, pos.shape[-1]), L_pos, ref_quirks)
    if dirs is None:
        return e_pos
    d = dirs[:, None, :].expand(B, n, dirs.shape[-1]).reshape(-1, dirs.shape[-1])
    return torch.cat([e_pos, embedder(d, L_dir, ref_quirks)], dim=-1)


def sinusoidal_freqs(n_freqs: int, min_exp=None, max_exp=None, dtype=torch.float32):
    """encoding/sinusoidal.py:27-28,49-51: 2 ** linspace(min_exp or 0, max_exp or n-1, n)."""
    mn = min_exp if min_exp else 0.0
    mx_ = max_exp if max_exp else float(n_freqs - 1)
    return 2.0 ** mlx_linspace(mn, mx_, n_freqs, dtype=dtype)              # host values; callers move them


def sinusoidal_encoding(x, n_freqs: int, min_exp: Optional[float] = None, max_exp: Optional[float] = None,
                        include_input: bool = False):
    """encoding/sinusoidal.py:13-66: freq = 2 ** linspace(min,max,n); s = x[...,None]*freq
    reshaped dim-major/freq-minor; out = sin(concat[s, s + pi/2]); raw input appended
    at the END.  (`min_exp if min_exp else 0.0`, `max_exp if max_exp else n-1`: :27-28)."""
    freq = sinusoidal_freqs(n_freqs, min_exp, max_exp, x.dtype).to(x.device)
    s = (x[..., None] * freq).reshape(x.shape[0], -1)
    out = torch.sin(torch.cat([s, s + math.pi / 2.0], dim=-1))
    if include_input:
        out = torch.cat([out, x], dim=-1)
    return out


def sh_encoding(d, n_degrees: int):
    """encoding/spherical_harmonics.py:33-94: real SH basis up to degree 4."""
    assert 0 <= n_degrees <= 4
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    o = [torch.full_like(x, 0.28209479177387814)]
    if n_degrees >= 1:
        o += [0.4886025119029199 * y, 0.4886025119029199 * z, 0.4886025119029199 * x]
    if n_degrees >= 2:
        o += [1.0925484305920792 * xy, 1.0925484305920792 * yz,
              0.9461746957575601 * zz - 0.31539156525251999,
              1.0925484305920792 * xz, 0.5462742152960396 * (xx - yy)]
    if n_degrees >= 3:
        o += [0.5900435899266435 * y * (3 * xx - yy), 2.890611442640554 * xy * z,
              0.4570457994644658 * y * (5 * zz - 1), 0.3731763325901154 * z * (5 * zz - 3),
              0.4570457994644658 * x * (5 * zz - 1), 1.445305721320277 * z * (xx - yy),
              0.5900435899266435 * x * (xx - 3 * yy)]
    if n_degrees >= 4:
        o += [2.5033429417967046 * xy * (xx - yy), 1.7701307697799304 * yz * (3 * xx - yy),
              0.9461746957575601 * xy * (7 * zz - 1), 0.6690465435572892 * yz * (7 * zz - 3),
              0.10578554691520431 * (35 * zz * zz - 30 * zz + 3),
              0.6690465435572892 * xz * (7 * zz - 3), 0.47308734787878004 * (xx - yy) * (7 * zz - 1),
              1.7701307697799304 * xz * (xx - 3 * yy),
              0.6258357354491761 * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    return torch.stack(o, dim=-1)


# ---- a22 multires hash grid (intended semantics; the committed class cannot run: Q13-15)

HASH_PRIMES = (1, 2654435761, 805459861)  # encoding/multi_hash.py:66-70


def hashgrid_resolutions(n_levels: int, min_res: int, max_res: int) -> List[int]:
    """encoding/multi_hash.py:35-40: b = exp((ln Nmax - ln Nmin)/(L-1)); N_l = floor(Nmin*b^l).
    Evaluated in float64 on the host, with the last level snapped so that
    floor() cannot land on Nmax-1 through rounding."""
    if n_levels == 1:
        return [int(min_res)]
    b = math.exp((math.log(max_res) - math.log(min_res)) / (n_levels - 1))
    res = [int(math.floor(min_res * (b ** l) + 1e-9)) for l in range(n_levels)]
    return res


def hash_coords(c: torch.Tensor, T: int) -> torch.Tensor:
    """encoding/multi_hash.py:61-77 with uint32 wrap-around:
    ((x*1) ^ (y*2654435761) ^ (z*805459861)) mod T, all products mod 2^32."""
    c = c.to(torch.int64) & 0xFFFFFFFF
    h = torch.zeros_like(c[..., 0])
    for i in range(c.shape[-1]):
        h = h ^ ((c[..., i] * HASH_PRIMES[i]) & 0xFFFFFFFF)
    return h % T


def hashgrid_encoding(x, tables: torch.Tensor, resolutions: Sequence[int]):
    """encoding/multi_hash.py:79-136, intended semantics (the committed class cannot run:
    Q13).  x [B,3]; tables [L,T,F].  x_l = x*N_l; corners from ceil/floor per axis (no
    +0.5, every level hashed); offset = x_l - floor(x_l) weights the CEIL corner; the
    nested lerps are restated literally (:122-131):
      grid_0=(c,c,c) 1=(c,f,c) 2=(f,f,c) 3=(f,c,c) 4=(c,c,f) 5=(c,f,f) 6=(f,f,f) 7=(f,c,f)."""
    L, T, F = tables.shape
    outs = []
    for l in range(L):
        xs = x * float(resolutions[l])
        fl = torch.floor(xs)
        ce = torch.ceil(xs)
        off = xs - fl
        fl_i, ce_i = fl.to(torch.int64), ce.to(torch.int64)

        def corner(cx, cy, cz):
            g = torch.stack([ce_i[:, 0] if cx else fl_i[:, 0], ce_i[:, 1] if cy else fl_i[:, 1],
                             ce_i[:, 2] if cz else fl_i[:, 2]], -1)
            return tables[l][hash_coords(g, T)]
        h0, h1, h2, h3 = corner(1, 1, 1), corner(1, 0, 1), corner(0, 0, 1), corner(0, 1, 1)
        h4, h5, h6, h7 = corner(1, 1, 0), corner(1, 0, 0), corner(0, 0, 0), corner(0, 1, 0)
        ox, oy, oz = off[:, 0:1], off[:, 1:2], off[:, 2:3]
        h03 = h0 * ox + h3 * (1 - ox)
        h12 = h1 * ox + h2 * (1 - ox)
        h56 = h5 * ox + h6 * (1 - ox)
        h47 = h4 * ox + h7 * (1 - ox)
        h0312 = h03 * oy + h12 * (1 - oy)
        h4756 = h47 * oy + h56 * (1 - oy)
        outs.append(h0312 * oz + h4756 * (1 - oz))
    return torch.cat(outs, dim=-1)


# --------------------------------------------------------------------------------------
# a11 / a12  the MLP
# --------------------------------------------------------------------------------------

class NerfArch:
    """Shapes of models/NeRF.py:160-199 for (n_layers=8, width=256, skips=[4])."""

    def __init__(self, channel_input=63, channel_input_views=27, channel_output=4, n_layers=8, width=256,
                 skips=(4,), use_viewdirs=True):
        self.cin, self.cdir, self.cout = channel_input, channel_input_views, channel_output
        self.D, self.W, self.skips, self.use_viewdirs = n_layers, width, tuple(skips), use_viewdirs

    def layer_shapes(self) -> List[Tuple[str, int, int]]:
        """(name, out, in) in flat-buffer order (include/nerf_hip.h "parameter layout")."""
        s = [("pos0", self.W, self.cin)]
        for i in range(self.D - 1):
            s.append((f"pos{i + 1}", self.W, self.W + self.cin if i in self.skips else self.W))
        if self.use_viewdirs:
            s += [("feature", self.W, self.W), ("alpha", 1, self.W),
                  ("dir0", self.W // 2, self.W + self.cdir), ("rgb", 3, self.W // 2)]
        else:
            s += [("output", self.cout, self.W)]
        return s

    def n_params(self) -> int:
        return sum(o * i + o for _, o, i in self.layer_shapes())


def init_params(arch: NerfArch, seed: int = 0, dtype=torch.float32) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
    """mlx.nn.Linear init (mlx 0.7.0, not in /root/reference): weight [out,in] and bias
    ~ U(-1/sqrt(in), 1/sqrt(in)).  Seeded numpy stream so GPU and oracle share weights."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, o, i in arch.layer_shapes():
        k = 1.0 / math.sqrt(i)
        w = rng.uniform(-k, k, size=(o, i)).astype(np.float32)
        b = rng.uniform(-k, k, size=(o,)).astype(np.float32)
        p[name] = (torch.from_numpy(w).to(dtype), torch.from_numpy(b).to(dtype))
    return p


def flatten_params(arch: NerfArch, p) -> torch.Tensor:
    return torch.cat([torch.cat([p[n][0].reshape(-1), p[n][1].reshape(-1)]) for n, _, _ in arch.layer_shapes()])


def unflatten_params(arch: NerfArch, flat: torch.Tensor):
    p, off = {}, 0
    for n, o, i in arch.layer_shapes():
        w = flat[off:off + o * i].reshape(o, i); off += o * i
        b = flat[off:off + o]; off += o
        p[n] = (w, b)
    return p


def _bf16(x):
    return x.to(torch.bfloat16).to(x.dtype)


class _RoundBothBF16(torch.autograd.Function):
    """Value AND incoming gradient rounded to bf16 (what `x.to(bf16).to(f32)` does under torch autograd, written out).
    On a layer input this is the kernel's arithmetic: the forward operand is bf16, and the gradient arriving here --
    dH = W^T dZ_next, which the ReLU mask then turns into this layer's dZ -- is stored as bf16 by the backward chain
    (csrc/mlp.hip: finish_quarter_bwd).  Mask-then-round equals round-then-mask."""

    @staticmethod
    def forward(ctx, x):
        return _bf16(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


class _RoundGradBF16(torch.autograd.Function):
    """Identity whose incoming gradient is rounded to bf16: the kernel's first dZ (d_raw -> bf16 rows of the rgb / alpha
    jobs, csrc/mlp.hip: bwd_tiles)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


def _bf16_value_only(w):
    """bf16-rounded value, straight-through gradient: the weight operand is bf16 but its gradient is accumulated and
    kept in fp32 (mlp_dw_kernel's accumulators), not rounded."""
    return w + (_bf16(w) - w).detach()


def nerf_forward(arch: NerfArch, p, x, emulate_bf16: bool = False, masks: Optional[Dict[str, torch.Tensor]] = None,
                 taps: Optional[Dict[str, torch.Tensor]] = None):
    """models/NeRF.py:201-243.  nn.Linear = x @ W.T + b.  Every pos layer is followed by
    ReLU; after layer idx in skips, h = concat[input_pos, h]; view head: alpha=Linear(h),
    feature=Linear(h) (no activation), h=ReLU(Linear([feature, input_dir])), rgb=Linear(h);
    output [rgb, alpha] raw (no sigmoid / ReLU).
    emulate_bf16: round MFMA operands (weights, layer inputs) to bf16 exactly where the
    HIP kernel does, keep fp32 accumulation, and in the backward round every dZ_l to bf16
    like the kernel's chain does (each layer input is rounded ONCE, so a tensor with two
    consumers -- h7 feeding alpha and feature -- gets the sum of both gradients rounded once,
    like dZ7) -- used to separate "bf16 by design" from bugs.
    masks (tests): {layer: bool [M, width]} replaces the ReLU decision of that layer
    (h = z * mask), so that a gradient comparison is not dominated by units whose
    pre-activation is ~0 and lands on the other side of zero in another implementation.
    taps (tests): dict that receives every layer's post-activation output."""
    r = _RoundBothBF16.apply if emulate_bf16 else (lambda t: t)
    rw = _bf16_value_only if emulate_bf16 else (lambda t: t)
    lin = lambda hr, n: hr @ rw(p[n][0]).T + p[n][1]

    def act(zz, name):
        out = zz * masks[name].to(zz.dtype) if masks is not None and name in masks else torch.relu(zz)
        if taps is not None:
            taps[name] = out
        return out

    if arch.use_viewdirs:
        input_pos, input_dir = x[..., :arch.cin], x[..., arch.cin:]
    else:
        input_pos, input_dir = x, None
    h = input_pos
    for i in range(arch.D):
        h = act(lin(r(h), f"pos{i}"), f"pos{i}")
        if i in arch.skips:
            h = torch.cat([input_pos, h], dim=-1)
    hr = r(h)
    if arch.use_viewdirs:
        alpha = lin(hr, "alpha")
        feature = lin(hr, "feature")
        if taps is not None:
            taps["feature"] = feature
        h = act(lin(r(torch.cat([feature, input_dir], dim=-1)), "dir0"), "dir0")
        rgb = lin(r(h), "rgb")
        out = torch.cat([rgb, alpha], dim=-1)
    else:
        out = lin(hr, "output")
    return _RoundGradBF16.apply(out) if emulate_bf16 else out


def run_model(arch, p, pos, dirs, netchunk: int = 65536, ref_quirks: bool = True, emulate_bf16: bool = False,
              masks=None, taps=None):
    """models/NeRF.py:10-48: assert rank 3; embed all points; forward in `netchunk` slices.
    masks / taps (tests, see nerf_forward) need the whole batch in one slice."""
    assert pos.dim() == 3, f"pos.shape={tuple(pos.shape)} should be [n_rays, n_depth_samples, 3]"
    B, n = pos.shape[:2]
    x = embed(pos, dirs, ref_quirks=ref_quirks)
    if masks is not None or taps is not None:
        assert x.shape[0] <= netchunk, "masks / taps: one netchunk slice only"
    outs = [nerf_forward(arch, p, x[i:i + netchunk], emulate_bf16, masks, taps) for i in range(0, x.shape[0], netchunk)]
    out = torch.cat(outs, 0)
    return out.reshape(B, n, out.shape[-1])


# --------------------------------------------------------------------------------------
# a13  alpha compositing
# --------------------------------------------------------------------------------------

def raw2outputs(raw, z_vals, rays_d, raw_noise_std: float = 0.0, white_bkgd: bool = False,
                noise: Optional[torch.Tensor] = None):
    """rendering/render.py:20-96.  delta_k = z_{k+1}-z_k, last 1e10, times |d|;
    x = delta*sigma; alpha = 1-exp(-relu(x)); T = exp(-exclusive_cumsum(x)) with x NOT
    ReLU'd (Q10); w = alpha*T; rgb = sum w*raw_rgb (no sigmoid, Q9); depth = sum w z;
    acc = sum w; disp = 1/max(1e-10, depth/acc); white => rgb += 1-acc.
    Returns rgb [B,3], disp [B,1], acc [B,1], weights [B,n,1], depth [B,1]."""
    raw_rgb = raw[..., :3]
    sigma = raw[..., 3]
    if raw_noise_std > 0.0:
        sigma = sigma + noise * raw_noise_std
    deltas = z_vals[..., 1:] - z_vals[..., :-1]
    deltas = torch.cat([deltas, torch.full_like(z_vals[..., :1], 1e10)], -1)
    deltas = deltas * torch.linalg.norm(rays_d[..., None, :], dim=-1)
    x = (deltas * sigma)[..., None]                                  # [B,n,1]
    alphas = 1.0 - torch.exp(-torch.relu(x))
    T = torch.cumsum(x[..., :-1, :], dim=-2)
    T = torch.cat([torch.zeros((*T.shape[:1], 1, 1), dtype=T.dtype, device=T.device), T], dim=-2)      # render.py:72-78
    T = torch.exp(-T)
    weights = alphas * T
    rgb = (weights * raw_rgb).sum(-2)
    depth = (weights[..., 0] * z_vals).sum(-1)[..., None]
    acc = weights.sum(-2)
    disp = 1.0 / torch.maximum(1e-10 * torch.ones_like(depth), depth / acc)
    if white_bkgd:
        rgb = rgb + (1.0 - acc)
    return rgb, disp, acc, weights, depth


# --------------------------------------------------------------------------------------
# a15 / a17  importance sampling + merge
# --------------------------------------------------------------------------------------

def inverse_cdf_parts(z_vals, weights, u, eps: float = 1e-5):
    """sampling/__init__.py:101-177 (the live torch-CPU sampler), with the uniform tensor
    `u` passed in instead of drawn by torch.rand (:140).  Returns (z_new, cdf, inds, below, above)."""
    w = weights[..., 0] + 0.01
    s = torch.sum(w, dim=-1, keepdim=True)
    pad = torch.relu(eps - s)
    w = w + pad / w.shape[-1]
    s = s + pad
    pdf = w / s
    cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    inds = torch.searchsorted(cdf, u.contiguous(), side="right")
    below = torch.clip(inds - 1, 0, cdf.shape[-1] - 1)
    above = torch.clip(inds, 0, cdf.shape[-1] - 1)
    c_from = torch.gather(cdf, -1, below)
    c_to = torch.gather(cdf, -1, above)
    zm = (z_vals[..., 1:] + z_vals[..., :-1]) / 2
    zm = torch.cat([zm[..., :1], zm, zm[..., -1:]], dim=-1)
    z_from = torch.gather(zm, -1, below)
    z_to = torch.gather(zm, -1, above)
    den = c_to - c_from
    den = torch.where(den < eps, torch.ones_like(den), den)
    t = torch.clip(torch.nan_to_num((u - c_from) / den, 0), 0.0, 1.0)
    return z_from + t * (z_to - z_from), cdf, inds, below, above


def sample_from_inverse_cdf(z_vals, weights, u, eps: float = 1e-5):
    return inverse_cdf_parts(z_vals, weights, u, eps)[0]


def merge_sorted(z_vals, z_imp):
    """entrypoints/__test_nerf.py:288, rendering/render.py:225: ascending sort of concat."""
    return torch.sort(torch.cat([z_vals, z_imp], dim=-1), dim=-1).values


# --------------------------------------------------------------------------------------
# a14 / a18 / a19  ray marching drivers
# --------------------------------------------------------------------------------------

def render_rays(arch, p_coarse, rays, n_samples: int, white_bkgd=False, lindisp=False, perturb=0.0,
                t_rand=None, ref_quirks=True, emulate_bf16=False, retraw=False):
    """rendering/render.py:112-162: coarse-only pass."""
    o, d, near, far, viewdirs = decompose_ray_batch(rays)
    z = (sample_z_lindisp if lindisp else sample_z_uniform)(near, far, n_samples)
    z = add_noise_z(z, perturb, t_rand)
    pos = o[..., None, :] + z[..., :, None] * d[..., None, :]
    raw = run_model(arch, p_coarse, pos, viewdirs, ref_quirks=ref_quirks, emulate_bf16=emulate_bf16)
    rgb, disp, acc, weights, depth = raw2outputs(raw, z, d, 0.0, white_bkgd)
    ret = {"rgb_map": rgb, "disp_map": disp, "acc_map": acc, "rgb_coarse": rgb, "disp_coarse": disp,
           "acc_coarse": acc, "z_vals": z, "weights": weights}
    if retraw:
        ret["raw"] = raw
    return ret


def render_rays_eval(arch, p_coarse, p_fine, rays, n_samples: int, n_importance: int, u,
                     white_bkgd=False, lindisp=False, ref_quirks=True, emulate_bf16=False):
    """rendering/render.py:164-241: coarse pass, importance sampling, sort, second pass
    through `network_fine or network_coarse`, kwargs' white_bkgd."""
    ret = render_rays(arch, p_coarse, rays, n_samples, white_bkgd, lindisp, 0.0, None, ref_quirks, emulate_bf16)
    o, d, near, far, viewdirs = decompose_ray_batch(rays)
    z_imp = sample_from_inverse_cdf(ret["z_vals"], ret["weights"], u)
    z = merge_sorted(ret["z_vals"], z_imp)
    pts = o[..., None, :] + d[..., None, :] * z[..., :, None]
    raw = run_model(arch, p_fine if p_fine is not None else p_coarse, pts, viewdirs, ref_quirks=ref_quirks,
                    emulate_bf16=emulate_bf16)
    rgb, disp, acc, w, depth = raw2outputs(raw, z, d, 0.0, white_bkgd)
    ret.update({"rgb_map": rgb, "disp_map": disp, "acc_map": acc, "z_fine": z})
    return ret


def render(arch, p_coarse, p_fine, H, W, K, c2w, near, far, n_samples, n_importance, u, chunk=32768,
           white_bkgd=True, ref_quirks=True, emulate_bf16=False):
    """rendering/render.py:243-345 (c2w path, ndc=False, use_viewdirs=True).
    `u` is [H*W, n_importance].  Returns [rgb [H,W,3], disp [H,W,1], acc [H,W,1], extras]."""
    o, d = get_rays(H, W, K, c2w)
    rays = pack_rays(o.reshape(-1, 3), d.reshape(-1, 3), near, far)
    outs: Dict[str, List[torch.Tensor]] = {}
    for i in range(0, rays.shape[0], chunk):
        r = render_rays_eval(arch, p_coarse, p_fine, rays[i:i + chunk], n_samples, n_importance, u[i:i + chunk],
                             white_bkgd, False, ref_quirks, emulate_bf16)
        for k, v in r.items():
            outs.setdefault(k, []).append(v)
    res = {k: torch.cat(v, 0) for k, v in outs.items()}
    res = {k: v.reshape(H, W, *v.shape[1:]) for k, v in res.items()}
    keys = ["rgb_map", "disp_map", "acc_map"]
    return [res[k] for k in keys] + [{k: v for k, v in res.items() if k not in keys}]


# --------------------------------------------------------------------------------------
# a20 / a21  losses, metrics, optimiser
# --------------------------------------------------------------------------------------

def mse(pred, gt):
    """ops/metric.py:12-14."""
    return torch.mean((pred - gt) ** 2)


def psnr(pred, gt):
    """ops/metric.py:16-18."""
    return 10.0 * torch.log10(1.0 / mse(pred, gt))


def ssim_window(w_size: int = 11, sigma: float = 1.5, ref_quirks: bool = True, dtype=torch.float32):
    """ops/metric.py:57-64 `gaussian`: [exp(-(x - w//2)**2) / (2 sigma**2)] / sum -- as committed the division is
    outside the exponential and cancels in the normalisation (effective window exp(-(x-c)^2)); ref_quirks=False is
    the textbook exp(-(x-c)^2 / (2 sigma^2))."""
    c = w_size // 2
    if ref_quirks:
        g = [math.exp(-(x - c) ** 2) / float(2 * sigma ** 2) for x in range(w_size)]
    else:
        g = [math.exp(-(x - c) ** 2 / float(2 * sigma ** 2)) for x in range(w_size)]
    g = torch.tensor(g, dtype=torch.float64)
    return (g / g.sum()).to(dtype)


def ssim(pred, gt, w_size: int = 11, size_average: bool = True, full: bool = False, ref_quirks: bool = True):
    """ops/metric.py:20-55 (SSIM.__call__ + create_window), with the part the reference leaves as "# TODO" (:44)
    completed by the formula its five moments exist for (Wang et al. 2004, the pytorch-ssim code the body follows):
    L from pred's range (:24-28); c1 = (0.01 L)^2, c2 = (0.03 L)^2; window = outer(g, g) per channel (:49-55);
    depthwise conv2d with padding 0 (:33-42); ssim_map = (2 mu_p mu_g + c1)(2 s_pg + c2) / ((mu_p^2 + mu_g^2 + c1)
    (s_p^2 + s_g^2 + c2)); mean over everything (size_average) or per image; full -> (ssim, cs)."""
    import torch.nn.functional as F
    _max = 255 if float(pred.max()) > 128 else 1
    _min = -1 if float(pred.min()) < -0.5 else 0
    L = _max - _min
    c1, c2 = (0.01 * L) ** 2, (0.03 * L) ** 2
    _, channel, height, width = pred.shape
    g = ssim_window(w_size, 1.5, ref_quirks, pred.dtype).to(pred.device)
    window = (g[:, None] @ g[None, :])[None, None].expand(channel, 1, w_size, w_size).contiguous()
    conv = lambda t: F.conv2d(t, window, padding=0, groups=channel)
    mu_p, mu_g = conv(pred), conv(gt)
    mu_pp, mu_gg, mu_pg = mu_p * mu_p, mu_g * mu_g, mu_p * mu_g
    s_pp, s_gg, s_pg = conv(pred * pred) - mu_pp, conv(gt * gt) - mu_gg, conv(pred * gt) - mu_pg
    v1, v2 = 2.0 * s_pg + c2, s_pp + s_gg + c2
    cs_map = v1 / v2
    ssim_map = ((2.0 * mu_pg + c1) * v1) / ((mu_pp + mu_gg + c1) * v2)
    if size_average:
        ret, cs = ssim_map.mean(), cs_map.mean()
    else:
        ret, cs = ssim_map.mean(dim=(1, 2, 3)), cs_map.mean(dim=(1, 2, 3))
    return (ret, cs) if full else ret


def adam_step(p, g, m, v, lr: float, b1=0.9, b2=0.999, eps=1e-8, bias_correction=False, step: int = 1):
    """mlx.optimizers.Adam.apply_single (mlx 0.7.0; not in /root/reference):
    m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr * m / (sqrt(v) + eps).
    No bias correction in that version; `bias_correction=True` is the
    non-quirk switch.  In-place on flat tensors."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    if bias_correction:
        mh = m / (1 - b1 ** step)
        vh = v / (1 - b2 ** step)
        p.sub_(lr * mh / (torch.sqrt(vh) + eps))
    else:
        p.sub_(lr * m / (torch.sqrt(v) + eps))


def lr_schedule(lrate: float, lrate_decay: int, i: int) -> float:
    """entrypoints/__test_nerf.py:302-305: lr_i = lrate * 0.1 ** (i / (lrate_decay*1000))."""
    return lrate * (0.1 ** (i / (lrate_decay * 1000)))


# --------------------------------------------------------------------------------------
# a25  poses
# --------------------------------------------------------------------------------------

def pose_spherical(theta: float, phi: float, radius: float) -> torch.Tensor:
    """ops/pose.py:7-58: c2w = swap . R_y(theta) . R_x(phi) . T_z(radius), float32."""
    t = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], dtype=torch.float32)
    ph = phi / 180.0 * np.pi
    rp = torch.tensor([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0],
                       [0, 0, 0, 1]], dtype=torch.float32)
    th = theta / 180.0 * np.pi
    rt = torch.tensor([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0],
                       [0, 0, 0, 1]], dtype=torch.float32)
    c2w = rt @ (rp @ t)
    swap = torch.tensor([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=torch.float32)
    return swap @ c2w


# --------------------------------------------------------------------------------------
# training step restatement (entrypoints/__test_nerf.py:47-145, 200-305) via autograd
# --------------------------------------------------------------------------------------

def coarse_loss(arch, p, rays, target, n_samples, white_bkgd=True, ref_quirks=True, emulate_bf16=False):
    """__test_nerf.py:47-90: loss = mean((rgb_coarse - y)^2), white_bkgd from kwargs."""
    r = render_rays(arch, p, rays, n_samples, white_bkgd, ref_quirks=ref_quirks, emulate_bf16=emulate_bf16)
    return mse(r["rgb_coarse"], target), r


def fine_loss(arch, p, rays, z_fine, target, ref_quirks=True, emulate_bf16=False):
    """__test_nerf.py:93-126: composites with white_bkgd=False hard-coded (Q8) in quirk
    mode; the non-quirk mode uses the white background like the eval path."""
    o, d, _, _, viewdirs = decompose_ray_batch(rays)
    pts = o[..., None, :] + d[..., None, :] * z_fine[..., :, None]
    raw = run_model(arch, p, pts, viewdirs, ref_quirks=ref_quirks, emulate_bf16=emulate_bf16)
    rgb, *_ = raw2outputs(raw, z_fine, d, 0.0, white_bkgd=not ref_quirks)
    return mse(rgb, target), rgb


class OracleTrainer:
    """The hot loop of entrypoints/__test_nerf.py:200-305 on flat fp32 parameter buffers:
    coarse step -> Adam -> re-render coarse (no grad) -> importance sample -> sort ->
    fine step -> Adam (same optimiser: shared m/v in quirk mode, Q7) -> lr update."""

    def __init__(self, arch: NerfArch, n_samples=64, n_importance=128, lrate=5e-4, lrate_decay=500, seed=0,
                 ref_quirks=True, emulate_bf16=False, near=2.0, far=6.0, device="cpu"):
        self.arch, self.n, self.N = arch, n_samples, n_importance
        self.lrate, self.decay, self.lr = lrate, lrate_decay, lrate
        self.q, self.emu, self.near, self.far = ref_quirks, emulate_bf16, near, far
        self.pc = flatten_params(arch, init_params(arch, seed)).to(device).requires_grad_(True)
        self.pf = flatten_params(arch, init_params(arch, seed + 1)).to(device).requires_grad_(True) if n_importance > 0 else None
        self.m = [torch.zeros_like(self.pc), torch.zeros_like(self.pc)]
        self.m2 = self.m if ref_quirks else [torch.zeros_like(self.pc), torch.zeros_like(self.pc)]
        self.it = 0

    def _adam(self, p, g, state):
        with torch.no_grad():
            adam_step(p, g, state[0], state[1], self.lr, bias_correction=False)

    def step(self, rays_o, rays_d, target, u):
        rays = pack_rays(rays_o, rays_d, self.near, self.far)
        loss, _ = coarse_loss(self.arch, unflatten_params(self.arch, self.pc), rays, target, self.n, True, self.q, self.emu)
        g, = torch.autograd.grad(loss, self.pc)
        self._adam(self.pc, g, self.m)
        out = {"loss_coarse": float(loss.detach())}
        if self.pf is not None:
            with torch.no_grad():
                r = render_rays(self.arch, unflatten_params(self.arch, self.pc), rays, self.n, True, ref_quirks=self.q,
                                emulate_bf16=self.emu)
                z_imp = sample_from_inverse_cdf(r["z_vals"], r["weights"], u)
                z_fine = merge_sorted(r["z_vals"], z_imp)
            lf, _ = fine_loss(self.arch, unflatten_params(self.arch, self.pf), rays, z_fine, target, self.q, self.emu)
            gf, = torch.autograd.grad(lf, self.pf)
            self._adam(self.pf, gf, self.m2)
            out["loss_fine"] = float(lf.detach())
        self.it += 1
        self.lr = lr_schedule(self.lrate, self.decay, self.it)
        return out


# --------------------------------------------------------------------------------------
# 2-D image fitting loop (entrypoints/__viser_image_learning.py:86-124, 198-236, 271-288)
# --------------------------------------------------------------------------------------

class OracleImageFitter:
    """embed = SinusoidalEncoding(2, 10, 0, 8) (:198); NeRF(40 -> 3, no view head) (:203-208);
    Adam(lr 1e-3, betas (0.9, 0.99)) (:224-227); loss = mean((model(embed(X)) - y)^2) (:210-219);
    X are INTEGER (row, col) pixel coordinates (:86-124)."""

    def __init__(self, seed: int = 0, lr: float = 1e-3, emulate_bf16: bool = False):
        self.arch = NerfArch(channel_input=40, channel_input_views=0, channel_output=3, use_viewdirs=False)
        self.p = flatten_params(self.arch, init_params(self.arch, seed)).requires_grad_(True)
        self.m, self.v = torch.zeros_like(self.p), torch.zeros_like(self.p)
        self.lr, self.emu = lr, emulate_bf16

    def forward(self, X, masks=None, taps=None):
        x = sinusoidal_encoding(X, 10, 0.0, 8.0, False)
        return nerf_forward(self.arch, unflatten_params(self.arch, self.p), x, self.emu, masks=masks, taps=taps)

    def step(self, X, y, masks=None):
        """masks (tests): the ReLU decisions of another implementation of the same forward (see nerf_forward)."""
        loss = mse(self.forward(X, masks=masks), y)
        g, = torch.autograd.grad(loss, self.p)
        with torch.no_grad():
            adam_step(self.p, g, self.m, self.v, self.lr, b1=0.9, b2=0.99)
        return float(loss.detach()), g


# --------------------------------------------------------------------------------------
# BASELINE configs[4]: hash grid + SH + the NeRF class at 2 x 64 (coarse-only loop)
# --------------------------------------------------------------------------------------

class OracleNGP:
    """MultiHashEncoding(3,L,Nmin,Nmax,F,log2T) (encoding/multi_hash.py, intended semantics) for positions,
    SphericalHarmonicsEncoding(3,3) for view directions, NeRF(n_layers=2, width=64, in 32+16, view head)
    (models/NeRF.py:160-243), render_rays' coarse pass (rendering/render.py:112-162) + raw2outputs + MSE, Adam WITH bias
    correction on the MLP and on the tables (this wiring is ours, not the reference's: without the correction the first
    steps are lr * sign(g), which amplifies bf16 noise in near-zero table gradients and can push sigma below zero
    everywhere -- a dead network under raw2outputs' un-activated sigma).  Tables are initialised by the caller (same values as the device)."""

    def __init__(self, tables: torch.Tensor, resolutions, seed=0, n_samples=64, lrate=5e-4, lrate_decay=500,
                 betas=(0.9, 0.99), eps=1e-8, bias_correction=True, emulate_bf16=False, near=2.0, far=6.0,
                 white_bkgd=True, bound=1.5):
        self.arch = NerfArch(channel_input=32, channel_input_views=16, n_layers=2, width=64, skips=(), use_viewdirs=True)
        self.p = flatten_params(self.arch, init_params(self.arch, seed)).requires_grad_(True)
        self.tables = tables.clone().float().requires_grad_(True)
        self.res = list(resolutions)
        self.n, self.lrate, self.decay, self.lr = n_samples, lrate, lrate_decay, lrate
        self.betas, self.eps, self.emu, self.bc = betas, eps, emulate_bf16, bias_correction
        self.near, self.far, self.white = near, far, white_bkgd
        self.pos_scale, self.pos_offset = (1.0, 0.0) if bound is None else (1.0 / (2.0 * bound), 0.5)   # scene box -> unit cube
        self.mp = [torch.zeros_like(self.p), torch.zeros_like(self.p)]
        self.mt = [torch.zeros_like(self.tables), torch.zeros_like(self.tables)]
        self.it = 0

    def render(self, rays, masks=None, taps=None):
        o, d, near, far, viewdirs = decompose_ray_batch(rays)
        z = sample_z_uniform(near, far, self.n)
        pts = o[..., None, :] + d[..., None, :] * z[..., :, None]
        B, n = z.shape
        feat = hashgrid_encoding(pts.reshape(-1, 3) * self.pos_scale + self.pos_offset, self.tables, self.res).reshape(B, n, -1)
        shf = sh_encoding(viewdirs, 3)
        x = torch.cat([feat, shf[:, None, :].expand(B, n, shf.shape[-1])], -1).reshape(B * n, -1)
        raw = nerf_forward(self.arch, unflatten_params(self.arch, self.p), x, self.emu, masks=masks, taps=taps).reshape(B, n, 4)
        if taps is not None:
            taps["raw"], taps["x"] = raw, x
        rgb, *_ = raw2outputs(raw, z, d, 0.0, white_bkgd=self.white)
        return rgb

    def loss_and_grads(self, rays_o, rays_d, target, masks=None):
        """masks (tests): the ReLU decisions of another implementation of the same forward (see nerf_forward)."""
        rays = pack_rays(rays_o, rays_d, self.near, self.far)
        loss = mse(self.render(rays, masks=masks), target)
        gp, gt = torch.autograd.grad(loss, [self.p, self.tables])
        return loss.detach(), gp, gt

    def step(self, rays_o, rays_d, target):
        loss, gp, gt = self.loss_and_grads(rays_o, rays_d, target)
        with torch.no_grad():
            adam_step(self.p, gp, self.mp[0], self.mp[1], self.lr, self.betas[0], self.betas[1], self.eps, self.bc, self.it + 1)
            adam_step(self.tables, gt, self.mt[0], self.mt[1], self.lr, self.betas[0], self.betas[1], self.eps, self.bc, self.it + 1)
        self.it += 1
        self.lr = lr_schedule(self.lrate, self.decay, self.it)
        return float(loss)
