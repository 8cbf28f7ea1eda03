"""Depth sampling with the reference's call surface (`mlx_nerf/sampling/__init__.py`).

`add_noise_z` (:10-31) and the live inverse-CDF sampler `sample_from_inverse_cdf_torch`
(:101-177) run as HIP kernels (csrc/sampling.hip); the reference's device->numpy->torch-CPU
round trip (entrypoints/__test_nerf.py:275-285) disappears.  The dead MLX sampler
(`sample_from_inverse_cdf`, :34-99) is not mirrored; the name aliases the live one.
"""
from typing import Optional

import torch

from .. import _native as N

__all__ = ["add_noise_z", "sample_from_inverse_cdf", "sample_from_inverse_cdf_torch", "importance_sample"]


def _rays_from_bounds(near: torch.Tensor, far: torch.Tensor) -> torch.Tensor:
    B = near.shape[0]
    rays = torch.zeros(B, 11, dtype=torch.float32, device=near.device)
    rays[:, 6] = near.reshape(B)
    rays[:, 7] = far.reshape(B)
    return rays


def sample_coarse(rays: torch.Tensor, n: int, lindisp: bool = False, perturb: float = 0.0,
                  t_rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """z [B,n] from packed rays [B,11] (a5-a7 fused: sample_z + add_noise_z)."""
    B = rays.shape[0]
    z = torch.empty(B, n, dtype=torch.float32, device=rays.device)
    if perturb > 0.0 and t_rand is None:
        t_rand = torch.rand(B, n, dtype=torch.float32, device=rays.device)     # mx.random.uniform (:17)
    N.check(N.lib().nerf_sample_coarse(N.ptr(rays), B, n, int(bool(lindisp)), float(perturb),
                                       N.ptr(t_rand) if perturb > 0.0 else None, N.ptr(z), N.stream()))
    return z


def add_noise_z(z_vals: torch.Tensor, strength: float = 1.0, t_rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Stratified jitter (`sampling/__init__.py:10-31`, intended semantics: SURVEY Q6)."""
    if strength <= 0.0:
        return z_vals
    z = N.f32(z_vals)
    if t_rand is None:
        t_rand = torch.rand_like(z)                                             # mx.random.uniform (:17)
    else:                                                                       # the kernel reads one uniform per depth:
        t_rand = N.f32(t_rand, z.device).expand_as(z).contiguous()              # broadcast like the reference's multiply (:29); a wrong shape raises
    n = z.shape[-1]
    out = torch.empty_like(z)
    N.check(N.lib().nerf_add_noise_z(N.ptr(z), N.ptr(t_rand), z.numel() // n, n, float(strength), N.ptr(out),
                                     N.stream()))
    return out


def importance_sample(z_vals, weights, n_importance_samples: int, u=None, eps: float = 1e-5, merge: bool = True,
                      return_parts: bool = False):
    """Fused a15 + a17: returns (z_new [B,N], z_merged [B,n+N]) (+ cdf, inds when asked)."""
    z = N.f32(z_vals)
    w = N.f32(weights[..., 0] if weights.dim() == 3 else weights)
    B, n = z.shape
    Nn = int(n_importance_samples)
    if u is None:
        u = torch.rand(B, Nn, dtype=torch.float32, device=z.device)            # torch.rand (:140)
    u = N.f32(u)
    z_new = torch.empty(B, Nn, dtype=torch.float32, device=z.device)
    z_m = torch.empty(B, n + Nn, dtype=torch.float32, device=z.device) if merge else None
    cdf = torch.empty(B, n + 1, dtype=torch.float32, device=z.device) if return_parts else None
    inds = torch.empty(B, Nn, dtype=torch.int64, device=z.device) if return_parts else None
    N.check(N.lib().nerf_importance_sample(N.ptr(z), N.ptr(w), N.ptr(u), B, n, Nn, float(eps), N.ptr(z_new),
                                           N.ptr(z_m), N.ptr(cdf), N.ptr(inds), N.stream()))
    if return_parts:
        return z_new, z_m, cdf, inds
    return z_new, z_m


def sample_from_inverse_cdf_torch(z_vals, weights, n_importance_samples, eps=1e-5, is_stratified_sampling=False,
                                  u=None) -> torch.Tensor:
    """Same signature as the reference (:102-108) plus an optional explicit `u`."""
    if is_stratified_sampling:
        # the reference's branch raises TypeError (:135-136); a working version:
        B = z_vals.shape[0]
        u = torch.linspace(0.0, 1.0, n_importance_samples, device=z_vals.device).expand(B, n_importance_samples)
    return importance_sample(z_vals, weights, n_importance_samples, u=u, eps=eps, merge=False)[0]


sample_from_inverse_cdf = sample_from_inverse_cdf_torch
