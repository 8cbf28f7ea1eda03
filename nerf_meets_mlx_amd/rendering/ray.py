"""Ray generation with the reference's call surface (`mlx_nerf/rendering/ray.py`).

`get_rays` (:7-35) returns the full [H,W,3] origin / direction images like the reference;
`gen_rays` is the hot-path form: packed [n,11] rays for a list of pixel indices, produced
on the device without materialising the image.
"""
import ctypes as C
from typing import Optional

import numpy as np
import torch

from .. import _native as N


def _host_cam(K, c2w):
    K = np.asarray(K.detach().cpu() if torch.is_tensor(K) else K, dtype=np.float64).reshape(3, 3)
    c = np.asarray(c2w.detach().cpu() if torch.is_tensor(c2w) else c2w, dtype=np.float32)[:3, :4]
    Kc = (C.c_double * 9)(*K.reshape(-1).tolist())
    cc = (C.c_float * 12)(*np.ascontiguousarray(c).reshape(-1).tolist())
    return Kc, cc


def gen_rays(H: int, W: int, K, c2w, near: float, far: float, pixel_idx: Optional[torch.Tensor] = None,
             device="cuda", return_coords: bool = False):
    """Packed rays [n,11] = [o, d, near, far, viewdirs] for `pixel_idx` (None = all H*W pixels)."""
    n = H * W if pixel_idx is None else pixel_idx.numel()
    if pixel_idx is not None:
        device = pixel_idx.device
    rays = torch.empty(n, 11, dtype=torch.float32, device=device)
    coords = torch.empty(n, 2, dtype=torch.int64, device=device) if return_coords else None
    Kc, cc = _host_cam(K, c2w)
    N.check(N.lib().nerf_ray_gen(N.ptr(pixel_idx), n, H, W, Kc, cc, float(near), float(far), N.ptr(rays),
                                 N.ptr(coords), N.stream()))
    return (rays, coords) if return_coords else rays


def sample_batch(H: int, W: int, K, c2w, near: float, far: float, image: torch.Tensor, n: int, seed: int, offset: int = 0,
                 return_idx: bool = False):
    """One training batch in one launch (`nerf_sample_batch`): n distinct pixels of `image` [H,W,3] (the keyed permutation
    of ops.index.pixel_permutation), their packed rays [n,11] and target colours [n,3] -- `__test_nerf.py:213-236, 60-82`."""
    img = N.f32(image).reshape(-1, 3)
    rays = torch.empty(n, 11, dtype=torch.float32, device=img.device)
    target = torch.empty(n, 3, dtype=torch.float32, device=img.device)
    idx = torch.empty(n, dtype=torch.int64, device=img.device) if return_idx else None
    Kc, cc = _host_cam(K, c2w)
    N.check(N.lib().nerf_sample_batch(n, H, W, seed & ((1 << 64) - 1), offset, Kc, cc, float(near), float(far), N.ptr(img),
                                      N.ptr(rays), N.ptr(target), N.ptr(idx), N.stream()))
    return (rays, target, idx) if return_idx else (rays, target)


def get_rays(H: int, W: int, K, c2w, device="cuda"):
    """(rays_o, rays_d), each [H,W,3] float32 (`rendering/ray.py:7-35`)."""
    rays = gen_rays(H, W, K, c2w, 0.0, 1.0, None, device)
    return rays[:, 0:3].reshape(H, W, 3), rays[:, 3:6].reshape(H, W, 3)


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """NDC warp (`rendering/ray.py:39-70`); returns new (rays_o, rays_d)."""
    shp = rays_o.shape
    n = rays_o.numel() // 3
    rays = torch.zeros(n, 11, dtype=torch.float32, device=rays_o.device)
    rays[:, 0:3] = rays_o.reshape(n, 3)
    rays[:, 3:6] = rays_d.reshape(n, 3)
    N.check(N.lib().nerf_ndc_rays(N.ptr(rays), n, int(H), int(W), float(focal), float(near), N.stream()))
    return rays[:, 0:3].reshape(shp), rays[:, 3:6].reshape(shp)
