"""`mlx_nerf/sampling/uniform.py:7-18`: z = near*(1-t) + far*t, t = linspace(0,1,n)."""
from . import _rays_from_bounds, sample_coarse


def sample_z(near, far, n_samples: int):
    return sample_coarse(_rays_from_bounds(near, far), n_samples, lindisp=False)
