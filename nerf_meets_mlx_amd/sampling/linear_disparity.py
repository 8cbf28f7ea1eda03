"""`mlx_nerf/sampling/linear_disparity.py:8-19`, restated literally."""
from . import _rays_from_bounds, sample_coarse


def sample_z(near, far, n_samples: int):
    return sample_coarse(_rays_from_bounds(near, far), n_samples, lindisp=True)
