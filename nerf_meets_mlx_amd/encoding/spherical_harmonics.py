"""`mlx_nerf/encoding/spherical_harmonics.py:13-94`: real SH basis, degree 0..4."""
import torch

from .. import _native as N
from . import Encoding


class SphericalHarmonicsEncoding(Encoding):
    def __init__(self, in_dim: int, n_degrees: int) -> None:
        super().__init__(in_dim)
        assert 0 <= n_degrees <= 4, f"[ERROR] {n_degrees=} must be in range [0, 4]!"
        self.n_degrees = n_degrees

    def get_out_dim(self):
        return (self.n_degrees + 1) ** 2

    def __call__(self, in_dirs: torch.Tensor):
        d = N.f32(in_dirs).reshape(-1, 3)
        out = torch.empty(d.shape[0], self.get_out_dim(), dtype=torch.float32, device=d.device)
        N.check(N.lib().nerf_sh_encode(N.ptr(d), d.shape[0], self.n_degrees, N.ptr(out), N.stream()))
        return out.reshape(*in_dirs.shape[:-1], self.get_out_dim())
