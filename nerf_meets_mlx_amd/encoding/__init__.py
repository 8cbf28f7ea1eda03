"""Encoders with the reference's `Encoding` interface (`mlx_nerf/encoding/__init__.py:10-24`):
`get_out_dim()` and `__call__(x)`; every `__call__` is one HIP kernel (csrc/encode.hip)."""
import torch


class Encoding:
    def __init__(self, in_dim: int) -> None:
        self.in_dim = in_dim

    def forward(self, in_array: torch.Tensor):
        raise NotImplementedError

    def get_out_dim(self):
        raise NotImplementedError


from .identity import IdentityEncoding            # noqa: E402
from .sinusoidal import SinusoidalEncoding        # noqa: E402
from .spherical_harmonics import SphericalHarmonicsEncoding  # noqa: E402
from .multi_hash import MultiHashEncoding         # noqa: E402
