"""`mlx_nerf/encoding/identity.py:13-31`: pass-through."""
from . import Encoding


class IdentityEncoding(Encoding):
    def get_out_dim(self):
        return self.in_dim

    def __call__(self, in_array):
        return in_array
