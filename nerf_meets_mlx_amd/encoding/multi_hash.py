"""Multiresolution hash grid (`mlx_nerf/encoding/multi_hash.py:13-136`), intended semantics.

The committed reference class cannot run; this follows its formulas
(growth factor :35-37, N_l :40, T :43, hash :61-77, corner / lerp order :93-131) with the
host evaluating b in float64 and the hash in uint32 wrap-around arithmetic.
"""
import ctypes as C
import math

import torch

from .. import _native as N
from . import Encoding


class MultiHashEncoding(Encoding):
    def __init__(self, in_dim: int, n_levels: int, min_res: int, max_res: int, n_features_per_level: int,
                 log2_hashmap_size: int, hash_init_scale: float = 0.0001, device="cuda", seed: int = 0) -> None:
        super().__init__(in_dim)
        assert in_dim == 3, "hash grid is implemented for 3-D inputs"
        self.n_levels, self.min_res, self.max_res = n_levels, min_res, max_res
        self.n_features_per_level, self.log2_hashmap_size = n_features_per_level, log2_hashmap_size
        self.growing_factor = math.exp((math.log(max_res) - math.log(min_res)) / (n_levels - 1)) if n_levels > 1 else 1.0
        self.scaled_res = [int(math.floor(min_res * self.growing_factor ** l + 1e-9)) for l in range(n_levels)]
        self.hash_table_size = 2 ** log2_hashmap_size
        g = torch.Generator(device="cpu").manual_seed(seed)
        t = (torch.rand(n_levels, self.hash_table_size, n_features_per_level, generator=g) * 2 - 1) * hash_init_scale
        self.tables = t.to(device)                                    # U(-1e-4, 1e-4): paper / intent (:51)
        self.grad = torch.zeros_like(self.tables)
        self._res_c = (C.c_int * n_levels)(*self.scaled_res)

    def get_out_dim(self):
        return self.n_levels * self.n_features_per_level

    def __call__(self, in_array: torch.Tensor):
        x = N.f32(in_array)
        out = torch.empty(x.shape[0], self.get_out_dim(), dtype=torch.float32, device=x.device)
        N.check(N.lib().nerf_hashgrid_forward(N.ptr(x), x.shape[0], N.ptr(self.tables), self.n_levels,
                                              self.log2_hashmap_size, self.n_features_per_level, self._res_c,
                                              N.ptr(out), N.stream()))
        return out

    def backward(self, in_array: torch.Tensor, d_out: torch.Tensor):
        """self.grad += d(out)/d(tables)^T d_out.  A float32 `self.grad` is accumulated with float atomics; an int64
        `self.grad` (engine/ngp.py, deterministic mode) holds 2^-52 fixed-point accumulators added with integer atomics."""
        x, g = N.f32(in_array), N.f32(d_out)
        N.check(N.lib().nerf_hashgrid_backward_ex(N.ptr(x), x.shape[0], N.ptr(g), self.n_levels, self.log2_hashmap_size,
                                                  self.n_features_per_level, self._res_c, 0, self.n_levels,
                                                  int(self.grad.dtype == torch.int64), N.ptr(self.grad), N.stream()))
        return self.grad
