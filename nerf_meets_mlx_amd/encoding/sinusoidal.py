"""`mlx_nerf/encoding/sinusoidal.py:13-66` (the image-learning PE, true 2^k frequencies)."""
import ctypes as C

import numpy as np
import torch

from .. import _native as N
from . import Encoding


class SinusoidalEncoding(Encoding):
    def __init__(self, in_dim: int, n_freqs: int, min_freq_exp: float = None, max_freq_exp: float = None,
                 is_include_input: bool = False) -> None:
        super().__init__(in_dim)
        self.n_freqs = n_freqs
        self.min_freq_exp = min_freq_exp if min_freq_exp else 0.0                  # :27 (0.0 is falsy)
        self.max_freq_exp = max_freq_exp if max_freq_exp else float(n_freqs - 1)   # :28
        self.is_include_input = is_include_input

    def get_out_dim(self):
        return self.in_dim * self.n_freqs * 2 + (self.in_dim if self.is_include_input else 0)

    def freq_bands(self) -> np.ndarray:
        """2 ** mx.linspace(min, max, n) in float32 (mx.linspace = arange * step + start)."""
        step = np.float32((self.max_freq_exp - self.min_freq_exp) / (self.n_freqs - 1)) if self.n_freqs > 1 else np.float32(0)
        e = np.arange(self.n_freqs, dtype=np.float32) * step + np.float32(self.min_freq_exp)
        return torch.pow(torch.tensor(2.0), torch.from_numpy(e)).numpy().astype(np.float32)

    def __call__(self, in_array: torch.Tensor):
        x = N.f32(in_array)
        M = x.shape[0]
        out = torch.empty(M, self.get_out_dim(), dtype=torch.float32, device=x.device)
        fr = self.freq_bands()
        N.check(N.lib().nerf_encode_sinusoidal(N.ptr(x), M, self.in_dim, self.n_freqs,
                                               (C.c_float * self.n_freqs)(*fr.tolist()),
                                               int(self.is_include_input), N.ptr(out), N.stream()))
        return out
