"""Pixel selection without replacement.

`np.random.choice(H*W, N_rand, replace=False)` (entrypoints/__test_nerf.py:229) builds a
full H*W permutation on the host every iteration.  Here the batch is N_rand consecutive
outputs of a keyed bijection of [0, H*W) (4-round Feistel network + cycle walking), O(N_rand)
on the device (`nerf_pixel_permutation`).  `pixel_permutation_host` is the bit-exact host
mirror used by the parity tests.
"""
import numpy as np
import torch

from .. import _native as N

_M64 = (1 << 64) - 1


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def _mix32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13); x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


def pixel_permutation_host(n: int, domain: int, seed: int, offset: int = 0) -> np.ndarray:
    """Host mirror of csrc/rays.hip perm_kernel (must stay bit-identical)."""
    assert 0 <= offset and offset + n <= domain
    bits = 2
    while bits < 64 and (1 << bits) < domain:
        bits += 1
    bits += bits & 1
    half = bits // 2
    mask = np.uint64((1 << half) - 1)
    keys = [np.uint64(_splitmix64((seed + r) & _M64) & 0xFFFFFFFF) for r in range(4)]
    v = np.arange(offset, offset + n, dtype=np.uint64)
    todo = np.ones(n, dtype=bool)
    while todo.any():
        x = v[todo]
        L, R = (x >> np.uint64(half)) & mask, x & mask
        for r in range(4):
            f = _mix32(((R * np.uint64(0x9E3779B1)) + keys[r]) & np.uint64(0xFFFFFFFF)) & mask
            L, R = R, L ^ f
        x = (L << np.uint64(half)) | R
        v[todo] = x
        todo[todo] = x >= np.uint64(domain)
    return v.astype(np.int64)


def pixel_permutation(n: int, domain: int, seed: int, offset: int = 0, device="cuda") -> torch.Tensor:
    out = torch.empty(n, dtype=torch.int64, device=device)
    N.check(N.lib().nerf_pixel_permutation(N.ptr(out), n, domain, seed & _M64, offset, N.stream()))
    return out


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[i] = src[idx[i]] for a [n_src, C] float tensor (target pixels, __test_nerf.py:236)."""
    src2 = src.reshape(-1, src.shape[-1])
    out = torch.empty(idx.numel(), src2.shape[1], dtype=torch.float32, device=src.device)
    N.check(N.lib().nerf_gather_rows(N.ptr(src2), src2.shape[0], N.ptr(idx), idx.numel(), src2.shape[1], N.ptr(out),
                                     N.stream()))
    return out
