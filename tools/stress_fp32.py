#!/usr/bin/env python3
"""Repeatability stress of the fp32 kernels (csrc/mlp32.hip): the same training forward, backward and inference forward
again and again, every result compared BIT FOR BIT with the first, while bf16 kernels of a second model run on another
stream every other iteration to perturb the timing.  The fp32 kernels wait for their asm-issued loads with their own
counted s_waitcnt: a wait that passes too early shows up here as a handful of differing outputs in some iterations (the
first version of the store-aware waits did), long before an accuracy test notices.

    python tools/stress_fp32.py 400        # 0 mismatches on the round's final kernels
    python tools/stress_fp32.py 400 22     # the same for the split-precision kernels (round 4)
"""
import sys

import torch

sys.path.insert(0, ".")


def rays(B, dev):
    o = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1) * 4
    d = -o / 4 + 0.2 * torch.randn(B, 3, device=dev)
    r = torch.zeros(B, 11, device=dev)
    r[:, :3] = o; r[:, 3:6] = d; r[:, 6] = 2; r[:, 7] = 6
    r[:, 8:] = d / d.norm(dim=-1, keepdim=True)
    return r


def run(iters=200, B=4096, n=192, dev="cuda", verbose=True, precision=32):
    """Returns the number of (iteration, tensor) pairs that differed from the first iteration.  precision 22: the split
    kernels (csrc/mlp22.hip, csrc/mlp_s16.hip: LDS ring with M0-clobbering DMA runs, half-tile pair-block staging in dW);
    their training and inference forwards are different kernels, so those two are not compared with each other."""
    from nerf_meets_mlx_amd.models.NeRF import NeRF
    torch.manual_seed(0)
    m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0, precision=precision)
    m16 = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=1, precision=16)
    r = rays(B, dev)
    z = torch.sort(torch.rand(B, n, device=dev) * 4 + 2, -1).values
    g = torch.randn(B, n, 4, device=dev)
    side = torch.cuda.Stream()
    ref, bad = None, 0
    for it in range(iters):
        if it % 2 == 1:
            with torch.cuda.stream(side):
                m16.query(r, z, train=True)
                m16.backward(g)
        raw = m.query(r, z, train=True).clone()
        gr = m.backward(g).clone()
        inf = m.query(r, z).clone()
        torch.cuda.synchronize()
        cur = (raw, gr, inf)
        if ref is None:
            ref = cur
            assert precision != 32 or torch.equal(raw, inf), "training and inference forward differ"
            continue
        for a, b, name in zip(ref, cur, ("raw_train", "grads", "raw_inference")):
            if not torch.equal(a, b):
                bad += 1
                if verbose:
                    print("MISMATCH", it, name, int((a != b).sum()), float((a - b).abs().max()), flush=True)
        if verbose and it % 50 == 0:
            print("it", it, "bad", bad, flush=True)
    return bad


if __name__ == "__main__":
    n_bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, precision=int(sys.argv[2]) if len(sys.argv) > 2 else 32)
    print("done bad =", n_bad)
    sys.exit(1 if n_bad else 0)
