#!/usr/bin/env python3
"""Development check of the split-bf16 training kernels (csrc/mlp_s16.hip, NeRF(precision=22) with train=True): errors
against the fp32 kernels on the same inputs and kernel timings at the bench's training size (4096 rays x 192 samples)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_meets_mlx_amd.models.NeRF import NeRF                      # noqa: E402

DEV = "cuda"


def mk(p):
    return NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=3, precision=p)


def main():
    B, n = int(os.environ.get("B", 4096)), int(os.environ.get("N", 192))
    g = torch.Generator().manual_seed(0)
    o = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * 4.0
    d = -o / 4.0 + 0.25 * torch.randn(B, 3, generator=g)
    vd = torch.nn.functional.normalize(d, dim=-1)
    rays = torch.cat([o, d, torch.full((B, 1), 2.0), torch.full((B, 1), 6.0), vd], -1).to(DEV)
    z = torch.sort(torch.rand(B, n, generator=g) * 4 + 2, -1).values.to(DEV)
    dr = (torch.randn(B, n, 4, generator=g) * 1e-4).to(DEV)
    res = {}
    for p in (32, 22, 16):
        m = mk(p)
        raw = m.query(rays, z, train=True)
        gr = m.backward(dr).clone()
        torch.cuda.synchronize()
        res[p] = (raw.clone(), gr)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        reps = 5
        tf = tb = 0.0
        for _ in range(reps):
            ev[0].record(); m.query(rays, z, train=True); ev[1].record(); m.backward(dr); ev[2].record()
            torch.cuda.synchronize()
            tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
        print(f"precision {p}: training forward {tf / reps:.3f} ms, backward (chain + dW) {tb / reps:.3f} ms  [{B} x {n} samples]", flush=True)
    r32, g32 = res[32]
    for p in (22, 16):
        r, gg = res[p]
        print(f"precision {p} vs 32: forward max err / scale {float((r - r32).abs().max() / r32.abs().max()):.2e}; "
              f"gradient rel-L2 {float((gg - g32).double().norm() / g32.double().norm()):.2e}, finite {bool(torch.isfinite(gg).all())}")


if __name__ == "__main__":
    main()
