#!/usr/bin/env python3
"""Launches of the precision-22 kernels at bench sizes, for rocprofv3 passes (tools/pmc_split22.sh) and quick timings:
split-fp16 inference forward (32768 x 192 and 32768 x 64 samples), split-bf16 training forward / chain + dW (4096 x 192)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_meets_mlx_amd.models.NeRF import NeRF                      # noqa: E402

DEV = "cuda"


def rays_z(B, n, g):
    o = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * 4.0
    d = -o / 4.0 + 0.25 * torch.randn(B, 3, generator=g)
    vd = torch.nn.functional.normalize(d, dim=-1)
    rays = torch.cat([o, d, torch.full((B, 1), 2.0), torch.full((B, 1), 6.0), vd], -1).to(DEV)
    return rays, torch.sort(torch.rand(B, n, generator=g) * 4 + 2, -1).values.to(DEV)


def main():
    reps = int(os.environ.get("REPS", 3))
    prec = int(os.environ.get("PREC", 22))
    g = torch.Generator().manual_seed(0)
    m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=3, precision=prec)
    r_big, z_big = rays_z(32768, 192, g)
    z_c = z_big[:, :64].contiguous()
    r_t, z_t = rays_z(4096, 192, g)
    dr = (torch.randn(4096, 192, 4, generator=g) * 1e-4).to(DEV)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    t = {"infer_fine": 0.0, "infer_coarse": 0.0, "train_fwd": 0.0, "train_bwd": 0.0}
    for it in range(reps + 1):
        e = [ev() for _ in range(5)]
        e[0].record(); m.query(r_big, z_big); e[1].record(); m.query(r_big, z_c); e[2].record()
        m.query(r_t, z_t, train=True); e[3].record(); m.backward(dr); e[4].record()
        torch.cuda.synchronize()
        if it > 0:
            for k, (a, b) in zip(t, ((0, 1), (1, 2), (2, 3), (3, 4))):
                t[k] += e[a].elapsed_time(e[b]) / reps
    fl = 2 * 593408
    print(f"precision {prec}: inference 32768x192 {t['infer_fine']:.3f} ms ({fl * 32768 * 192 / t['infer_fine'] / 1e9:.0f} TF), "
          f"32768x64 {t['infer_coarse']:.3f} ms ({fl * 32768 * 64 / t['infer_coarse'] / 1e9:.0f} TF); training 4096x192: forward "
          f"{t['train_fwd']:.3f} ms, chain + dW {t['train_bwd']:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
