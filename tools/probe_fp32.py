#!/usr/bin/env python3
"""Throughput of the fp32 reference-precision mode (NeRF(precision=32) = nerf_mlp_arch.precision 32, csrc/mlp32.hip) against the fp32
matrix peak (157.3 TFLOP/s, MI355X_MICROARCH.md), next to the bf16 mode on the same inputs.

    python tools/probe_fp32.py > profiles/r02_fp32_mode.csv
"""
import sys, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd.models.NeRF import NeRF
dev = "cuda"
FLOP = 2 * 593408


def rays(B):
    o = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1) * 4
    d = -o / 4 + 0.2 * torch.randn(B, 3, device=dev)
    r = torch.zeros(B, 11, device=dev); r[:, :3] = o; r[:, 3:6] = d; r[:, 6] = 2; r[:, 7] = 6
    r[:, 8:] = d / d.norm(dim=-1, keepdim=True); return r


def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


print("mode,pass,rays,samples_per_ray,ms,TFLOP_per_s,frac_of_peak")
for bits, peak in ((32, 157.3), (16, 2500.0)):
    m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0, precision=bits)
    for B, n in ((4096, 64), (4096, 192), (32768, 192)):
        r = rays(B); z = torch.sort(torch.rand(B, n, device=dev) * 4 + 2, -1).values
        g = torch.randn(B, n, 4, device=dev)
        t = timeit(lambda: m.query(r, z))
        print(f"fp{bits},forward (render),{B},{n},{t:.3f},{FLOP * B * n / t / 1e9:.1f},{FLOP * B * n / t / 1e9 / peak:.3f}")
        if B <= 4096:
            def step():
                m.query(r, z, train=True); m.backward(g)
            t = timeit(step)
            print(f"fp{bits},forward+backward (train),{B},{n},{t:.3f},{3 * FLOP * B * n / t / 1e9:.1f},{3 * FLOP * B * n / t / 1e9 / peak:.3f}")
