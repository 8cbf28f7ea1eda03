#!/usr/bin/env python3
"""Round-5 probe: matrix-pipe fraction of the split-precision forwards, 16x16x32 (mlp22.hip) against 32x32x16 (mlp_s16x.hip's image
model without stores), back to back on one box.  python tools/probe_r5.py [M]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_meets_mlx_amd.models.NeRF import NeRF

dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768 * 192


def timeit(fn, n=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


view = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0, precision=22)
img = NeRF(channel_input=40, channel_input_views=0, channel_output=3, is_use_view_directions=False, device=dev, seed=0, precision=22)
B, n = M // 192, 192
g = torch.Generator(device=dev).manual_seed(0)
o = torch.nn.functional.normalize(torch.randn(B, 3, device=dev, generator=g), dim=-1) * 4.0
d = -o / 4.0 + 0.2 * torch.randn(B, 3, device=dev, generator=g)
rays = torch.cat([o, d, torch.full((B, 1), 2.0, device=dev), torch.full((B, 1), 6.0, device=dev), torch.nn.functional.normalize(d, dim=-1)], -1).contiguous()
z = torch.sort(torch.rand(B, n, device=dev, generator=g) * 4 + 2, -1).values.contiguous()
x40 = torch.randn(M, 40, device=dev, generator=g)
for rep in range(2):
    t = timeit(lambda: view.query(rays, z))
    print(f"view p22 inference (mlp22, 16x16x32 f16): {t:.3f} ms  {2 * 593408 * M / t / 1e9:.0f} TF algorithmic = {2 * 593408 * M / t / 1e9 / 833.3:.3f} of 833")
    t = timeit(lambda: img.forward(x40))
    print(f"image p22 inference (s16x, 32x32x16 bf16, no stores): {t:.3f} ms  {2 * 480000 * M / t / 1e9:.0f} TF algorithmic = {2 * 480000 * M / t / 1e9 / 833.3:.3f} of 833")
Mt = 4096 * 192
xt = x40[:Mt].contiguous()
t = timeit(lambda: img.forward(xt, train=True))
print(f"image p22 training forward (stores), {Mt} samples: {t:.3f} ms  {2 * 480000 * Mt / t / 1e9 / 833.3:.3f} of 833")
t = timeit(lambda: view.query(rays[:4096].contiguous(), z[:4096].contiguous(), train=True))
print(f"view p22 training forward (stores), {Mt} samples: {t:.3f} ms  {2 * 593408 * Mt / t / 1e9 / 833.3:.3f} of 833")
