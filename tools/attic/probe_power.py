#!/usr/bin/env python3
"""Is the split-fp16 render forward limited by its instruction schedule or by chip power?  The SAME binary and launch on
(a) the benchmark's randomly initialised network and (b) an all-zero network (identical instruction stream: the chain is branch-free;
only the operand bits differ).  Operand toggling drives MFMA power, power drives the clock / the issue throttle.
   python tools/probe_power.py random|zero|small [16|22]      (under tools/pmc_split22.sh for matrix-busy cycles and the clock)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_meets_mlx_amd.models.NeRF import NeRF

mode = sys.argv[1] if len(sys.argv) > 1 else "random"
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 22
reps = int(os.environ.get("REPS", "6"))
dev = "cuda"
B, n = 32768, 192
m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=4, precision=prec)
if mode == "zero":
    m.load_flat(torch.zeros_like(m.params))
elif mode == "small":
    m.load_flat(m.params * 1e-3)
g = torch.Generator(device=dev).manual_seed(0)
o = torch.nn.functional.normalize(torch.randn(B, 3, device=dev, generator=g), dim=-1) * 4.0
d = -o / 4.0 + 0.2 * torch.randn(B, 3, device=dev, generator=g)
rays = torch.cat([o, d, torch.full((B, 1), 2.0, device=dev), torch.full((B, 1), 6.0, device=dev), torch.nn.functional.normalize(d, dim=-1)], -1).contiguous()
z = torch.sort(torch.rand(B, n, device=dev, generator=g) * 4 + 2, -1).values.contiguous()
for _ in range(3):
    m.query(rays, z)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    raw = m.query(rays, z)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps
peak = {16: 2500.0, 22: 2500.0 / 3, 32: 157.3}[prec]
print(f"precision {prec} weights {mode:6s}: {t:.3f} ms per 32768 x 192 samples = {2 * 593408 * B * n / t / 1e9:.0f} TF algorithmic = "
      f"{2 * 593408 * B * n / t / 1e9 / peak:.3f} of {peak:.0f}; |raw|max {float(raw.abs().max()):.3g}")
