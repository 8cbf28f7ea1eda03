#!/usr/bin/env python3
"""Time of the image model's precision-22 forward without stores (mlp_s16x.hip, 32x32x16 split bf16) for the library selected with
NERF_HIP_LIB -- the simplest ring kernel, used for timing-only ablations (tools/ab_one.sh).  Prints ms and the MFMA fraction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_meets_mlx_amd.models.NeRF import NeRF
M = 32768 * 192
m = NeRF(channel_input=40, channel_input_views=0, channel_output=3, is_use_view_directions=False, device="cuda", seed=0, precision=22)
x = torch.randn(M, 40, device="cuda")
mode = os.environ.get("PROBE_DATA", "random")      # random | zero_weights | small_weights: operand toggling drives chip power, hence the clock
if mode == "zero_weights":
    m.load_flat(torch.zeros_like(m.params))
elif mode == "small_weights":
    m.load_flat(m.params * 1e-3)
for _ in range(3):
    m.forward(x)
torch.cuda.synchronize()
ts = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        m.forward(x)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5)
t = min(ts)
# executed MFMA work: 2880 v_mfma_f32_32x32x16 per 32 samples (incl. padding), 32 cycles each, one wave per SIMD
print(f"{os.path.basename(os.environ.get('NERF_HIP_LIB', 'shipped')):22s} {mode:14s} {t:7.3f} ms   algorithmic {2 * 480000 * M / t / 1e9 / 833.3:.3f} of 833 TF   "
      f"(MFMA-only time at 2.4 GHz: {2880 * 32 * (M / 32) / 1024 / 2.4e9 * 1e3:.2f} ms)")
