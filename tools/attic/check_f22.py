#!/usr/bin/env python3
"""Split-fp16 (precision 22) inference forward: accuracy against the fp32 oracle / the fp32 kernels and speed against the
fp32 and bf16 kernels at render-chunk size.  Development tool; the parity tests are in tests/test_gpu_round4.py."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_meets_mlx_amd.models.NeRF import NeRF          # noqa: E402
from oracle import nerf_oracle as O                      # noqa: E402

DEV = "cuda"


def main():
    mk = lambda p: NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=4, precision=p)
    m16, m22, m32 = mk(16), mk(22), mk(32)
    g = torch.Generator().manual_seed(1)
    for B, n in ((37, 45), (1024, 64), (4096, 192)):
        o = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * 4.0
        d = -o / 4.0 + 0.25 * torch.randn(B, 3, generator=g)
        rays = O.pack_rays(o, d, 2.0, 6.0).to(DEV)
        z = (torch.sort(torch.rand(B, n, generator=g), -1).values * 4 + 2).to(DEV)
        r22, r32, r16 = m22.query(rays, z), m32.query(rays, z), m16.query(rays, z)
        torch.cuda.synchronize()
        rec = {"case": f"query B={B} n={n}", "scale": float(r32.abs().max())}
        if B * n <= 70000:
            # float64 GEMMs on the float32 embedding (the reference's float32 x * f products and sin / cos, then exact layers)
            arch = O.NerfArch()
            p = O.unflatten_params(arch, m32.params.cpu().double())
            pts = rays[:, None, 0:3].cpu() + z.cpu()[:, :, None] * rays[:, None, 3:6].cpu()
            xe = O.embed(pts, rays[:, 8:11].cpu(), ref_quirks=True)
            ref = O.nerf_forward(arch, p, xe.double()).float().reshape(B, n, 4).to(DEV)
            sc = float(ref.abs().max())
            rec.update({"err22_vs_oracle64": float((r22 - ref).abs().max()) / sc, "err32_vs_oracle64": float((r32 - ref).abs().max()) / sc,
                        "err16_vs_oracle64": float((r16 - ref).abs().max()) / sc})
        sc = float(r32.abs().max())
        rec.update({"err22_vs_f32kernel": float((r22 - r32).abs().max()) / sc, "err16_vs_f32kernel": float((r16 - r32).abs().max()) / sc,
                    "finite": bool(torch.isfinite(r22).all())})
        print(json.dumps(rec), flush=True)
    # NeRF.forward(x) entry (embedded rows)
    x = torch.randn(5000, 90, generator=g).to(DEV)
    y22, y32 = m22.forward(x), m32.forward(x)
    print(json.dumps({"case": "forward(x) M=5000", "err22_vs_f32kernel": float((y22 - y32).abs().max() / y32.abs().max())}), flush=True)
    # speed at the render fine pass size
    B, n = 32768, 192
    o = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * 4.0
    d = -o / 4.0 + 0.25 * torch.randn(B, 3, generator=g)
    rays = O.pack_rays(o, d, 2.0, 6.0).to(DEV)
    z = (torch.sort(torch.rand(B, n, generator=g), -1).values * 4 + 2).to(DEV)
    flop = 2 * 593408 * B * n
    for name, m, reps in (("bf16", m16, 20), ("split_fp16", m22, 10), ("fp32", m32, 4)):
        for _ in range(2):
            m.query(rays, z)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            m.query(rays, z)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(json.dumps({"case": f"speed {name}", "ms_per_launch": ms, "tflops_equiv": flop / ms / 1e9, "samples": B * n}), flush=True)


if __name__ == "__main__":
    main()
