#!/usr/bin/env python3
"""What each fused MLP kernel returns for non-finite INPUTS (round 6, review item 1a): a NaN / Inf of either sign in a ray
origin, a view direction, a depth or an embedded row, per precision and entry point.  The reference's nn.relu = mx.maximum
propagates NaN (models/NeRF.py:222,236): position NaN -> all four outputs NaN; direction NaN -> rgb NaN, alpha finite.
    python tools/probe_nan_inputs.py"""
import os
import struct
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_meets_mlx_amd.models.NeRF import NeRF                       # noqa: E402
from nerf_meets_mlx_amd.models import embedding                       # noqa: E402


def bits(u):
    return struct.unpack("f", struct.pack("I", u))[0]


POISON = {"+nan": 0x7FC00000, "-nan": 0xFFC00000, "+inf": 0x7F800000, "-inf": 0xFF800000}


def put(t, idx, u):
    t.view(torch.int32)[idx] = u if u < 2 ** 31 else u - 2 ** 32


def main():
    dev = "cuda"
    torch.manual_seed(0)
    B, n = 96, 64
    o = torch.randn(B, 3, device=dev) * 0.3
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
    rays = torch.cat([o, d, torch.full((B, 1), 2.0, device=dev), torch.full((B, 1), 6.0, device=dev), d], -1).contiguous()
    z = torch.linspace(2, 6, n, device=dev).expand(B, n).contiguous()
    for prec in (22, 32, 16):
        m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0, precision=prec)
        for train in (False, True):
            clean = m.query(rays, z, train=train).clone()
            for name, u in POISON.items():
                for what, col in (("origin", 1), ("dir(pts)", 4), ("viewdir", 9)):
                    r2 = rays.clone()
                    put(r2, (17, col), u)
                    raw = m.query(r2, z, train=train)
                    bad = torch.isnan(raw[17])
                    others = torch.equal(torch.cat([raw[:17], raw[18:]]), torch.cat([clean[:17], clean[18:]]))
                    print(f"prec {prec} train {int(train)} {name} in {what:9s}: ray 17 NaN rgb {int(bad[:, :3].all(-1).sum())}/{n} "
                          f"alpha {int(bad[:, 3].sum())}/{n}; finite-wrong rgb {int((~bad[:, :3].any(-1)).sum())}; others bit-identical {others}")
                z2 = z.clone()
                put(z2, (17, 5), u)
                raw = m.query(rays, z2, train=train)
                bad = torch.isnan(raw[17, 5])
                print(f"prec {prec} train {int(train)} {name} in z[17,5]: sample NaN {bad.tolist()}")
        # embedded rows (NeRF.forward(x))
        ep, _ = embedding.get_embedder(10)
        ed, _ = embedding.get_embedder(4)
        pts = (rays[:, None, 0:3] + z[..., None] * rays[:, None, 3:6])
        x = embedding.embed(pts, ep, rays[:, 8:11], ed).reshape(-1, 90).contiguous()
        clean = m.forward(x).clone()
        for name, u in POISON.items():
            for what, col in (("pos ch 0", 0), ("pos ch 62", 62), ("dir ch 0", 63), ("dir ch 26", 89)):
                x2 = x.clone()
                put(x2, (1000, col), u)
                out = m.forward(x2)
                bad = torch.isnan(out[1000])
                keep = torch.ones(x.shape[0], dtype=torch.bool, device=dev)
                keep[1000] = False
                print(f"prec {prec} rows {name} in {what:9s}: out NaN {bad.tolist()} value {out[1000].tolist()}; others bit-identical "
                      f"{torch.equal(out[keep], clean[keep])}")
    # image model and 2 x 64 model
    for prec in (22, 32, 16):
        m = NeRF(channel_input=40, channel_input_views=0, channel_output=3, is_use_view_directions=False, device=dev, seed=0, precision=prec)
        x = torch.randn(512, 40, device=dev)
        clean = m.forward(x).clone()
        for name, u in POISON.items():
            x2 = x.clone()
            put(x2, (100, 7), u)
            out = m.forward(x2)
            print(f"image prec {prec} {name}: out {out[100].tolist()}")
    for prec in (22, 16):
        m = NeRF(n_layers=2, width_layers=64, channel_input=32, channel_input_views=16, list_skip_connection_layers=[],
                 is_use_view_directions=True, device=dev, seed=0, precision=prec)
        x = torch.randn(512, 48, device=dev)
        for name, u in POISON.items():
            for col in (3, 40):
                x2 = x.clone()
                put(x2, (100, col), u)
                out = m.forward(x2)
                print(f"2x64 prec {prec} {name} col {col}: out {out[100].tolist()}")


if __name__ == "__main__":
    main()
