"""f22_tiles 2 | 3 (32 | 48 samples per wave in the split-fp16 render forward): bit-identical outputs?  time per launch at the bench's two
launch sizes, alternating on one process / one box.    python tools/ab_f22_tiles.py"""
import sys, hashlib, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import _native
from nerf_meets_mlx_amd.models.NeRF import NeRF
from oracle import nerf_oracle as O
lib = _native.lib()
g = torch.Generator().manual_seed(1)
m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device="cuda", seed=4, precision=22)
B = 32768
o = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * 4.0
d = -o / 4.0 + 0.25 * torch.randn(B, 3, generator=g)
rays = O.pack_rays(o, d, 2.0, 6.0).cuda()
zs = {n: (torch.sort(torch.rand(B, n, generator=g), -1).values * 4 + 2).cuda() for n in (64, 192)}
# ragged sizes: same values?
for Bq, n in ((1, 1), (37, 45), (1000, 64), (4099, 192)):
    zz = (torch.sort(torch.rand(Bq, n, generator=g), -1).values * 4 + 2).cuda()
    outs = []
    for t in (2, 3):
        _native.check(lib.nerf_set_option(b"f22_tiles", t))
        outs.append(m.query(rays[:Bq].contiguous(), zz).clone())
    print(f"B={Bq} n={n}: tiles 3 == tiles 2 bit for bit: {torch.equal(outs[0], outs[1])}  max |diff| {float((outs[0]-outs[1]).abs().max()):.3e}")
res = {2: {64: [], 192: []}, 3: {64: [], 192: []}}
for r in range(4):
    for t in (2, 3):
        _native.check(lib.nerf_set_option(b"f22_tiles", t))
        for n in (64, 192):
            for _ in range(3): m.query(rays, zs[n])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): m.query(rays, zs[n])
            e1.record(); torch.cuda.synchronize()
            res[t][n].append(e0.elapsed_time(e1) / 10)
for t in (2, 3):
    print(f"f22_tiles {t}: coarse " + " ".join(f"{x:.3f}" for x in res[t][64]) + " ms | fine " + " ".join(f"{x:.3f}" for x in res[t][192]) + " ms")
_native.check(lib.nerf_set_option(b"f22_tiles", 0))
