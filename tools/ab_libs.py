"""A/B of two BUILDS of the library (tools/ab_one.sh) on one box: the render chunk of the bench (coarse 32768 x 64 + fine 32768 x 192
samples through the precision-22 inference forward) in alternating child processes, outputs compared bit for bit.
    python tools/ab_libs.py tools/diag/libnerf_A.so tools/diag/libnerf_B.so [--rounds 3]      ("default" = the shipped library)"""
import argparse, hashlib, json, os, subprocess, sys
CHILD = r'''
import sys, json, hashlib, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd.models.NeRF import NeRF
from oracle import nerf_oracle as O
g = torch.Generator().manual_seed(1)
m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device="cuda", seed=4, precision=int(sys.argv[1]))
B = 32768
o = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * 4.0
d = -o / 4.0 + 0.25 * torch.randn(B, 3, generator=g)
rays = O.pack_rays(o, d, 2.0, 6.0).cuda()
res = {}
for n in (64, 192):
    z = (torch.sort(torch.rand(B, n, generator=g), -1).values * 4 + 2).cuda()
    for _ in range(3): m.query(rays, z)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(12): out = m.query(rays, z)
    e1.record(); torch.cuda.synchronize()
    res[f"ms_{n}"] = e0.elapsed_time(e1) / 12
    res[f"sha_{n}"] = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
print(json.dumps(res))
'''
ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--rounds", type=int, default=3); ap.add_argument("--precision", type=int, default=22)
a = ap.parse_args()
acc = {l: [] for l in a.libs}
for r in range(a.rounds):
    for l in a.libs:
        env = dict(os.environ)
        if l != "default": env["NERF_HIP_LIB"] = os.path.abspath(l)
        out = subprocess.run([sys.executable, "-c", CHILD, str(a.precision)], capture_output=True, text=True, env=env)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if not line: print(l, "FAILED", out.stderr[-500:]); continue
        acc[l].append(json.loads(line[-1]))
for l in a.libs:
    rs = acc[l]
    print(f"{l}: coarse " + " ".join(f"{x['ms_64']:.3f}" for x in rs) + " ms | fine " + " ".join(f"{x['ms_192']:.3f}" for x in rs) + f" ms | sha {rs[0]['sha_64']} {rs[0]['sha_192']}")
