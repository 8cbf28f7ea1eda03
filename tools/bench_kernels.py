#!/usr/bin/env python3
"""Achieved HBM rate of the non-MLP (bandwidth-bound) kernels at bench.py's sizes: algorithmic bytes / time.

    python tools/bench_kernels.py > profiles/r01_hbm_kernels.csv

Algorithmic bytes = compulsory reads + writes of the call's tensors (each counted once); peak 8 TB/s HBM3E
(MI355X_MICROARCH.md); the streaming ceilings measured with tools/attic/hbm_probe.hip on this pool are 5.4-5.8 (stores) and
6.2-7.1 TB/s (loads).  Timing: HIP events around 20 back-to-back launches on the launch stream."""
import sys, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import sampling
from nerf_meets_mlx_amd.rendering import render, ray
from nerf_meets_mlx_amd.dataset import synthetic
from nerf_meets_mlx_amd.encoding.multi_hash import MultiHashEncoding
from nerf_meets_mlx_amd.encoding.spherical_harmonics import SphericalHarmonicsEncoding
from nerf_meets_mlx_amd.models.NeRF import NeRF, Adam

dev = "cuda"
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
rows = []
def report(name, shape, nbytes, fn):
    t = timeit(fn)
    rows.append((name, shape, nbytes, t * 1e6, nbytes / t / 1e9, nbytes / t / 8e12))

H = W = 800
imgs, poses, rposes, hwf, K = synthetic.make_dataset(H, W, 1, seed=0, device=dev)
B = 32768 * 8                      # 8 render chunks per call: long enough to leave the launch-latency regime
idx = torch.arange(B, device=dev, dtype=torch.int64)
c2w = rposes[40][:3, :4]
report("ray_gen", f"{B} rays", B * (8 + 44), lambda: ray.gen_rays(H, W, K, c2w, 2.0, 6.0, idx))
rays = ray.gen_rays(H, W, K, c2w, 2.0, 6.0, idx)
report("sample_coarse n=64", f"{B} rays", B * (8 + 256), lambda: sampling.sample_coarse(rays, 64))
z64 = sampling.sample_coarse(rays, 64)
torch.manual_seed(0)
raw64 = torch.randn(B, 64, 4, device=dev)
report("composite fwd n=64 (rgb, acc, weights)", f"{B} rays", B * (64 * 20 + 44 + 16 + 64 * 4), lambda: render.composite(raw64, z64, rays, 0.0, True))
w64 = render.composite(raw64, z64, rays, 0.0, True)[3]
u = torch.rand(B, 128, device=dev)
report("importance sample + merge 64 -> 128 / 192", f"{B} rays", B * 4 * (64 + 64 + 128 + 128 + 192),
       lambda: sampling.importance_sample(z64, w64, 128, u=u))
zf = sampling.importance_sample(z64, w64, 128, u=u)[1]
B2 = 32768 * 4
raw192 = torch.randn(B2, 192, 4, device=dev); zf2 = zf[:B2].contiguous(); rays2 = rays[:B2].contiguous()
report("composite fwd n=192 (rgb only)", f"{B2} rays", B2 * (192 * 20 + 44 + 12),
       lambda: render.composite(raw192, zf2, rays2, 0.0, True, need_weights=False))
d_rgb = torch.randn(B2, 3, device=dev)
report("composite bwd n=192", f"{B2} rays", B2 * (192 * 20 + 44 + 12 + 192 * 16),
       lambda: render.composite_backward(raw192, zf2, rays2, d_rgb, True))
m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0)
opt = Adam(5e-4, shared_state=True)
m.grads.normal_()
report("adam step (595 844 params)", "1 network", 595844 * 28, lambda: opt.update(m))
# hash grid: 16 levels x 2^19 x 2 features (Instant-NGP defaults), M points in [-1.5, 1.5]^3
M = 1 << 22
enc = MultiHashEncoding(3, 16, 16, 2048, 2, 19, device=dev)
x = (torch.rand(M, 3, device=dev) * 3 - 1.5)
report("hashgrid fwd (16 levels x 8 corners x 8 B gathered)", f"{M} points", M * (12 + 128 + 16 * 8 * 8), lambda: enc(x))
g = torch.randn(M, 32, device=dev)
report("hashgrid bwd (float atomics)", f"{M} points", M * (12 + 128 + 16 * 8 * 8), lambda: enc.backward(x, g))
sh = SphericalHarmonicsEncoding(3, 4)
d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev), dim=-1)
report("sh_encode degree 4", f"{M} dirs", M * (12 + 100), lambda: sh(d))
from nerf_meets_mlx_amd.ops.metric import SSIM
ia = torch.rand(4, 3, 800, 800, device=dev); ib = (ia + 0.05 * torch.randn_like(ia)).clamp(0, 1)
ssim = SSIM()
report("ssim 11x11 window (4 x 3 x 800 x 800, both images read once)", "4 images", 2 * ia.numel() * 4, lambda: ssim(ia, ib))
print("kernel,work,algorithmic_bytes,avg_us,GB_per_s,frac_of_8TBps")
for r in rows:
    print(f"{r[0]},{r[1]},{r[2]},{r[3]:.1f},{r[4]:.0f},{r[5]:.3f}")
