"""Does the fine network's backward chain run slower INSIDE the training step than alone because of what runs in front of it?  (round 6:
the kernel takes 2.4 ms in a PMC pass -- kernels serialised -- and 3.0-3.1 ms in the bench's kernel trace.)  Times the chain (bwd_stage
1) and the weight gradients (bwd_stage 2) of the fine pass with events, in the step's own order, with and without an HBM-bound filler
(a device-to-device copy of --filler-mb) between the training forward and the chain.    python tools/probe_chain_context.py"""
import argparse, sys
import numpy as np, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import _native, sampling
from nerf_meets_mlx_amd.dataset import synthetic
from nerf_meets_mlx_amd.engine.trainer import Trainer
from nerf_meets_mlx_amd.rendering import render
ap = argparse.ArgumentParser(); ap.add_argument("--filler-mb", type=int, default=4096); ap.add_argument("--steps", type=int, default=12)
a = ap.parse_args()
dev = torch.device("cuda", 0)
lib = _native.lib()
imgs, poses, rposes, hwf, K = synthetic.make_dataset(800, 800, 4, seed=0, device=dev)
tr = Trainer(imgs, poses, K, N_rand=4096, n_depth_samples=64, N_importance=128, seed=4, device=dev, precision=22)
src = torch.empty(a.filler_mb << 20, dtype=torch.uint8, device=dev); dst = torch.empty_like(src)


def step(filler):
    rays, target = tr.sample_batch()
    z = sampling.sample_coarse(rays, 64)
    tr._step_net(tr.coarse, rays, z, target, True)
    raw = tr.coarse.query(rays, z)
    _, _, _, w, _ = render.composite(raw, z, rays, 0.0, True)
    _, zf = sampling.importance_sample(z, w, 128, u=tr.train_uniforms(rays.shape[0]))
    m = tr._fine
    e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    e[0].record()
    rawf = m.query(rays, zf, train=True)
    loss, d_raw, _ = render.composite_mse_backward(rawf, zf, rays, target, False)
    e[1].record()
    if filler:
        dst.copy_(src)
    e[2].record()
    _native.check(lib.nerf_set_option(b"bwd_stage", 1)); m.backward(d_raw)
    e[3].record()
    _native.check(lib.nerf_set_option(b"bwd_stage", 2)); g = m.backward(d_raw)
    e[4].record()
    _native.check(lib.nerf_set_option(b"bwd_stage", 0))
    tr._opt.update(m, g)
    tr.it += 1
    return e


for _ in range(4):
    step(False)
for filler in (False, True, False, True):
    ev = [step(filler) for _ in range(a.steps)]
    torch.cuda.synchronize()
    f = lambda i, j: float(np.mean([x[i].elapsed_time(x[j]) for x in ev]))
    print(f"filler {int(filler)} ({a.filler_mb} MiB copy): forward+composite {f(0, 1):.3f} ms | filler {f(1, 2):.3f} | chain {f(2, 3):.3f} | weight gradients {f(3, 4):.3f}", flush=True)
