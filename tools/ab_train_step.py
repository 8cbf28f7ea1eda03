"""A/B of a nerf_set_option setting in the bench's own step (train N_rand 4096 rays + render one 32 768-ray chunk, precision 22 by
default): the settings alternate in blocks of `--steps` steps on ONE trainer, so that both see the same clocks / temperature; prints
the training and the render phase per block (HIP events).   python tools/ab_train_step.py dw22_variant 0 1 [--precision 16]"""
import argparse, sys, numpy as np, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import _native, sampling
from nerf_meets_mlx_amd.dataset import synthetic
from nerf_meets_mlx_amd.engine.trainer import Trainer
from nerf_meets_mlx_amd.rendering import ray, render
ap = argparse.ArgumentParser()
ap.add_argument("key"); ap.add_argument("values", type=int, nargs="+")
ap.add_argument("--steps", type=int, default=12); ap.add_argument("--rounds", type=int, default=3); ap.add_argument("--precision", type=int, default=22)
a = ap.parse_args()
dev = torch.device("cuda", 0)
H = W = 800
imgs, poses, rposes, hwf, K = synthetic.make_dataset(H, W, 4, seed=0, device=dev)
ridx = torch.arange(0, 32768, device=dev, dtype=torch.int64)
rrays = ray.gen_rays(H, W, K, rposes[40][:3, :4], 2.0, 6.0, ridx)
tr = Trainer(imgs, poses, K, N_rand=4096, n_depth_samples=64, N_importance=128, seed=4, device=dev, chunk=32768, precision=a.precision)
def render_chunk():
    z = sampling.sample_coarse(rrays, 64)
    raw = tr.coarse.query(rrays, z)
    _, _, _, w, _ = render.composite(raw, z, rrays, 0.0, True)
    u = torch.rand(rrays.shape[0], 128, device=dev, generator=tr.gen)
    _, zf = sampling.importance_sample(z, w, 128, u=u)
    raw = tr.fine.query(rrays, zf, ref_quirks=True)
    return render.composite(raw, zf, rrays, 0.0, True, need_weights=False)[0]
def block(n):
    ev = []
    for _ in range(n):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); tr.train_step(); e[1].record(); render_chunk(); e[2].record()
        ev.append(e)
    torch.cuda.synchronize()
    return float(np.mean([e[0].elapsed_time(e[1]) for e in ev])), float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
block(5)
res = {v: [] for v in a.values}
for r in range(a.rounds):
    for v in a.values:
        _native.check(_native.lib().nerf_set_option(a.key.encode(), v))
        block(2)
        res[v].append(block(a.steps))
for v in a.values:
    t = [x[0] for x in res[v]]; rn = [x[1] for x in res[v]]
    print(f"{a.key} = {v}: train {np.mean(t):.3f} ms/step ({4096 / np.mean(t) / 1e3:.4f} M rays/s; blocks " + " ".join(f"{x:.3f}" for x in t) + f") | render {np.mean(rn):.3f} ms", flush=True)
