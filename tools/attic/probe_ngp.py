"""Probe: step time of the hash-grid + 2x64 path at bench sizes, by stage."""
import sys, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import sampling
from nerf_meets_mlx_amd.dataset import synthetic
from nerf_meets_mlx_amd.engine.ngp import NGPTrainer
from nerf_meets_mlx_amd.rendering import ray, render
from nerf_meets_mlx_amd.ops.metric import mse_loss_grad
dev = "cuda"
def timeit(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
H = W = 800
imgs, poses, rposes, hwf, K = synthetic.make_dataset(H, W, 2, seed=0, device=dev)
tr = NGPTrainer(imgs, poses, K, N_rand=4096, n_depth_samples=64, seed=0, device=dev)
rays, target = tr.sample_batch()
print(f"train step (4096 rays x 64): {timeit(lambda: tr.train_step(rays, target)):.3f} ms", flush=True)
idx = torch.arange(32768, device=dev, dtype=torch.int64) + 200 * 800
rr = ray.gen_rays(H, W, K, rposes[40][:3, :4], 2.0, 6.0, idx)
print(f"render chunk (32768 rays x 64): {timeit(lambda: tr.render_rays(rr)):.3f} ms", flush=True)
f = tr.field
z = sampling.sample_coarse(rays, 64)
pts, x = f.features(rays, z)
print(f"  train: features (o+zd, hash fwd, sh, cat) {timeit(lambda: f.features(rays, z)):.3f} | hash fwd only {timeit(lambda: f.enc(pts)):.3f}")
print(f"  train: mlp fwd-train {timeit(lambda: f.mlp.forward(x, train=True)):.3f}")
raw = f.query(rays, z, train=True)
rgb = render.composite(raw, z, rays, 0.0, True)[0]
loss, d_rgb = mse_loss_grad(rgb, target)
d_raw = render.composite_backward(raw, z, rays, d_rgb, True)
print(f"  train: mlp bwd + dW + dx {timeit(lambda: f.mlp.backward(d_raw, need_input_grad=True)):.3f}")
g, d_x = f.mlp.backward(d_raw, need_input_grad=True)
print(f"  train: hash bwd {timeit(lambda: f.enc.backward(pts, d_x)):.3f} | grad zero {timeit(lambda: f.enc.grad.zero_()):.3f} | adam table {timeit(lambda: tr.opt.update(f.table, f.enc.grad.view(-1))):.3f}")
zr = sampling.sample_coarse(rr, 64)
ptsr, xr = f.features(rr, zr)
print(f"  render: features {timeit(lambda: f.features(rr, zr)):.3f} | hash fwd only {timeit(lambda: f.enc(ptsr)):.3f} | mlp fwd {timeit(lambda: f.mlp.forward(xr)):.3f}")
print(f"  render: fused query {timeit(lambda: f.query(rr, zr)):.3f} | unfused query {timeit(lambda: f.query(rr, zr, fused=False)):.3f}")
print(f"  train: fused query {timeit(lambda: f.query(rays, z, train=True)):.3f} | unfused {timeit(lambda: f.query(rays, z, train=True, fused=False)):.3f}")
