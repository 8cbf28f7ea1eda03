"""Probe: the weight-gradient kernel alone (bwd_stage 2), whole job list and per job, B=4096 x n=192 samples."""
import os, sys, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import _native
from nerf_meets_mlx_amd.models.NeRF import NeRF
dev = "cuda"
def rays(B):
    o = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1) * 4
    d = -o / 4 + 0.2 * torch.randn(B, 3, device=dev)
    r = torch.zeros(B, 11, device=dev); r[:, :3] = o; r[:, 3:6] = d; r[:, 6] = 2; r[:, 7] = 6
    r[:, 8:] = d / d.norm(dim=-1, keepdim=True); return r
def timeit(fn, it=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
tag = os.path.basename(os.environ.get("NERF_HIP_LIB", "default"))
opt = lambda k, v: _native.check(_native.lib().nerf_set_option(k, v))
m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0)
B, n = 4096, 192
r = rays(B); z = torch.sort(torch.rand(B, n, device=dev) * 4 + 2, -1).values
g = torch.randn(B, n, 4, device=dev)
m.query(r, z, train=True)
m.backward(g)
names = ["pos0", "pos1", "pos2", "pos3", "pos4", "pos5|H4", "pos5|PE", "pos6", "pos7", "feature", "alpha", "dir0|feat", "dir0|dPE", "rgb"]
frags = [20, 32, 32, 32, 32, 32, 20, 32, 32, 32, 17, 24, 10, 9]
opt(b"bwd_stage", 1); t_chain = timeit(lambda: m.backward(g))
opt(b"bwd_stage", 2); t_dw = timeit(lambda: m.backward(g))
tot = sum(frags) * 1024 * (B * n / 32)
print(f"[{tag}] chain {t_chain:.3f} ms | dW all jobs {t_dw:.3f} ms = {tot/t_dw/1e9:.2f} TB/s", flush=True)
if "--jobs" in sys.argv:
    for j, (nm, fr) in enumerate(zip(names, frags)):
        opt(b"dw_job_mask", 1 << j)
        t = timeit(lambda: m.backward(g))
        print(f"[{tag}]   job {nm:10s} {fr:2d} frags/tile: {t:.3f} ms = {fr*1024*(B*n/32)/t/1e9:.2f} TB/s", flush=True)
    opt(b"dw_job_mask", 0)
opt(b"bwd_stage", 0)
