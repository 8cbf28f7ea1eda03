"""A/B of nerf_set_option("ring_split", 1 | 2): one 8-wave workgroup per CU behind a 128 KiB weight ring, or two independent
4-wave workgroups behind 64 KiB rings (training forward with activation stores, backward chain)."""
import sys, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import _native
from nerf_meets_mlx_amd.models.NeRF import NeRF
dev = "cuda"
def rays(B):
    o = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1) * 4
    d = -o / 4 + 0.2 * torch.randn(B, 3, device=dev)
    r = torch.zeros(B, 11, device=dev); r[:, :3] = o; r[:, 3:6] = d; r[:, 6] = 2; r[:, 7] = 6
    r[:, 8:] = d / d.norm(dim=-1, keepdim=True); return r
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
opt = lambda k, v: _native.check(_native.lib().nerf_set_option(k, v))
m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0)
ref = {}
for rep in range(2):
    for split in (1, 2):
        opt(b"ring_split", split)
        for B, n in ((4096, 64), (4096, 192), (1024, 192)):
            torch.manual_seed(B + n)
            r = rays(B); z = torch.sort(torch.rand(B, n, device=dev) * 4 + 2, -1).values
            g = torch.randn(B, n, 4, device=dev)
            f = timeit(lambda: m.query(r, z, train=True))
            raw = m.query(r, z, train=True)
            opt(b"bwd_stage", 1)
            c = timeit(lambda: m.backward(g))
            opt(b"bwd_stage", 0)
            grads = m.backward(g).clone()
            key = (B, n)
            if split == 1 and rep == 0:
                ref[key] = (raw.clone(), grads)
            elif split == 2 and rep == 0:
                print(f"   split 2 vs 1: raw max|d| {(raw - ref[key][0]).abs().max().item():.3e}  grads max|d| {(grads - ref[key][1]).abs().max().item():.3e}")
            print(f"[split {split}] B={B} n={n}: fwd-train {f:.3f} ms  chain {c:.3f} ms", flush=True)
opt(b"ring_split", 1)
