"""Probe: training-kernel timings (fwd-train, bwd chain + dW) for the library selected by NERF_HIP_LIB."""
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
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0)
tag = os.path.basename(os.environ.get("NERF_HIP_LIB", "default"))
for rep in range(2):
    for B, n in ((4096, 64), (4096, 192)):
        r = rays(B); z = torch.sort(torch.rand(B, n, device=dev) * 4 + 2, -1).values
        g = torch.randn(B, n, 4, device=dev)
        f = timeit(lambda: m.query(r, z, train=True))
        m.query(r, z, train=True)
        b = timeit(lambda: m.backward(g))
        _native.check(_native.lib().nerf_set_option(b"bwd_stage", 1))
        c = timeit(lambda: m.backward(g))
        _native.check(_native.lib().nerf_set_option(b"bwd_stage", 0))
        _native.check(_native.lib().nerf_set_option(b"mlp_variant", 3))
        i = timeit(lambda: m.query(r, z))
        _native.check(_native.lib().nerf_set_option(b"mlp_variant", 0))
        print(f"[{tag}] B={B} n={n}: fwd-train {f:.3f} ms  bwd+dW {b:.3f} ms (chain {c:.3f})  fwd-infer(v3) {i:.3f} ms", flush=True)
