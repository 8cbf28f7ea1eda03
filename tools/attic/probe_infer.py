import os, sys, torch
sys.path.insert(0, ".")
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
B, n = 32768, 192
r = rays(B); z = torch.sort(torch.rand(B, n, device=dev) * 4 + 2, -1).values
from nerf_meets_mlx_amd import _native
ref = None
for variant in (3, 4, 5, 3, 4, 5):
    _native.check(_native.lib().nerf_set_option(b"mlp_variant", variant))
    out = m.query(r, z)
    if ref is None: ref = out
    else: print("max |v - v3| =", (out - ref).abs().max().item(), "rel", ((out - ref).norm() / ref.norm()).item())
    for rep in range(2):
        ms = timeit(lambda: m.query(r, z))
        print(f"[{tag}] variant {variant} fwd-infer B={B} n={n}: {ms:.3f} ms  {2*593408*B*n/ms/1e9:.0f} TFLOP/s", flush=True)
