"""Probe: effect of the fragment-store tile stride (tile_pad16) on the training kernels."""
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
def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
_native.check(_native.lib().nerf_set_option(b"mlp_variant", 1))
B, n = 4096, 192
r = rays(B); z = torch.sort(torch.rand(B, n, device=dev) * 4 + 2, -1).values
g = torch.randn(B, n, 4, device=dev)
for pad in (0, 16, 17, 80, 272, 1040):
    _native.check(_native.lib().nerf_set_option(b"tile_pad16", pad))
    m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0)
    msf = timeit(lambda: m.query(r, z, train=True))
    for wgs in (256, 768):
        _native.check(_native.lib().nerf_set_option(b"dw_workgroups", wgs))
        msb = timeit(lambda: m.backward(g))
        print(f"pad16={pad:5d} wgs={wgs}: fwd(train) {msf:.3f} ms   bwd chain + dW {msb:.3f} ms", flush=True)
    del m
