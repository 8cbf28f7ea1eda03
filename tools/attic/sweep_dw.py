"""dW job split sweep (cost-model bias x workgroup count); PREC=22 sweeps the split-bf16 dW kernel."""
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
opt = lambda k, v: _native.check(_native.lib().nerf_set_option(k, v))
PREC = int(os.environ.get("PREC", 16))
m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0, precision=PREC)
opt(b"bwd_stage", 2)
for B, n in ((4096, 64), (4096, 192)):
    r = rays(B); z = torch.sort(torch.rand(B, n, device=dev) * 4 + 2, -1).values
    g = torch.randn(B, n, 4, device=dev)
    m.query(r, z, train=True); opt(b"bwd_stage", 0); m.backward(g); opt(b"bwd_stage", 2)
    for bias in ((64, 32, 128) if PREC == 16 else (32, 64, 128, 256, 512)):
        opt(b"dw_unit_bias", bias)
        res = []
        for wgs in ((0, 256, 512, 768, 1024) if PREC == 16 else (0, 384, 512)):
            opt(b"dw_workgroups", wgs)
            res.append(f"{wgs}:{timeit(lambda: m.backward(g)):.3f}")
        print(f"n={n} bias={bias}: dW ms by workgroups  " + "  ".join(res), flush=True)
