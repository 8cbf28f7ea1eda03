"""Quick on-box probe: kernel timings of the MLP variants (not part of the product)."""
import sys, time, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import _native, sampling
from nerf_meets_mlx_amd.models.NeRF import NeRF
from nerf_meets_mlx_amd.rendering import render
dev = "cuda"
m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0)
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
variants = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
for variant in variants:
    _native.check(_native.lib().nerf_set_option(b"mlp_variant", variant))
    for B, n in ((1024, 64), (4096, 64), (4096, 192), (32768, 64), (32768, 192)):
        r = rays(B); z = sampling.sample_coarse(r, n) if n == 64 else torch.sort(torch.rand(B, n, device=dev) * 4 + 2, -1).values
        ms = timeit(lambda: m.query(r, z))
        fl = 2 * 593408 * B * n
        print(f"variant {variant} fwd  B={B:6d} n={n:3d}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)
    for B, n in ((1024, 64), (1024, 192), (4096, 192)):
        r = rays(B); z = torch.sort(torch.rand(B, n, device=dev) * 4 + 2, -1).values
        g = torch.randn(B, n, 4, device=dev)
        msf = timeit(lambda: m.query(r, z, train=True))
        m.query(r, z, train=True)
        msb = timeit(lambda: m.backward(g))
        fl = 2 * 593408 * B * n
        print(f"variant {variant} train B={B:6d} n={n:3d}  fwd {msf:8.3f} ms ({fl / msf / 1e9:7.1f} TF)  bwd+dW {msb:8.3f} ms ({2 * fl / msb / 1e9:7.1f} TF)", flush=True)
