"""Static split of the split-bf16 weight-gradient jobs: kernel time over the cost model's fixed per-tile part ("dw_unit_bias") at the
bench's fine- and coarse-pass sizes."""
import os, sys, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import _native
from nerf_meets_mlx_amd.models.NeRF import NeRF
dev="cuda"
opt=lambda k,v: _native.check(_native.lib().nerf_set_option(k,v))
if len(sys.argv) > 1: opt(b"dw22_variant", int(sys.argv[1]))     # default: the shipped kernel
for B,n in ((4096,192),(4096,64)):
    m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0, precision=22)
    g=torch.Generator().manual_seed(0)
    o=torch.nn.functional.normalize(torch.randn(B,3,generator=g),dim=-1)*4; d=-o/4+0.25*torch.randn(B,3,generator=g)
    rays=torch.cat([o,d,torch.full((B,1),2.0),torch.full((B,1),6.0),torch.nn.functional.normalize(d,dim=-1)],-1).to(dev)
    z=torch.sort(torch.rand(B,n,generator=g)*4+2,-1).values.to(dev); dr=(torch.randn(B,n,4,generator=g)*1e-4).to(dev)
    m.query(rays,z,train=True); m.backward(dr)
    opt(b"bwd_stage", 2)
    def t_dw(reps=10):
        m.backward(dr); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): m.backward(dr)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)/reps
    for rep in range(2):
        line=[]
        for bias in (0, 8, 16, 24, 32, 48, 64, 96, 128, 192):
            opt(b"dw_unit_bias", bias)
            line.append(f"{bias}:{t_dw():.3f}")
        print(f"B={B} n={n}  " + "  ".join(line))
    opt(b"dw_unit_bias", -1); opt(b"bwd_stage",0)
