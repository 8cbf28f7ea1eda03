"""Do cache / TLB state or a hot chip slow the two weight-gradient launches?  Each timed behind nothing, behind 48 GB of unrelated
traffic, and behind 13 ms of dense bf16 GEMM (round 5: neither does)."""
import os, sys, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import _native
from nerf_meets_mlx_amd.models.NeRF import NeRF
dev="cuda"
m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0, precision=22)
g=torch.Generator().manual_seed(0)
opt=lambda k,v: _native.check(_native.lib().nerf_set_option(k,v))
B,n=4096,192
o=torch.nn.functional.normalize(torch.randn(B,3,generator=g),dim=-1)*4; d=-o/4+0.25*torch.randn(B,3,generator=g)
rays=torch.cat([o,d,torch.full((B,1),2.0),torch.full((B,1),6.0),torch.nn.functional.normalize(d,dim=-1)],-1).to(dev)
z=torch.sort(torch.rand(B,n,generator=g)*4+2,-1).values.to(dev); dr=(torch.randn(B,n,4,generator=g)*1e-4).to(dev)
m.query(rays,z,train=True); m.backward(dr)
opt(b"bwd_stage", 2)
junk = torch.empty(6 * 1024**3 // 4, device=dev)
hot = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
wide = sum(1<<j for j in (1,2,3,4,5,7,8,9)); narrow = sum(1<<j for j in (0,6,10,11,12,13))
def t(mask, pre):
    opt(b"dw_job_mask", mask)
    ts=[]
    for _ in range(6):
        if pre == "thrash": junk.add_(1.0)                 # 48 GB of other traffic: caches and TLBs see other pages
        elif pre == "gemm":
            for _ in range(12): hot @ hot                  # ~13 ms of dense bf16 GEMM: a hot, power-limited chip, no memory thrash
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); m.backward(dr); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts)//2]
for pre in ("none", "thrash", "gemm", "none"):
    print(f"before each launch: {pre:7s} | 256x256 jobs {t(wide, pre):.3f} ms | other jobs {t(narrow, pre):.3f} ms", flush=True)
opt(b"dw_job_mask", 0); opt(b"bwd_stage", 0)
