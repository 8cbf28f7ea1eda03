import os, sys, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import _native
from nerf_meets_mlx_amd.models.NeRF import NeRF
dev="cuda"
m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0, precision=22)
g=torch.Generator().manual_seed(0)
opt=lambda k,v: _native.check(_native.lib().nerf_set_option(k,v))
for B,n in ((4096,192),(4096,64)):
    o=torch.nn.functional.normalize(torch.randn(B,3,generator=g),dim=-1)*4; d=-o/4+0.25*torch.randn(B,3,generator=g)
    rays=torch.cat([o,d,torch.full((B,1),2.0),torch.full((B,1),6.0),torch.nn.functional.normalize(d,dim=-1)],-1).to(dev)
    z=torch.sort(torch.rand(B,n,generator=g)*4+2,-1).values.to(dev); dr=(torch.randn(B,n,4,generator=g)*1e-4).to(dev)
    m.query(rays,z,train=True); m.backward(dr)
    opt(b"bwd_stage", 2)
    def t_dw(reps=8):
        m.backward(dr); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): m.backward(dr)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)/reps
    wide = sum(1<<j for j in (1,2,3,4,5,7,8,9)); narrow = sum(1<<j for j in (0,6,10,11,12,13))
    for rep in range(2):
        opt(b"dw_job_mask", 0); ta=t_dw()
        opt(b"dw_job_mask", wide); tw=t_dw()
        opt(b"dw_job_mask", narrow); tn=t_dw()
        print(f"B={B} n={n}: all {ta:.3f} ms | 256x256 jobs (12.9 GB at n=192) {tw:.3f} | other jobs (4.6 GB) {tn:.3f}", flush=True)
    opt(b"dw_job_mask", 0); opt(b"bwd_stage", 0)
