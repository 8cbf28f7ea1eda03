"""dW (split bf16) timing probe for the NERF_DWX timing-only builds (tools/ab_one.sh dwxN mlp_s16 -DNERF_DWX=N, or mlp_dww): all
jobs, and ONE job alone on 256 workgroups (256 x 256: 192 stages each; 256 x 64; alpha), for "dw22_variant" 1 (256 x 256 jobs on the
one-wave-per-SIMD kernel) and 0 (every job on the 16-wave kernel)."""
import os, sys, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import _native
from nerf_meets_mlx_amd.models.NeRF import NeRF
dev="cuda"
m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0, precision=22)
g=torch.Generator().manual_seed(0)
B,n=4096,192
o=torch.nn.functional.normalize(torch.randn(B,3,generator=g),dim=-1)*4; d=-o/4+0.25*torch.randn(B,3,generator=g)
rays=torch.cat([o,d,torch.full((B,1),2.0),torch.full((B,1),6.0),torch.nn.functional.normalize(d,dim=-1)],-1).to(dev)
z=torch.sort(torch.rand(B,n,generator=g)*4+2,-1).values.to(dev); dr=(torch.randn(B,n,4,generator=g)*1e-4).to(dev)
opt=lambda k,v: _native.check(_native.lib().nerf_set_option(k,v))
m.query(rays,z,train=True); m.backward(dr)
opt(b"bwd_stage", 2)
def t_dw(reps=8):
    m.backward(dr); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): m.backward(dr)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
tag = os.path.basename(os.environ.get("NERF_HIP_LIB", "shipped"))
for variant in [int(v) for v in (sys.argv[1:] or ["0", "1"])]:
    opt(b"dw22_variant", variant)
    opt(b"dw_job_mask", 0); ta = t_dw()
    opt(b"dw_job_mask", 1 << 1); t1 = t_dw()
    opt(b"dw_job_mask", 1 << 0); t0 = t_dw()
    opt(b"dw_job_mask", 1 << 10); t10 = t_dw()
    print(f"{tag:24s} variant {variant}: all jobs {ta:.3f} ms | 256x256 job alone {t1*1e3:.0f} us ({t1*1e3/192:.2f} us/stage) | 256x64 job {t0*1e3:.0f} us | alpha job {t10*1e3:.0f} us", flush=True)
opt(b"dw_job_mask", 0); opt(b"bwd_stage",0)
