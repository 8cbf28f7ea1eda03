"""Per-job times of the split-bf16 weight-gradient kernel at the bench's fine-pass size ("dw_job_mask": one job at a time over all
256 workgroups), their sum against the all-jobs launch, and the static split the launcher chose ("dw_unit_bias")."""
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
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1     # "dw22_variant": 1 the default (two kernels), 0 every job on the 16-wave kernel
opt(b"dw22_variant", variant)
m.query(rays,z,train=True); m.backward(dr)
opt(b"bwd_stage", 2)
def t_dw(reps=6):
    m.backward(dr); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): m.backward(dr)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
names=["pos0","pos1","pos2","pos3","pos4","pos5|H4","pos5|PE","pos6","pos7","feature","alpha","dir0|feat","dir0|dirPE","rgb"]
frags=[(16,4),(16,16),(16,16),(16,16),(16,16),(16,16),(16,4),(16,16),(16,16),(16,16),(1,16),(8,16),(8,2),(1,8)]
tot=0.0
print(f"variant {variant}; all jobs: {t_dw():.3f} ms")
for j,(nm,(nf,kf)) in enumerate(zip(names,frags)):
    opt(b"dw_job_mask", 1<<j)
    t=t_dw(4)
    gb = 786432/32 * 2*(nf+kf) * 1024 / 1e9
    print(f"  job {j:2d} {nm:10s} nf {nf:2d} kf {kf:2d}: {t:.3f} ms alone on 256 workgroups  ({gb:.2f} GB -> {gb/t:.2f} TB/s)")
    tot+=t
opt(b"dw_job_mask", 0)
print(f"sum of single-job launches: {tot:.3f} ms")
for bias in (32, 64, 128, 256, 512):
    opt(b"dw_unit_bias", bias)
    print(f"dw_unit_bias {bias}: {t_dw():.3f} ms")
opt(b"dw_unit_bias", -1)
for wgs in (256, 512):
    opt(b"dw_workgroups", wgs)
    print(f"dw_workgroups {wgs}: {t_dw():.3f} ms")
opt(b"dw_workgroups", 0); opt(b"bwd_stage",0)
