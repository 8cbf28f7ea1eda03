"""Precision-22 training kernels at the bench's fine-pass size (4096 x 192 samples): forward-with-stores, chain, dW, back to back
(nerf_set_option "bwd_stage"); with NERF_HIP_LIB an A/B probe."""
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
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
m.query(rays,z,train=True); m.backward(dr)
for rep in range(2):
    tf = timeit(lambda: m.query(rays,z,train=True))
    opt(b"bwd_stage",1); tc = timeit(lambda: m.backward(dr))
    opt(b"bwd_stage",2); tw = timeit(lambda: m.backward(dr))
    opt(b"bwd_stage",0)
    ti = timeit(lambda: m.query(rays,z))
    print(f"{os.path.basename(os.environ.get('NERF_HIP_LIB','shipped')):28s} forward+stores {tf:.3f}  chain {tc:.3f}  dW {tw:.3f}  inference {ti:.3f} ms")
