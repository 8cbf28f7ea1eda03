"""Split-bf16 backward at the bench's fine-pass size (4096 x 192 samples), chain and dW timed separately (nerf_set_option
"bwd_stage"), the weight gradients in both settings of "dw22_variant" (1 = 256 x 256 jobs on the one-wave-per-SIMD kernel + the rest
on the 16-wave kernel, the default; 0 = every job on the 16-wave kernel); with NERF_HIP_LIB pointing at a timing-only build an A/B probe."""
import os, sys, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import _native
from nerf_meets_mlx_amd.models.NeRF import NeRF
dev="cuda"
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 22          # 22: split-bf16 stores ("dw22_variant"); 16: bf16 stores ("dw16_variant")
key = b"dw22_variant" if prec == 22 else b"dw16_variant"
m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0, precision=prec)
g=torch.Generator().manual_seed(0)
B,n=4096,192
o=torch.nn.functional.normalize(torch.randn(B,3,generator=g),dim=-1)*4; d=-o/4+0.25*torch.randn(B,3,generator=g)
rays=torch.cat([o,d,torch.full((B,1),2.0),torch.full((B,1),6.0),torch.nn.functional.normalize(d,dim=-1)],-1).to(dev)
z=torch.sort(torch.rand(B,n,generator=g)*4+2,-1).values.to(dev); dr=(torch.randn(B,n,4,generator=g)*1e-4).to(dev)
opt=lambda k,v: _native.check(_native.lib().nerf_set_option(k,v))
ref = None
for variant in (0, 1, 0, 1):
    opt(key, variant)
    opt(b"bwd_stage", 0)
    m.query(rays,z,train=True); g_ = m.backward(dr).clone()
    if ref is None: ref = g_
    same = torch.equal(ref, g_)
    rel = float((g_ - ref).norm() / ref.norm())
    for stage,name in ((1,"chain"),(2,"dW")):
        opt(b"bwd_stage",stage)
        m.backward(dr); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): m.backward(dr)
        e1.record(); torch.cuda.synchronize()
        print(f"variant {variant} {name}: {e0.elapsed_time(e1)/10:.3f} ms" + (f"   gradient vs variant 0: bit-equal {same}, rel-L2 {rel:.1e}" if stage == 2 else ""), flush=True)
opt(b"bwd_stage",0); opt(key, 1)
