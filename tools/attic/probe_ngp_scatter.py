#!/usr/bin/env python3
"""configs[4] table-gradient scatter by level and by mode (float atomics vs deterministic int64 fixed-point atomics),
and the training step in both modes.

    python tools/probe_ngp_scatter.py > profiles/r03_ngp_scatter.csv
"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from nerf_meets_mlx_amd import _native as N
from nerf_meets_mlx_amd import sampling
from nerf_meets_mlx_amd.dataset import synthetic
from nerf_meets_mlx_amd.engine.ngp import NGPTrainer
from nerf_meets_mlx_amd.rendering import render

dev = "cuda"


def timeit(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


H = W = 800
imgs, poses, rposes, hwf, K = synthetic.make_dataset(H, W, 2, seed=0, device=dev)
print("what,mode,levels,ms,atomics_G_per_s")
# render chunk (32768 adjacent pixels x 64 samples) through the fused inference query: tile order A/B
_tr = NGPTrainer(imgs, poses, K, N_rand=4096, n_depth_samples=64, seed=4, device=dev)
for _ in range(20):
    _tr.train_step()
from nerf_meets_mlx_amd.rendering import ray as _ray
_idx = torch.arange(32768, device=dev, dtype=torch.int64) + 200 * 800
_rr = _ray.gen_rays(H, W, K, rposes[40][:3, :4], 2.0, 6.0, _idx)
_z = sampling.sample_coarse(_rr, 64)
for _mode in (0, 1):
    N.check(N.lib().nerf_set_option(b"ngp_ray_major", _mode))
    _t = timeit(lambda: _tr.field.query(_rr, _z))
    print(f"fused_inference_query_32768x64,{'ray_major (32 rays x 1 depth)' if _mode else 'sample_major (1 ray x 32 depths)'},all,{_t:.4f},")
N.check(N.lib().nerf_set_option(b"ngp_ray_major", 1))
del _tr
for det in (False, True):
    tr = NGPTrainer(imgs, poses, K, N_rand=4096, n_depth_samples=64, seed=4, device=dev, deterministic=det)
    for _ in range(20):
        tr.train_step()                                   # a trained-in state: realistic d_x
    rays, target = tr.sample_batch()
    f, e = tr.field, tr.field.enc
    z = sampling.sample_coarse(rays, 64)
    raw = f.query(rays, z, train=True)
    _, d_raw, _ = render.composite_mse_backward(raw, z, rays, target, True)
    _, d_x = f.mlp.backward(d_raw, need_input_grad=True)
    M = z.numel()
    mode = "fixed_point_int64" if det else "float_atomics"

    def scatter(lo, hi):
        N.check(N.lib().nerf_hashgrid_backward_rays_ex(N.ptr(rays), N.ptr(z), z.shape[0], z.shape[1], N.ptr(d_x), e.n_levels,
                                                       e.log2_hashmap_size, e.n_features_per_level, e._res_c, f.pos_scale,
                                                       f.pos_offset, lo, hi, int(det), N.ptr(e.grad), N.stream()))
    for comb in (0, 64):
        N.check(N.lib().nerf_set_option(b"hash_combine_max_res", comb))
        t = timeit(lambda: scatter(0, 16))
        print(f"scatter,{mode} combine<={comb},0-15,{t:.4f},{M * 256 / t / 1e6:.2f}")
        for l in range(6):
            t = timeit(lambda: scatter(l, l + 1))
            print(f"scatter,{mode} combine<={comb},{l} (res {e.scaled_res[l]}),{t:.4f},{M * 16 / t / 1e6:.2f}")
    t = timeit(lambda: scatter(0, 16))
    print(f"scatter,{mode},0-15,{t:.4f},{M * 256 / t / 1e6:.2f}")
    for lo in range(0, 16, 4):
        t = timeit(lambda: scatter(lo, lo + 4))
        print(f"scatter,{mode},{lo}-{lo + 3},{t:.4f},{M * 64 / t / 1e6:.2f}")
    for l in range(16):
        t = timeit(lambda: scatter(l, l + 1))
        print(f"scatter,{mode},{l} (res {e.scaled_res[l]}),{t:.4f},{M * 16 / t / 1e6:.2f}")
    e.grad.zero_()
    t = timeit(lambda: tr.opt.update(f.table, e.grad.view(-1), zero_grads=True))
    print(f"adam_tables_zeroing,{mode},all,{t:.4f},")
    t = timeit(lambda: tr.train_step(rays, target), it=20)
    print(f"train_step_4096_rays,{mode},all,{t:.4f},")
