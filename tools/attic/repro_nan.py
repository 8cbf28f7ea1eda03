#!/usr/bin/env python3
"""Find where a training run of a given (seed, precision) first produces a non-finite loss / gradient / parameter
(round 4: seed 71's fp32 arm of the converged-regime ensemble reported NaN PSNR from its first checkpoint on).
    python tools/repro_nan.py --seed 71 --precision 32 --iters 1000"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_meets_mlx_amd.dataset import synthetic                      # noqa: E402
from nerf_meets_mlx_amd.engine.trainer import Trainer                 # noqa: E402


def dissect(tr, sd, rays, target, u):
    """Replay the failing iteration from the state before it, piece by piece."""
    from nerf_meets_mlx_amd import sampling
    from nerf_meets_mlx_amd.rendering import render
    tr.load_state_dict(sd)
    st = lambda name, t: print(f"    {name}: finite {bool(torch.isfinite(t).all())} nan {int(torch.isnan(t).sum())} inf {int(torch.isinf(t).sum())} "
                               f"max|finite| {float(t[torch.isfinite(t)].abs().max()) if torch.isfinite(t).any() else float('nan'):.4e}", flush=True)
    z = sampling.sample_coarse(rays, 64)
    for name, net, zz, white in (("coarse", tr.coarse, z, True),):
        st(name + " params", net.params)
        raw = net.query(rays, zz, train=True)
        st(name + " raw rgb", raw[..., :3]); st(name + " raw sigma", raw[..., 3])
        loss, d_raw, rgb = render.composite_mse_backward(raw, zz, rays, target, white, need_rgb=True)
        st(name + " rgb_map", rgb); st(name + " loss", loss.reshape(1)); st(name + " d_raw rgb", d_raw[..., :3]); st(name + " d_raw sigma", d_raw[..., 3])
        bad = ~torch.isfinite(d_raw).all(-1)
        if bad.any():
            b, i = [int(x[0]) for x in torch.nonzero(bad, as_tuple=True)]
            print(f"    first bad d_raw at ray {b} sample {i}: raw {raw[b, i].tolist()} z {float(zz[b, i]):.6f} d_raw {d_raw[b, i].tolist()}")
            print(f"    that ray: sigma min {float(raw[b, :, 3].min()):.4e} max {float(raw[b, :, 3].max()):.4e}; |d| {float(rays[b, 3:6].norm()):.4f}; rgb_map {rgb[b].tolist()} target {target[b].tolist()}")
        g = net.backward(d_raw)
        st(name + " grads", g)
    # the fine half of the iteration (__test_nerf.py:270-292): updated coarse net -> weights -> importance samples -> fine step
    tr.load_state_dict(sd)
    tr._step_net(tr.coarse, rays, z, target, True)
    raw = tr.coarse.query(rays, z)
    _, _, _, w, _ = render.composite(raw, z, rays, 0.0, True)
    st("coarse weights (updated net)", w)
    z_imp, z_fine = sampling.importance_sample(z, w, 128, u=u)
    st("z_imp", z_imp); st("z_fine", z_fine)
    bad = ~torch.isfinite(z_fine).all(-1)
    if bad.any():
        b = int(torch.nonzero(bad)[0])
        print(f"    first bad ray {b}: weights sum {float(w[b].sum()):.4e} min {float(w[b].min()):.4e} max {float(w[b].max()):.4e}; "
              f"z_imp[:4] {z_imp[b, :4].tolist()} u[:4] {u[b, :4].tolist()}")
    fine = tr._fine
    st("fine params", fine.params)
    rawf = fine.query(rays, z_fine, train=True)
    st("fine raw rgb", rawf[..., :3]); st("fine raw sigma", rawf[..., 3])
    loss, d_raw, rgb = render.composite_mse_backward(rawf, z_fine, rays, target, False, need_rgb=True)
    st("fine rgb_map", rgb); st("fine loss", loss.reshape(1)); st("fine d_raw", d_raw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle", action="store_true", help="run the torch fp32 OracleTrainer on the device on the same batches")
    ap.add_argument("--dissect", action="store_true", help="replay the first non-finite iteration piece by piece")
    ap.add_argument("--seed", type=int, default=71)
    ap.add_argument("--precision", type=int, default=32)
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--hw", type=int, default=800)
    ap.add_argument("--n-rand", type=int, default=1024)
    ap.add_argument("--every", type=int, default=1)
    a = ap.parse_args()
    dev = "cuda"
    imgs, poses, _, _, K = synthetic.make_dataset(a.hw, a.hw, 14, seed=0, device=dev)
    tr = Trainer(imgs[:-2], poses[:-2], K, N_rand=a.n_rand, n_depth_samples=64, N_importance=128, seed=a.seed, device=dev,
                 lrate_decay=500, precision=a.precision)
    fin = lambda t: bool(torch.isfinite(t).all())
    ot = None
    if a.oracle:
        from oracle import nerf_oracle as O
        torch.backends.cuda.matmul.allow_tf32 = False
        ot = O.OracleTrainer(O.NerfArch(), 64, 128, seed=a.seed, lrate_decay=500, ref_quirks=True, device=dev)
    for it in range(1, a.iters + 1):
        sd = tr.state_dict() if a.dissect else None
        rays, target = tr.sample_batch()
        u = tr.train_uniforms(rays.shape[0])
        out = tr.train_step(rays, target, u)
        if ot is not None:
            lo = ot.step(rays[:, 0:3], rays[:, 3:6], target, u)
            okc = all(map(lambda v: v == v and abs(v) != float("inf"), (float(lo["loss_coarse"]), float(lo["loss_fine"])))) and fin(ot.pc) and fin(ot.pf)
            if not okc or it % 100 == 0:
                print(f"it {it} ORACLE (torch fp32 on the device): loss {float(lo['loss_coarse']):.5f} / {float(lo['loss_fine']):.5f} params finite {fin(ot.pc)} / {fin(ot.pf)}", flush=True)
        if it % a.every == 0 or it == a.iters:
            ok = {"loss_c": fin(out["loss_coarse"]), "loss_f": fin(out["loss_fine"]), "pc": fin(tr.coarse.params), "pf": fin(tr.fine.params),
                  "gc": fin(tr.coarse.grads), "gf": fin(tr.fine.grads)}
            if not all(ok.values()) or it % 100 == 0:
                print(f"it {it} precision {a.precision}: loss {float(out['loss_coarse']):.5f} / {float(out['loss_fine']):.5f} "
                      f"|p| {float(tr.coarse.params.abs().max()):.3f} / {float(tr.fine.params.abs().max()):.3f} "
                      f"|g| {float(tr.coarse.grads.abs().max()):.3e} / {float(tr.fine.grads.abs().max()):.3e} finite {ok}", flush=True)
            if not all(ok.values()):
                print("first non-finite at iteration", it)
                if a.dissect:
                    dissect(tr, sd, rays, target, u)
                return 1
    print("all finite")
    return 0


if __name__ == "__main__":
    sys.exit(main())
