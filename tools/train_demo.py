#!/usr/bin/env python3
"""End-to-end convergence run of the HIP trainers on the synthetic Lego-like scene (no oracle involved): PSNR on
held-out views over training, rays/s of the loop as it really runs (pixel sampling, ray generation, both network steps,
Adam, LR schedule).

    python tools/train_demo.py --config nerf --hw 200 --iters 3000          (a minute on the GPU box)
    python tools/train_demo.py --config ngp  --hw 200 --iters 3000
"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_meets_mlx_amd.dataset import synthetic
from nerf_meets_mlx_amd.engine.trainer import Trainer
from nerf_meets_mlx_amd.engine.ngp import NGPTrainer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["nerf", "ngp"], default="nerf")
    ap.add_argument("--hw", type=int, default=200)
    ap.add_argument("--views", type=int, default=24)
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--every", type=int, default=500)
    ap.add_argument("--n-rand", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=-1)
    ap.add_argument("--precision", type=int, default=16, choices=[16, 22, 32], help="8 x 256 model only (config nerf)")
    a = ap.parse_args()
    dev = "cuda"
    H = W = a.hw
    imgs, poses, rposes, hwf, K = synthetic.make_dataset(H, W, a.views + 2, seed=0, device=dev)
    test_imgs, test_poses = imgs[-2:], poses[-2:]
    if a.config == "nerf":
        seed = 4 if a.seed < 0 else a.seed                  # coarse 4 / fine 5 start alive
        tr = Trainer(imgs[:-2], poses[:-2], K, N_rand=a.n_rand, n_depth_samples=64, N_importance=128, seed=seed, device=dev,
                     precision=a.precision)
    else:
        seed = 7 if a.seed < 0 else a.seed
        tr = NGPTrainer(imgs[:-2], poses[:-2], K, N_rand=a.n_rand, n_depth_samples=64, seed=seed, device=dev)
    print(json.dumps({"config": a.config, "hw": H, "views": a.views, "n_rand": a.n_rand, "seed": seed,
                      "precision": a.precision if a.config == "nerf" else 16}), flush=True)
    psnr = lambda: float(np.mean([tr.psnr(p[:3, :4].cpu().numpy(), im) for im, p in zip(test_imgs, test_poses)]))
    print(json.dumps({"iter": 0, "psnr": psnr()}), flush=True)
    torch.cuda.synchronize()
    t_train = 0.0
    for it in range(1, a.iters + 1):
        if (it - 1) % a.every == 0:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        out = tr.train_step()
        if it % a.every == 0 or it == a.iters:
            torch.cuda.synchronize(); dt = time.perf_counter() - t0; t_train += dt
            rec = {"iter": it, "psnr": psnr(), "loss_coarse": float(out["loss_coarse"]),
                   "train_rays_per_s": a.n_rand * (a.every if it % a.every == 0 else it % a.every) / dt}
            if "loss_fine" in out:
                rec["loss_fine"] = float(out["loss_fine"])
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
