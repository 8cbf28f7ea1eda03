#!/usr/bin/env python3
"""PSNR parity at equal iterations: the HIP Trainer (bf16 MFMA) vs the CPU oracle trainer (fp32 autograd
restatement of entrypoints/__test_nerf.py:200-305) fed the SAME rays, targets and importance uniforms.

    python tools/psnr_parity.py --iters 300 --hw 64 --n-rand 256   (run on the GPU box; a few minutes)

Prints one JSON line per checkpoint with the PSNR of both on held-out views of the synthetic scene.
The oracle is the checker here, never the product.
"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_meets_mlx_amd.dataset import synthetic
from nerf_meets_mlx_amd.engine.trainer import Trainer
from oracle import nerf_oracle as O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--n-rand", type=int, default=256)
    ap.add_argument("--views", type=int, default=12)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--lrate-decay", type=int, default=500)
    ap.add_argument("--seed", type=int, default=-1, help="-1: first seed whose coarse AND fine nets start with sigma > 0")
    ap.add_argument("--no-quirks", action="store_true")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    dev = "cuda"
    H = W = a.hw
    imgs, poses, rposes, hwf, K = synthetic.make_dataset(H, W, a.views + 2, seed=0, device=dev)
    test_imgs, test_poses = imgs[-2:].cpu(), poses[-2:]
    arch = O.NerfArch()
    q = not a.no_quirks
    seed = a.seed
    if seed < 0:
        # The reference feeds raw sigma (no activation, Q9/Q10) into alpha = 1 - exp(-relu(sigma delta)): a network whose
        # initial sigma is negative everywhere has alpha == 0 and an exactly-zero gradient -- it never trains (in the
        # reference too).  A deep ReLU net at init is nearly constant over its inputs, so that is a coin flip per seed;
        # pick the first seed where both nets are alive so that the PSNR comparison is not vacuous.
        probe_pos = (torch.rand(64, 8, 3, generator=torch.Generator().manual_seed(0)) - 0.5) * 3.0
        probe_dir = torch.nn.functional.normalize(torch.randn(64, 3, generator=torch.Generator().manual_seed(1)), dim=-1)
        for seed in range(100):
            ok = True
            for sd in (seed, seed + 1):
                raw = O.run_model(arch, O.init_params(arch, sd), probe_pos, probe_dir, ref_quirks=q)
                ok = ok and float((raw[..., 3] > 0).float().mean()) > 0.95
            if ok:
                break
    print(json.dumps({"seed": seed, "ref_quirks": q}), flush=True)
    tr = Trainer(imgs[:-2], poses[:-2], K, N_rand=a.n_rand, n_depth_samples=64, N_importance=128, seed=seed, device=dev,
                 lrate_decay=a.lrate_decay, ref_quirks=q)
    ot = O.OracleTrainer(arch, 64, 128, seed=seed, lrate_decay=a.lrate_decay, ref_quirks=q)
    assert torch.equal(tr.coarse.params.cpu(), ot.pc.detach())
    g = torch.Generator().manual_seed(123)

    def oracle_psnr():
        with torch.no_grad():
            pc = O.unflatten_params(arch, ot.pc.detach()); pf = O.unflatten_params(arch, ot.pf.detach())
            vals = []
            for img, pose in zip(test_imgs, test_poses):
                u = torch.rand(H * W, 128, generator=torch.Generator().manual_seed(7))
                rgb = O.render(arch, pc, pf, H, W, K, pose[:3, :4], 2.0, 6.0, 64, 128, u, chunk=4096, white_bkgd=True, ref_quirks=q)[0]
                vals.append(float(O.psnr(rgb, img)))
        return float(np.mean(vals))

    def hip_psnr():
        return float(np.mean([tr.psnr(p[:3, :4].numpy(), im) for im, p in zip(test_imgs, test_poses)]))

    t0 = time.time()
    for it in range(1, a.iters + 1):
        rays, target = tr.sample_batch()
        u = torch.rand(a.n_rand, 128, generator=g)
        lh = tr.train_step(rays, target, u.to(dev))
        lo = ot.step(rays[:, 0:3].cpu(), rays[:, 3:6].cpu(), target.cpu(), u)
        if it % a.every == 0 or it == a.iters:
            rec = {"iter": it, "psnr_hip": hip_psnr(), "psnr_oracle": oracle_psnr(),
                   "loss_coarse_hip": float(lh["loss_coarse"]), "loss_coarse_oracle": lo["loss_coarse"],
                   "loss_fine_hip": float(lh["loss_fine"]), "loss_fine_oracle": lo["loss_fine"], "elapsed_s": time.time() - t0}
            rec["delta_db"] = rec["psnr_hip"] - rec["psnr_oracle"]
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
