"""Headless NeRF training + rendering entrypoint: `mlx_nerf/entrypoints/__test_nerf.py:25-342` without the
matplotlib / mp4 output (IO, out of scope).  Same steps: parse defaults -> load `configs/lego.txt` -> update args
(called with BOTH arguments) -> load Blender data (or the synthetic Lego-like scene when no dataset is
given: there is no Lego data offline) -> white-background composite -> K -> Trainer (= create_NeRF + the hot loop of
:200-305 on the device) -> optional periodic full-frame render (:308-322) -> final render poses (:326-341).
"""
import os
import time
from typing import Optional

import numpy as np
import torch

from .. import config_parser
from ..dataset import synthetic
from ..dataset.dataloader import load_blender_data, post_load_blender_data
from ..engine.runlog import RunLog
from ..engine.trainer import Trainer


def checkpoint_dir(args) -> str:
    """`{basedir}/{expname}` like nerf-pytorch, whose flags the reference keeps (config_parser.py:7-8)."""
    return os.path.join(args.basedir, args.expname or "nerf")


def latest_checkpoint(ckpt_dir: str) -> Optional[str]:
    """Highest-numbered `{iteration:06d}.npz` in ckpt_dir, or None."""
    if not os.path.isdir(ckpt_dir):
        return None
    names = sorted(f for f in os.listdir(ckpt_dir) if f.endswith(".npz") and f[:-4].isdigit())
    return os.path.join(ckpt_dir, names[-1]) if names else None


def main(path_dataset: Optional[str] = None, max_iter: int = 5000, device="cuda", ref_quirks: bool = True,
         hw_synthetic: int = 800, n_train_synthetic: int = 100, render_every: int = 50000, n_render_poses: int = 0,
         log_every: int = 100, seed: int = 4, argv=None, precision: Optional[int] = None, write_log: bool = True):
    """argv: extra command-line flags of `config_parser` (e.g. ["--basedir", d, "--expname", "lego", "--i_weights",
    "1000", "--ft_path", f, "--no_reload"]).  Checkpoint flags (config_parser.py:25-26,75; declared upstream, with
    `models/NeRF.py:122-125` left as TODOs): every `--i_weights` iterations the trainer state goes to
    `{basedir}/{expname}/{it:06d}.npz`; at start `--ft_path` (if given) or, unless `--no_reload`, the newest
    checkpoint of that directory is loaded and training continues from its iteration.  In quirk mode a config file
    forces no_reload like upstream does (config_parser.py:120).
    Run log (`engine/runlog.py`; the reference keeps `loss.item()` of every iteration in a list and plots it at i_render,
    `__test_nerf.py:298-299,314-322`): rank 0 appends to `{basedir}/{expname}/log.jsonl` one `train` record {it, loss_coarse,
    loss_fine, psnr_*, lr, rays_per_s} every `log_every` iterations and one `eval` record {it, psnr, view} per periodic frame
    (PSNR of the rendered training view against its image); `write_log=False` turns the file off.
    precision (ours; default from NERF_PRECISION, else 22): 22 = the reference's float32 tolerance on the 16-bit matrix pipe
    (what bench.py measures), 32 = float32 operands on the fp32 MFMA, 16 = bf16 operands (declared reduced precision)."""
    if precision is None:
        precision = int(os.environ.get("NERF_PRECISION", "22"))
    args = config_parser.config_parser().parse_args(args=list(argv or []))
    if path_dataset is not None:
        configs = config_parser.load_config(None, os.path.join(path_dataset, "configs/lego.txt"))
        args = config_parser.update_NeRF_args(args, configs, ref_quirks=ref_quirks)
        # upstream never forwards half_res (Q3): training is at native resolution in quirk mode
        half = False if ref_quirks else bool(args.half_res)
        images, poses, render_poses, hwf, i_split = load_blender_data(os.path.join(path_dataset, configs["datadir"]), half)
        i_train, i_val, i_test, near, far, images = post_load_blender_data(i_split, images, bool(args.white_bkgd))
        images, poses = torch.from_numpy(images), torch.from_numpy(poses)
        H, W, focal = int(hwf[0]), int(hwf[1]), hwf[2]
        K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
        train_imgs, train_poses = images[i_train], poses[i_train]
    else:
        args.n_depth_samples, args.N_importance, args.N_rand, args.lrate_decay = 64, 128, 1024, 500   # lego.txt values
        args.white_bkgd, args.use_viewdirs = True, True
        train_imgs, train_poses, render_poses, hwf, K = synthetic.make_dataset(hw_synthetic, hw_synthetic,
                                                                               n_train_synthetic, seed=0, device=device)
        near, far = 2.0, 6.0
    tr = Trainer(train_imgs, train_poses, K, near=near, far=far, N_rand=args.N_rand, n_depth_samples=args.n_depth_samples,
                 N_importance=args.N_importance, lrate=args.lrate, lrate_decay=args.lrate_decay,
                 white_bkgd=bool(args.white_bkgd), ref_quirks=ref_quirks, seed=seed, device=device, chunk=args.chunk,
                 precision=precision)
    ckpt_dir = checkpoint_dir(args)
    resume = args.ft_path if args.ft_path else (None if args.no_reload else latest_checkpoint(ckpt_dir))
    if resume is not None:
        tr.load(resume)
    losses, frames, saved = [], [], []
    log = RunLog(ckpt_dir, rank=tr.rank, world=tr.world, enabled=write_log)
    log.run(n_rand=int(args.N_rand), n_depth_samples=int(args.n_depth_samples), n_importance=int(args.N_importance), precision=int(precision),
            lrate=float(args.lrate), lrate_decay=int(args.lrate_decay), seed=int(seed), resumed_from=resume, start_it=int(tr.it),
            max_iter=int(max_iter), hw=[int(tr.H), int(tr.W)], dataset=path_dataset or "synthetic")
    log.mark(tr.it)
    for i in range(tr.it + 1, max_iter + 1):
        out = tr.train_step()
        if args.i_weights and i % args.i_weights == 0:
            tr.sync_optimizer_state()        # collective (a no-op for replicated Adam): every rank, then rank 0 writes
            if tr.rank == 0:
                os.makedirs(ckpt_dir, exist_ok=True)
                saved.append(tr.save(os.path.join(ckpt_dir, f"{i:06d}.npz")))
        if i % log_every == 0 or i == max_iter:
            lc = float(out["loss_coarse"])                    # (reaching the host waits for the device: the rate below is honest)
            lf = float(out["loss_fine"]) if "loss_fine" in out else None
            losses.append((i, lc, 0.0 if lf is None else lf))
            log.train(i, lc, lf, tr._opt.learning_rate, args.N_rand)
        if render_every and i % render_every == 0:
            view = len(train_poses) // 2
            t0 = time.perf_counter()
            frame = tr.render_frame(train_poses[view], shard=False)
            mse = float(torch.mean((frame - tr.images[view]) ** 2))
            frames.append(frame.clamp(0, 1).cpu())
            log.eval(i, None if not mse > 0 else -10.0 * np.log10(mse), view, time.perf_counter() - t0)
    video = [tr.render_frame(p, shard=False).clamp(0, 1).cpu() for p in render_poses[:n_render_poses]]
    return {"trainer": tr, "losses": losses, "frames": frames, "video": video, "checkpoints": saved, "resumed_from": resume,
            "log": log.path if log.active else None}
