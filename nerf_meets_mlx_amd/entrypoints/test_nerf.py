"""Headless NeRF training + rendering entrypoint: `mlx_nerf/entrypoints/__test_nerf.py:25-342` without the
matplotlib / mp4 output (IO, out of scope).  Same steps: parse defaults -> load `configs/lego.txt` -> update args
(called with BOTH arguments) -> load Blender data (or the synthetic Lego-like scene when no dataset is
given: there is no Lego data offline) -> white-background composite -> K -> Trainer (= create_NeRF + the hot loop of
:200-305 on the device) -> optional periodic full-frame render (:308-322) -> final render poses (:326-341).
"""
import os
from typing import Optional

import numpy as np
import torch

from .. import config_parser
from ..dataset import synthetic
from ..dataset.dataloader import load_blender_data, post_load_blender_data
from ..engine.trainer import Trainer


def main(path_dataset: Optional[str] = None, max_iter: int = 5000, device="cuda", ref_quirks: bool = True,
         hw_synthetic: int = 800, n_train_synthetic: int = 100, render_every: int = 50000, n_render_poses: int = 0,
         log_every: int = 100, seed: int = 4):
    args = config_parser.config_parser().parse_args(args=[])
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
                 white_bkgd=bool(args.white_bkgd), ref_quirks=ref_quirks, seed=seed, device=device, chunk=args.chunk)
    losses, frames = [], []
    for i in range(1, max_iter + 1):
        out = tr.train_step()
        if i % log_every == 0 or i == max_iter:
            losses.append((i, float(out["loss_coarse"]), float(out.get("loss_fine", torch.zeros(1)))))
        if render_every and i % render_every == 0:
            frames.append(tr.render_frame(train_poses[len(train_poses) // 2], shard=False).clamp(0, 1).cpu())
    video = [tr.render_frame(p, shard=False).clamp(0, 1).cpu() for p in render_poses[:n_render_poses]]
    return {"trainer": tr, "losses": losses, "frames": frames, "video": video}
