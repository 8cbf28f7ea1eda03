"""Blender-synthetic reader (`mlx_nerf/dataset/dataloader.py:20-111`): host IO only.

`transforms_{train,val,test}.json` + RGBA PNGs -> float32 images, 4x4 poses, the 160 spherical render
poses, [H, W, focal] and the split indices.  `imageio` is not required (PIL reads the PNGs).
Differences from upstream, both flagged in SURVEY Q18: `half_res` is honoured for a real bool and the
half-resolution resize runs on the uint8 image (upstream calls PIL on a float RGBA array, which PIL
rejects) with LANCZOS like upstream.
"""
import json
import os

import numpy as np
import torch
from PIL import Image

from ..ops import pose


def load_blender_data(basedir, half_res: bool = False, testskip: int = 1):
    splits = ["train", "val", "test"]
    metas = {}
    for s in splits:
        with open(os.path.join(basedir, f"transforms_{s}.json"), "r") as fp:
            metas[s] = json.load(fp)
    all_imgs, all_poses, counts = [], [], [0]
    for s in splits:
        meta = metas[s]
        skip = 1 if (s == "train" or testskip == 0) else testskip
        imgs, poses = [], []
        for frame in meta["frames"][::skip]:
            im = Image.open(os.path.join(basedir, frame["file_path"] + ".png")).convert("RGBA")
            if half_res is True:
                im = im.resize((im.size[0] // 2, im.size[1] // 2), Image.Resampling.LANCZOS)
            imgs.append(np.asarray(im))
            poses.append(np.array(frame["transform_matrix"]))
        all_imgs.append((np.array(imgs) / 255.0).astype(np.float32))          # keep all 4 channels
        all_poses.append(np.array(poses).astype(np.float32))
        counts.append(counts[-1] + all_imgs[-1].shape[0])
    i_split = [np.arange(counts[i], counts[i + 1]) for i in range(len(splits))]
    imgs = np.concatenate(all_imgs, 0)
    poses = np.concatenate(all_poses, 0)
    H, W = imgs[0].shape[:2]
    camera_angle_x = float(meta["camera_angle_x"])
    full_w = W * 2 if half_res is True else W
    focal = 0.5 * full_w / np.tan(0.5 * camera_angle_x)
    if half_res is True:
        focal = focal / 2.0
    render_poses = torch.stack([pose.pose_spherical(theta=a, phi=-30.0, radius=4.0)
                                for a in np.linspace(-180, 180, 160 + 1)[:-1]], dim=0)
    return imgs, poses, render_poses, [H, W, focal], i_split


def post_load_blender_data(i_split, images, is_white_bkgd):
    """`dataloader.py:95-111`: near = 2, far = 6, optional white-background composite of the RGBA images."""
    i_train, i_val, i_test = i_split
    if is_white_bkgd:
        images = images[..., :3] * images[..., -1:] + (1.0 - images[..., -1:])
    else:
        images = images[..., :3]
    return i_train, i_val, i_test, 2.0, 6.0, images
