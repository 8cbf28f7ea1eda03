"""Training / rendering engine.  The reference's `mlx_nerf/engine/trainer.py` is an empty file;
its de-facto engine is the loop inlined in `mlx_nerf/entrypoints/__test_nerf.py:200-341`, whose
per-iteration ordering this class reproduces on the device:

    pick image (:202) -> pick N_rand pixels w/o replacement (:229) -> rays + targets (:208-236)
    -> coarse step: render_rays -> MSE -> grads -> Adam (:240 / :129-135)
    -> re-render coarse with the UPDATED net, no grad (:270)
    -> inverse-CDF importance samples + sort (:278-288)
    -> fine step (white_bkgd=False in quirk mode, Q8) -> same Adam object (:292 / :139-145)
    -> lr_i = lrate * 0.1 ** (i / (lrate_decay*1000)) (:302-305)

Differences from the reference are mechanical only: ray generation, pixel selection and the
importance sampler run as HIP kernels instead of numpy / torch-CPU, nothing leaves the device, and
with world_size > 1 each rank draws its own rays and the flat gradients are summed with ONE RCCL
all-reduce per network step (mean over ranks).
"""
from typing import Dict, Optional

import numpy as np
import torch

from .. import parallel, sampling
from ..models.NeRF import Adam, NeRF
from ..ops import index
from ..ops.metric import mse_loss_grad
from ..rendering import ray, render


class Trainer:
    def __init__(self, images: torch.Tensor, poses: torch.Tensor, K, near: float = 2.0, far: float = 6.0,
                 N_rand: int = 1024, n_depth_samples: int = 64, N_importance: int = 128, lrate: float = 5e-4,
                 lrate_decay: int = 500, white_bkgd: bool = True, ref_quirks: bool = True, seed: int = 0,
                 device="cuda", chunk: int = 1024 * 32):
        self.device = torch.device(device)
        self.images = images.to(self.device, torch.float32).contiguous()      # [N,H,W,3], white-composited
        self.poses = poses.float().cpu()
        self.K = np.asarray(K, dtype=np.float64)
        self.H, self.W = int(images.shape[1]), int(images.shape[2])
        self.near, self.far = near, far
        self.N_rand, self.n, self.N = N_rand, n_depth_samples, N_importance
        self.lrate, self.lrate_decay = lrate, lrate_decay
        self.white_bkgd, self.q, self.chunk = white_bkgd, ref_quirks, chunk
        self.rank, self.world = parallel.world()
        self.seed = seed
        self.rng = np.random.default_rng(parallel.rank_seed(seed, self.rank, 1))          # image choice
        mk = lambda s: NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True,
                            device=self.device, seed=s)
        self.coarse = mk(seed)                                  # identical initial weights on every rank
        self.fine = mk(seed + 1) if N_importance > 0 else None
        self.opt = Adam(lrate, betas=(0.9, 0.999), shared_state=ref_quirks)
        self.it = 0
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(parallel.rank_seed(seed, self.rank, 2))

    # ---------------------------------------------------------------- one network step
    def _step_net(self, model: NeRF, rays, z, target, white: bool):
        raw = model.query(rays, z, ref_quirks=self.q, train=True)
        rgb, _, _, weights, _ = render.composite(raw, z, rays, 0.0, white)
        loss, d_rgb = mse_loss_grad(rgb, target)
        d_raw = render.composite_backward(raw, z, rays, d_rgb, white)
        grads = model.backward(d_raw)
        parallel.allreduce_sum_(grads)                                         # C1: the only collective
        self.opt.update(model, grads, grad_scale=1.0 / self.world)
        return loss

    def sample_batch(self, pixel_idx: Optional[torch.Tensor] = None, img_i: Optional[int] = None):
        """(rays [B,11], target [B,3]); by default one random image, N_rand distinct pixels."""
        if img_i is None:
            img_i = int(self.rng.integers(0, self.images.shape[0]))            # np.random.choice(i_train)
        if pixel_idx is None:
            pixel_idx = index.pixel_permutation(self.N_rand, self.H * self.W,
                                                parallel.rank_seed(self.seed, self.rank, 3) + self.it, 0, self.device)
        rays = ray.gen_rays(self.H, self.W, self.K, self.poses[img_i, :3, :4], self.near, self.far, pixel_idx)
        target = index.gather_rows(self.images[img_i].reshape(-1, 3), pixel_idx)
        return rays, target

    def train_step(self, rays=None, target=None, u=None) -> Dict[str, torch.Tensor]:
        """One iteration of `__test_nerf.py:200-305`.  Losses are device scalars (no host sync)."""
        if rays is None:
            rays, target = self.sample_batch()
        self.opt.learning_rate = self.lrate * (0.1 ** (self.it / (self.lrate_decay * 1000)))   # set after iter it-1
        z = sampling.sample_coarse(rays, self.n)                               # perturb = 0 in quirk mode (Q5)
        out = {"loss_coarse": self._step_net(self.coarse, rays, z, target, self.white_bkgd)}
        if self.fine is not None:
            raw = self.coarse.query(rays, z, ref_quirks=self.q)                # updated coarse net, no grad (:270)
            _, _, _, weights, _ = render.composite(raw, z, rays, 0.0, self.white_bkgd)
            if u is None:
                u = torch.rand(rays.shape[0], self.N, device=self.device, generator=self.gen)
            _, z_fine = sampling.importance_sample(z, weights, self.N, u=u)
            white_fine = self.white_bkgd if not self.q else False               # Q8
            out["loss_fine"] = self._step_net(self.fine, rays, z_fine, target, white_fine)
        self.it += 1
        return out

    # ---------------------------------------------------------------- rendering
    def render_rays(self, rays: torch.Tensor, u=None):
        """render_rays_eval on packed rays, chunked like batchify_rays (`chunk` rays per call of the fused
        renderer `nerf_render_rays_fused`); returns rgb [B,3]."""
        outs = []
        for s in range(0, rays.shape[0], self.chunk):
            r = rays[s:s + self.chunk]
            uu = None if u is None else u[s:s + self.chunk]
            if uu is None and self.N > 0:
                uu = torch.rand(r.shape[0], self.N, device=self.device, generator=self.gen)
            outs.append(render.render_rays_fused(r, self.coarse, self.fine, self.n, self.N, u=uu,
                                                 white_bkgd=self.white_bkgd, ref_quirks=self.q, with_coarse=False)["rgb_map"])
        return torch.cat(outs, 0)

    def render_frame(self, c2w, shard: bool = True) -> Optional[torch.Tensor]:
        """Full frame [H,W,3]; with world_size > 1 each rank renders a contiguous slice of the
        pixel list and rank 0 receives the image (others return None)."""
        lo, hi = parallel.shard_range(self.H * self.W, self.rank, self.world) if shard else (0, self.H * self.W)
        idx = torch.arange(lo, hi, device=self.device, dtype=torch.int64)
        rays = ray.gen_rays(self.H, self.W, self.K, np.asarray(c2w)[:3, :4], self.near, self.far, idx)
        rgb = self.render_rays(rays)
        if shard and self.world > 1:
            rgb = parallel.gather_rows_to_rank0(rgb, self.H * self.W)
            if rgb is None:
                return None
        return rgb.reshape(self.H, self.W, 3)

    def psnr(self, c2w, gt: torch.Tensor) -> float:
        img = self.render_frame(c2w, shard=False)
        mse = torch.mean((img - gt.to(self.device)) ** 2)
        return float(10.0 * torch.log10(1.0 / mse))

    # ---------------------------------------------------------------- checkpoint
    def state_dict(self):
        sd = {"it": self.it, "coarse": self.coarse.params.cpu(), "adam": {k: [t.cpu() for t in v] for k, v in self.opt.state.items() if isinstance(k, str)}}
        if self.fine is not None:
            sd["fine"] = self.fine.params.cpu()
        return sd

    def save(self, path: str):
        """Flat-buffer checkpoint as .npz (the reference declares --i_weights / --ft_path but never implements them,
        `models/NeRF.py:122-125`)."""
        sd = self.state_dict()
        arrays = {"it": np.array(sd["it"]), "coarse": sd["coarse"].numpy()}
        if "fine" in sd:
            arrays["fine"] = sd["fine"].numpy()
        for k, (m, v) in sd["adam"].items():
            arrays[f"adam_{k}_m"], arrays[f"adam_{k}_v"] = m.numpy(), v.numpy()
        np.savez(path, **arrays)

    def load(self, path: str):
        z = np.load(path)
        sd = {"it": int(z["it"]), "coarse": torch.from_numpy(z["coarse"]), "adam": {}}
        if "fine" in z:
            sd["fine"] = torch.from_numpy(z["fine"])
        for k in z.files:
            if k.startswith("adam_") and k.endswith("_m"):
                name = k[5:-2]
                sd["adam"][name] = [torch.from_numpy(z[k]), torch.from_numpy(z[f"adam_{name}_v"])]
        self.load_state_dict(sd)

    def load_state_dict(self, sd):
        self.it = int(sd["it"])
        self.coarse.load_flat(sd["coarse"])
        if self.fine is not None and "fine" in sd:
            self.fine.load_flat(sd["fine"])
        for k, v in sd.get("adam", {}).items():
            self.opt.state[k] = [t.to(self.device) for t in v]
            self.opt.step_count.setdefault(k, 0)
