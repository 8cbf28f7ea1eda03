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

Round 4: with world_size > 1 the collectives and the Adam steps run on a COMM STREAM, in program order
(AR_coarse, Adam_coarse, AR_fine, Adam_fine, AR_coarse', ...).  The compute stream joins it after the coarse step
(the re-render at :270 reads the updated coarse weights) but NOT after the fine step: the fine network's all-reduce +
Adam overlap the next iteration's batch sampling and coarse forward / backward, which read neither the fine weights
nor -- until the next coarse Adam, which is queued behind on the same stream -- the shared Adam moments (Q7).  Same
arithmetic in the same order as the serial schedule: weights are bit-identical (tests/test_gpu_multirank.py).
"""
import os
from typing import Dict, Optional

import numpy as np
import torch

from .. import parallel, sampling
from ..models.NeRF import Adam, NeRF
from ..ops import index
from ..rendering import ray, render


class Trainer:
    def __init__(self, images: torch.Tensor, poses: torch.Tensor, K, near: float = 2.0, far: float = 6.0,
                 N_rand: int = 1024, n_depth_samples: int = 64, N_importance: int = 128, lrate: float = 5e-4,
                 lrate_decay: int = 500, white_bkgd: bool = True, ref_quirks: bool = True, seed: int = 0,
                 device="cuda", chunk: int = 1024 * 32, precision: int = 22, overlap_comm: bool = True):
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
        # Everything random of iteration `it` (image, pixels, importance uniforms) comes from generators seeded with
        # parallel.counter_seed(seed, rank, stream, it): (seed, rank, it) IS the RNG state, so a checkpoint written by
        # rank 0 resumes every rank on its own streams.
        mk = lambda s: NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True,
                            device=self.device, seed=s, precision=precision)
        self.precision = precision
        self.coarse = mk(seed)                                  # identical initial weights on every rank
        self.fine = mk(seed + 1) if N_importance > 0 else None
        self.coarse.name = "coarse"                             # Adam state keys when the state is not shared
        if self.fine is not None:
            self.fine.name = "fine"
        self.opt = Adam(lrate, betas=(0.9, 0.999), shared_state=ref_quirks)
        self.it = 0
        # collectives + Adam steps of a multi-rank run (see the module docstring); None: everything on the caller's stream
        self._comm = torch.cuda.Stream(device=self.device) if (self.world > 1 and overlap_comm) else None
        self.gen = torch.Generator(device=self.device)                 # evaluation-time uniforms (render_rays without u)
        self.gen.manual_seed(parallel.rank_seed(seed, self.rank, 2))
        self._train_gen = torch.Generator(device=self.device)          # re-seeded per iteration (train_uniforms)

    # `fine` and `opt` are touched by work that may still be queued on the comm stream when train_step returns: reading
    # them from outside joins that stream first (a stream-side wait, no host block).  The hot loop uses _fine / _opt.
    def _join_comm(self):
        if getattr(self, "_comm", None) is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._comm)

    @property
    def fine(self):
        self._join_comm()
        return self._fine

    @fine.setter
    def fine(self, m):
        self._fine = m

    @property
    def opt(self):
        self._join_comm()
        return self._opt

    @opt.setter
    def opt(self, o):
        self._opt = o

    def image_choice(self) -> int:
        """np.random.choice(i_train) of iteration `self.it` (`__test_nerf.py:202`)."""
        return parallel.counter_seed(self.seed, self.rank, 1, self.it) % int(self.images.shape[0])

    def train_uniforms(self, B: int) -> torch.Tensor:
        """torch.rand [B, N_importance] of iteration `self.it` (`sampling/__init__.py:140`)."""
        self._train_gen.manual_seed(parallel.counter_seed(self.seed, self.rank, 2, self.it))
        return torch.rand(B, self.N, device=self.device, generator=self._train_gen)

    # ---------------------------------------------------------------- one network step
    def _step_net(self, model: NeRF, rays, z, target, white: bool, join: bool = True):
        raw = model.query(rays, z, ref_quirks=self.q, train=True)
        loss, d_raw, _ = render.composite_mse_backward(raw, z, rays, target, white)      # raw2outputs + MSE + adjoint
        grads = model.backward(d_raw)
        if self._comm is None:
            parallel.allreduce_sum_(grads)                                     # C1: the only collective
            self._opt.update(model, grads, grad_scale=1.0 / self.world)
            return loss
        cur = torch.cuda.current_stream(self.device)
        self._comm.wait_stream(cur)                                            # the gradient is complete
        with torch.cuda.stream(self._comm):
            parallel.allreduce_sum_(grads)
            self._opt.update(model, grads, grad_scale=1.0 / self.world)       # Adam kernel on the comm stream (N.stream())
        if join:
            cur.wait_stream(self._comm)
        return loss

    def sample_batch(self, pixel_idx: Optional[torch.Tensor] = None, img_i: Optional[int] = None):
        """(rays [B,11], target [B,3]); by default one random image, N_rand distinct pixels."""
        if img_i is None:
            img_i = self.image_choice()
        if pixel_idx is None:                                                  # the usual case: one fused launch
            return ray.sample_batch(self.H, self.W, self.K, self.poses[img_i, :3, :4], self.near, self.far,
                                    self.images[img_i], self.N_rand, parallel.counter_seed(self.seed, self.rank, 3, self.it))
        rays = ray.gen_rays(self.H, self.W, self.K, self.poses[img_i, :3, :4], self.near, self.far, pixel_idx)
        target = index.gather_rows(self.images[img_i].reshape(-1, 3), pixel_idx)
        return rays, target

    def train_step(self, rays=None, target=None, u=None) -> Dict[str, torch.Tensor]:
        """One iteration of `__test_nerf.py:200-305`.  Losses are device scalars (no host sync)."""
        if rays is None:
            rays, target = self.sample_batch()
        self._opt.learning_rate = self.lrate * (0.1 ** (self.it / (self.lrate_decay * 1000)))  # a host float, passed by value
        z = sampling.sample_coarse(rays, self.n)                               # perturb = 0 in quirk mode (Q5)
        out = {"loss_coarse": self._step_net(self.coarse, rays, z, target, self.white_bkgd)}
        if self._fine is not None:
            raw = self.coarse.query(rays, z, ref_quirks=self.q)                # updated coarse net, no grad (:270)
            _, _, _, weights, _ = render.composite(raw, z, rays, 0.0, self.white_bkgd)
            if u is None:
                u = self.train_uniforms(rays.shape[0])
            _, z_fine = sampling.importance_sample(z, weights, self.N, u=u)
            white_fine = self.white_bkgd if not self.q else False               # Q8
            # no join: the fine all-reduce + Adam run under the next iteration's sampling and coarse forward / backward
            out["loss_fine"] = self._step_net(self._fine, rays, z_fine, target, white_fine, join=False)
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
    def _checkpoint_buffers(self):
        """name -> object with `.params` (flat fp32 device buffer) and `.load_flat(t)`; the names are also the Adam keys."""
        b = {"coarse": self.coarse}
        if self.fine is not None:
            b["fine"] = self.fine
        return b

    def sync_optimizer_state(self):
        """COLLECTIVE hook (every rank, same iteration) that makes this rank's optimiser state complete before a rank-0
        `state_dict()` / `save()`.  Replicated Adam (this class): nothing to do.  NGPTrainer with sharded table updates
        gathers the other ranks' moments here."""
        self._join_comm()

    def state_dict(self):
        """Everything a bit-identical continuation needs: iteration, flat parameters, Adam (m, v) + step counts per
        state key.  The random streams are functions of (seed, rank, iteration) (parallel.counter_seed), so there is no
        RNG state to save and a file written by rank 0 is right for every rank."""
        return {"it": self.it, "seed": int(self.seed),
                "params": {k: m.params.detach().cpu().clone() for k, m in self._checkpoint_buffers().items()},
                "adam": self.opt.state_dict()}

    def load_state_dict(self, sd, allow_legacy_rng: bool = False):
        """allow_legacy_rng: accept a round-1/2 state (saved RNG states `rng_numpy*` / `rng_torch`, or no `seed`): its
        continuation here is NOT the one the file's writer would have produced (the streams are now functions of
        (seed, rank, iteration)); refused unless the caller says so."""
        legacy = [k for k in sd if str(k).startswith("rng_")]
        if (legacy or "seed" not in sd) and not allow_legacy_rng:
            raise ValueError("checkpoint predates the stateless random streams (" + (", ".join(legacy) or "no `seed` entry") +
                             "): resuming it is not a bit-identical continuation of the run that wrote it; pass "
                             "allow_legacy_rng=True to continue on this trainer's (seed, rank, iteration) streams anyway")
        self.it = int(sd["it"])
        bufs = self._checkpoint_buffers()
        params = sd.get("params", {k: sd[k] for k in bufs if k in sd})       # round-1 layout: parameters at top level
        missing = [k for k in bufs if k not in params]
        if missing:
            raise KeyError(f"checkpoint has no parameters for {missing} (has {sorted(params)})")
        for k, m in bufs.items():
            m.load_flat(params[k])
        adam = sd.get("adam", {})
        if "state" not in adam:                                              # round-1 layout: {key: [m, v]}
            adam = {"state": adam, "step_count": {}}
        self.opt.load_state_dict(adam, device=self.device)
        if "seed" in sd:
            self.seed = int(sd["seed"])      # the streams are functions of (seed, rank, it): adopt the saved run's seed so that
                                             # this trainer CONTINUES it, whatever seed its own (now overwritten) init used
            self.gen.manual_seed(parallel.rank_seed(self.seed, self.rank, 2))      # evaluation-time uniforms follow the adopted seed

    @staticmethod
    def _npz_path(path: str) -> str:
        return path if path.endswith(".npz") else path + ".npz"

    def save(self, path: str) -> str:
        """Flat-buffer checkpoint as ONE .npz (the reference declares --i_weights / --ft_path / --no_reload but never
        implements them, `models/NeRF.py:122-125`).  Returns the path written (".npz" is appended when missing, which is
        also what load() does)."""
        import json
        sd = self.state_dict()
        arrays = {"it": np.array(sd["it"], dtype=np.int64), "seed": np.array(sd["seed"], dtype=np.int64)}
        for k, t in sd["params"].items():
            arrays[f"params/{k}"] = t.numpy()
        for k, (m, v) in sd["adam"]["state"].items():
            arrays[f"adam/{k}/m"], arrays[f"adam/{k}/v"] = m.numpy(), v.numpy()
            arrays[f"adam/{k}/steps"] = np.array(sd["adam"]["step_count"].get(k, 0), dtype=np.int64)
        arrays["adam_lr"] = np.array(sd["adam"]["learning_rate"], dtype=np.float64)
        path = self._npz_path(path)
        tmp = path + ".tmp.npz"
        np.savez(tmp, **arrays)
        os.replace(tmp, path)                                   # never leave a half-written checkpoint under the final name
        return path

    def load(self, path: str, allow_legacy_rng: bool = False):
        import json
        path = path if os.path.exists(path) else self._npz_path(path)
        z = np.load(path)
        sd = {"it": int(z["it"]), "params": {}, "adam": {"state": {}, "step_count": {}}}
        for k in z.files:
            if k.startswith("params/"):
                sd["params"][k[7:]] = torch.from_numpy(z[k])
            elif k.startswith("adam/") and k.endswith("/m"):
                name = k[5:-2]
                sd["adam"]["state"][name] = [torch.from_numpy(z[k]), torch.from_numpy(z[f"adam/{name}/v"])]
                sd["adam"]["step_count"][name] = int(z[f"adam/{name}/steps"])
        if "adam_lr" in z.files:
            sd["adam"]["learning_rate"] = float(z["adam_lr"])
        if "seed" in z.files:
            sd["seed"] = int(z["seed"])
        for k in z.files:
            if k.startswith("rng_"):
                sd[k] = True                 # round-1/2 file: saved RNG states this trainer no longer uses
        self.load_state_dict(sd, allow_legacy_rng=allow_legacy_rng)
        return sd["it"]
