"""BASELINE configs[4]: multiresolution hash-grid encoding + tiny fused MLP.

The reference ships the pieces but never wires them (`encoding/multi_hash.py` cannot run, SURVEY Q13;
`encoding/spherical_harmonics.py` has no caller): this module is that wiring, with the reference's own classes --
`MultiHashEncoding(3, 16, 16, 2048, 2, 19)` for positions, `SphericalHarmonicsEncoding(3, 3)` for view directions,
and its `NeRF` class at Instant-NGP size (`NeRF(n_layers=2, width_layers=64, channel_input=32,
channel_input_views=16, list_skip_connection_layers=[], is_use_view_directions=True)`, models/NeRF.py:160-243) --
trained by the coarse-only loop of `render_rays` (rendering/render.py:112-162) + `raw2outputs` + MSE + Adam.

Device work: nerf_hashgrid_forward / nerf_sh_encode -> nerf_mlp_forward_train (2 x 64 kernels) ->
nerf_composite_mse_backward (raw2outputs + MSE + adjoint, one launch) -> nerf_mlp_backward_inputs (dZ chain, dW,
dL/dfeatures) -> nerf_hashgrid_backward_rays_ex per LEVEL GROUP (float atomics, or -- `deterministic=True` -- int64
fixed-point integer atomics whose sum does not depend on the order of the requests) -> nerf_adam_step (MLP) and
nerf_adam_step_ex (tables: reads the accumulated gradient and clears it in the same pass, no memset launch).
With world_size > 1 the table gradient (64 MB float32 / 128 MB int64) is all-reduced group by group on a second
stream while the next group's scatter, the MLP all-reduce and the MLP Adam run (NGPTrainer.train_step).
"""
import ctypes as C
import os
from typing import Dict, Optional

import torch

from .. import _native as N
from .. import parallel, sampling
from ..encoding.multi_hash import MultiHashEncoding
from ..encoding.spherical_harmonics import SphericalHarmonicsEncoding
from ..models.NeRF import Adam, NeRF
from ..rendering import render
from .trainer import Trainer


class _Flat:
    """Adam-compatible view of a flat trainable buffer (the hash tables)."""

    def __init__(self, params: torch.Tensor, grads: torch.Tensor, half: bool = False):
        self.params, self.grads, self.n_params = params.view(-1), grads.view(-1), params.numel()     # grads: float32 or int64 fixed point
        self.name = "tables"
        # fp16 shadow image of the tables for the forward gathers (4 instead of 8 bytes per entry pair): written by the Adam
        # pass that updates the float32 master (`nerf_adam_step_shadow`), rebuilt from the master whenever somebody else wrote it
        self.half = torch.empty(self.n_params, dtype=torch.float16, device=params.device) if half else None
        self._half_version = -1              # params._version at which the shadow is known to equal fp16(params); -1: stale

    def mark_updated(self, shadow_written: bool = False):
        """Called by Adam after it wrote the master through raw pointers (no torch version bump): with shadow_written it
        wrote EVERY shadow entry in the same pass, so the shadow is current; otherwise it is stale."""
        self._half_version = self.params._version if shadow_written else -1

    def shadow(self) -> Optional[torch.Tensor]:
        """The fp16 image, current with the float32 master (None when the model was built without one).  Rebuilt by one
        conversion pass after construction, load_flat, or a torch in-place edit of the tables (version counter)."""
        if self.half is None:
            return None
        if self._half_version != self.params._version:
            self.half.copy_(self.params)
            self._half_version = self.params._version
        return self.half

    def load_flat(self, flat: torch.Tensor):
        assert flat.numel() == self.n_params
        with torch.no_grad():
            self.params.copy_(flat.to(self.params.device, torch.float32).reshape(-1))
        self._half_version = -1


class HashNeRF:
    """positions -> hash grid (32) | view directions -> SH degree 3 (16) -> NeRF 2 x 64 -> raw [rgb, sigma]."""

    def __init__(self, device="cuda", seed: int = 0, n_levels: int = 16, min_res: int = 16, max_res: int = 2048,
                 n_features_per_level: int = 2, log2_hashmap_size: int = 19, hash_init_scale: float = 1e-4,
                 bound: Optional[float] = 1.5, deterministic: bool = True, level_groups: int = 4,
                 half_tables: Optional[bool] = None, precision: int = 22):
        """precision: arithmetic of the 2 x 64 network (`NeRF(precision=...)`).  22 (default): the reference's float32 tolerance
        -- float32 gathers from the master tables, float32 interpolation (encoding/multi_hash.py:112-131), split-bf16 MLP
        (csrc/mlp_s16x.hip); 16: the declared reduced-precision mode (bf16 MLP operands, interpolated features rounded to bf16).
        half_tables (precision 16 only; default on there): the fused query gathers from an fp16 shadow image of the tables (half
        the gather bytes; float32 master, float32 interpolation; the bf16 MLP rounds the interpolated features to 8 bits anyway)
        -- SURVEY 8(d)'s 512 B per sample.  Refused at precision 22, whose point is not to round the features.
        bound: half-extent of the scene box that is mapped onto the grid's unit cube before hashing
        (x' = (x + bound) / (2 bound)), so that N_l is the level's resolution ACROSS the scene (without it the reference's
        x * N_l sees world units: 3 x finer cells, and a 24-view run memorises its training rays: held-out PSNR 13.8 dB
        at a training loss of 1e-3).  None = world coordinates, the bare reference formula."""
        assert n_levels * n_features_per_level == 32, "the 2 x 64 kernels take 32 position features"
        self.enc = MultiHashEncoding(3, n_levels, min_res, max_res, n_features_per_level, log2_hashmap_size,
                                     hash_init_scale, device=device, seed=seed)
        self.sh = SphericalHarmonicsEncoding(3, 3)
        self.mlp = NeRF(n_layers=2, width_layers=64, channel_input=32, channel_input_views=16,
                        list_skip_connection_layers=[], is_use_view_directions=True, device=device, seed=seed, precision=precision)
        self.precision = int(precision)
        if half_tables is None:
            half_tables = self.precision == 16
        if half_tables and self.precision != 16:
            raise ValueError("HashNeRF: half_tables (fp16 shadow gathers) is a reduced-precision option of precision=16")
        self.pos_scale, self.pos_offset = (1.0, 0.0) if bound is None else (1.0 / (2.0 * bound), 0.5)
        # table-gradient accumulator: int64 2^-52 fixed point added with integer atomics (the default: bit-reproducible, and
        # measured no slower than float atomics -- both are bound by the atomic request rate, profiles/r03_ngp_scatter.csv)
        # or float32 (float atomics, order-dependent rounding).  The Adam pass that consumes it clears it: no memset.
        self.deterministic = bool(deterministic)
        if self.deterministic:
            self.enc.grad = torch.zeros(self.enc.tables.shape, dtype=torch.int64, device=self.enc.tables.device)
        self.table = _Flat(self.enc.tables, self.enc.grad, half=bool(half_tables) and n_features_per_level == 2)
        # True while the accumulator is known to hold zeros (fresh, or consumed-and-cleared by nerf_adam_step_ex); any
        # scatter makes it dirty.  NGPTrainer.train_step only skips the clear when this says so: a public backward() call,
        # or a step that raised between the scatter and Adam, leaves a gradient behind that must not be added to the next one.
        self._grad_clean = True
        ng = max(1, min(int(level_groups), n_levels))
        self.level_groups = [(n_levels * i // ng, n_levels * (i + 1) // ng) for i in range(ng)]
        self.on_group_done = None           # NGPTrainer: called after each level group's scatter is enqueued (lo, hi)
        self.on_mlp_grads = None            # NGPTrainer: called with the MLP gradient as soon as it is enqueued (before the scatters)
        self._pts, self._rz = None, None
        self.fused = os.environ.get("NERF_NGP_FUSED", "1") != "0"      # rows inside the forward kernel (default) or through HBM
        self.timing = None                  # bench.py: list that receives (start, end) events around the table scatter

    def features(self, rays: torch.Tensor, z: torch.Tensor, need_pts: bool = True):
        """x [B n, 48] = [MultiHashEncoding(o + z d) | SphericalHarmonicsEncoding(viewdirs)] in one call
        (`nerf_ngp_encode`), and the sample positions for the table-gradient pass."""
        B, n = z.shape
        e = self.enc
        x = torch.empty(B * n, 48, dtype=torch.float32, device=z.device)
        pts = torch.empty(B * n, 3, dtype=torch.float32, device=z.device) if need_pts else None
        N.check(N.lib().nerf_ngp_encode(N.ptr(N.f32(rays)), N.ptr(N.f32(z)), B, n, N.ptr(e.tables), e.n_levels,
                                        e.log2_hashmap_size, e.n_features_per_level, e._res_c, 3, self.pos_scale,
                                        self.pos_offset, N.ptr(x), N.ptr(pts), N.stream()))
        return pts, x

    def features_unfused(self, rays: torch.Tensor, z: torch.Tensor):
        """The same rows from the stand-alone encoder classes (tests compare the two)."""
        B, n = z.shape
        pts = (rays[:, None, 0:3] + z[:, :, None] * rays[:, None, 3:6]).reshape(-1, 3)     # render.py:142
        pts = pts * self.pos_scale + self.pos_offset                                       # scene box -> unit cube
        feat = self.enc(pts)                                                               # [B n, 32]
        shf = self.sh(rays[:, 8:11].contiguous())                                          # [B, 16]
        x = torch.cat([feat.view(B, n, 32), shf[:, None, :].expand(B, n, 16)], dim=-1).reshape(B * n, 48)
        return pts, x

    def query(self, rays: torch.Tensor, z: torch.Tensor, train: bool = False, fused: Optional[bool] = None) -> torch.Tensor:
        """raw [B,n,4].  fused (default): hash gathers and SH are evaluated inside the 2 x 64 forward kernel
        (`nerf_ngp_query_fused`), no [B n, 48] rows in HBM; fused=False goes through `features` + `mlp.forward`."""
        B, n = z.shape
        if fused is None:
            fused = self.fused
        if not fused:
            pts, x = self.features(rays, z, need_pts=train)
            self._pts, self._rz = (pts if train else None), None
            return self.mlp.forward(x, train=train).view(B, n, 4)
        e, m = self.enc, self.mlp
        rays, z = N.f32(rays), N.f32(z)
        raw = torch.empty(B, n, 4, dtype=torch.float32, device=z.device)
        acts = None
        if train:
            acts = m._begin_train_pass(B * n)
            self._pts, self._rz = None, (rays, z)
        N.check(N.lib().nerf_ngp_query_fused_h(C.byref(m.arch), N.ptr(m.packed()), N.ptr(rays), N.ptr(z), B, n,
                                               N.ptr(e.tables), N.ptr(self.table.shadow()), e.n_levels, e.log2_hashmap_size,
                                               e.n_features_per_level, e._res_c, 3, self.pos_scale, self.pos_offset,
                                               N.ptr(raw), N.ptr(acts), N.stream()))
        return raw

    def table_grad(self) -> torch.Tensor:
        """The accumulated table gradient as float32 [L,T,F] (a copy when the accumulators are int64 fixed point)."""
        g = self.enc.grad
        return (g.double() * 2.0 ** -52).float() if g.dtype == torch.int64 else g

    def backward(self, d_raw: torch.Tensor, accumulate: bool = False):
        """(MLP gradient [13188], table gradient [L,T,F]: float32, or int64 2^-52 fixed point when deterministic) of the
        last query(train=True).  accumulate=True adds into the gradient buffer as it stands (NGPTrainer: the Adam pass that
        consumed the previous gradient left it zeroed); the default clears it first."""
        grads, d_x = self.mlp.backward(d_raw, need_input_grad=True)
        if self.on_mlp_grads is not None:
            self.on_mlp_grads(grads)
        e = self.enc
        if not accumulate:
            e.grad.zero_()
        self._grad_clean = False
        if self.timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self._rz is None:                                 # unfused rows: positions were kept by features()
            e.backward(self._pts, d_x)
            if self.on_group_done is not None:
                self.on_group_done(0, e.n_levels)
        else:
            rays, z = self._rz
            for lo, hi in self.level_groups:
                N.check(N.lib().nerf_hashgrid_backward_rays_ex(
                    N.ptr(rays), N.ptr(z), z.shape[0], z.shape[1], N.ptr(d_x), e.n_levels, e.log2_hashmap_size,
                    e.n_features_per_level, e._res_c, self.pos_scale, self.pos_offset, lo, hi, int(self.deterministic),
                    N.ptr(e.grad), N.stream()))
                if self.on_group_done is not None:
                    self.on_group_done(lo, hi)
        if self.timing is not None:
            e1.record()
            self.timing.append((e0, e1))
        return grads, e.grad


class NGPTrainer(Trainer):
    """Coarse-only training / rendering loop on a HashNeRF (one network, `n_depth_samples` stratified-grid samples per
    ray, no importance pass).  Rays shard across ranks; MLP and table gradients are sum-all-reduced before Adam."""

    def __init__(self, images, poses, K, near: float = 2.0, far: float = 6.0, N_rand: int = 4096,
                 n_depth_samples: int = 64, lrate: float = 5e-4, lrate_decay: int = 500, white_bkgd: bool = True,
                 seed: int = 0, device="cuda", chunk: int = 1024 * 32, table_sync: str = "shard", precision: int = 22,
                 **hash_kw):
        super().__init__(images, poses, K, near=near, far=far, N_rand=N_rand, n_depth_samples=n_depth_samples,
                         N_importance=0, lrate=lrate, lrate_decay=lrate_decay, white_bkgd=white_bkgd, ref_quirks=True,
                         seed=seed, device=device, chunk=chunk, precision=precision)
        self.coarse = None                                   # the 8 x 256 network of the base class is not used
        self._field = HashNeRF(device=self.device, seed=seed, precision=precision, **hash_kw)
        self._field.mlp.name = "mlp"
        # Adam WITH bias correction: without it the first steps are lr * sign(g), which turns bf16 noise in near-zero
        # table gradients into full-size steps and can drive sigma negative everywhere (a dead network under the
        # reference's un-activated sigma).  This loop is our wiring, so the choice is ours; lrate is
        # the reference's 5e-4 (at 2e-3 both this trainer and the fp32 oracle collapse to sigma < 0 within 200 iterations).
        self.opt = Adam(lrate, betas=(0.9, 0.99), eps=1e-8, bias_correction=True, shared_state=False)
        self._comm = torch.cuda.Stream(device=self.device) if self.world > 1 else None
        # How the table gradient is combined across ranks (world_size > 1):
        #   "allreduce": every level group's accumulator slice is sum-all-reduced, every rank steps all 16.8 M entries
        #                (int64 accumulators: 134 MB on the wire twice per step);
        #   "shard"    : reduce-scatter of the accumulators (each rank receives the sum of ITS 1/W of every level group), Adam on
        #                the owned shards only, all-gather of the updated float32 tables: 134 (W-1)/W + 67 (W-1)/W MB instead of
        #                2 x 134 (W-1)/W, a W-th of the Adam work, and exact integer sums as before (bit-identical tables on all
        #                ranks, run to run, and identical to the all-reduce schedule).  Adam moments are sharded with the
        #                parameters: `sync_optimizer_state()` gathers them -- an explicit COLLECTIVE every rank calls at the same
        #                iteration; state_dict() / save() never communicate and RAISE while the moments are stale, so the
        #                usual "if rank == 0: save()" idiom cannot hang in a collective the other ranks never enter.
        if table_sync not in ("shard", "allreduce"):
            raise ValueError("NGPTrainer: table_sync must be 'shard' or 'allreduce'")
        per_group = [(hi - lo) * self._field.enc.hash_table_size * self._field.enc.n_features_per_level for lo, hi in self._field.level_groups]
        self.table_sync = table_sync if (self.world > 1 and all(n % self.world == 0 for n in per_group)) else "allreduce"
        self._rs_bufs = None
        self._moments_synced = True          # no step taken yet: nothing sharded

    # `field` (tables, MLP) may still be receiving the other ranks' updated shards on the comm stream when train_step returns:
    # reading it from outside joins that stream first (a stream-side wait, no host block).  The hot loop uses _field.
    @property
    def field(self):
        self._join_comm()
        return self._field

    @field.setter
    def field(self, f):
        self._field = f

    def train_step(self, rays=None, target=None, u=None) -> Dict[str, torch.Tensor]:
        if rays is None:
            rays, target = self.sample_batch()
        self._opt.learning_rate = self.lrate * (0.1 ** (self.it / (self.lrate_decay * 1000)))
        z = sampling.sample_coarse(rays, self.n)
        self._join_comm()                                    # the previous step's table all-gathers (sharded updates)
        raw = self._field.query(rays, z, train=True)
        loss, d_raw, _ = render.composite_mse_backward(raw, z, rays, target, self.white_bkgd)
        pending, mlp_work = [], []
        shard = self.world > 1 and self.table_sync == "shard"
        if self.world > 1:
            # Collectives of one process group run in issue order, so the small MLP gradient goes FIRST (it is complete
            # before the scatters start), then each level group's slice of the table gradient as soon as its scatter is
            # enqueued -- all on the comm stream, behind events: the transfers overlap the following groups' scatters,
            # and the MLP Adam runs while the table slices are still on the wire.
            per_level = self._field.enc.hash_table_size * self._field.enc.n_features_per_level
            flat = self._field.table.grads

            def on_comm(t):
                ev = torch.cuda.Event()
                ev.record()
                self._comm.wait_event(ev)
                with torch.cuda.stream(self._comm):
                    return torch.distributed.all_reduce(t, async_op=True)
            self._field.on_mlp_grads = lambda g: mlp_work.append(on_comm(g))
            if shard:
                if self._rs_bufs is None or self._rs_bufs[0].dtype != flat.dtype:
                    self._rs_bufs = [torch.empty((hi - lo) * per_level // self.world, dtype=flat.dtype, device=flat.device)
                                     for lo, hi in self._field.level_groups]

                def on_group(lo, hi):
                    gi = self._field.level_groups.index((lo, hi))
                    ev = torch.cuda.Event()
                    ev.record()
                    self._comm.wait_event(ev)
                    with torch.cuda.stream(self._comm):
                        pending.append(torch.distributed.reduce_scatter_tensor(self._rs_bufs[gi], flat[lo * per_level:hi * per_level],
                                                                               async_op=True))
                self._field.on_group_done = on_group
            else:
                self._field.on_group_done = lambda lo, hi: pending.append(on_comm(flat[lo * per_level:hi * per_level]))
        # accumulate (no clear) only when the accumulator is KNOWN to be zero: left so by the previous step's table Adam
        g_mlp, g_tab = self._field.backward(d_raw, accumulate=self._field._grad_clean)
        self._field.on_group_done = self._field.on_mlp_grads = None
        for w in mlp_work:
            w.wait()
        self._opt.update(self._field.mlp, g_mlp, grad_scale=1.0 / self.world)      # _opt: the `opt` property would join the comm stream
        for w in pending:
            w.wait()
        if pending:
            torch.cuda.current_stream().wait_stream(self._comm)
        if shard:
            # Adam on the owned 1/W of every level group (sums received by the reduce-scatter), then every rank gets the others'
            # updated entries; the accumulator was consumed by the collectives: clear it (one 134 MB memset, ~30 us)
            params = self._field.table.params
            spans = []
            for (lo, hi), buf in zip(self._field.level_groups, self._rs_bufs):
                n = buf.numel()
                spans.append((lo * per_level + self.rank * n, lo * per_level + (self.rank + 1) * n, buf))
            self._opt.update_spans(self._field.table, spans, grad_scale=1.0 / self.world)
            self._moments_synced = False
            flat.zero_()
            # all-gathers of the updated shards on the comm stream, one per level group, behind the Adam launches; the next
            # step's forward (the first reader of the tables) joins that stream
            half = self._field.table.half
            ev = torch.cuda.Event()
            ev.record()
            self._comm.wait_event(ev)
            with torch.cuda.stream(self._comm):
                for (lo, hi), (a, b, _) in zip(self._field.level_groups, spans):
                    torch.distributed.all_gather_into_tensor(params[lo * per_level:hi * per_level], params[a:b].clone())
                    if half is not None:     # the fp16 shadow shards Adam wrote in the same pass travel too: no re-conversion
                        torch.distributed.all_gather_into_tensor(half[lo * per_level:hi * per_level], half[a:b].clone())
            if half is not None:
                self._field.table.mark_updated(shadow_written=True)
        else:
            self._opt.update(self._field.table, g_tab.view(-1), grad_scale=1.0 / self.world, zero_grads=True)    # reads g, writes 0
        self._field._grad_clean = True
        self.it += 1
        return {"loss_coarse": loss}

    def render_rays(self, rays: torch.Tensor, u=None):
        outs = []
        self._join_comm()
        for s in range(0, rays.shape[0], self.chunk):
            r = rays[s:s + self.chunk]
            z = sampling.sample_coarse(r, self.n)
            raw = self._field.query(r, z)
            outs.append(render.composite(raw, z, r, 0.0, self.white_bkgd, need_weights=False)[0])
        return torch.cat(outs, 0)

    def sync_optimizer_state(self):
        """COLLECTIVE (every rank, same iteration): with sharded table updates (table_sync "shard", world_size > 1) a rank's
        Adam moments are current on its own shards only; this gathers the others'.  A no-op otherwise.  Call it on all ranks
        before a rank-0 `state_dict()` / `save()` (entrypoints/test_nerf.py does)."""
        if self.world > 1 and self.table_sync == "shard" and "tables" in self._opt.state and not self._moments_synced:
            self._join_comm()
            per_level = self._field.enc.hash_table_size * self._field.enc.n_features_per_level
            for t in self._opt.state["tables"]:
                for lo, hi in self._field.level_groups:
                    n = (hi - lo) * per_level // self.world
                    a = lo * per_level + self.rank * n
                    torch.distributed.all_gather_into_tensor(t[lo * per_level:hi * per_level], t[a:a + n].clone())
        self._moments_synced = True

    def state_dict(self):
        """Never communicates.  With sharded table updates the Adam moments of the other ranks' shards are stale after a
        training step: raises until `sync_optimizer_state()` has run (on every rank) at this iteration."""
        if self.world > 1 and self.table_sync == "shard" and not self._moments_synced:
            raise RuntimeError("NGPTrainer.state_dict / save: the Adam moments of the hash tables are sharded across ranks "
                               "(table_sync='shard'); call tr.sync_optimizer_state() on EVERY rank first (a collective), then "
                               "save from rank 0")
        return super().state_dict()

    def load_state_dict(self, sd, allow_legacy_rng: bool = False):
        """Joins the comm stream first (the table all-gathers of a sharded step may still be writing the buffers this loads into),
        and a loaded state is complete on every rank: the moments count as synced until the next sharded step."""
        self._join_comm()
        super().load_state_dict(sd, allow_legacy_rng=allow_legacy_rng)
        self._moments_synced = True

    def _checkpoint_buffers(self):
        """Trainer.save / load / state_dict / load_state_dict work on these: the 2 x 64 MLP and the hash tables, with
        their two Adam (m, v) pairs and step counts (bias correction is on here, so the counts matter)."""
        return {"mlp": self._field.mlp, "tables": self._field.table}
