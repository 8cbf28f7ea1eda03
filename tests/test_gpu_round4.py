"""Round-4 GPU tests.

(1) configs[4]: a table gradient left behind by a public HashNeRF.backward() call (or by a step that raised between the
    scatter and Adam) is NOT added to the next training step's gradient (advisor, round 3).
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"


def _ngp(det, groups=4, seed=7, **kw):
    from nerf_meets_mlx_amd.dataset import synthetic
    from nerf_meets_mlx_amd.engine.ngp import NGPTrainer
    imgs, poses, _, _, K = synthetic.make_dataset(24, 24, 3, seed=0, device=DEV)
    return NGPTrainer(imgs, poses, K, N_rand=128, n_depth_samples=64, seed=seed, device=DEV, log2_hashmap_size=14,
                      deterministic=det, level_groups=groups, **kw)


@pytest.mark.parametrize("det", [True, False])
def test_ngp_stale_table_gradient_is_not_added_to_the_next_step(det):
    """NGPTrainer.train_step scatters into the accumulator WITHOUT clearing it when the previous step's Adam left it
    zero.  A public field.backward() in between (default accumulate=False: clears, then leaves ITS gradient behind)
    must not leak into the step: parameters after (backward; train_step) == parameters after (train_step) alone."""
    from nerf_meets_mlx_amd import sampling
    from nerf_meets_mlx_amd.rendering import render
    a, b = _ngp(det), _ngp(det)
    for tr in (a, b):
        tr.train_step()                                  # one ordinary step first: Adam state exists, accumulator cleared
    assert a.field._grad_clean and b.field._grad_clean
    rays, target = a.sample_batch()
    rb, tb = b.sample_batch()
    assert torch.equal(rays, rb) and torch.equal(target, tb)
    # a: three stray gradient evaluations on another batch (what tests/test_gpu_parity.py does before train_step)
    other = torch.roll(rays, 7, 0)
    z = sampling.sample_coarse(other, 64)
    for _ in range(3):
        raw = a.field.query(other, z, train=True)
        _, d_raw, _ = render.composite_mse_backward(raw, z, other, torch.roll(target, 3, 0), True)
        a.field.backward(d_raw)
    assert not a.field._grad_clean
    assert float(a.field.table_grad().abs().max()) > 0    # a stale gradient IS sitting in the accumulator
    a.train_step(rays, target)
    b.train_step(rays, target)
    torch.cuda.synchronize()
    assert a.field._grad_clean
    if det:
        assert torch.equal(a.field.enc.tables, b.field.enc.tables)
    else:
        # float atomics: order-dependent rounding, nothing more.  An Adam update is lr * m / (sqrt(v) + eps) with lr = 5e-4: the
        # sum of a cancelling set of addends in another order differs by ~1e-4 relative, i.e. 5e-8 of movement (measured 3.6e-8
        # at the default precision); a leaked stale gradient would move entries by the full lr = 5e-4
        assert float((a.field.enc.tables - b.field.enc.tables).abs().max()) <= 2e-7
    assert torch.equal(a.field.mlp.params, b.field.mlp.params) or not det
    assert float(a.field.table_grad().abs().max()) == 0   # consumed and cleared


# ------------------------------------------------------------------------------------------------ a20: the loss closures
def _flat_from_seed(layers, seed, checksum, alpha):
    rng = np.random.default_rng(seed)
    chunks, tot = [], 0.0
    for name, o, i in layers:
        k = 1.0 / np.sqrt(i)
        w = rng.uniform(-k, k, size=(o, i)).astype(np.float32)
        b = rng.uniform(-k, k, size=(o,)).astype(np.float32)
        tot += float(np.abs(w.astype(np.float64)).sum() + np.abs(b.astype(np.float64)).sum())
        if name == "alpha":
            w = w * np.float32(alpha[0]); b = b * np.float32(alpha[0]) + np.float32(alpha[1])
        chunks += [w.reshape(-1), b]
    assert abs(tot - checksum) <= 1e-9 * checksum
    return torch.from_numpy(np.concatenate(chunks))


@pytest.mark.parametrize("precision,tol", [(32, 1e-4), (22, 1e-4), (16, 2e-2)])
def test_training_losses_vs_reference_loss_closures(golden_dir, precision, tol):
    """The HIP training path's loss values against the reference's own `mlx_mse_coarse` / `mlx_mse_fine`
    (entrypoints/__test_nerf.py:47-126, AST-extracted and executed over the shim: tests/golden/make_golden_losses.py) on the
    same rays, targets, weights and uniforms: coarse loss with the white background of the kwargs, importance samples +
    sort of :275-288, fine loss WITHOUT the white background (Q8), fine rgb."""
    import json
    from nerf_meets_mlx_amd import sampling
    from nerf_meets_mlx_amd.models.NeRF import NeRF
    from nerf_meets_mlx_amd.rendering import render
    g = np.load(os.path.join(golden_dir, "ref_mx_losses.npz"))
    with open(os.path.join(golden_dir, "ref_mx_losses.json")) as fp:
        m = json.load(fp)
    layers = [tuple(l) for l in m["layers"]]
    nets = {}
    for name in ("coarse", "fine"):
        net = NeRF(channel_input=63, channel_input_views=27, channel_output=5, is_use_view_directions=True, device=DEV, seed=0,
                   precision=precision)
        net.load_flat(_flat_from_seed(layers, m["seeds"][name], m["checksum"][name], m["alpha_scale_bias"]))
        nets[name] = net
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    rays = O.pack_rays(torch.from_numpy(g["rays_o"]), torch.from_numpy(g["rays_d"]), m["near"], m["far"]).to(DEV)
    y = T(g["target"])
    z = sampling.sample_coarse(rays, m["n_depth_samples"])
    assert torch.equal(z.cpu(), torch.from_numpy(g["z_vals"]))
    for train in (True, False):                          # the training forward (stores activations) and the inference forward
        raw = nets["coarse"].query(rays, z, train=train)
        loss, d_raw, rgb = render.composite_mse_backward(raw, z, rays, y, m["kwargs_white_bkgd"], need_rgb=True)
        assert float((rgb.cpu() - torch.from_numpy(g["rgb_coarse"])).abs().max()) < tol
        assert abs(float(loss) - m["loss_coarse"]) < tol * max(1.0, m["loss_coarse"])
    _, _, _, w, _ = render.composite(raw, z, rays, 0.0, True)
    z_imp, z_fine = sampling.importance_sample(z, w, m["N_importance"], u=T(g["u"]))
    # the reference's samples came from ITS weights (float32 numpy); ours from this precision's weights: same bins almost everywhere
    assert float((z_imp.cpu() - torch.from_numpy(g["z_imp"])).abs().median()) < 10 * tol
    # the sampler itself, on the reference's weights: exact bins, values to float32 noise
    z_imp2, z_fine2 = sampling.importance_sample(T(g["z_vals"]), T(g["weights"])[..., 0], m["N_importance"], u=T(g["u"]))
    np.testing.assert_allclose(z_imp2.cpu().numpy(), g["z_imp"], atol=5e-6, rtol=0)
    np.testing.assert_allclose(z_fine2.cpu().numpy(), g["z_fine"], atol=5e-6, rtol=0)
    zf = T(g["z_fine"])
    for train in (True, False):
        raw = nets["fine"].query(rays, zf, train=train)
        loss, _, rgb = render.composite_mse_backward(raw, zf, rays, y, False, need_rgb=True)       # Q8: no white background here
        assert float((rgb.cpu() - torch.from_numpy(g["fine_rgb"])).abs().max()) < tol
        assert abs(float(loss) - m["loss_fine"]) < tol * max(1.0, m["loss_fine"])


@pytest.mark.parametrize("kind", ["nan", "huge"])
def test_ngp_deterministic_scatter_surfaces_nonfinite_and_out_of_range_gradients(kind):
    """int64 fixed-point accumulators must not MASK a diverged run (advisor, round 3): a NaN addend, or one beyond the
    representable +-256, comes back from nerf_adam_step_ex as a NaN gradient -> NaN table entries, as float atomics give."""
    from nerf_meets_mlx_amd import sampling
    from nerf_meets_mlx_amd.rendering import render
    tr = _ngp(True)
    tr.train_step()
    rays, target = tr.sample_batch()
    z = sampling.sample_coarse(rays, 64)
    raw = tr.field.query(rays, z, train=True)
    _, d_raw, _ = render.composite_mse_backward(raw, z, rays, target, True)
    if kind == "nan":
        d_raw[3, 5, :] = float("nan")
    else:
        d_raw *= 1e12
    _, g_tab = tr.field.backward(d_raw)
    assert g_tab.dtype == torch.int64
    before = tr.field.enc.tables.clone()
    assert torch.isfinite(before).all()
    tr._opt.update(tr.field.table, g_tab.view(-1), grad_scale=1.0, zero_grads=True)
    tr.field._grad_clean = True
    torch.cuda.synchronize()
    t = tr.field.enc.tables
    assert torch.isnan(t).any(), "a non-finite / out-of-range table gradient was silently turned into finite integers"
    untouched = (t == before)
    assert bool(untouched.any())                          # entries the batch never touched keep their values
    assert int(tr.field.enc.grad.abs().max()) == 0        # consumed and cleared all the same


def test_ngp_deterministic_scatter_single_negative_saturated_addend_is_nan():
    """The poisoned window is symmetric (advisor, round 4): ONE addend below -256 on an otherwise untouched entry must come back as
    NaN, like one above +256 does -- not as a finite -256 gradient.  Round 6 (advisor, round 5): saturated addends sit at
    +-1.5 x 2^60 units, half a window OUTSIDE (-2^60, 2^60), so that ordinary addends of either sign on the SAME entry (the
    common case on coarse levels) cannot pull the sum back inside: a second sample in the same cell adds +-100 per corner."""
    import ctypes as C
    from nerf_meets_mlx_amd import _native as N
    L, log2T, F = 2, 8, 2
    res = (C.c_int * L)(4, 8)
    for sign, other in ((-1.0, 0.0), (1.0, 0.0), (-1.0, 100.0), (-1.0, -100.0), (1.0, 100.0), (1.0, -100.0)):
        # sample 0 carries the saturating gradient; samples 1..3 sit at the same point (same eight corners, same weights) with an
        # ordinary gradient of +-100 each: 3 x 100 x w <= 300 x w with every corner weight w <= 0.4 here -> |sum| < 128
        x = torch.tensor([[0.3, 0.6, 0.2]] * 4, device=DEV)
        d_out = torch.zeros(4, L * F, device=DEV)
        d_out[0, 0] = sign * 1e9                              # feature 0 of level 0: every corner weight x 1e9 is far beyond 256
        d_out[1:, 0] = other
        acc = torch.zeros(L, 1 << log2T, F, dtype=torch.int64, device=DEV)
        N.check(N.lib().nerf_hashgrid_backward_ex(N.ptr(x), 4, N.ptr(d_out), L, log2T, F, res, 0, L, 1, N.ptr(acc), N.stream()))
        touched = acc[0, :, 0] != 0
        assert 1 <= int(touched.sum()) <= 8
        assert bool((acc[0, touched, 0] * int(sign) >= (1 << 60)).all()), (sign, other)       # outside the window WITH the ordinary addends
        if other == 0.0:
            assert bool((acc[0, touched, 0] * int(sign) == (1 << 60) + (1 << 59)).all())     # a corner alone on its entry: exactly +-1.5 x 2^60
        params = torch.zeros(acc.numel(), device=DEV)
        m, v = torch.zeros_like(params), torch.zeros_like(params)
        N.check(N.lib().nerf_adam_step_ex(N.ptr(params), N.ptr(acc), N.ptr(m), N.ptr(v), params.numel(), 1e-3, 0.9, 0.99, 1e-8, 1, 1, 1.0,
                                          1, 1, N.stream()))
        torch.cuda.synchronize()
        p = params.view(L, 1 << log2T, F)
        assert bool(torch.isnan(p[0, touched, 0]).all()), (sign, other)
        assert bool(torch.isfinite(p[0, ~touched, 0]).all()) and bool(torch.isfinite(p[1]).all())
        assert int(acc.abs().max()) == 0


# ------------------------------------------------------------------------------------------------ a11 adjoint: split-bf16 training
@pytest.mark.parametrize("B,n", [(64, 96), (37, 45), (300, 64)])
def test_split_bf16_training_kernels_vs_fp32_oracle(B, n):
    """NeRF(precision=22) trains on csrc/mlp_s16.hip (every float32 GEMM operand as a bf16 (hi, lo) pair, three bf16 MFMAs
    per product).  Against the fp32 oracle (torch autograd):
    (1) training forward <= 1e-4 of the output scale (the bar of the fp32 MFMA kernels), every stored activation <= 1e-4 of
        its layer's scale, the encodings to 1e-5 of theirs;
    (2) the ReLU decisions agree except for units whose pre-activation is ~0: < 1e-4 of the units per layer;
    (3) with the oracle's backward run on the KERNEL's ReLU decisions (masks=...), dW / db agree for EVERY tensor at rel-L2
        <= 1e-4 and rel-max <= 1e-3, every dZ at rel-L2 <= 1e-4 -- ten times inside the bars of the fp32 MFMA kernels
        (tests/test_gpu_round2.py::test_fp32_mode_backward_and_training_step: 1e-3 / 1e-2).  Free-running (oracle on its own
        decisions) the comparison measures the handful of flipped units, not arithmetic (a random-signed upstream gradient
        makes each flip count in full: tests/test_oracle_golden.py::test_oracle_gradient_noise_floor): <= 2e-2 there.
    (37, 45): a ragged last sample tile (1665 samples = 52 tiles + 1) and a workgroup with idle waves."""
    from tests.test_gpu_round2 import LAYER_NAMES, _model_pair, _rays, _rel_l2, _relmax
    from nerf_meets_mlx_amd.models.NeRF import debug_layer
    m, arch, flat = _model_pair(3, 1.5, precision=22)
    torch.manual_seed(5 + B)
    rays = _rays(B, 77)
    z = torch.sort(torch.rand(B, n) * 4 + 2, -1).values
    g = torch.randn(B, n, 4)
    raw = m.query(rays.to(DEV), z.to(DEV), train=True)
    grads = m.backward(g.to(DEV)).cpu()
    o, d, _, _, vd = O.decompose_ray_batch(rays)
    pos = o[:, None, :] + z[:, :, None] * d[:, None, :]
    # (1) + (2): free-running oracle
    fl0 = flat.clone().requires_grad_(True)
    taps = {}
    out = O.run_model(arch, O.unflatten_params(arch, fl0), pos, vd, taps=taps)
    (out * g).sum().backward()
    e_fwd = _relmax(raw.cpu(), out.detach())
    assert e_fwd < 1e-4, e_fwd
    # the inference kernel (split fp16) and the training kernel (split bf16) of the same model agree to the same bar
    assert _relmax(m.query(rays.to(DEV), z.to(DEV)).cpu(), raw.cpu()) < 1e-4
    masks, flips_max = {}, 0.0
    for li, name in enumerate(LAYER_NAMES):
        act = debug_layer(m, "acts", li).cpu()
        want = taps[name].detach()
        assert act.shape == want.shape and _relmax(act, want) < 1e-4, (name, _relmax(act, want))
        if name != "feature":
            masks[name] = act > 0
            flips = float((masks[name] != (want > 0)).float().mean())
            flips_max = max(flips_max, flips)
            assert flips < 1e-4, (name, flips)
    pe = debug_layer(m, "acts", 10).cpu()
    xe = O.embed(pos, vd)
    # 16 significand bits: 2^-17 of the value (the identity channels reach |x| ~ 6)
    assert _relmax(pe[:, :63], xe[:, :63]) < 1e-5 and float(pe[:, 63].abs().max()) == 0.0
    dpe = debug_layer(m, "acts", 11).cpu()
    assert _relmax(dpe[:, :27], xe[:, 63:]) < 1e-5 and float(dpe[:, 27:].abs().max()) == 0.0
    free = _rel_l2(grads, fl0.grad)
    assert free < 2e-2, free
    # (3): oracle backward on the kernel's ReLU decisions
    fl = flat.clone().requires_grad_(True)
    taps2 = {}
    out2 = O.run_model(arch, O.unflatten_params(arch, fl), pos, vd, masks=masks, taps=taps2)
    for t in taps2.values():
        t.retain_grad()
    (out2 * g).sum().backward()
    off, worst = 0, (0.0, None)
    for name, o_, i_ in arch.layer_shapes():
        for part, cnt in (("W", o_ * i_), ("b", o_)):
            a, b = grads[off:off + cnt], fl.grad[off:off + cnt]
            l2, mx = _rel_l2(a, b), _relmax(a, b)
            worst = max(worst, (l2, (name, part)))
            assert l2 < 1e-4 and mx < 1e-3, (name, part, l2, mx)
            off += cnt
    assert off == 595844
    for li, name in enumerate(LAYER_NAMES):
        dz = debug_layer(m, "dz", li).cpu()
        ref = taps2[name].grad if name == "feature" else taps2[name].grad * masks[name].float()
        assert _rel_l2(dz, ref) < 1e-4, (name, _rel_l2(dz, ref))
    print(f"[s16 B={B} n={n}] forward {e_fwd:.2e} of scale; ReLU flips <= {flips_max:.1e}; mask-aligned worst dW/db rel-L2 "
          f"{worst[0]:.2e} at {worst[1]}, total {_rel_l2(grads, fl.grad):.2e}; free-running total {free:.2e}")


def test_split_bf16_rows_entry_trainer_and_reproducibility():
    """precision 22: NeRF.forward(x, train=True) on embedded rows takes the same kernels; three Trainer iterations follow the
    fp32 OracleTrainer (losses within 1e-3, weights rel-L2 <= 1e-3); the gradient of one batch computed twice is bit-identical
    (plain-store split-K partial tiles + fixed-order reduce, as for the other precisions)."""
    from tests.test_gpu_round2 import _model_pair, _rays, _rel_l2, _relmax
    from nerf_meets_mlx_amd.dataset import synthetic
    from nerf_meets_mlx_amd.engine.trainer import Trainer
    m, arch, flat = _model_pair(3, 1.5, precision=22)
    rows = torch.randn(100, 90, generator=torch.Generator().manual_seed(3))
    want = O.nerf_forward(arch, O.unflatten_params(arch, flat), rows)
    assert _relmax(m.forward(rows.to(DEV), train=True).cpu(), want) < 1e-4
    assert _relmax(m.forward(rows.to(DEV)).cpu(), want) < 1e-4
    rays, z = _rays(200, 5).to(DEV), torch.sort(torch.rand(200, 64, generator=torch.Generator().manual_seed(1)) * 4 + 2, -1).values.to(DEV)
    g = torch.randn(200, 64, 4, generator=torch.Generator().manual_seed(2)).to(DEV)
    m.query(rays, z, train=True)
    g1 = m.backward(g).clone()
    m.grads.fill_(float("nan"))
    m.query(rays, z, train=True)
    g2 = m.backward(g)
    assert torch.equal(g1, g2) and torch.isfinite(g2).all()
    H = W = 24
    imgs, poses, _, _, K = synthetic.make_dataset(H, W, 3, seed=0, device=DEV)
    tr = Trainer(imgs, poses, K, N_rand=128, n_depth_samples=64, N_importance=128, seed=4, device=DEV, precision=22)
    ot = O.OracleTrainer(arch, 64, 128, seed=4)
    gen = torch.Generator().manual_seed(3)
    for it in range(3):
        r, t = tr.sample_batch()
        u = torch.rand(128, 128, generator=gen)
        lh = tr.train_step(r, t, u.to(DEV))
        lo = ot.step(r[:, 0:3].cpu(), r[:, 3:6].cpu(), t.cpu(), u)
        assert abs(float(lh["loss_coarse"]) - lo["loss_coarse"]) <= 1e-3 * abs(lo["loss_coarse"]), (it, lh, lo)
        assert abs(float(lh["loss_fine"]) - lo["loss_fine"]) <= 1e-3 * abs(lo["loss_fine"]), (it, lh, lo)
    # Adam without bias correction steps ~ lr * g / (|g| + eps): where |g| ~ eps a 1e-5 relative gradient difference is a
    # visible step difference; 3 steps of lr 5e-4 on weights of scale 0.06 bound the drift (measured 2.5e-4)
    assert _rel_l2(tr.coarse.params.cpu(), ot.pc.detach()) < 1e-3


def test_checkpoint_from_before_the_stateless_streams_is_refused_unless_opted_in(tmp_path):
    """A round-1/2 checkpoint (saved RNG states, no `seed`) cannot be continued bit-identically any more: load() says so
    instead of silently switching streams (advisor, round 3); adopting a saved seed also reseeds the evaluation generator."""
    from nerf_meets_mlx_amd.dataset import synthetic
    from nerf_meets_mlx_amd.engine.trainer import Trainer
    imgs, poses, _, _, K = synthetic.make_dataset(16, 16, 2, seed=0, device=DEV)
    a = Trainer(imgs, poses, K, N_rand=32, seed=11, device=DEV)
    a.train_step()
    sd = a.state_dict()
    legacy = {k: v for k, v in sd.items() if k != "seed"}
    legacy["rng_torch"] = torch.zeros(8, dtype=torch.uint8)
    b = Trainer(imgs, poses, K, N_rand=32, seed=99, device=DEV)
    with pytest.raises(ValueError, match="predates the stateless random streams"):
        b.load_state_dict(legacy)
    b.load_state_dict(legacy, allow_legacy_rng=True)
    assert b.it == 1 and b.seed == 99 and torch.equal(b.coarse.params, a.coarse.params)
    c = Trainer(imgs, poses, K, N_rand=32, seed=99, device=DEV)
    c.load_state_dict(sd)
    assert c.seed == 11
    rays = a.sample_batch()[0]
    assert torch.equal(torch.rand(4, device=DEV, generator=c.gen), torch.rand(4, device=DEV, generator=Trainer(imgs, poses, K, N_rand=32, seed=11, device=DEV).gen))
    path = a.save(str(tmp_path / "ck"))
    assert c.load(path) == 1


# ------------------------------------------------------------------------------------------------ a22: fp16 shadow tables
def test_ngp_half_shadow_tables_follow_the_float32_master():
    """configs[4]: the fused query gathers from an fp16 shadow image of the hash tables (4 bytes per entry pair, SURVEY 8(d)'s
    512 B of gathers per sample) that the tables' Adam pass writes next to the float32 master.  (1) shadow == fp16(master)
    after construction, after training steps, after a torch in-place edit and after load_flat; (2) the query through the
    shadow agrees with the query through the float32 tables to fp16 resolution of the table values (the MLP rounds the
    interpolated features to bf16 = 2^-9 anyway); (3) master tables and MLP of a run with and without the shadow stay
    within the tolerance the bf16 arithmetic of the step allows."""
    from nerf_meets_mlx_amd import sampling
    a, b = _ngp(True, half_tables=True, precision=16), _ngp(True, half_tables=False, precision=16)      # an option of the bf16 mode
    assert a.field.table.half is not None and b.field.table.half is None
    rays, target = a.sample_batch()
    z = sampling.sample_coarse(rays, 64)
    with torch.no_grad():                                   # values large enough to matter: the 1e-4 initialisation is all fp16 denormals
        a.field.enc.tables.mul_(3000.0); b.field.enc.tables.mul_(3000.0)
    ra, rb = a.field.query(rays, z), b.field.query(rays, z)
    assert torch.equal(a.field.table.half, a.field.enc.tables.view(-1).half())
    sc = float(rb.abs().max())
    assert float((ra - rb).abs().max()) <= 4e-3 * sc, float((ra - rb).abs().max()) / sc
    assert torch.equal(b.field.query(rays, z), rb)
    for _ in range(3):
        la, lb = a.train_step(), b.train_step()
        assert torch.equal(a.field.table.half, a.field.enc.tables.view(-1).half())      # written by nerf_adam_step_shadow
        assert abs(float(la["loss_coarse"]) - float(lb["loss_coarse"])) <= 2e-2 * float(lb["loss_coarse"])
    flat = b.field.enc.tables.view(-1).clone()
    a.field.table.load_flat(flat)
    a.field.query(rays, z)
    assert torch.equal(a.field.table.half, flat.half())


# ------------------------------------------------------------------------------------------------ north star, headline mode
def test_psnr_paired_ensemble_precision22_vs_reference_arithmetic():
    """The bench's headline mode (precision 22: split-bf16 training, split-fp16 rendering) against the reference's float32
    arithmetic (precision 32) on identical batches, 4 seeds fixed before any outcome (the first four alive-at-init seeds) x 400
    iterations: its per-forward error is 1e-5 of the output scale, so the trajectories stay together far longer than bf16's --
    at 200 iterations every seed is within 0.1 dB and the median within 0.03 dB; at 400 the mean is zero within its confidence
    interval widened by the target's 0.1 dB.  The 32-seed x 2500-iteration run of the same tool is
    profiles/r04_psnr_p22_vs_fp32_*.jsonl."""
    import argparse
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import psnr_ensemble as E
    a = argparse.Namespace(hw=100, views=12, test_views=2, n_rand=1024, n_importance=128, lrate_decay=500, no_quirks=False,
                           iters=400, every=200, dead_every=20, bridge_iters=0, resync=False, lead_precision=22)
    seeds = E.alive_seeds(4, True, 0)
    assert seeds == [4, 10, 18, 21]
    recs = []
    for sd in seeds:
        r, _ = E.run_seed(sd, a, emit=lambda line: None)
        recs += r
    stats = {st["ensemble_iter"]: st for st in E.summarise(recs)}          # key "bf16" = the lead arm = precision 22 here
    print({k: (round(v["mean_delta_db"], 3), round(v["ci95_half_width_db"], 3), round(v["median_delta_db"], 3),
               round(v["max_abs_delta_db"], 3)) for k, v in stats.items()})
    assert stats[200]["n"] == 4 and not stats[200]["seeds_non_finite"]
    assert stats[200]["max_abs_delta_db"] <= 0.1 and abs(stats[200]["median_delta_db"]) <= 0.03, stats[200]
    for it in (200, 400):
        st = stats[it]
        assert abs(st["mean_delta_db"]) <= 0.1 + st["ci95_half_width_db"], st
        assert st["mean_a"] > 11.0 and st["mean_b"] > 11.0


@pytest.mark.parametrize("B,n", [(1, 1), (1, 31), (3, 11), (1, 129), (5, 64)])
def test_split_bf16_training_kernels_ragged_and_tiny_sizes(B, n):
    """precision 22 at sizes below / across one 32-sample tile and one 4-wave workgroup (1, 31, 33, 129, 320 samples): the
    training forward equals the inference forward and the fp32 MFMA kernels to 1e-4 of the output scale, gradients are finite
    and agree with the fp32 kernels' (same batch) to 2e-3 rel-L2 (free-running: a flipped ReLU unit counts in full), samples
    past M contribute nothing (a NaN-poisoned gradient buffer is overwritten everywhere)."""
    from tests.test_gpu_round2 import _model_pair, _rays, _rel_l2, _relmax
    m22, _, _ = _model_pair(3, 1.5, precision=22)
    m32, _, _ = _model_pair(3, 1.5, precision=32)
    g = torch.Generator().manual_seed(100 * B + n)
    rays = _rays(B, 7).to(DEV)
    z = torch.sort(torch.rand(B, n, generator=g) * 4 + 2, -1).values.to(DEV)
    dr = torch.randn(B, n, 4, generator=g).to(DEV)
    r22 = m22.query(rays, z, train=True)
    m22.grads.fill_(float("nan"))
    g22 = m22.backward(dr).clone()
    r32 = m32.query(rays, z, train=True)
    g32 = m32.backward(dr)
    assert _relmax(r22.cpu(), r32.cpu()) < 1e-4 and _relmax(m22.query(rays, z).cpu(), r32.cpu()) < 1e-4
    assert torch.isfinite(g22).all()
    assert _rel_l2(g22, g32) < 2e-3, _rel_l2(g22, g32)


def test_split_bf16_training_propagates_non_finite_gradients():
    """A NaN upstream gradient must surface as NaN parameter gradients, not as silently finite numbers.  (A NaN in a sample
    POSITION does not survive the first ReLU in any precision of this library -- hardware max(NaN, 0) returns 0, the fp32 MFMA
    kernels included -- where torch.relu in the oracle propagates it: an input-validation difference, not an arithmetic one;
    noted in DESIGN.md 10.)"""
    from tests.test_gpu_round2 import _model_pair, _rays
    m, _, _ = _model_pair(3, 1.5, precision=22)
    rays = _rays(40, 9).to(DEV)
    z = torch.sort(torch.rand(40, 64, generator=torch.Generator().manual_seed(3)) * 4 + 2, -1).values.to(DEV)
    dr = torch.randn(40, 64, 4, generator=torch.Generator().manual_seed(4)).to(DEV)
    m.query(rays, z, train=True)
    dr[7, 3, 1] = float("nan")
    assert torch.isnan(m.backward(dr)).any()
    dr[7, 3, 1] = float("inf")
    m.query(rays, z, train=True)
    assert not torch.isfinite(m.backward(dr)).all()


def test_split_kernels_repeat_bit_for_bit_under_concurrent_load():
    """tools/stress_fp32.py at precision 22, short: 24 x (training forward, backward, inference forward) at 1024 x 192 with bf16
    kernels of a second model on another stream every other iteration; every result bit-identical to the first (the LDS ring's
    counted waits with M0-clobbering DMA runs, the dW kernel's half-tile staging; 400 iterations at 4096 x 192: 0 mismatches)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_fp32
    assert stress_fp32.run(iters=24, B=1024, n=192, dev=DEV, verbose=False, precision=22) == 0
