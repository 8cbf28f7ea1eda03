"""GPU parity tests added in round 2 (same rules as tests/test_gpu_parity.py: HIP path through the C ABI vs the
oracle on identical seeded inputs; tolerances stated per test).

  * per-layer parity of the fused MLP's stored activations and dZ (nerf_mlp_debug_read) and a MASK-ALIGNED gradient
    check at rel-L2 <= 2e-2 / rel-max <= 8e-2 for every tensor;
  * the fp32 reference-precision mode of the fused chain (<= 1e-4 of the output scale vs the fp32 oracle);
  * PSNR parity over a training run (|delta| <= 0.1 dB at every checkpoint, >= 20 dB reached);
  * render() of a full 800 x 800 frame at chunk 32768 (configs[2]'s path) against the oracle on sampled pixels;
  * SSIM, checkpoint round trips (both trainers, .npz on disk, bit-identical continuation), the entrypoint's
    --i_weights / --ft_path / --no_reload flags, stale-activation and stale-weight guards, gather bounds;
  * RCCL with two ranks (needs two visible devices; skipped with the reason printed on a one-GPU box).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from nerf_meets_mlx_amd import _native
    assert _native.lib().nerf_abi_version() == 3


def _relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _rays(B, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * 4.0
    d = -o / 4.0 + 0.25 * torch.randn(B, 3, generator=g)
    return O.pack_rays(o, d, 2.0, 6.0)


def _model_pair(seed=0, scale=1.0, precision=16):
    from nerf_meets_mlx_amd.models.NeRF import NeRF
    arch = O.NerfArch()
    m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=seed, precision=precision)
    flat = O.flatten_params(arch, O.init_params(arch, seed))
    assert torch.equal(m.params.cpu(), flat)
    if scale != 1.0:
        flat = flat * scale
        m.load_flat(flat)
    return m, arch, flat


LAYER_NAMES = [f"pos{i}" for i in range(8)] + ["feature", "dir0"]


# ------------------------------------------------------------------------------ a11: every layer, and the adjoint
@pytest.mark.parametrize("B,n", [(6, 40), (64, 96), (300, 64)])
def test_mlp_every_layer_and_mask_aligned_backward(B, n):
    """The training kernels keep every layer's activation and dZ (fragment blocks); `nerf_mlp_debug_read` decodes them.
    (1) every layer's stored activation == the bf16-emulating oracle's, to bf16 resolution (rel-to-max 1e-2; the stored
        value is bf16, the oracle's tap is fp32 before the next layer's rounding: 2^-9 relative on top of the
        accumulation-order noise);
    (2) the ReLU decisions agree except for units whose pre-activation is ~0 (a last-bit difference in a bf16 input
        moves them across zero): < 0.5 % of the units per layer;
    (3) with the oracle's backward run on the KERNEL's ReLU decisions (masks=...), dW / db agree for EVERY tensor at
        rel-L2 <= 5e-3 and rel-max <= 2e-2 (measured: <= 1.1e-3 worst tensor, 5e-4 whole gradient), and every layer's dZ
        at rel-L2 <= 5e-3 -- four times inside the 2e-2 / 8e-2 the round-1 review asked for.  Without the alignment a
        random-signed upstream gradient makes this comparison measure mask flips, not arithmetic: the oracle's own
        gradient moves by 1.2-4.4 % rel-L2 when its input positions are perturbed by 3e-7 (tests/test_oracle_golden.py::
        test_oracle_gradient_noise_floor), which is why test_mlp_backward_matches_autograd carries a 3e-2 / 6e-2 bar."""
    from nerf_meets_mlx_amd.models.NeRF import debug_layer
    m, arch, flat = _model_pair(3, 1.5)
    torch.manual_seed(1000 * B + n)
    rays = _rays(B, 77)
    z = torch.sort(torch.rand(B, n) * 4 + 2, -1).values
    g = torch.randn(B, n, 4)
    raw = m.query(rays.to(DEV), z.to(DEV), train=True)
    grads = m.backward(g.to(DEV)).cpu()
    o, d, _, _, vd = O.decompose_ray_batch(rays)
    pos = o[:, None, :] + z[:, :, None] * d[:, None, :]
    # (1) + (2): free-running oracle
    taps = {}
    out = O.run_model(arch, O.unflatten_params(arch, flat), pos, vd, emulate_bf16=True, taps=taps)
    assert _relmax(raw.cpu(), out) < 1e-2
    masks = {}
    for li, name in enumerate(LAYER_NAMES):
        got = debug_layer(m, "acts", li).cpu()
        want = taps[name]
        assert got.shape == want.shape, (name, got.shape, want.shape)
        assert _relmax(got, want) < 1e-2, (name, _relmax(got, want))
        if name != "feature":
            masks[name] = got > 0
            flips = float((masks[name] != (want > 0)).float().mean())
            assert flips < 5e-3, (name, flips)
    pe = debug_layer(m, "acts", 10).cpu()
    x = O.embed(pos, vd)
    assert _relmax(pe[:, :63], x[:, :63]) < 1e-2 and float(pe[:, 63:].abs().max()) == 0.0
    dpe = debug_layer(m, "acts", 11).cpu()
    assert _relmax(dpe[:, :27], x[:, 63:]) < 1e-2 and float(dpe[:, 27:].abs().max()) == 0.0
    # (3): oracle backward on the kernel's ReLU decisions
    fl = flat.clone().requires_grad_(True)
    taps2 = {}
    out2 = O.run_model(arch, O.unflatten_params(arch, fl), pos, vd, emulate_bf16=True, masks=masks, taps=taps2)
    for t in taps2.values():
        t.retain_grad()
    (out2 * g).sum().backward()
    want = fl.grad
    off = 0
    worst = (0.0, None)
    for name, o_, i_ in arch.layer_shapes():
        for part, cnt in (("W", o_ * i_), ("b", o_)):
            a, b = grads[off:off + cnt], want[off:off + cnt]
            l2, mx = _rel_l2(a, b), _relmax(a, b)
            worst = max(worst, (l2, (name, part)))
            assert l2 < 5e-3 and mx < 2e-2, (name, part, l2, mx)
            off += cnt
    assert off == 595844
    assert _rel_l2(grads, want) < 2e-3, _rel_l2(grads, want)
    for li, name in enumerate(LAYER_NAMES):
        dz = debug_layer(m, "dz", li).cpu()
        ref = taps2[name].grad if name == "feature" else taps2[name].grad * masks[name].float()
        assert _rel_l2(dz, ref) < 5e-3, (name, _rel_l2(dz, ref))
    print(f"[layers B={B} n={n}] worst dW/db rel-L2 {worst[0]:.2e} at {worst[1]}; total {_rel_l2(grads, want):.2e}")


# ------------------------------------------------------------------------------ fp32 reference-precision mode
@pytest.mark.parametrize("quirk", [True, False])
def test_fp32_mode_forward_matches_fp32_oracle(quirk):
    """NeRF(precision=32) (nerf_mlp_arch.precision): the fused chain on v_mfma_f32_32x32x2_f32 with fp32 operands -- the
    reference's own arithmetic (MLX computes in float32).  <= 1e-4 of the output scale against the fp32 oracle
    (fp32 accumulation-order noise over 12 layers; the encodings use the hardware sin with fp32 range reduction,
    <= 2e-6 absolute per channel)."""
    m, arch, flat = _model_pair(1, 1.5, precision=32)
    p = O.unflatten_params(arch, flat)
    for B, n in [(3, 64), (5, 192), (1, 7), (100, 64)]:
        rays = _rays(B, 10 + B)
        z = torch.sort(torch.rand(B, n) * 4 + 2, -1).values
        raw = m.query(rays.to(DEV), z.to(DEV), ref_quirks=quirk).cpu()
        o, d, _, _, vd = O.decompose_ray_batch(rays)
        pos = o[:, None, :] + z[:, :, None] * d[:, None, :]
        ref = O.run_model(arch, p, pos, vd, ref_quirks=quirk)
        assert raw.shape == (B, n, 4)
        assert _relmax(raw, ref) < 1e-4, (B, n, _relmax(raw, ref))


def test_fp32_mode_render_rays_eval_end_to_end():
    """The whole render path (coarse pass -> compositing -> inverse-CDF importance sampling -> sort -> fine pass ->
    compositing, rendering/render.py:164-241) in fp32 mode against the fp32 oracle on the same rays and uniforms:
    rgb / acc within 5e-4 absolute (bf16 mode: 3e-2), the integer bin indices of the importance sampler -- hence z_fine --
    identical except where a uniform lands within float32 noise of a CDF step."""
    from nerf_meets_mlx_amd.models.NeRF import NeRF
    from nerf_meets_mlx_amd.rendering import render
    arch = O.NerfArch()
    mc = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=4, precision=32)
    mf = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=5, precision=32)
    for mm in (mc, mf):
        mm.load_flat(mm.params * 1.5)
    B = 1500
    rays = _rays(B, 21)
    u = torch.rand(B, 128, generator=torch.Generator().manual_seed(2))
    out = render.render_rays_fused(rays.to(DEV), mc, mf, 64, 128, u=u.to(DEV), white_bkgd=True, ref_quirks=True,
                                   with_coarse=True)
    ref = O.render_rays_eval(arch, O.unflatten_params(arch, mc.params.cpu()), O.unflatten_params(arch, mf.params.cpu()),
                             rays, 64, 128, u, white_bkgd=True)
    assert float((out["rgb_map"].cpu() - ref["rgb_map"]).abs().max()) < 5e-4
    assert float((out["acc_map"].cpu().reshape(-1) - ref["acc_map"].reshape(-1)).abs().max()) < 5e-4
    assert float((out["rgb_coarse"].cpu() - ref["rgb_coarse"]).abs().max()) < 5e-4


def test_fp32_mode_backward_and_training_step():
    """fp32 mode, adjoint: dW / db against torch autograd through the fp32 oracle, rel-L2 <= 1e-3 and rel-max <= 1e-2
    for every tensor (no bf16 anywhere, so no rounding-induced ReLU flips: what remains is fp32 summation order over
    thousands of samples), then three Trainer iterations against the OracleTrainer with losses within 1e-3 relative."""
    from nerf_meets_mlx_amd.dataset import synthetic
    from nerf_meets_mlx_amd.engine.trainer import Trainer
    m, arch, flat = _model_pair(3, 1.5, precision=32)
    B, n = 64, 96
    torch.manual_seed(5)
    rays = _rays(B, 77)
    z = torch.sort(torch.rand(B, n) * 4 + 2, -1).values
    g = torch.randn(B, n, 4)
    raw = m.query(rays.to(DEV), z.to(DEV), train=True)
    grads = m.backward(g.to(DEV)).cpu()
    fl = flat.clone().requires_grad_(True)
    o, d, _, _, vd = O.decompose_ray_batch(rays)
    pos = o[:, None, :] + z[:, :, None] * d[:, None, :]
    out = O.run_model(arch, O.unflatten_params(arch, fl), pos, vd)
    (out * g).sum().backward()
    assert _relmax(raw.cpu(), out.detach()) < 1e-4
    off = 0
    for name, o_, i_ in arch.layer_shapes():
        for part, cnt in (("W", o_ * i_), ("b", o_)):
            a, b = grads[off:off + cnt], fl.grad[off:off + cnt]
            assert _rel_l2(a, b) < 1e-3 and _relmax(a, b) < 1e-2, (name, part, _rel_l2(a, b), _relmax(a, b))
            off += cnt
    # every stored layer (float32 [tile][row][32] stores decoded by nerf_mlp_debug_read) against the oracle's taps:
    # activations to 1e-4 of the layer's scale, dZ to 1e-3 rel-L2 (a handful of units flip at |z| ~ 1e-7)
    from nerf_meets_mlx_amd.models.NeRF import debug_layer
    fl2 = flat.clone().requires_grad_(True)
    taps = {}
    out2 = O.run_model(arch, O.unflatten_params(arch, fl2), pos, vd, taps=taps)
    for t in taps.values():
        t.retain_grad()
    (out2 * g).sum().backward()
    for li, name in enumerate(LAYER_NAMES):
        act = debug_layer(m, "acts", li).cpu()
        assert act.shape == taps[name].shape and _relmax(act, taps[name].detach()) < 1e-4, (name, _relmax(act, taps[name].detach()))
        dz = debug_layer(m, "dz", li).cpu()
        ref = taps[name].grad if name == "feature" else taps[name].grad * (taps[name].detach() > 0).float()
        assert _rel_l2(dz, ref) < 1e-3, (name, _rel_l2(dz, ref))
    pe = debug_layer(m, "acts", 10).cpu()
    xe = O.embed(pos, vd)
    assert float((pe[:, :63] - xe[:, :63]).abs().max()) < 2e-6 and float(pe[:, 63].abs().max()) == 0.0
    # NeRF.forward(x) on embedded rows takes the same fp32 kernels
    rows = torch.randn(100, 90, generator=torch.Generator().manual_seed(3))
    got = m.forward(rows.to(DEV)).cpu()
    assert _relmax(got, O.nerf_forward(arch, O.unflatten_params(arch, flat), rows)) < 1e-4
    H = W = 24
    imgs, poses, _, _, K = synthetic.make_dataset(H, W, 3, seed=0, device=DEV)
    tr = Trainer(imgs, poses, K, N_rand=128, n_depth_samples=64, N_importance=128, seed=4, device=DEV, precision=32)
    ot = O.OracleTrainer(arch, 64, 128, seed=4)
    gen = torch.Generator().manual_seed(3)
    for it in range(3):
        r, t = tr.sample_batch()
        u = torch.rand(128, 128, generator=gen)
        lh = tr.train_step(r, t, u.to(DEV))
        lo = ot.step(r[:, 0:3].cpu(), r[:, 3:6].cpu(), t.cpu(), u)
        assert abs(float(lh["loss_coarse"]) - lo["loss_coarse"]) <= 1e-3 * abs(lo["loss_coarse"]), (it, lh, lo)
        assert abs(float(lh["loss_fine"]) - lo["loss_fine"]) <= 1e-3 * abs(lo["loss_fine"]), (it, lh, lo)
    assert _rel_l2(tr.coarse.params.cpu(), ot.pc.detach()) < 1e-4


# ------------------------------------------------------------------------------ PSNR parity over a training run
def test_same_weights_parity_over_a_training_run():
    """The arithmetic half of the north-star PSNR claim (the trajectory half is the paired ensemble of
    tests/test_gpu_round3.py::test_psnr_paired_ensemble_bf16_vs_reference_arithmetic; round 2's 300-iteration lockstep check
    against one oracle trajectory is gone: it was a statement about one chaotic trajectory and had been tuned to a
    configuration where it passed).  At EVERY checkpoint of a 2000-iteration run of the bf16 HIP trainer, on the SAME
    weights: the bf16 HIP renderer and the fp32 oracle renderer give the same held-out PSNR within 0.05 dB, and the two
    compute the same gradient of the current batch (cosine >= 0.98 for both networks, losses within 2 %) -- equal
    arithmetic along the whole run; and the run learns the scene (>= 20 dB held-out), so this is not a statement about
    untrained networks.  The configs[2] / configs[1]-scale runs (800^2 / 400^2, N_rand 1024, 5000 iterations) are
    tools/psnr_parity.py, logged under profiles/."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import psnr_parity
    show = lambda r: print("[psnr]", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()
                                      if k.startswith(("iter", "psnr", "delta", "grad"))}, flush=True)
    # the HIP trainer runs; the oracle is evaluated on ITS weights at every checkpoint
    recs = psnr_parity.run(hw=200, n_rand=4096, iters=2000, every=100, views=24, test_views=2, n_importance=128,
                           oracle_device="cuda", eval_chunk=20000, cross=True, oracle_until=0,
                           emit=lambda r: show(r) if "iter" in r else None)
    assert len(recs) == 20
    for r in recs:
        assert abs(r["delta_db_same_weights"]) <= 0.05, r
        assert r["grad_coarse_cos"] >= 0.98 and r["grad_fine_cos"] >= 0.98, r
        assert abs(r["loss_coarse_hip_at_w"] - r["loss_coarse_oracle_at_w"]) <= 0.02 * r["loss_coarse_oracle_at_w"], r
        assert abs(r["loss_fine_hip_at_w"] - r["loss_fine_oracle_at_w"]) <= 0.02 * r["loss_fine_oracle_at_w"], r
    assert max(r["psnr_hip"] for r in recs[-3:]) >= 20.0, [r["psnr_hip"] for r in recs]


# ------------------------------------------------------------------------------ a19 at configs[2] size
@pytest.mark.parametrize("precision", [None, 16])
def test_render_full_frame_800_chunk_32768(precision):
    """`render(H, W, K, chunk=32768, c2w=...)` (rendering/render.py:268-345) on a full 800 x 800 frame (configs[2] size): 20
    chunks, the last one ragged (640000 = 19 x 32768 + 17408); output structure of the reference ([rgb, disp, acc, extras]),
    and the rgb / acc / z_vals of 2048 sampled pixels against the oracle's render_rays_eval on the same rays and uniforms.
    precision None = the constructors' default (22): the FLOAT32 oracle, 1e-4 of the output scale on the fine pass's raw
    values and 1e-3 on rgb / acc (the inverse-CDF samples move with the coarse weights; measured ~1e-5);
    16: the bf16-emulating oracle at 3e-2."""
    from nerf_meets_mlx_amd.models import embedding
    from nerf_meets_mlx_amd.models.NeRF import NeRF, NetworkQuery
    from nerf_meets_mlx_amd.rendering import render
    H = W = 800
    K = np.array([[1111.111, 0, 400.0], [0, 1111.111, 400.0], [0, 0, 1]])
    c2w = O.pose_spherical(30.0, -30.0, 4.0)[:3, :4]
    arch = O.NerfArch()
    pk = {} if precision is None else {"precision": precision}
    mc = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=4, **pk)
    mf = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=5, **pk)
    assert mc.precision == (22 if precision is None else 16)
    emu = precision == 16
    tol = 3e-2 if emu else 1e-3
    for mm in (mc, mf):                      # weights x1.5 so that sigma varies along the rays
        mm.load_flat(mm.params * 1.5)
    gen = torch.Generator(device=DEV).manual_seed(11)
    u = torch.rand(H * W, 128, device=DEV, generator=gen)
    fp, _ = embedding.get_embedder(10); fd, _ = embedding.get_embedder(4)
    kw = dict(network_coarse=mc, network_fine=mf, network_query_fn=NetworkQuery(fp, fd, 65536), n_depth_samples=64,
              N_importance=128, white_bkgd=True, use_viewdirs=True, ndc=False, near=2.0, far=6.0,
              render_rays_func=render.render_rays_eval, perturb=0.0, raw_noise_std=0.0, u=u)
    rgb, disp, acc, extras = render.render(H, W, K, chunk=1024 * 32, c2w=c2w, **kw)
    assert rgb.shape == (H, W, 3) and disp.shape[:2] == (H, W) and acc.shape[:2] == (H, W)
    assert extras["z_vals"].shape == (H, W, 64) and extras["weights"].shape[:3] == (H, W, 64)
    assert torch.isfinite(rgb).all()
    pick = torch.randperm(H * W, generator=torch.Generator().manual_seed(0))[:2048]
    ro, rd = O.get_rays(H, W, K, c2w)
    rays = O.pack_rays(ro.reshape(-1, 3)[pick], rd.reshape(-1, 3)[pick], 2.0, 6.0)
    pc = O.unflatten_params(arch, mc.params.cpu()); pf = O.unflatten_params(arch, mf.params.cpu())
    ref = O.render_rays_eval(arch, pc, pf, rays, 64, 128, u.cpu()[pick], white_bkgd=True, emulate_bf16=emu)
    got_rgb = rgb.reshape(-1, 3).cpu()[pick]
    assert float((got_rgb - ref["rgb_map"]).abs().max()) < tol, float((got_rgb - ref["rgb_map"]).abs().max())
    assert float((acc.reshape(-1).cpu()[pick] - ref["acc_map"].reshape(-1)).abs().max()) < tol
    assert torch.equal(extras["z_vals"].reshape(-1, 64).cpu()[pick], ref["z_vals"])
    if not emu:
        # the network itself at full size: raw of the 2048 rays' coarse samples through the same fused query, 1e-4 of scale
        rays_d = rays.to(DEV)
        raw = mc.query(rays_d, extras["z_vals"].reshape(-1, 64)[pick.to(DEV)].contiguous()).cpu()
        o, d, _, _, vd = O.decompose_ray_batch(rays)
        pos = o[:, None, :] + ref["z_vals"][:, :, None] * d[:, None, :]
        want = O.run_model(arch, pc, pos, vd)
        e = float((raw - want).abs().max() / want.abs().max())
        assert e < 1e-4, e
        print(f"[800 x 800 frame, default precision] rgb max abs err {float((got_rgb - ref['rgb_map']).abs().max()):.1e}, raw {e:.1e} of scale")


# ------------------------------------------------------------------------------ SSIM (8f-4)
@pytest.mark.parametrize("quirk", [True, False])
def test_ssim_matches_oracle(quirk):
    """ops/metric.py:20-64 finished.  float32 windowed moments in a different summation order than torch's conv2d:
    |delta| <= 2e-5 on the mean (values are in [-1, 1])."""
    from nerf_meets_mlx_amd.ops.metric import SSIM
    g = torch.Generator().manual_seed(1)
    for shape in [(1, 3, 64, 64), (2, 3, 37, 53), (1, 1, 11, 11), (3, 4, 100, 12)]:
        a = torch.rand(*shape, generator=g)
        b = (a + 0.1 * torch.randn(*shape, generator=g)).clamp(0, 1)
        for w in (11, 5):
            if min(shape[2:]) < w:
                continue
            got, got_cs = SSIM(quirk)(a.to(DEV), b.to(DEV), w_size=w, full=True)
            want, want_cs = O.ssim(a, b, w_size=w, full=True, ref_quirks=quirk)
            assert abs(float(got) - float(want)) < 2e-5 and abs(float(got_cs) - float(want_cs)) < 2e-5, (shape, w)
            per = SSIM(quirk)(a.to(DEV), b.to(DEV), w_size=w, size_average=False).cpu()
            np.testing.assert_allclose(per.numpy(), O.ssim(a, b, w_size=w, size_average=False, ref_quirks=quirk).numpy(), atol=2e-5)
    x = torch.rand(1, 3, 32, 32, generator=g)
    assert abs(float(SSIM()(x.to(DEV), x.to(DEV))) - 1.0) < 1e-6
    big = x * 255.0                                               # dynamic range 255 branch (:24)
    assert abs(float(SSIM()(big.to(DEV), (big * 0.9).to(DEV))) - float(O.ssim(big, big * 0.9))) < 2e-5
    with pytest.raises(ValueError):
        SSIM()(x.to(DEV)[0], x.to(DEV)[0])
    with pytest.raises(ValueError):
        SSIM()(x.to(DEV)[:, :, :8, :8], x.to(DEV)[:, :, :8, :8])          # image smaller than the window


# ------------------------------------------------------------------------------ checkpoints (8f-3)
def _mini_trainer(kind="nerf", shared=True, seed=4):
    from nerf_meets_mlx_amd.dataset import synthetic
    from nerf_meets_mlx_amd.engine.ngp import NGPTrainer
    from nerf_meets_mlx_amd.engine.trainer import Trainer
    imgs, poses, _, _, K = synthetic.make_dataset(20, 20, 3, seed=0, device=DEV)
    if kind == "ngp":
        return NGPTrainer(imgs, poses, K, N_rand=64, n_depth_samples=64, seed=7, device=DEV, log2_hashmap_size=12)
    return Trainer(imgs, poses, K, N_rand=64, n_depth_samples=64, N_importance=128, seed=seed, device=DEV, ref_quirks=shared)


@pytest.mark.parametrize("kind,shared", [("nerf", True), ("nerf", False), ("ngp", True)])
def test_checkpoint_npz_roundtrip_continues(tmp_path, kind, shared):
    """save() mid-training -> load() into a FRESH trainer: the restored state is BIT-identical to the saved one --
    parameters, Adam (m, v) per state key (shared, or coarse / fine, or mlp / tables), step counts (the NGP loop uses
    bias correction) and iteration (LR schedule; it is also the whole RNG state: every random draw of iteration `it` is a
    function of (seed, rank, it), so the next batch is the same rays, bit for bit) --
    and training continues on the same trajectory: BIT-identical parameters after two more iterations for the 8 x 256
    trainer (its kernels are deterministic since the split-K partial tiles are reduced in a fixed order; the reported
    loss scalar is still an atomic sum, compared at 1e-5), fp32 summation-noise level for the hash-grid trainer (its
    table gradients are scattered with float atomics, whose order differs from launch to launch).
    The path is given without ".npz" on purpose (np.savez appends it; load() must find it)."""
    a = _mini_trainer(kind, shared)
    for _ in range(3):
        a.train_step()
    path = a.save(str(tmp_path / "ck"))
    assert path.endswith(".npz") and os.path.exists(path)
    snap = {k: m.params.clone() for k, m in a._checkpoint_buffers().items()}
    snap_adam = {k: [t.clone() for t in v] for k, v in a.opt.state.items()}
    b = _mini_trainer(kind, shared)
    assert b.load(str(tmp_path / "ck")) == 3
    assert b.it == 3 and set(b.opt.state) == set(a.opt.state) and b.opt.step_count == a.opt.step_count
    assert all(v >= 3 for v in b.opt.step_count.values())
    for k, t in snap.items():
        assert torch.equal(t, b._checkpoint_buffers()[k].params), k
    for k, (m_, v_) in snap_adam.items():
        assert torch.equal(m_, b.opt.state[k][0]) and torch.equal(v_, b.opt.state[k][1]), k
    ra, ta = a.sample_batch()
    rb, tb = b.sample_batch()
    assert torch.equal(ra, rb) and torch.equal(ta, tb)                     # same image, same pixels: RNG streams restored
    assert torch.equal(a.train_uniforms(4), b.train_uniforms(4)) and a.image_choice() == b.image_choice()
    cont = [a.train_step() for _ in range(2)]
    again = [b.train_step() for _ in range(2)]
    for x, y in zip(cont, again):
        for k in x:
            assert abs(float(x[k]) - float(y[k])) <= 1e-5 * abs(float(x[k])), (k, float(x[k]), float(y[k]))
    for k, ma in a._checkpoint_buffers().items():
        # both trainers are bit-reproducible: the 8 x 256 one since its split-K partial tiles are reduced in a fixed order, the
        # hash-grid one since its table gradient is accumulated in int64 fixed point with integer atomics (the default)
        assert torch.equal(ma.params, b._checkpoint_buffers()[k].params), (kind, k)
    # state_dict round trip in memory too
    c = _mini_trainer(kind, shared)
    c.load_state_dict(a.state_dict())
    la, lc = float(a.train_step()["loss_coarse"]), float(c.train_step()["loss_coarse"])
    assert abs(la - lc) <= 1e-5 * abs(la)


def test_entrypoint_checkpoint_flags(tmp_path):
    """--i_weights / --basedir / --expname / --ft_path / --no_reload (config_parser.py:7-8,25-26,75) through the
    headless entrypoint: periodic .npz checkpoints, automatic resume from the newest one, explicit file, and opt-out."""
    from nerf_meets_mlx_amd.entrypoints import test_nerf
    base = ["--basedir", str(tmp_path), "--expname", "run", "--i_weights", "4"]
    kw = dict(hw_synthetic=40, n_train_synthetic=3, log_every=1, seed=4)      # N_rand = 1024 of 1600 pixels
    full = test_nerf.main(None, max_iter=10, argv=["--basedir", str(tmp_path), "--expname", "full", "--i_weights", "0"], **kw)
    r1 = test_nerf.main(None, max_iter=6, argv=base, **kw)
    assert [os.path.basename(p) for p in r1["checkpoints"]] == ["000004.npz"] and r1["resumed_from"] is None
    r2 = test_nerf.main(None, max_iter=10, argv=base, **kw)                        # resumes at iteration 4
    assert r2["resumed_from"].endswith("000004.npz") and r2["losses"][0][0] == 5
    assert [os.path.basename(p) for p in r2["checkpoints"]] == ["000008.npz"]
    for x, y in zip(r2["losses"][-1][1:], full["losses"][-1][1:]):                 # same trajectory as the uninterrupted run
        assert abs(x - y) <= 1e-4 * abs(y), (r2["losses"][-1], full["losses"][-1])   # (float atomics: not bitwise)
    assert torch.equal(r2["trainer"].fine.params, full["trainer"].fine.params)      # resumed == uninterrupted, bit for bit
    assert torch.equal(r2["trainer"].coarse.params, full["trainer"].coarse.params)
    r3 = test_nerf.main(None, max_iter=5, argv=base + ["--no_reload"], **kw)
    assert r3["resumed_from"] is None and r3["losses"][0][0] == 1
    r4 = test_nerf.main(None, max_iter=9, argv=base + ["--no_reload", "--ft_path", r1["checkpoints"][0]], **kw)
    assert r4["resumed_from"] == r1["checkpoints"][0] and r4["losses"][0][0] == 5
    # round 6: the run log (engine/runlog.py; __test_nerf.py:298-299 keeps loss.item() per iteration): the four runs of
    # {basedir}/run appended to ONE log.jsonl -- a `run` record each, a `train` record per logged iteration whose losses are
    # the returned ones, a positive whole-job rays/s, the scheduled learning rate
    from nerf_meets_mlx_amd.engine import runlog
    recs = runlog.read(os.path.join(str(tmp_path), "run", "log.jsonl"))
    assert r2["log"] == os.path.join(str(tmp_path), "run", "log.jsonl")
    runs = [r for r in recs if r["kind"] == "run"]
    assert len(runs) == 4 and [r["start_it"] for r in runs] == [0, 4, 0, 4] and runs[1]["resumed_from"].endswith("000004.npz")
    assert all(r["precision"] == 22 and r["n_rand"] == 1024 and r["world_size"] == 1 for r in runs)
    train = [r for r in recs if r["kind"] == "train"]
    assert [r["it"] for r in train] == list(range(1, 7)) + list(range(5, 11)) + list(range(1, 6)) + list(range(5, 10))
    by_it = {r["it"]: r for r in train[6:12]}
    for it, lc, lf in r2["losses"]:
        assert by_it[it]["loss_coarse"] == lc and by_it[it]["loss_fine"] == lf
        assert abs(by_it[it]["lr"] - 5e-4 * 0.1 ** ((it - 1) / 500000)) < 1e-12 and by_it[it]["rays_per_s"] > 0
        assert abs(by_it[it]["psnr_fine"] + 10 * np.log10(lf)) < 1e-9
    full_log = runlog.read(full["log"])
    assert [r["it"] for r in full_log if r["kind"] == "train"] == list(range(1, 11))
    off = test_nerf.main(None, max_iter=2, argv=["--basedir", str(tmp_path), "--expname", "nolog", "--i_weights", "0"], write_log=False, **kw)
    assert off["log"] is None and not os.path.exists(os.path.join(str(tmp_path), "nolog", "log.jsonl"))


# ------------------------------------------------------------------------------ guards
def test_stale_activation_and_stale_weight_guards():
    from nerf_meets_mlx_amd import autograd as A
    from nerf_meets_mlx_amd.ops import index
    m, arch, flat = _model_pair(4, 1.5)
    rays = _rays(8, 3).to(DEV)
    params = m.trainable()
    out1 = A.render_rays_grad(rays, m, 64, True)
    out2 = A.render_rays_grad(rays, m, 64, False)            # second graph on the same model, same B*n
    with pytest.raises(RuntimeError, match="overwritten"):
        (out1["rgb_map"].sum() + out2["rgb_map"].sum()).backward()
    params.grad = None
    out = A.render_rays_grad(rays, m, 64, True)
    out["rgb_map"].sum().backward()                          # a single graph still works
    assert torch.isfinite(params.grad).all() and float(params.grad.abs().max()) > 0
    # in-place edits through the parameter views re-pack the bf16 image without mark_updated()
    m2, _, _ = _model_pair(5, 1.5)
    z = torch.sort(torch.rand(8, 64) * 4 + 2, -1).values.to(DEV)
    before = m2.query(rays, z)
    with torch.no_grad():
        m2.parameters()["rgb_linear"]["bias"].add_(1.0)
    after = m2.query(rays, z)
    assert float((after[..., :3] - before[..., :3] - 1.0).abs().max()) < 2e-2 and torch.equal(after[..., 3], before[..., 3])
    # option validation and the refusals of the new entry points
    from nerf_meets_mlx_amd import _native as NV
    L = NV.lib()
    assert L.nerf_set_option(b"mlp_precision", 32) == -3 and b"nerf_mlp_arch.precision" in L.nerf_last_error()   # ABI 3: not an option
    assert L.nerf_set_option(b"ring_split", 3) == -3
    assert L.nerf_get_option(b"mlp_precision") == -2 ** 31 and L.nerf_get_option(b"ring_split") == 1
    zz = torch.rand(4, 8, device=DEV)
    assert L.nerf_add_noise_z(NV.ptr(zz), NV.ptr(zz), 4, 8, 1.0, NV.ptr(zz), None) == -2        # in place is refused
    # precision belongs to the model: an arch with an unknown precision, or fp32 for a model without fp32 kernels, is refused
    import ctypes as C
    bad = NV.MlpArch(8, 256, 63, 27, 4, 1, 4, 24)
    assert L.nerf_mlp_packed_bytes(C.byref(bad)) == -1
    ngp32 = NV.MlpArch(2, 64, 32, 16, -1, 1, 4, 32)                      # the 2 x 64 model has no fp32-MFMA kernels (16 / 22 only)
    assert L.nerf_mlp_packed_bytes(C.byref(ngp32)) == -1
    assert L.nerf_mlp_packed_bytes(C.byref(NV.MlpArch(8, 256, 40, 0, 4, 0, 3, 32))) > 0      # round 5: the image model has
    from nerf_meets_mlx_amd.models.NeRF import NeRF
    with pytest.raises(ValueError):
        NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=0, precision=8)
    with pytest.raises(ValueError):
        NeRF(n_layers=2, width_layers=64, channel_input=32, channel_input_views=16, list_skip_connection_layers=[],
             is_use_view_directions=True, device=DEV, seed=0, precision=32).packed()
    # gather_rows: out-of-range indices never read, they give NaN rows
    src = torch.arange(12, dtype=torch.float32, device=DEV).reshape(4, 3)
    got = index.gather_rows(src, torch.tensor([0, 3, 4, -1], device=DEV))
    assert torch.equal(got[:2].cpu(), src[[0, 3]].cpu()) and torch.isnan(got[2:]).all()


# ------------------------------------------------------------------------------ C1 with two ranks (RCCL)
def test_rccl_two_ranks_allreduce_and_training():
    """torch.distributed "nccl" (= RCCL) and libnerf_hip's own communicator with TWO ranks on two devices: summed
    gradients, then two Trainer iterations whose weights must be bit-identical on both ranks.  RCCL refuses two ranks on
    one device, so on a one-GPU box this is skipped (the 2-rank logic is covered with gloo in
    tests/test_gpu_multirank.py and tests/test_parallel_gloo.py)."""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"RCCL needs one device per rank: {n} device visible on this box (driver SCALE runs exercise N > 1)")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631",
                        os.path.join(ROOT, "tests", "_rccl_two_ranks.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "RCCL2 OK" in r.stdout, r.stdout[-3000:]


# ------------------------------------------------------------------------------ 8f-2: the on-disk format, end to end
def _write_blender_scene(root, hw=40, n_train=6, n_val=2, n_test=2):
    """A Blender-synthetic dataset on disk made from the synthetic scene: `transforms_{train,val,test}.json`
    (camera_angle_x + frames[file_path, transform_matrix]) and straight-alpha RGBA PNGs, plus `configs/lego.txt` with the 14
    keys `update_NeRF_args` copies (config_parser.py:106-119)."""
    import json
    from PIL import Image
    from nerf_meets_mlx_amd.dataset import synthetic
    os.makedirs(os.path.join(root, "configs"))
    data = os.path.join(root, "data", "lego")
    poses = synthetic.train_poses(n_train + n_val + n_test, seed=3)
    K, _ = synthetic.intrinsics(hw, hw)
    k = 0
    for split, cnt in (("train", n_train), ("val", n_val), ("test", n_test)):
        os.makedirs(os.path.join(data, split))
        frames = []
        for i in range(cnt):
            c2w = poses[k]; k += 1
            # rgb accumulated over the teacher field and its opacity, un-premultiplied: the loader composites on white itself
            j, ii = torch.meshgrid(torch.arange(hw, dtype=torch.float64), torch.arange(hw, dtype=torch.float64), indexing="ij")
            dirs = torch.stack([(ii - K[0, 2]) / K[0, 0], -(j - K[1, 2]) / K[1, 1], -torch.ones_like(ii)], -1).reshape(-1, 3)
            d = (dirs @ c2w[:3, :3].double().T).float(); o = c2w[:3, 3].float()
            t = torch.linspace(2.0, 6.0, 192)
            sigma, rgb = synthetic.teacher_field(o + d[:, None, :] * t[None, :, None])
            delta = torch.cat([t[1:] - t[:-1], torch.tensor([1e10])]) * d.norm(dim=-1, keepdim=True)
            alpha = 1.0 - torch.exp(-sigma * delta)
            T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
            w = alpha * T
            acc = w.sum(1, keepdim=True)
            col = (w[..., None] * rgb).sum(1) / acc.clamp_min(1e-6)
            rgba = torch.cat([col, acc], -1).clamp(0, 1).reshape(hw, hw, 4)
            Image.fromarray((rgba.numpy() * 255.0 + 0.5).astype(np.uint8), "RGBA").save(os.path.join(data, split, f"r_{i}.png"))
            frames.append({"file_path": f"./{split}/r_{i}", "transform_matrix": c2w.tolist()})
        with open(os.path.join(data, f"transforms_{split}.json"), "w") as fp:
            json.dump({"camera_angle_x": synthetic.CAMERA_ANGLE_X, "frames": frames}, fp)
    with open(os.path.join(root, "configs", "lego.txt"), "w") as fp:
        fp.write(f"expname = blender_paper_lego\nbasedir = {root}/logs\ndatadir = ./data/lego\ndataset_type = blender\n\n"
                 "no_batching = True\n\nuse_viewdirs = True\nwhite_bkgd = True\nlrate_decay = 500\n\n"
                 "N_samples = 64\nN_importance = 128\nN_rand = 1024\n\nprecrop_iters = 500\nprecrop_frac = 0.5\n\nhalf_res = True\n")


def test_entrypoint_on_a_blender_dataset_on_disk(tmp_path):
    """`entrypoints/__test_nerf.py:25-342` from the on-disk format the reference reads (dataset/dataloader.py:20-111):
    configs/lego.txt -> load_config -> update_NeRF_args (called with both arguments, Q1; string booleans, Q2) ->
    load_blender_data WITHOUT half_res although the file says `half_res = True` (Q3) -> white composite -> K from
    camera_angle_x -> the device trainer -> a rendered frame.  Loss falls over 60 iterations; the loaded images equal the
    PNGs; the trainer's intrinsics are the reference's focal formula."""
    from PIL import Image
    from nerf_meets_mlx_amd.entrypoints import test_nerf
    root = str(tmp_path / "NeRF")
    os.makedirs(root)
    _write_blender_scene(root)
    res = test_nerf.main(root, max_iter=60, log_every=10, render_every=60, n_render_poses=1, seed=4)
    assert res["log"] == os.path.join(root, "logs", "blender_paper_lego", "log.jsonl")      # basedir / expname of lego.txt
    tr = res["trainer"]
    from nerf_meets_mlx_amd.engine import runlog
    recs = runlog.read(res["log"])                          # {basedir}/{expname from lego.txt}/log.jsonl: 6 train records + the i_render frame's PSNR
    ev = [r for r in recs if r["kind"] == "eval"]
    assert [r["it"] for r in recs if r["kind"] == "train"] == [10, 20, 30, 40, 50, 60] and len(ev) == 1 and ev[0]["it"] == 60
    assert 5.0 < ev[0]["psnr"] < 60.0 and ev[0]["view"] == 3 and recs[0]["dataset"] == root
    assert tr.H == 40 and tr.W == 40 and tr.N_rand == 1024 and tr.n == 64 and tr.N == 128          # native resolution (Q3)
    assert abs(tr.K[0, 0] - 0.5 * 40 / np.tan(0.5 * 0.6911112070083618)) < 1e-9 and tr.images.shape == (6, 40, 40, 3)
    png = np.asarray(Image.open(os.path.join(root, "data", "lego", "train", "r_0.png"))).astype(np.float32) / 255.0
    want = png[..., :3] * png[..., 3:] + (1.0 - png[..., 3:])
    np.testing.assert_allclose(tr.images[0].cpu().numpy(), want, atol=1e-6)
    assert res["resumed_from"] is None                     # a config file forces no_reload (config_parser.py:120)
    losses = [l[1] for l in res["losses"]]
    assert all(np.isfinite(losses)) and min(losses[-3:]) < 0.75 * losses[0], losses         # per-batch losses: noisy
    assert len(res["frames"]) == 1 and res["frames"][0].shape == (40, 40, 3) and len(res["video"]) == 1


# ------------------------------------------------------------------------------ a13 + a20 fused for the training step
@pytest.mark.parametrize("n,white", [(64, True), (192, False), (7, True), (300, True), (1, True), (2, False), (65, True), (1024, False)])
def test_composite_mse_backward_equals_the_staged_kernels(n, white):
    """`nerf_composite_mse_backward` (raw2outputs + MSE + their adjoint in one launch per ray, what the trainers call) is
    bit-identical in rgb and d_raw to nerf_composite_forward -> nerf_mse_loss_grad -> nerf_composite_backward, and its
    loss agrees to float32 summation order; against the oracle's autograd the gradient is within 1e-5 relative."""
    from nerf_meets_mlx_amd.ops.metric import mse_loss_grad
    from nerf_meets_mlx_amd.rendering import render
    B = 133 if n not in (1, 1024) else 3                               # also the one-wave-per-ray corner cases: n = 1, n = 16 x 64
    g = torch.Generator().manual_seed(n)
    rays = _rays(B, 5)
    z = torch.sort(torch.rand(B, n, generator=g) * 4 + 2, -1).values
    raw = torch.randn(B, n, 4, generator=g)
    raw[..., 3] = raw[..., 3] * (3.0 if n <= 300 else 0.05)            # signed sigma: the un-ReLU'd transmittance matters (Q10)
    target = torch.rand(B, 3, generator=g)
    rd, zd, rwd, td = rays.to(DEV), z.to(DEV), raw.to(DEV), target.to(DEV)
    rgb, _, _, _, _ = render.composite(rwd, zd, rd, 0.0, white)
    loss_s, d_rgb = mse_loss_grad(rgb, td)
    d_raw_s = render.composite_backward(rwd, zd, rd, d_rgb, white)
    loss_f, d_raw_f, rgb_f = render.composite_mse_backward(rwd, zd, rd, td, white, need_rgb=True)
    assert torch.equal(rgb_f, rgb) and torch.equal(d_raw_f, d_raw_s)
    assert abs(float(loss_f) - float(loss_s)) <= 1e-6 * abs(float(loss_s))
    rw = raw.clone().double().requires_grad_(True)
    o_rgb, *_ = O.raw2outputs(rw, z.double(), rays[:, 3:6].double(), 0.0, white)
    O.mse(o_rgb, target.double()).backward()
    assert _rel_l2(d_raw_f.cpu(), rw.grad) < 1e-5
    half = render.composite_mse_backward(rwd, zd, rd, td, white, grad_scale=0.5)[1]
    assert torch.equal(half, 0.5 * d_raw_f)


def test_sample_batch_equals_permutation_raygen_gather():
    """`nerf_sample_batch` (pixel permutation + ray generation + target gather in one launch) is bit-identical to the
    three separate entry points, integer pixel indices included (the contract's "bit-exact ray/pixel indices")."""
    from nerf_meets_mlx_amd.ops import index
    from nerf_meets_mlx_amd.rendering import ray
    H, W = 37, 53
    K = np.array([[60.0, 0, W / 2], [0, 61.0, H / 2], [0, 0, 1]])
    c2w = O.pose_spherical(70.0, -25.0, 4.0)[:3, :4]
    img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(0)).to(DEV)
    for n, seed, off in [(1024, 12345, 0), (H * W, 7, 0), (5, 99, 100)]:
        rays, target, idx = ray.sample_batch(H, W, K, c2w, 2.0, 6.0, img, n, seed, off, return_idx=True)
        want_idx = index.pixel_permutation(n, H * W, seed, off, DEV)
        assert torch.equal(idx, want_idx) and len(set(idx.tolist())) == n
        assert torch.equal(rays, ray.gen_rays(H, W, K, c2w, 2.0, 6.0, want_idx))
        assert torch.equal(target, index.gather_rows(img.reshape(-1, 3), want_idx))
    with pytest.raises(ValueError):
        ray.sample_batch(H, W, K, c2w, 2.0, 6.0, img, H * W + 1, 0)
