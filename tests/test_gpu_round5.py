"""Round-5 GPU tests: the reference's float32 TOLERANCE for the two network shapes that had bf16 kernels only.

  * image-fitting model (entrypoints/__viser_image_learning.py:198-208: 8 x 256, in 40, no view head, out 3) and the 2 x 64
    hash-grid model (BASELINE configs[4]) at `precision=22` (csrc/mlp_s16x.hip: split-bf16 operands, three bf16 MFMAs per
    float32 product) and -- image model -- `precision=32` (csrc/mlp32.hip: float32 operands on the fp32 MFMA), against the
    NON-emulating float32 oracle: forward <= 1e-4 of the output scale, every stored activation <= 1e-4 of its layer's scale,
    dW / db (and dL/dx of the 2 x 64 model) <= 1e-3 rel-L2 per tensor with the oracle's backward run on the kernel's ReLU
    decisions (measured ~3e-5: asserted at 1e-4 where the kernel is the split-bf16 one).
  * ImageFitter and NGPTrainer (defaults: precision 22) against the float32 oracle loops: loss 1e-3, mask-aligned gradients
    1e-3 rel-L2, then a few Adam iterations.
  * the drop-in surface defaults to the reference-tolerance arithmetic.
"""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"


def _relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _image_pair(precision, scale=1.5, out_ch=3):
    from nerf_meets_mlx_amd.models.NeRF import NeRF
    arch = O.NerfArch(channel_input=40, channel_input_views=0, channel_output=out_ch, use_viewdirs=False)
    m = NeRF(channel_input=40, channel_input_views=0, channel_output=out_ch, is_use_view_directions=False, device=DEV, seed=0,
             precision=precision)
    flat = O.flatten_params(arch, O.init_params(arch, 0)) * scale
    m.load_flat(flat)
    return m, arch, flat


def _small_pair(precision, scale=1.5):
    from nerf_meets_mlx_amd.models.NeRF import NeRF
    arch = O.NerfArch(channel_input=32, channel_input_views=16, n_layers=2, width=64, skips=(), use_viewdirs=True)
    m = NeRF(n_layers=2, width_layers=64, channel_input=32, channel_input_views=16, list_skip_connection_layers=[],
             is_use_view_directions=True, device=DEV, seed=0, precision=precision)
    flat = O.flatten_params(arch, O.init_params(arch, 0)) * scale
    m.load_flat(flat)
    return m, arch, flat


def _grad_check(arch, grads, want, tol_l2, tol_max, n_params):
    off, worst = 0, (0.0, None)
    for name, o_, i_ in arch.layer_shapes():
        for part, cnt in (("W", o_ * i_), ("b", o_)):
            a, b = grads[off:off + cnt], want[off:off + cnt]
            l2, mx = _rel_l2(a, b), _relmax(a, b)
            worst = max(worst, (l2, (name, part)))
            assert l2 < tol_l2 and mx < tol_max, (name, part, l2, mx)
            off += cnt
    assert off == n_params
    return worst


@pytest.mark.parametrize("precision", [22, 32])
@pytest.mark.parametrize("M", [1, 33, 2500])
def test_image_model_forward_at_reference_tolerance(precision, M):
    m, arch, flat = _image_pair(precision)
    x = torch.randn(M, 40, generator=torch.Generator().manual_seed(11 + M))
    want = O.nerf_forward(arch, O.unflatten_params(arch, flat), x)
    got = m.forward(x.to(DEV)).cpu()
    assert got.shape == (M, 3)
    assert _relmax(got, want) < 1e-4, _relmax(got, want)
    assert _relmax(m.forward(x.to(DEV), train=True).cpu(), want) < 1e-4


@pytest.mark.parametrize("precision,out_ch,M", [(22, 3, 2500), (22, 4, 1000), (22, 1, 97), (32, 3, 2500), (32, 4, 333)])
def test_image_model_gradients_vs_fp32_oracle(precision, out_ch, M):
    """dW / db of the image model against torch autograd through the float32 oracle, the oracle's backward on the KERNEL's ReLU
    decisions (stored activations > 0), so that the comparison measures arithmetic and not the handful of units whose
    pre-activation is ~0 (tests/test_oracle_golden.py::test_oracle_gradient_noise_floor)."""
    from nerf_meets_mlx_amd.models.NeRF import debug_layer
    m, arch, flat = _image_pair(precision, out_ch=out_ch)
    gen = torch.Generator().manual_seed(7 + M)
    x, g = torch.randn(M, 40, generator=gen), torch.randn(M, out_ch, generator=gen)
    out = m.forward(x.to(DEV), train=True).cpu()
    grads = m.backward(g.to(DEV)).cpu()
    taps = {}
    want = O.nerf_forward(arch, O.unflatten_params(arch, flat), x, taps=taps)
    assert _relmax(out, want) < 1e-4
    masks, flips = {}, 0.0
    for l in range(8):
        act = debug_layer(m, "acts", l).cpu()
        ref = taps[f"pos{l}"]
        assert act.shape == ref.shape and _relmax(act, ref) < 1e-4, (l, _relmax(act, ref))
        masks[f"pos{l}"] = act > 0
        flips = max(flips, float((masks[f"pos{l}"] != (ref > 0)).float().mean()))
    assert flips < 1e-3, flips
    fl = flat.clone().requires_grad_(True)
    (O.nerf_forward(arch, O.unflatten_params(arch, fl), x, masks=masks) * g).sum().backward()
    tol = 1e-4 if precision == 22 else 1e-3
    worst = _grad_check(arch, grads, fl.grad, tol, 10 * tol, 481280 + 257 * out_ch)
    # bit-reproducible (plain-store split-K partial tiles + fixed-order reduce, as for every other mode)
    m.grads.fill_(float("nan"))
    m.forward(x.to(DEV), train=True)
    assert torch.equal(m.backward(g.to(DEV)).cpu(), grads)
    print(f"[image p{precision} out {out_ch} M={M}] forward {_relmax(out, want):.1e}; flips <= {flips:.1e}; worst dW/db rel-L2 {worst[0]:.1e} at {worst[1]}")


@pytest.mark.parametrize("M", [1, 33, 4100])
def test_small_model_forward_at_reference_tolerance(M):
    m, arch, flat = _small_pair(22)
    x = torch.randn(M, 48, generator=torch.Generator().manual_seed(21 + M))
    want = O.nerf_forward(arch, O.unflatten_params(arch, flat), x)
    got = m.forward(x.to(DEV)).cpu()
    assert got.shape == (M, 4)
    assert _relmax(got, want) < 1e-4, _relmax(got, want)
    assert _relmax(m.forward(x.to(DEV), train=True).cpu(), want) < 1e-4


def test_small_model_gradients_and_input_grads_vs_fp32_oracle():
    from nerf_meets_mlx_amd.models.NeRF import debug_layer
    m, arch, flat = _small_pair(22)
    M = 4100
    gen = torch.Generator().manual_seed(5)
    x, g = torch.randn(M, 48, generator=gen), torch.randn(M, 4, generator=gen)
    out = m.forward(x.to(DEV), train=True).cpu()
    grads, d_x = m.backward(g.to(DEV), need_input_grad=True)
    grads, d_x = grads.cpu().clone(), d_x.cpu()
    taps = {}
    want = O.nerf_forward(arch, O.unflatten_params(arch, flat), x, taps=taps)
    assert _relmax(out, want) < 1e-4
    masks = {}
    for name, layer in (("pos0", 0), ("pos1", 1), ("dir0", 9)):
        act = debug_layer(m, "acts", layer).cpu()
        assert act.shape == taps[name].shape and _relmax(act, taps[name]) < 1e-4, (name, _relmax(act, taps[name]))
        masks[name] = act > 0
        assert float((masks[name] != (taps[name] > 0)).float().mean()) < 1e-3
    assert _relmax(debug_layer(m, "acts", 8).cpu(), taps["feature"]) < 1e-4
    assert _relmax(debug_layer(m, "acts", 10).cpu(), x[:, :32]) < 1e-5 and _relmax(debug_layer(m, "acts", 11).cpu(), x[:, 32:]) < 1e-5
    fl = flat.clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    (O.nerf_forward(arch, O.unflatten_params(arch, fl), xr, masks=masks) * g).sum().backward()
    worst = _grad_check(arch, grads, fl.grad, 1e-4, 1e-3, 13188)
    assert d_x.shape == (M, 32)
    assert _rel_l2(d_x, xr.grad[:, :32]) < 1e-4, _rel_l2(d_x, xr.grad[:, :32])
    m.forward(x.to(DEV), train=True)
    assert torch.equal(m.backward(g.to(DEV)).cpu(), grads)
    print(f"[2x64 p22] forward {_relmax(out, want):.1e}; worst dW/db rel-L2 {worst[0]:.1e} at {worst[1]}; d_x {_rel_l2(d_x, xr.grad[:, :32]):.1e}")


def _make_alive(tr, orc):
    """The seed-0 2 x 64 network is DEAD at initialisation under the reference's un-activated sigma (sigma in [-0.22, -0.10] on
    every sample -> all compositing weights 0, rgb = white, every gradient exactly 0: DESIGN.md section 7): a gradient comparison
    on it is 0 == 0.  Lift the alpha bias (flat index 10496 = LN::P_BA) on both sides."""
    with torch.no_grad():
        orc.p[10496] += 0.6
    tr.field.mlp.load_flat(orc.p.detach())


def test_image_fitter_default_tracks_the_float32_oracle_loop():
    """entrypoints/__viser_image_learning.py:198-236 headless at the DEFAULT precision against the float32 oracle loop: first
    loss to 1e-3 (measured ~1e-6), mask-aligned gradient of the first batch <= 1e-3 rel-L2, losses of six Adam steps within
    1 % (Adam's first steps are lr * sign(g): the trajectories separate at the units with |g| ~ 0, not through arithmetic)."""
    from nerf_meets_mlx_amd.entrypoints.image_learning import ImageFitter
    from nerf_meets_mlx_amd.models.NeRF import debug_layer
    from nerf_meets_mlx_amd.ops.metric import mse_loss_grad
    H = W = 40
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    img = torch.stack([0.5 + 0.5 * torch.sin(6 * xx), yy, 0.5 + 0.5 * torch.cos(5 * (xx + yy))], -1)
    fit = ImageFitter(img.to(DEV), batch_downsample_factor=4, seed=0)
    assert fit.model.precision == 22
    orc = O.OracleImageFitter(seed=0)
    assert torch.equal(fit.model.params.cpu(), orc.p.detach())
    batches = list(fit.batch_iterate())
    X, y = batches[0]
    pred = fit.model.forward(fit.embed(X), train=True)
    loss, d_pred = mse_loss_grad(pred, y)
    g = fit.model.backward(d_pred).cpu().clone()
    masks = {f"pos{l}": debug_layer(fit.model, "acts", l).cpu() > 0 for l in range(8)}
    want_loss = O.mse(orc.forward(X.cpu()), y.cpu())
    want_loss = want_loss.detach()
    assert abs(float(loss) - float(want_loss)) < 1e-3 * float(want_loss), (float(loss), float(want_loss))
    gw, = torch.autograd.grad(O.mse(orc.forward(X.cpu(), masks=masks), y.cpu()), orc.p)
    assert _rel_l2(g, gw) < 1e-3, _rel_l2(g, gw)
    hip, ora = [], []
    for X, y in batches[:3] + batches[:3]:
        hip.append(float(fit.step(X, y)))
        ora.append(orc.step(X.cpu(), y.cpu())[0])
    for a, b in zip(hip, ora):
        assert abs(a - b) < 1e-2 * b, (hip, ora)
    assert hip[-1] < hip[0]
    pr = fit.predict()
    assert pr.shape == (H, W, 3) and torch.isfinite(pr).all()
    print(f"[image fitter p22] loss {float(loss):.6f} vs {float(want_loss):.6f}; gradient rel-L2 {_rel_l2(g, gw):.1e}; losses {hip} vs {ora}")


def test_ngp_default_field_gradients_and_training_track_the_float32_oracle():
    """configs[4] at the DEFAULT precision (22: float32 table gathers + interpolation, split-bf16 2 x 64 MLP) against the
    float32 OracleNGP: loss 1e-3, MLP and table gradients <= 1e-3 rel-L2 with the oracle on the kernel's ReLU decisions, six Adam
    iterations on identical batches within 2 %."""
    from nerf_meets_mlx_amd import sampling
    from nerf_meets_mlx_amd.dataset import synthetic
    from nerf_meets_mlx_amd.engine.ngp import NGPTrainer
    from nerf_meets_mlx_amd.models.NeRF import debug_layer
    from nerf_meets_mlx_amd.ops.metric import mse_loss_grad
    from nerf_meets_mlx_amd.rendering import render
    H = W = 32
    imgs, poses, _, hwf, K = synthetic.make_dataset(H, W, 2, seed=0, device=DEV)
    kw = dict(n_levels=16, min_res=4, max_res=128, n_features_per_level=2, log2_hashmap_size=12, hash_init_scale=0.5)
    tr = NGPTrainer(imgs, poses, K, N_rand=256, n_depth_samples=32, seed=0, device=DEV, **kw)
    assert tr.field.precision == 22 and tr.field.mlp.precision == 22 and tr.field.table.half is None
    orc = O.OracleNGP(tr.field.enc.tables.cpu(), tr.field.enc.scaled_res, seed=0, n_samples=32)
    assert torch.equal(tr.field.mlp.params.cpu(), orc.p.detach())
    _make_alive(tr, orc)
    rays, target = tr.sample_batch()
    ro, rd, tg = rays[:, 0:3].cpu(), rays[:, 3:6].cpu(), target.cpu()
    z = sampling.sample_coarse(rays, 32)
    # rows computed inside the forward kernel vs rows through HBM: same float32 features, same split, same MFMAs
    raw_f = tr.field.query(rays, z, train=True, fused=True)
    raw_u = tr.field.query(rays, z, train=True, fused=False)
    assert torch.equal(raw_f, raw_u)
    taps = {}
    rgb_o = orc.render(O.pack_rays(ro, rd, 2.0, 6.0), taps=taps)
    assert _relmax(raw_f.cpu().reshape(-1, 4), taps["raw"].detach().reshape(-1, 4)) < 1e-4
    raw = tr.field.query(rays, z, train=True)
    rgb = render.composite(raw, z, rays, 0.0, True)[0]
    loss, d_rgb = mse_loss_grad(rgb, target)
    g_mlp, _ = tr.field.backward(render.composite_backward(raw, z, rays, d_rgb, True))
    g_mlp, g_tab = g_mlp.cpu().clone(), tr.field.table_grad().cpu().clone()
    masks = {n: debug_layer(tr.field.mlp, "acts", l).cpu() > 0 for n, l in (("pos0", 0), ("pos1", 1), ("dir0", 9))}
    lo, gp, gt = orc.loss_and_grads(ro, rd, tg, masks=masks)
    assert abs(float(loss) - float(lo)) < 1e-3 * float(lo), (float(loss), float(lo))
    assert float(gp.norm()) > 1e-3 and float(gt.norm()) > 1e-4, "dead network: the comparison would be 0 == 0"
    assert _rel_l2(g_mlp, gp) < 1e-3, _rel_l2(g_mlp, gp)
    assert _rel_l2(g_tab, gt) < 1e-3, _rel_l2(g_tab, gt)
    hip, ora = [], []
    for it in range(6):
        rays, target = tr.sample_batch()
        hip.append(float(tr.train_step(rays, target)["loss_coarse"]))
        ora.append(orc.step(rays[:, 0:3].cpu(), rays[:, 3:6].cpu(), target.cpu()))
    for a, b in zip(hip, ora):
        assert abs(a - b) < 2e-2 * b, (hip, ora)
    fixed = [float(tr.train_step(rays, target)["loss_coarse"]) for _ in range(8)]      # the SAME batch eight times: the loss falls
    assert fixed[-1] < fixed[0], fixed
    img = tr.render_frame(poses[0], shard=False)
    assert img.shape == (H, W, 3) and torch.isfinite(img).all()
    print(f"[ngp p22] loss {float(loss):.6f} vs {float(lo):.6f}; g_mlp {_rel_l2(g_mlp, gp):.1e}, g_tab {_rel_l2(g_tab, gt):.1e}; losses {hip} vs {ora}")


def test_ngp_ray_major_and_sample_major_fused_queries_agree():
    """The inference query walks (32 adjacent rays x 1 depth) tiles by default and (1 ray x 32 depths) tiles with
    ngp_ray_major = 0: same value per sample, bit for bit, at precision 22 as at 16."""
    from nerf_meets_mlx_amd import _native, sampling
    from nerf_meets_mlx_amd.engine.ngp import HashNeRF
    f = HashNeRF(device=DEV, seed=3, log2_hashmap_size=14, hash_init_scale=0.5)
    g = torch.Generator().manual_seed(1)
    o = torch.nn.functional.normalize(torch.randn(70, 3, generator=g), dim=-1) * 4.0
    rays = O.pack_rays(o, -o / 4.0 + 0.2 * torch.randn(70, 3, generator=g), 2.0, 6.0).to(DEV)
    z = sampling.sample_coarse(rays, 48)
    a = f.query(rays, z)
    _native.check(_native.lib().nerf_set_option(b"ngp_ray_major", 0))
    try:
        b = f.query(rays, z)
    finally:
        _native.check(_native.lib().nerf_set_option(b"ngp_ray_major", 1))
    assert torch.equal(a, b) and torch.equal(a, f.query(rays, z, train=True))


def test_drop_in_surface_defaults_to_the_reference_tolerance():
    """`create_NeRF(args)` of the reference returns float32 networks (models/NeRF.py:86-112): the mirror's constructors default
    to precision 22 (float32 tolerance); bf16 is opt-in; half_tables is refused outside the bf16 mode."""
    from nerf_meets_mlx_amd import config_parser as CP
    from nerf_meets_mlx_amd.engine.ngp import HashNeRF
    from nerf_meets_mlx_amd.engine.trainer import Trainer
    from nerf_meets_mlx_amd.models.NeRF import NeRF, create_NeRF
    assert NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=0).precision == 22
    args = CP.config_parser().parse_args(args=[])
    args.use_viewdirs = True; args.N_importance = 128
    kw, _, _, _ = create_NeRF(args, device=DEV)
    assert kw["network_coarse"].precision == 22 and kw["network_fine"].precision == 22
    imgs = torch.rand(2, 8, 8, 3)
    poses = torch.stack([O.pose_spherical(10.0, -30.0, 4.0), O.pose_spherical(100.0, -40.0, 4.0)])
    K = np.array([[11.0, 0, 4], [0, 11.0, 4], [0, 0, 1]])
    assert Trainer(imgs, poses, K, N_rand=16, device=DEV).coarse.precision == 22
    assert HashNeRF(device=DEV, log2_hashmap_size=10).mlp.precision == 22
    assert HashNeRF(device=DEV, log2_hashmap_size=10, precision=16).table.half is not None
    with pytest.raises(ValueError):
        HashNeRF(device=DEV, log2_hashmap_size=10, half_tables=True)
    with pytest.raises(ValueError):                      # the 2 x 64 model has no fp32-MFMA kernels: refused, never a fallback
        NeRF(n_layers=2, width_layers=64, channel_input=32, channel_input_views=16, list_skip_connection_layers=[],
             is_use_view_directions=True, device=DEV, seed=0, precision=32).packed()


@pytest.mark.parametrize("kind,precision", [("view", 22), ("view", 16), ("image", 22), ("image", 16), ("small", 22), ("small", 16)])
@pytest.mark.parametrize("M", [1, 32, 97, 5 * 32, 37 * 32 + 5, 8192 + 33])
def test_weight_gradients_of_the_two_dw_launch_forms_agree(kind, precision, M):
    """"dw22_variant" / "dw16_variant" 1 (default): the 256 x 256 jobs on the one-wave-per-SIMD kernel (csrc/mlp_dww.hip: its own
    prologue / steady state / tail split, odd stage counts for the bf16 stores, sample counts of one ragged tile), the other jobs
    on the 16-wave kernel, two launches over disjoint partial-tile slots; 0: every job on the 16-wave kernel.  The two forms sum a
    tile's products in different orders: equal to float32 rounding of a sum (<= 1e-5 rel-L2 per tensor), each bit-reproducible.
    The 2 x 64 model has no 256 x 256 job: there the A/B is "dw_private_tiles" (jobs of at most four output tiles as sixteen
    independent wave pipelines with a fixed-order tree reduction through the LDS, csrc/mlp_s16.hip:dw_private) against the shared
    16-wave form."""
    from nerf_meets_mlx_amd import _native
    from nerf_meets_mlx_amd.models.NeRF import NeRF
    gen = torch.Generator().manual_seed(100 + M)
    if kind == "view":
        arch = O.NerfArch()
        m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=0, precision=precision)
        flat = O.flatten_params(arch, O.init_params(arch, 0)) * 1.5
        m.load_flat(flat)
        x, g = torch.randn(M, 90, generator=gen), torch.randn(M, 4, generator=gen)
    elif kind == "image":
        m, arch, flat = _image_pair(precision)
        x, g = torch.randn(M, 40, generator=gen), torch.randn(M, 3, generator=gen)
    else:
        m, arch, flat = _small_pair(precision)
        x, g = torch.randn(M, 48, generator=gen), torch.randn(M, 4, generator=gen)
    key = b"dw22_variant" if precision == 22 else b"dw16_variant"
    lib = _native.lib()
    got = {}
    if kind == "small" and precision == 22:
        key = b"dw_private_tiles"       # the 2 x 64 model has no 256 x 256 job: its A/B is the wave-private form of the tiny jobs (0 = off)
    # precision 16: "dw16_variant" 0 = every job on round 2's mlp_dw_kernel, 1 = the round-5 kernels (for the 2 x 64 model: the
    # wave-private form over the bf16 stores)
    try:
        for v in ((0, 4, 4) if key == b"dw_private_tiles" else (0, 1, 1, 3, 3) if precision == 16 else (0, 1, 1)):
            _native.check(lib.nerf_set_option(key, v))
            m.grads.fill_(float("nan"))
            m.forward(x.to(DEV), train=True)
            gr = m.backward(g.to(DEV)).cpu().clone()
            assert torch.isfinite(gr).all()
            if v in got:
                assert torch.equal(got[v], gr), "not bit-reproducible"
            got[v] = gr
    finally:
        _native.check(lib.nerf_set_option(key, 4 if key == b"dw_private_tiles" else 1))
    for form in sorted(k for k in got if k != 0):
        off, worst = 0, 0.0
        for name, o_, i_ in arch.layer_shapes():
            for cnt in (o_ * i_, o_):
                a, b = got[form][off:off + cnt], got[0][off:off + cnt]
                if float(b.norm()) > 0:
                    worst = max(worst, _rel_l2(a, b))
                else:
                    assert float(a.norm()) == 0
                off += cnt
        assert worst < 1e-5, (form, worst)


@pytest.mark.parametrize("precision", [22, 16])
@pytest.mark.parametrize("wgs", [7, 96, 300, 512])
def test_weight_gradients_do_not_depend_on_the_workgroup_count_beyond_rounding(precision, wgs):
    """`dw_workgroups` (split-K width of the weight-gradient launches; with the two-launch form each launch is clamped to its half
    of the partial-tile slots): any setting gives the gradient of the default to float32 summation rounding."""
    from nerf_meets_mlx_amd import _native
    from nerf_meets_mlx_amd.models.NeRF import NeRF
    lib = _native.lib()
    arch = O.NerfArch()
    m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=0, precision=precision)
    m.load_flat(O.flatten_params(arch, O.init_params(arch, 0)) * 1.5)
    gen = torch.Generator().manual_seed(5)
    M = 3000 * 32 + 17
    x, g = torch.randn(M, 90, generator=gen).to(DEV), torch.randn(M, 4, generator=gen).to(DEV)
    m.forward(x, train=True)
    ref = m.backward(g).cpu().clone()
    try:
        _native.check(lib.nerf_set_option(b"dw_workgroups", wgs))
        m.grads.fill_(float("nan"))
        m.forward(x, train=True)
        got = m.backward(g).cpu()
    finally:
        _native.check(lib.nerf_set_option(b"dw_workgroups", 0))
    assert torch.isfinite(got).all()
    assert _rel_l2(got, ref) < 1e-5
