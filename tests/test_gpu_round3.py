"""Round-3 GPU tests.

(1) The HIP path against fixtures produced by the reference's OWN source (rendering/render.py, models/embedding.py,
    models/NeRF.py, encoding/*.py, ops/*.py, sampling/*.py executed over a numpy `mx` shim in the build container:
    tests/golden/make_golden_mx.py).  fp32 models (NeRF(precision=32), the reference's arithmetic) are held to
    float32 noise, bf16 models to the bf16 tolerances of tests/test_gpu_parity.py; round 4: split-fp16 models
    (NeRF(precision=22): inference on csrc/mlp22.hip) are held to the SAME tolerances as the fp32 models.
(2) Precision is part of the model (ABI 3): a bf16 and an fp32 model interleaved on two streams.
(3) The pixel permutation as a SAMPLER (uniformity / independence), next to numpy's choice(replace=False).
(4) The north-star PSNR statement as a paired ensemble cut (bf16 arm vs fp32 arm on identical batches).
"""
import json
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
    from nerf_meets_mlx_amd import _native
    assert _native.lib().nerf_abi_version() == 3


@pytest.fixture(scope="module")
def meta(golden_dir):
    with open(os.path.join(golden_dir, "ref_mx_meta.json")) as fp:
        return json.load(fp)


def _npz(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def D(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close(got, want, atol, rtol=0.0):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    np.testing.assert_allclose(got, np.asarray(want), atol=atol, rtol=rtol, equal_nan=True)


def _flat_from_seed(layers, seed, checksum, alpha=(1.0, 0.0)):
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


def _model(ctor, layers, seed, checksum, precision, alpha=(1.0, 0.0)):
    from nerf_meets_mlx_amd.models.NeRF import NeRF
    m = NeRF(n_layers=ctor["n_layers"], width_layers=ctor["width_layers"], channel_input=ctor["channel_input"],
             channel_input_views=ctor["channel_input_views"], channel_output=ctor["channel_output"],
             list_skip_connection_layers=ctor["list_skip_connection_layers"],
             is_use_view_directions=ctor["is_use_view_directions"], device=DEV, seed=0, precision=precision)
    assert [list(t) for t in m.shapes] == [list(l) for l in layers]             # our constructor == the reference's shapes
    m.load_flat(_flat_from_seed(layers, seed, checksum, alpha))
    return m


# ------------------------------------------------------------------------------------------------ a13
def test_raw2outputs_vs_reference_source_fixture(golden_dir, meta):
    from nerf_meets_mlx_amd.rendering import render
    g = _npz(golden_dir, "ref_mx_raw2outputs.npz")
    for c in meta["raw2outputs"]["cases"]:
        t = c["tag"]
        rgb, disp, acc, w, depth = render.raw2outputs(D(g[f"{t}_raw"]), D(g[f"{t}_z"]), D(g[f"{t}_d"]), 0, c["white_bkgd"])
        assert tuple(w.shape) == g[f"{t}_weights"].shape and tuple(disp.shape) == g[f"{t}_disp"].shape      # [B,n,1], [B,1]
        sc = max(1.0, float(np.nanmax(np.abs(g[f"{t}_weights"]))))
        close(w, g[f"{t}_weights"], atol=3e-6 * sc, rtol=1e-4)            # exp(+large) for negative sigma: relative
        close(acc, g[f"{t}_acc"], atol=2e-5 * sc, rtol=1e-4)
        close(rgb, g[f"{t}_rgb"], atol=2e-5 * sc, rtol=2e-4)
        close(depth, g[f"{t}_depth"], atol=2e-5 * sc, rtol=2e-4)
        assert np.array_equal(np.isnan(disp.cpu().numpy()), np.isnan(g[f"{t}_disp"]))                       # Q11
    rgb, disp, acc, w, depth = render.raw2outputs(D(g["noise_raw"]), D(g["noise_z"]), D(g["noise_d"]), meta["raw2outputs"]["noise_std"],
                                                  True, noise=D(g["noise_noise"]))
    close(w, g["noise_weights"], atol=3e-6, rtol=1e-4)
    close(rgb, g["noise_rgb"], atol=2e-5, rtol=2e-4)


# ------------------------------------------------------------------------------------------------ a9 a10 a23 a24
def test_encodings_vs_reference_source_fixture(golden_dir, meta):
    from nerf_meets_mlx_amd.encoding.identity import IdentityEncoding
    from nerf_meets_mlx_amd.encoding.sinusoidal import SinusoidalEncoding
    from nerf_meets_mlx_amd.encoding.spherical_harmonics import SphericalHarmonicsEncoding
    from nerf_meets_mlx_amd.models import embedding
    g = _npz(golden_dir, "ref_mx_encodings.npz")
    x3, dirs = D(g["x3"]), D(g["dirs"])
    f10, d10 = embedding.get_embedder(10)
    f4, d4 = embedding.get_embedder(4)
    f2, d2 = embedding.get_embedder(6, n_input_dims=2)
    fid, did = embedding.get_embedder(-1)
    e = meta["embedder"]
    assert (d10, d4, d2, did) == (e["out_dim_10"], e["out_dim_4"], e["out_dim_2d_6"], e["out_dim_identity"])
    close(f10(x3), g["emb10"], atol=4e-6)                      # |arg| <= 81 * 4: cosf / sinf with fp32 range reduction
    close(f4(dirs), g["emb4"], atol=1e-6)
    close(f2(D(g["x2"])), g["emb2d_6"], atol=2e-6)             # 2-d inputs: no raw-input block (embedding.py:79)
    assert torch.equal(fid(x3), x3)
    close(embedding.embed(D(g["embed_pos"]), f10, D(g["embed_dir"]), f4), g["embed_out"], atol=4e-6)
    close(embedding.embed(D(g["embed_pos"]), f10, None, None), g["embed_out_nodir"], atol=4e-6)
    s = meta["sinusoidal"]
    enc = SinusoidalEncoding(2, 10, 0.0, 8.0, False)
    assert enc.get_out_dim() == s["img"]["out_dim"]
    got = enc(D(g["sin_xi"]))
    close(got[:, :5], g["sin_img"][:, :5], atol=2e-4)          # small arguments
    close(got, g["sin_img"], atol=1e-2)                        # |arg| up to 1e5: one fp32 ulp of the argument is 8e-3 rad
    enc2 = SinusoidalEncoding(3, 4, is_include_input=True)
    assert enc2.get_out_dim() == s["inc"]["out_dim"]
    close(enc2(x3), g["sin_inc"], atol=3e-6)
    enc3 = SinusoidalEncoding(3, 5, -1.0, 2.5, False)
    close(enc3(x3), g["sin_frac"], atol=6e-6)
    for deg in range(5):
        sh = SphericalHarmonicsEncoding(3, deg)
        close(sh(dirs), g[f"sh{deg}"], atol=2e-6)
    assert torch.equal(IdentityEncoding(3)(x3), x3)


# ------------------------------------------------------------------------------------------------ a11 a12
@pytest.mark.parametrize("tag,precision,tol", [("view", 32, 1e-4), ("view", 22, 1e-4), ("view", 16, 1.5e-2),
                                                 ("image", 32, 1e-4), ("image", 22, 1e-4), ("image", 16, 1.5e-2),
                                                 ("ngp", 22, 1e-4), ("ngp", 16, 1.5e-2)])
def test_mlp_forward_vs_reference_source_fixture(golden_dir, meta, tag, precision, tol):
    """NeRF.forward (models/NeRF.py:201-243) on rows the reference's own class produced outputs for -- all three shapes the
    reference instantiates or BASELINE names: the view model, the image-fitting model (:196-197,241) and the 2 x 64 model.
    Precision 32 (fp32 MFMA) and 22 (split 16-bit operands): 1e-4 of the output scale (float32 summation order over the
    layers); bf16 models (declared reduced precision): 1.5e-2 (bf16 operands, 8 mantissa bits,
    random walk over the layers; against the bf16-EMULATING oracle the same kernels are at 1e-2, test_gpu_parity)."""
    g = _npz(golden_dir, "ref_mx_mlp.npz")
    net = meta["mlp"]["nets"][tag]
    m = _model(net["ctor"], net["layers"], net["seed"], net["checksum"], precision)
    out = m.forward(D(g[f"{tag}_x"]))
    want = g[f"{tag}_out"]
    assert tuple(out.shape) == want.shape
    assert float(np.abs(out.cpu().numpy() - want).max() / np.abs(want).max()) < tol


@pytest.mark.parametrize("precision,tol", [(32, 1e-4), (22, 1e-4), (16, 1.5e-2)])
def test_run_model_vs_reference_source_fixture(golden_dir, meta, precision, tol):
    from nerf_meets_mlx_amd.models import embedding
    from nerf_meets_mlx_amd.models.NeRF import run_model
    g = _npz(golden_dir, "ref_mx_mlp.npz")
    net = meta["mlp"]["nets"]["view"]
    m = _model(net["ctor"], net["layers"], net["seed"], net["checksum"], precision)
    f10, _ = embedding.get_embedder(10)
    f4, _ = embedding.get_embedder(4)
    out = run_model(D(g["run_pos"]), f10, D(g["run_dir"]), f4, m, netchunk=meta["mlp"]["run_model_netchunk"])
    assert tuple(out.shape) == g["run_out"].shape
    assert float(np.abs(out.cpu().numpy() - g["run_out"]).max() / np.abs(g["run_out"]).max()) < tol
    with pytest.raises(AssertionError):                                         # models/NeRF.py:31
        run_model(D(g["run_pos"]).reshape(-1, 3), f10, D(g["run_dir"]), f4, m)


# ------------------------------------------------------------------------------------------------ a2 a4 a5 a6 a7 a20 a25
def test_sampling_rays_metric_pose_vs_reference_source_fixture(golden_dir, meta):
    from nerf_meets_mlx_amd import sampling
    from nerf_meets_mlx_amd.ops import metric
    from nerf_meets_mlx_amd.ops.pose import pose_spherical
    from nerf_meets_mlx_amd.rendering import ray, render
    from nerf_meets_mlx_amd.sampling import linear_disparity, uniform
    g = _npz(golden_dir, "ref_mx_misc.npz")
    near, far = D(g["near"]), D(g["far"])
    assert torch.equal(uniform.sample_z(near, far, 64).cpu(), torch.from_numpy(g["z_uniform_64"]))          # same fp32 op sequence
    assert torch.equal(uniform.sample_z(near, far, 5).cpu(), torch.from_numpy(g["z_uniform_5"]))
    zl = linear_disparity.sample_z(near, far, 64).cpu().numpy()
    close(zl, g["z_lindisp_64"], atol=0, rtol=2e-6)
    assert np.all(zl[:, 0] == 0) and np.all(zl[:, -1] == 0)                                                 # Q12
    z = D(g["z_uniform_64"])
    assert sampling.add_noise_z(z, 0.0) is z                                                                # :13-14 returns z_vals itself
    n = meta["sampling"]["ndc"]
    no, nd = ray.ndc_rays(n["H"], n["W"], n["focal"], n["near"], D(g["ndc_o_in"]), D(g["ndc_d_in"]))
    close(no, g["ndc_o"], atol=2e-6, rtol=2e-5); close(nd, g["ndc_d"], atol=2e-6, rtol=2e-5)
    o, d, nr, fr, vd, ft = render.decompose_ray_batch(D(g["rays_linear"]))
    assert ft is None and tuple(nr.shape) == (6, 1)
    for got, key in ((o, "dec_o"), (d, "dec_d"), (nr, "dec_near"), (fr, "dec_far"), (vd, "dec_viewdirs")):
        assert np.array_equal(got.cpu().numpy(), g[key])
    close(metric.MSE()(D(g["metric_a"]), D(g["metric_b"])), g["mse"], atol=0, rtol=2e-6)
    close(metric.PSNR()(D(g["metric_a"]), D(g["metric_b"])), g["psnr"], atol=0, rtol=2e-6)
    for args, want in zip(g["pose_args"], g["pose_out"]):
        close(np.asarray(pose_spherical(*[float(a) for a in args])), want, atol=1e-6)


# ------------------------------------------------------------------------------------------------ create_NeRF, a14 a18 a19
def _render_kwargs(meta, precision):
    """create_NeRF exactly as the fixture's generator called the reference's (args from config_parser defaults + the
    same overrides), then the fixture's weights loaded into the two networks."""
    from nerf_meets_mlx_amd import config_parser as C
    from nerf_meets_mlx_amd.models.NeRF import create_NeRF
    r = meta["render"]
    args = C.config_parser().parse_args(args=[])
    args.use_viewdirs = True; args.white_bkgd = True; args.dataset_type = "blender"; args.N_importance = r["N_importance"]
    args.n_depth_samples = r["n_depth_samples"]; args.netchunk = r["netchunk"]; args.lindisp = False
    kw_train, kw_test, idx_iter, opt = create_NeRF(args, device=DEV, precision=precision)
    c = meta["create_NeRF"]
    assert kw_test is kw_train and set(kw_train.keys()) == set(c["keys"]) and idx_iter == c["idx_iter"]      # Q5 alias
    assert kw_train["render_rays_func"].__name__ == c["render_rays_func"]
    for k, v in c["values"].items():
        assert kw_train[k] == v, (k, kw_train[k], v)
    assert opt.learning_rate == c["optimizer"]["learning_rate"] and list(opt.betas) == c["optimizer"]["betas"]
    layers = [tuple(l) for l in r["layers"]]
    for name in ("coarse", "fine"):
        kw_train[f"network_{name}"].load_flat(_flat_from_seed(layers, r["seeds"][name], r["checksum"][name], r["alpha_scale_bias"][name]))
    kw_train.update({"near": r["near"], "far": r["far"]})
    return kw_train, r


@pytest.mark.parametrize("precision,tol_raw,tol_w", [(32, 1e-4, 2e-5), (22, 1e-4, 2e-5), (16, 2e-2, 1.5e-2)])
def test_render_rays_vs_reference_source_fixture(golden_dir, meta, precision, tol_raw, tol_w):
    from nerf_meets_mlx_amd.rendering import render
    g = _npz(golden_dir, "ref_mx_render.npz")
    kw, r = _render_kwargs(meta, precision)
    call = {k: v for k, v in kw.items() if k not in ("use_viewdirs", "is_test", "ndc", "near", "far", "render_rays_func")}
    ret = render.render_rays(D(g["rr_rays"]), retraw=True, **call)
    assert sorted(ret.keys()) == sorted(["raw", "rgb_map", "disp_map", "acc_map", "rgb_coarse", "disp_coarse", "acc_coarse", "z_vals", "weights"])
    assert torch.equal(ret["z_vals"].cpu(), torch.from_numpy(g["rr_z_vals"]))
    assert tuple(ret["weights"].shape) == (40, 64, 1) and tuple(ret["disp_map"].shape) == (40, 1)
    assert float(np.abs(ret["raw"].cpu().numpy() - g["rr_raw"]).max() / np.abs(g["rr_raw"]).max()) < tol_raw
    close(ret["weights"], g["rr_weights"], atol=tol_w)
    close(ret["rgb_map"], g["rr_rgb_map"], atol=tol_w); close(ret["acc_map"], g["rr_acc_map"], atol=tol_w)


@pytest.mark.parametrize("precision,tol", [(32, 1e-4), (22, 1e-4), (16, 2e-2)])
def test_render_rays_eval_and_render_vs_reference_source_fixture(golden_dir, meta, precision, tol):
    """render_rays_eval (a18) and render() (a19: get_rays -> packing -> ragged chunks -> reshape) against what the
    reference's own render.py returned for the same weights, rays and uniforms."""
    from nerf_meets_mlx_amd.rendering import render
    g = _npz(golden_dir, "ref_mx_render.npz")
    kw, r = _render_kwargs(meta, precision)
    call = {k: v for k, v in kw.items() if k not in ("use_viewdirs", "is_test", "ndc", "near", "far", "render_rays_func")}
    ret = render.render_rays_eval(D(g["rr_rays"]), u=D(g["re_u"]), **call)
    close(ret["rgb_coarse"], g["re_rgb_coarse"], atol=tol)
    close(ret["rgb_map"], g["re_rgb_map"], atol=tol); close(ret["acc_map"], g["re_acc_map"], atol=tol)
    H, W = (int(v) for v in g["render_HW"])
    rgb, disp, acc, extras = render.render(H, W, g["render_K"], chunk=int(g["render_chunk"]), c2w=g["render_c2w"], u=D(g["render_u"]), **kw)
    assert tuple(rgb.shape) == (H, W, 3) and tuple(disp.shape) == (H, W, 1) and tuple(acc.shape) == (H, W, 1)
    close(rgb, g["render_rgb"], atol=tol); close(acc, g["render_acc"], atol=tol)
    assert set(extras.keys()) == set(meta["render"]["extras_keys"])
    assert tuple(extras["weights"].shape) == (H, W, 64, 1) and tuple(extras["z_vals"].shape) == (H, W, 64)
    assert torch.equal(extras["z_vals"].cpu(), torch.from_numpy(g["render_extra_z_vals"]))
    close(extras["rgb_coarse"], g["render_extra_rgb_coarse"], atol=tol)
    close(extras["weights"], g["render_extra_weights"], atol=tol)


# ------------------------------------------------------------------------------------------------ ABI 3: precision per model
def test_bf16_and_fp32_models_interleaved_on_two_streams():
    """No process-wide precision switch: a bf16 and an fp32 model (same weights) are queried and trained alternately on
    two streams; every result equals the one the same model gives when it runs alone."""
    from nerf_meets_mlx_amd.models.NeRF import NeRF
    mk = lambda p: NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=4, precision=p)
    g = torch.Generator().manual_seed(0)
    o = torch.nn.functional.normalize(torch.randn(256, 3, generator=g), dim=-1) * 4.0
    d = -o / 4.0 + 0.25 * torch.randn(256, 3, generator=g)
    rays = O.pack_rays(o, d, 2.0, 6.0).to(DEV)
    z = torch.sort(torch.rand(256, 96, generator=g) * 4 + 2, -1).values.to(DEV)
    d_raw = torch.randn(256, 96, 4, generator=g).to(DEV)
    alone = {}
    for p in (16, 32):
        m = mk(p)
        raw = m.query(rays, z, train=True)
        alone[p] = (raw.clone(), m.backward(d_raw).clone())
    torch.cuda.synchronize()
    m16, m32 = mk(16), mk(32)
    assert m16.packed().numel() < m32.packed().numel()                        # the fp32 image carries the fp32 streams too
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    got = {16: [], 32: []}
    for _ in range(3):
        for m, s, p in ((m16, s1, 16), (m32, s2, 32)):
            with torch.cuda.stream(s):
                raw = m.query(rays, z, train=True)
                got[p].append((raw.clone(), m.backward(d_raw).clone()))
    torch.cuda.synchronize()
    for raw, gr in got[16]:
        assert torch.equal(raw, alone[16][0]) and torch.equal(gr, alone[16][1])           # bf16 path is bit-deterministic
    for raw, gr in got[32]:
        assert torch.equal(raw, alone[32][0])
        assert float((gr - alone[32][1]).norm() / alone[32][1].norm()) < 1e-5             # fp32 dW: float atomics (order)
    rel = float((alone[16][0] - alone[32][0]).abs().max() / alone[32][0].abs().max())
    assert 1e-5 < rel < 3e-2                                                               # and they ARE different arithmetics


def test_add_noise_z_broadcasts_or_refuses_t_rand():
    from nerf_meets_mlx_amd import sampling
    z = torch.sort(torch.rand(9, 16) * 4 + 2, -1).values
    t_row, t_col = torch.rand(16), torch.rand(9, 1)
    for t in (t_row, t_col):
        got = sampling.add_noise_z(z.to(DEV), 1.0, t.to(DEV)).cpu()
        np.testing.assert_allclose(got.numpy(), O.add_noise_z(z, 1.0, t.expand_as(z)).numpy(), atol=5e-7)
    with pytest.raises(RuntimeError):
        sampling.add_noise_z(z.to(DEV), 1.0, torch.rand(9, 5).to(DEV))                    # not broadcastable: refused, no OOB read


# ------------------------------------------------------------------------------------------------ a3 as a sampler
def test_pixel_permutation_device_is_a_uniform_sampler():
    """Device side of tests/test_host_cpu.py::test_pixel_permutation_is_a_uniform_sampler: 512 keys x the first 1024 of
    640 000 outputs, chi-square over 64 equal bins of the pixel range, next to numpy's choice(replace=False)."""
    from nerf_meets_mlx_amd.ops import index
    dom, n, bins = 640000, 1024, 64
    cnt = np.zeros(bins)
    for seed in range(512):
        v = index.pixel_permutation(n, dom, 1000 + seed, 0, DEV).cpu().numpy()
        assert len(np.unique(v)) == n and np.array_equal(v, index.pixel_permutation_host(n, dom, 1000 + seed, 0))
        cnt += np.bincount(v * bins // dom, minlength=bins)
    e = 512 * n / bins
    chi2 = float(((cnt - e) ** 2 / e).sum())
    assert chi2 < 110.0, chi2                                                              # chi2(63): mean 63, 99.98 % quantile ~ 110


# ------------------------------------------------------------------------------------------------ configs[4]: determinism, level groups
def _ngp(det, groups=4, seed=7, **kw):
    from nerf_meets_mlx_amd.dataset import synthetic
    from nerf_meets_mlx_amd.engine.ngp import NGPTrainer
    imgs, poses, _, _, K = synthetic.make_dataset(24, 24, 3, seed=0, device=DEV)
    return NGPTrainer(imgs, poses, K, N_rand=128, n_depth_samples=64, seed=seed, device=DEV, log2_hashmap_size=14,
                      deterministic=det, level_groups=groups, **kw)


def test_ngp_deterministic_scatter_is_bit_reproducible_and_matches_float_atomics():
    """HashNeRF(deterministic=True): the table gradient is accumulated in int64 2^-52 fixed point with INTEGER atomics
    (associative), so two runs give bit-identical tables, whatever the level grouping of the launches; against the float
    atomics the gradient agrees to float32 summation noise.  The float mode is NOT bit-reproducible (documented)."""
    from nerf_meets_mlx_amd import sampling
    from nerf_meets_mlx_amd.rendering import render
    from nerf_meets_mlx_amd import _native
    runs = []
    try:
        for groups, combine in ((4, 64), (1, 64), (16, 64), (4, 0), (4, 30)):
            # combine: coarse levels (N_l <= value) go through the LDS write-combining kernel, 0 = all levels directly
            _native.check(_native.lib().nerf_set_option(b"hash_combine_max_res", combine))
            tr = _ngp(True, groups)
            for _ in range(4):
                tr.train_step()
            torch.cuda.synchronize()
            runs.append((tr.field.enc.tables.clone(), tr.field.mlp.params.clone()))
    finally:
        _native.check(_native.lib().nerf_set_option(b"hash_combine_max_res", 64))
    assert float((runs[0][0] - _ngp(True).field.enc.tables).abs().max()) > 0             # the tables did train
    for t, p in runs[1:]:
        assert torch.equal(t, runs[0][0]) and torch.equal(p, runs[0][1])
    # one gradient, both modes, same inputs
    a, b = _ngp(True), _ngp(False)
    rays, target = a.sample_batch()
    z = sampling.sample_coarse(rays, 64)
    gs = []
    for tr in (a, b):
        raw = tr.field.query(rays, z, train=True)
        _, d_raw, _ = render.composite_mse_backward(raw, z, rays, target, True)
        _, gt = tr.field.backward(d_raw)
        gs.append(gt.double() * 2.0 ** -52 if gt.dtype == torch.int64 else gt.double())
    assert a.field.enc.grad.dtype == torch.int64 and b.field.enc.grad.dtype == torch.float32
    assert float(gs[0].abs().max()) > 0
    assert float((gs[0] - gs[1]).norm() / gs[0].norm()) < 1e-5
    # the Adam pass that consumes the accumulators clears them (no memset launch in the training step)
    a.train_step(rays, target)
    assert int(a.field.enc.grad.abs().max()) == 0
    b.train_step(rays, target)
    assert float(b.field.enc.grad.abs().max()) == 0.0


def test_ngp_checkpoint_resume_is_bit_identical_in_deterministic_mode(tmp_path):
    a = _ngp(True)
    for _ in range(3):
        a.train_step()
    path = a.save(str(tmp_path / "ngp"))
    b = _ngp(True)
    assert b.load(path) == 3
    for _ in range(3):
        a.train_step(); b.train_step()
    torch.cuda.synchronize()
    assert torch.equal(a.field.enc.tables, b.field.enc.tables) and torch.equal(a.field.mlp.params, b.field.mlp.params)


def test_fused_renderer_on_two_streams_uses_separate_workspaces():
    """`render_rays_fused` keeps its scratch (z, raw, weights, z_fine, raw_fine) per (device, stream): two streams
    rendering different rays concurrently give the results they give alone."""
    from nerf_meets_mlx_amd.models.NeRF import NeRF
    from nerf_meets_mlx_amd.rendering import render
    mk = lambda s: NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=s)
    mc, mf = mk(4), mk(5)
    g = torch.Generator().manual_seed(3)

    def rays_of(B):
        o = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * 4.0
        d = -o / 4.0 + 0.25 * torch.randn(B, 3, generator=g)
        return O.pack_rays(o, d, 2.0, 6.0).to(DEV)
    ra, rb = rays_of(3000), rays_of(3000)
    ua, ub = torch.rand(3000, 128, generator=g).to(DEV), torch.rand(3000, 128, generator=g).to(DEV)
    alone_a = render.render_rays_fused(ra, mc, mf, 64, 128, u=ua, white_bkgd=True)["rgb_map"].clone()
    alone_b = render.render_rays_fused(rb, mc, mf, 64, 128, u=ub, white_bkgd=True)["rgb_map"].clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(4):
        with torch.cuda.stream(s1):
            xa = render.render_rays_fused(ra, mc, mf, 64, 128, u=ua, white_bkgd=True)["rgb_map"]
        with torch.cuda.stream(s2):
            xb = render.render_rays_fused(rb, mc, mf, 64, 128, u=ub, white_bkgd=True)["rgb_map"]
        outs.append((xa, xb))
    torch.cuda.synchronize()
    for xa, xb in outs:
        assert torch.equal(xa, alone_a) and torch.equal(xb, alone_b)


# ------------------------------------------------------------------------------------------------ north star: PSNR at equal iterations
def test_psnr_paired_ensemble_bf16_vs_reference_arithmetic():
    """north_star: "PSNR within 0.1 dB of the MLX reference at equal iterations", as a DISTRIBUTIONAL statement
    (one trajectory cannot carry it: training under the reference's formulas is chaotic).
    8 seeds x 600 iterations, two HIP trainers per seed on identical batches / uniforms / initial weights:
      bf16 arm = the product path; fp32 arm = the reference's own arithmetic (fp32 MFMA kernels, 1e-4 per forward
      against the fp32 oracle).  Paired delta_s = PSNR_bf16 - PSNR_fp32 on held-out views (tools/psnr_ensemble.py; the
      tracked 170-seed x 2500-iteration run of the same tool is profiles/r03_psnr_ensemble_170seeds_*.jsonl).
    Seeds: the first 8 that are alive at INITIALISATION (chosen before any outcome; round 3's list was post-selected).
    Asserted: (a) at 200 and 400 iterations the MEDIAN paired difference is within the target's 0.1 dB; (b) at every
    checkpoint the mean paired difference is zero within its own 95 % confidence interval widened by those 0.1 dB -- a
    systematic bf16 deficit or gain of a few tenths of a dB fails; (c) the number of seeds that END in the dead-sigma
    state differs by at most one between the arms (the unconditional part: survival itself is compared).  The converged
    regime (24 seeds x 20 000 iterations at 800 x 800) is profiles/r04_psnr_converged_*.jsonl, DESIGN.md 5.3."""
    import argparse
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import psnr_ensemble as E
    a = argparse.Namespace(hw=100, views=12, test_views=2, n_rand=1024, n_importance=128, lrate_decay=500, no_quirks=False,
                           iters=600, every=200, dead_every=20, bridge_iters=0, resync=False)
    # Seeds fixed BEFORE any outcome is known: the first 8 whose two networks start with sigma > 0 (alive at INITIALISATION,
    # DESIGN.md 7 -- a property of the initial weights, the same for both arms).  Round 3 used a list post-selected for being
    # alive at iteration 2500 in both arms; a precision-induced change of the dead-sigma rate could not have failed it.
    seeds = E.alive_seeds(8, True, 0)
    assert seeds == [4, 10, 18, 21, 28, 33, 47, 58]
    recs, deads = [], {}
    for sd in seeds:
        r, dead = E.run_seed(sd, a, emit=lambda line: None)
        recs += r
        deads[sd] = dead
    stats = {st["ensemble_iter"]: st for st in E.summarise(recs)}
    print({k: (round(v["mean_delta_db"], 3), round(v["ci95_half_width_db"], 3), round(v["median_delta_db"], 3),
               round(v["max_abs_delta_db"], 3), round(v["mean_a"], 2), round(v["mean_b"], 2)) for k, v in stats.items()})
    assert stats[200]["n"] == len(seeds) and not stats[200]["seeds_non_finite"]
    for it in (200, 400):
        # early, while most seeds are still one trajectory: the MEDIAN paired difference (robust against one seed leaving the
        # dead-sigma state at different iterations in the two arms) is within the target's 0.1 dB
        assert abs(stats[it]["median_delta_db"]) <= 0.1, stats[it]
    for it in (200, 400, 600):
        st = stats[it]
        assert st["n"] == len(seeds)
        assert abs(st["mean_delta_db"]) <= 0.1 + st["ci95_half_width_db"], st
        # both arms are training: the empty-volume image scores 10.2 dB; measured ensemble means (8 seeds, one of them dead in both
        # arms and pinned at 10.2) 12.05 / 12.10 dB at 200, 12.37 / 12.40 at 400, 12.44 / 12.54 at 600 iterations -- the 11.0 floor
        # is one dead seed's worth below them (a second dead seed in one arm costs 0.25 dB of the mean and is caught by (c))
        assert st["mean_a"] > 11.0 and st["mean_b"] > 11.0
    # per-seed guard (advisor, round 4: restored): no single seed's arms may separate early -- measured max |delta| 0.33 dB at 200
    # and 0.24 dB at 400 iterations (0.65 at 600, where one seed leaves the dead-sigma state at different iterations in the two arms)
    assert stats[200]["max_abs_delta_db"] < 0.6 and stats[400]["max_abs_delta_db"] < 0.6, (stats[200], stats[400])
    assert stats[600]["max_abs_delta_db"] < 1.5, stats[600]
    # the unconditional part: the number of seeds in the dead-sigma state at the end must not differ between the arms by more
    # than one seed (a bf16-induced rise of the dead rate would show here; the 170-seed file has 34 / 34)
    n_dead = {arm: sum(1 for d in deads.values() if d[arm]["coarse"]["dead_at_end"] or d[arm]["fine"]["dead_at_end"]) for arm in ("bf16", "fp32")}
    print("dead at end:", n_dead)
    assert abs(n_dead["bf16"] - n_dead["fp32"]) <= 1, n_dead


def test_ngp_fused_inference_ray_major_tiles_are_bit_identical():
    """The fused configs[4] inference query walks (32 adjacent rays x one depth) tiles by default (cache locality of the
    hash gathers across neighbouring pixels); per sample nothing changes, so the output equals the (one ray x 32
    depths) order bit for bit -- also with a ray count that is not a multiple of 32 and a tiny n."""
    from nerf_meets_mlx_amd import _native, sampling
    from nerf_meets_mlx_amd.rendering import ray
    tr = _ngp(True)
    for _ in range(3):
        tr.train_step()
    K = tr.K
    L = _native.lib()
    for B, n in ((100, 64), (33, 7), (31, 64), (4096, 64)):
        idx = torch.arange(0, B, device=DEV, dtype=torch.int64)
        rays = ray.gen_rays(tr.H, tr.W, K, tr.poses[0, :3, :4], 2.0, 6.0, idx % (tr.H * tr.W))
        z = sampling.sample_coarse(rays, n)
        outs = []
        try:
            for mode in (1, 0):
                _native.check(L.nerf_set_option(b"ngp_ray_major", mode))
                outs.append(tr.field.query(rays, z).clone())
        finally:
            _native.check(L.nerf_set_option(b"ngp_ray_major", 1))
        assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1]), (B, n)


@pytest.mark.parametrize("B,n,N,kind", [(50, 64, 128, "rand"), (7, 64, 128, "const"), (5, 64, 128, "linspace"), (9, 256, 512, "rand"),
                                        (4, 16, 5, "rand"), (3, 64, 128, "nan"), (6, 64, 128, "negcdf"), (4, 65, 100, "rand")])
def test_importance_merge_counting_sort_paths(B, n, N, kind):
    """The merge of the importance sampler (bitonic sort of the N new depths + rank merge with the ascending coarse list)
    must give sort(concat(z, z_new)) for every kind of input: iid uniforms, a constant u, linspace incl. 1.0, NaN /
    negative / > 1 uniforms, weights below -0.01 (non-monotone CDF), the largest supported sizes, odd n / N.
    (Round 3 tried a rank sort -- slower, 151 vs 97 us per 32 768-ray chunk -- and a counting sort on u -- no faster:
    the sort is not what bounds the kernel; both reverted, this test stays.)"""
    from nerf_meets_mlx_amd import sampling
    g = torch.Generator().manual_seed(B * 1000 + N)
    z = torch.sort(torch.rand(B, n, generator=g) * 4 + 2, -1).values
    w = torch.rand(B, n, generator=g) ** 4
    u = torch.rand(B, N, generator=g)
    if kind == "const":
        u = torch.full((B, N), 0.37)
    elif kind == "linspace":
        u = torch.linspace(0.0, 1.0, N).expand(B, N).contiguous()
    elif kind == "nan":
        u[:, ::7] = float("nan"); u[:, 3::11] = -0.25; u[:, 5::13] = 1.5
    elif kind == "negcdf":
        w = torch.randn(B, n, generator=g) * 0.3
    z_new, z_m = sampling.importance_sample(z.to(DEV), w.to(DEV), N, u=u.to(DEV))
    want_new = O.sample_from_inverse_cdf(z, w[..., None], u)
    ok = torch.isfinite(want_new)
    np.testing.assert_allclose(z_new.cpu().numpy()[ok.numpy()], want_new.numpy()[ok.numpy()], rtol=0, atol=5e-5)   # CDF in float64 here
    assert torch.equal(z_m.cpu(), torch.sort(torch.cat([z, z_new.cpu()], -1), -1).values)


def test_fp32_models_are_bit_reproducible():
    """precision=32: the gradient of one batch computed twice is bit-identical (the dW kernel writes per-unit partial blocks
    that a second kernel adds in split order: no float atomics), every parameter is written (NaN-poisoned gradient buffer
    comes back finite), and two trainers with the reference's arithmetic stay bit-identical over training iterations."""
    from nerf_meets_mlx_amd.dataset import synthetic
    from nerf_meets_mlx_amd.engine.trainer import Trainer
    from nerf_meets_mlx_amd.models.NeRF import NeRF
    m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=DEV, seed=3, precision=32)
    g = torch.Generator().manual_seed(5)
    B, n = 37, 45                                         # ragged: 1665 samples = 52 tiles + 1
    x = torch.randn(B * n, 90, generator=g).to(DEV)
    d = torch.randn(B * n, 4, generator=g).to(DEV)
    grads = []
    for k in range(3):
        m.forward(x, train=True)
        if k == 1:
            m.grads.fill_(float("nan"))                   # the backward must overwrite every element, not accumulate
        grads.append(m.backward(d).clone())
    assert torch.isfinite(grads[1]).all()
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    imgs, poses, _, _, K = synthetic.make_dataset(20, 20, 3, seed=0, device=DEV)
    ts = [Trainer(imgs, poses, K, N_rand=64, n_depth_samples=64, N_importance=128, seed=4, device=DEV, precision=32) for _ in range(2)]
    for _ in range(4):
        for t in ts:
            t.train_step()
    for k, ma in ts[0]._checkpoint_buffers().items():
        assert torch.equal(ma.params, ts[1]._checkpoint_buffers()[k].params), k


def test_fp32_kernels_repeat_bit_for_bit_under_concurrent_load():
    """tools/stress_fp32.py, short: 24 x (training forward, backward, inference forward) of an fp32 model at 1024 x 192
    samples, each compared bit for bit with the first, bf16 kernels on a second stream every other iteration.  Guards
    the counted waits of the asm-loaded weight / operand fragments (a wait that passes early = a few stale values)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_fp32
    assert stress_fp32.run(iters=24, B=1024, n=192, dev=DEV, verbose=False) == 0
