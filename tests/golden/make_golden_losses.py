#!/usr/bin/env python3
"""Pin the reference's two LOSS CLOSURES (entrypoints/__test_nerf.py:46-126: `mlx_mse_coarse`, `mlx_mse_fine`) as
fixtures  ->  tests/golden/ref_mx_losses.npz + ref_mx_losses.json.

They are nested functions of `main()` in a module that cannot be imported here (imageio / tyro are absent, and main()
reads a dataset from disk), so the two FunctionDef nodes are cut out of the module's AST, compiled on their own and
executed -- unchanged -- over the numpy `mx` shim with the free names they use (`mx`, `render_kwargs_train`,
`render_rays`, `raw2outputs`, `ray`) bound to the reference's own objects, exactly as main() binds them.  What this pins
that the restatement (oracle.coarse_loss / fine_loss) only asserted so far: the viewdir normalisation of :62-64 / :98-101,
the ray packing [o, d, near, far, viewdirs] of :72-82, `white_bkgd` taken from the kwargs in the coarse loss but
HARD-CODED False in the fine loss (:106, Q8), `raw_noise_std = 0` (:107), the loss = mean over B x 3 (:88, :124), and
the importance-sample construction of the training loop (:275-288: sampler on the re-rendered coarse weights, sort of
the concatenation).  Arithmetic is float32 numpy (not MLX's): see make_golden_mx.py.

Build container only (needs /root/reference; listed in .gpurunignore).  Only arrays and scalars are stored.

    python tests/golden/make_golden_losses.py
"""
import ast
import json
import os
import sys
import warnings

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
import mlx_numpy_shim as shim      # noqa: E402
import make_golden_mx as G         # noqa: E402


def extract_closures(path, outer="main", names=("mlx_mse_coarse", "mlx_mse_fine")):
    """The FunctionDef nodes `names` nested in function `outer` of the module at `path`, compiled as a module of their own."""
    with open(path) as fp:
        tree = ast.parse(fp.read(), filename=path)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == outer)
    found = [n for n in main.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in found) == sorted(names), [n.name for n in found]
    mod = ast.Module(body=found, type_ignores=[])
    return compile(mod, path, "exec"), {n.name: (n.lineno, n.end_lineno) for n in found}


def main():
    assert os.path.isdir(REF), "reference checkout not present (build container only)"
    shim.install()
    sys.path.insert(0, REF)
    import mlx.core as mx
    from mlx_nerf import config_parser as ref_cfg
    from mlx_nerf import sampling as ref_sampling
    from mlx_nerf.models import NeRF as ref_nerf
    from mlx_nerf.ops import pose as ref_pose
    from mlx_nerf.rendering import ray as ref_ray
    from mlx_nerf.rendering import render as ref_render

    A = lambda x: mx.array(np.asarray(x, dtype=np.float32))
    rng = np.random.default_rng(20261005)
    # the networks of the render fixtures (make_golden_mx.py: seeds 4 / 5, sigma sharpened so that the CDF is not flat)
    args = ref_cfg.config_parser().parse_args(args=[])
    args.use_viewdirs = True; args.white_bkgd = True; args.dataset_type = "blender"; args.N_importance = 24
    args.n_depth_samples = 64; args.netchunk = 3000; args.lindisp = False
    kw_train, kw_test, _, _ = ref_nerf.create_NeRF(args)
    L = G.layer_list(63, 27, 5, 8, 256, [4], True)
    seeds, alpha = {"coarse": 4, "fine": 5}, (40.0, 1.0)
    pc, pf = G.draw_weights(L, seeds["coarse"]), G.draw_weights(L, seeds["fine"])
    G.inject(kw_train["network_coarse"], pc, *alpha); G.inject(kw_train["network_fine"], pf, *alpha)
    kw_train.update({"near": 2.0, "far": 6.0})                                   # entrypoints/__test_nerf.py:153-158

    # the closures, cut out of main() and bound to the names main() binds them to
    code, lines = extract_closures(os.path.join(REF, "mlx_nerf", "entrypoints", "__test_nerf.py"))
    captured = {}

    def raw2outputs_recording(*a, **k):                  # the reference's raw2outputs, its outputs and flags also kept
        out = ref_render.raw2outputs(*a, **k)
        captured["fine_rgb"] = np.asarray(out[0]); captured["fine_args"] = [float(a[3]), bool(a[4])]
        return out
    ns = {"mx": mx, "render_kwargs_train": kw_train, "render_rays": ref_render.render_rays, "raw2outputs": raw2outputs_recording,
          "ray": ref_ray}
    exec(code, ns)
    mlx_mse_coarse, mlx_mse_fine = ns["mlx_mse_coarse"], ns["mlx_mse_fine"]

    # a batch like the training loop's (:202-236): rays of random pixels of one pose, white-composited-looking targets
    H, W = 20, 24
    f = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
    K = np.array([[f, 0, 0.5 * W], [0, f, 0.5 * H], [0, 0, 1]], dtype=np.float64)
    c2w = np.asarray(ref_pose.pose_spherical(70.0, -30.0, 4.0))[:3, :4]
    ro, rd = ref_ray.get_rays(H, W, K, c2w)
    ro = np.asarray(ro, np.float32).reshape(-1, 3); rd = np.asarray(rd, np.float32).reshape(-1, 3)
    choice = rng.choice(H * W, size=48, replace=False)
    rays_o, rays_d = ro[choice], rd[choice]
    y = rng.random((48, 3)).astype(np.float32)
    batch_rays = mx.stack([A(rays_o), A(rays_d)], axis=0)                         # :235
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        loss_c = mlx_mse_coarse(kw_train["network_coarse"], batch_rays, A(y))
        # the loop's re-render + importance samples + sort (:245-288), with the reference's own functions
        vd = rays_d / np.linalg.norm(rays_d, axis=-1, keepdims=True)
        rays_linear = np.concatenate([rays_o, rays_d, np.full_like(rays_d[:, :1], 2.0), np.full_like(rays_d[:, :1], 6.0), vd], -1)
        res = ref_render.render_rays(A(rays_linear), **kw_train)
        z_vals, weights = np.asarray(res["z_vals"]), np.asarray(res["weights"])
        torch.manual_seed(2026)
        z_imp = ref_sampling.sample_from_inverse_cdf_torch(torch.from_numpy(z_vals), torch.from_numpy(weights), 24).numpy()
        torch.manual_seed(2026); u = torch.rand(48, 24).numpy()
        z_fine = mx.sort(mx.concatenate([A(z_vals), A(z_imp)], axis=-1), axis=-1)
        loss_f = mlx_mse_fine(kw_train["network_fine"], batch_rays, z_fine, A(y))
    d = {"rays_o": rays_o, "rays_d": rays_d, "target": y, "u": u.astype(np.float32), "z_vals": G.f32(z_vals), "weights": G.f32(weights),
         "z_imp": z_imp.astype(np.float32), "z_fine": G.f32(z_fine), "rgb_coarse": G.f32(res["rgb_coarse"]),
         "loss_coarse": np.float32(np.asarray(loss_c)), "loss_fine": np.float32(np.asarray(loss_f)), "fine_rgb": G.f32(captured["fine_rgb"])}
    np.savez_compressed(os.path.join(OUT, "ref_mx_losses.npz"), **d)
    meta = {"note": "generated by tests/golden/make_golden_losses.py: the reference's mlx_mse_coarse / mlx_mse_fine, AST-extracted from "
                    "entrypoints/__test_nerf.py main() and executed unchanged over the numpy mx shim",
            "closure_lines": lines, "layers": L, "seeds": seeds, "alpha_scale_bias": list(alpha),
            "checksum": {"coarse": G.checksum(pc, L), "fine": G.checksum(pf, L)}, "n_depth_samples": 64, "N_importance": 24,
            "near": 2.0, "far": 6.0, "kwargs_white_bkgd": bool(kw_train["white_bkgd"]),
            "fine_raw2outputs_raw_noise_std_and_white_bkgd": captured["fine_args"],
            "loss_coarse": float(np.asarray(loss_c)), "loss_fine": float(np.asarray(loss_f))}
    with open(os.path.join(OUT, "ref_mx_losses.json"), "w") as fp:
        json.dump(meta, fp, indent=1, sort_keys=True)
    print("loss-closure fixtures written:", {k: meta[k] for k in ("closure_lines", "loss_coarse", "loss_fine", "fine_raw2outputs_raw_noise_std_and_white_bkgd")})


if __name__ == "__main__":
    main()
