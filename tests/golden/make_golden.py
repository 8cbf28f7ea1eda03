#!/usr/bin/env python3
"""Generate tests/golden/ref_*.npz / ref_config_defaults.json from the REFERENCE's own code.

Runs only in the build container (needs /root/reference; never on the GPU box).
The reference's hot-path modules do `import mlx.core as mx` at import time and mlx is
not installable here.  Three of its functions have bodies that use no mlx arithmetic:

  * mlx_nerf/rendering/ray.py:7-35        get_rays                (numpy only)
  * mlx_nerf/sampling/__init__.py:101-177 sample_from_inverse_cdf_torch (torch only)
  * mlx_nerf/config_parser.py:3-80        config_parser           (stdlib only)

To get past the module-level `import mlx...` lines an EMPTY placeholder module is put in
sys.modules (it provides the name `array` for a type annotation and nothing else -- no
arithmetic; any reference function that really needs MLX would raise AttributeError).
Those three functions are then executed unchanged and their inputs/outputs are saved
as data.  No reference source text is stored.

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _placeholder_mlx():
    mlx = types.ModuleType("mlx")
    core = types.ModuleType("mlx.core")
    nn = types.ModuleType("mlx.nn")
    core.array = type("array", (), {})        # used only in a type annotation (ray.py:7)
    mlx.core, mlx.nn = core, nn
    sys.modules.update({"mlx": mlx, "mlx.core": core, "mlx.nn": nn})


def main():
    assert os.path.isdir(REF), "reference checkout not present (build container only)"
    _placeholder_mlx()
    sys.path.insert(0, REF)
    from mlx_nerf.rendering import ray as ref_ray
    from mlx_nerf import sampling as ref_sampling
    from mlx_nerf import config_parser as ref_cfg

    rng = np.random.default_rng(1234)

    # ---- get_rays -------------------------------------------------------------------
    cases = {}
    def rand_pose():
        a = rng.normal(size=(3, 3)); q, _ = np.linalg.qr(a)
        return np.concatenate([q, rng.normal(size=(3, 1)) * 3], axis=1).astype(np.float32)
    for name, (H, W, f) in {"small": (6, 8, 7.5), "rect": (16, 12, 20.0), "lego64": (64, 64, 88.888885)}.items():
        K = np.array([[f, 0, 0.5 * W], [0, f, 0.5 * H], [0, 0, 1]], dtype=np.float64)
        c2w = rand_pose()
        o, d = ref_ray.get_rays(H, W, K, c2w)
        cases[f"{name}_K"] = K; cases[f"{name}_c2w"] = c2w
        cases[f"{name}_o"] = np.ascontiguousarray(o); cases[f"{name}_d"] = np.ascontiguousarray(d)
        cases[f"{name}_HW"] = np.array([H, W])
    # 800x800 lego intrinsics, sparse pixel subset (keeps the fixture small)
    H = W = 800
    f = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
    K = np.array([[f, 0, 0.5 * W], [0, f, 0.5 * H], [0, 0, 1]], dtype=np.float64)
    c2w = rand_pose()
    o, d = ref_ray.get_rays(H, W, K, c2w)
    idx = rng.choice(H * W, size=512, replace=False)
    idx[:4] = [0, W - 1, (H // 2) * W + W // 2, H * W - 1]
    cases.update({"lego800_K": K, "lego800_c2w": c2w, "lego800_idx": idx.astype(np.int64),
                  "lego800_o": o.reshape(-1, 3)[idx].copy(), "lego800_d": d.reshape(-1, 3)[idx].copy(),
                  "lego800_HW": np.array([H, W])})
    np.savez_compressed(os.path.join(OUT, "ref_get_rays.npz"), **cases)

    # ---- sample_from_inverse_cdf_torch ------------------------------------------------
    s = {}
    def run(tag, z, w, N, seed):
        torch.manual_seed(seed)
        out = ref_sampling.sample_from_inverse_cdf_torch(z.clone(), w.clone(), N)
        torch.manual_seed(seed)
        u = torch.rand(list(z.shape[:-1]) + [N])       # same stream the reference drew (:140)
        s[f"{tag}_z"] = z.numpy(); s[f"{tag}_w"] = w.numpy(); s[f"{tag}_u"] = u.numpy()
        s[f"{tag}_out"] = out.numpy()
    B, n, N = 8, 64, 128
    z_lego = torch.linspace(2.0, 6.0, n).expand(B, n).contiguous()
    run("const", z_lego, torch.full((B, n, 1), 0.02), N, 1)
    run("zero", z_lego, torch.zeros(B, n, 1), N, 2)
    w = torch.from_numpy(rng.random((B, n, 1)).astype(np.float32)) ** 8
    run("peaky", z_lego, w, N, 3)
    w = torch.zeros(B, n, 1); w[:, 17, 0] = 1.0; w[:, 40, 0] = 0.25
    run("spike", z_lego, w, N, 4)
    zj = torch.sort(torch.from_numpy((2 + 4 * rng.random((B, n))).astype(np.float32)), dim=-1).values
    w = torch.from_numpy(rng.random((B, n, 1)).astype(np.float32))
    run("jitter", zj, w, N, 5)
    run("small", torch.linspace(0.5, 1.5, 8).expand(3, 8).contiguous(),
        torch.from_numpy(rng.random((3, 8, 1)).astype(np.float32)), 16, 6)
    w = torch.from_numpy((rng.normal(size=(B, n, 1)) * 0.3).astype(np.float32))   # T>1 quirk can give w<0 / >1
    run("signed", z_lego, w, N, 7)
    np.savez_compressed(os.path.join(OUT, "ref_inverse_cdf.npz"), **s)

    # ---- config_parser defaults -------------------------------------------------------
    args = ref_cfg.config_parser().parse_args(args=[])
    with open(os.path.join(OUT, "ref_config_defaults.json"), "w") as fp:
        json.dump({k: v for k, v in sorted(vars(args).items())}, fp, indent=1)
    # load_config / update_NeRF_args on an upstream-style lego.txt (written here, not reference text)
    lego = {"expname": "blender_paper_lego", "basedir": "./logs", "datadir": "./data/nerf_synthetic/lego",
            "dataset_type": "blender", "no_batching": "True", "use_viewdirs": "True", "white_bkgd": "True",
            "lrate_decay": "500", "N_samples": "64", "N_importance": "128", "N_rand": "1024",
            "precrop_iters": "500", "precrop_frac": "0.5", "half_res": "True"}
    tmp = os.path.join(OUT, "_lego_tmp.txt")
    with open(tmp, "w") as fp:
        fp.write("\n".join(f"{k} = {v}" for k, v in lego.items()) + "\n\n")
    cfg = ref_cfg.load_config(None, tmp)
    args2 = ref_cfg.update_NeRF_args(ref_cfg.config_parser().parse_args(args=[]), cfg)
    os.remove(tmp)
    with open(os.path.join(OUT, "ref_config_lego.json"), "w") as fp:
        json.dump({"file": lego, "loaded": cfg, "args": {k: v for k, v in sorted(vars(args2).items())}}, fp, indent=1)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
