"""Volume rendering with the reference's call surface (`mlx_nerf/rendering/render.py`):

    render(...) -> batchify_rays(...) -> render_rays / render_rays_eval(...) -> raw2outputs(...)

Every stage is one HIP launch: coarse depths (`nerf_sample_coarse`), the fused
PE + MLP query (`nerf_query_fused`), alpha compositing (`nerf_composite_forward`) and the
importance sampler + merge (`nerf_importance_sample`).  The reference's numpy ray-gen and the
torch-CPU sampler round trip (:215-223) do not exist here.
"""
from typing import Dict, Optional

import torch

from .. import _native as N
from .. import sampling
from . import ray


def composite(raw, z_vals, rays, raw_noise_std=0.0, white_bkgd=False, noise=None, need_weights=True):
    """raw [B,n,4], z [B,n], packed rays [B,11] -> (rgb [B,3], disp [B], acc [B], weights [B,n], depth [B])."""
    B, n = z_vals.shape
    dev = raw.device
    rgb = torch.empty(B, 3, dtype=torch.float32, device=dev)
    disp = torch.empty(B, dtype=torch.float32, device=dev)
    acc = torch.empty(B, dtype=torch.float32, device=dev)
    depth = torch.empty(B, dtype=torch.float32, device=dev)
    weights = torch.empty(B, n, dtype=torch.float32, device=dev) if need_weights else None
    if raw_noise_std > 0.0 and noise is None:
        noise = torch.randn(B, n, dtype=torch.float32, device=dev)            # mx.random.normal (:42)
    N.check(N.lib().nerf_composite_forward(N.ptr(raw), N.ptr(z_vals), N.ptr(rays), B, n, float(raw_noise_std),
                                           N.ptr(noise) if raw_noise_std > 0.0 else None, int(bool(white_bkgd)),
                                           N.ptr(rgb), N.ptr(disp), N.ptr(acc), N.ptr(weights), N.ptr(depth), N.stream()))
    return rgb, disp, acc, weights, depth


def composite_backward(raw, z_vals, rays, d_rgb, white_bkgd=False, d_acc=None, d_depth=None, raw_noise_std=0.0,
                       noise=None):
    """d_raw [B,n,4] for upstream gradients of rgb (and optionally acc / depth)."""
    B, n = z_vals.shape
    d_raw = torch.empty(B, n, 4, dtype=torch.float32, device=raw.device)
    N.check(N.lib().nerf_composite_backward(N.ptr(raw), N.ptr(z_vals), N.ptr(rays), B, n, float(raw_noise_std),
                                            N.ptr(noise) if raw_noise_std > 0.0 else None, int(bool(white_bkgd)),
                                            N.ptr(d_rgb), N.ptr(d_acc), N.ptr(d_depth), N.ptr(d_raw), N.stream()))
    return d_raw


def composite_mse_backward(raw, z_vals, rays, target, white_bkgd=False, grad_scale: float = 1.0, need_rgb: bool = False):
    """(loss [1], d_raw [B,n,4], rgb [B,3] or None): raw2outputs + MSE + their gradient w.r.t. raw in one launch
    (`nerf_composite_mse_backward`); same values as composite + mse_loss_grad + composite_backward."""
    B, n = z_vals.shape
    loss = torch.zeros(1, dtype=torch.float32, device=raw.device)
    d_raw = torch.empty(B, n, 4, dtype=torch.float32, device=raw.device)
    rgb = torch.empty(B, 3, dtype=torch.float32, device=raw.device) if need_rgb else None
    N.check(N.lib().nerf_composite_mse_backward(N.ptr(raw), N.ptr(z_vals), N.ptr(rays), B, n, int(bool(white_bkgd)),
                                                N.ptr(N.f32(target)), float(grad_scale), N.ptr(loss), N.ptr(rgb),
                                                N.ptr(d_raw), N.stream()))
    return loss, d_raw, rgb


_WS = {}


def render_rays_fused(rays, network_coarse, network_fine, n_depth_samples, N_importance, u=None, white_bkgd=False,
                      ref_quirks=True, with_coarse=True):
    """render_rays_eval as ONE C call (`nerf_render_rays_fused`): same kernels, same results as the staged path."""
    import ctypes as C
    rays = N.f32(rays)
    B, n, Nn = rays.shape[0], int(n_depth_samples), int(N_importance or 0)
    dev = rays.device
    if Nn > 0 and u is None:
        u = torch.rand(B, Nn, dtype=torch.float32, device=dev)
    nbytes = N.lib().nerf_render_workspace_bytes(B, n, Nn)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)      # one scratch buffer per (device, stream): two streams that
    ws = _WS.get(key)                                            # render concurrently must not share intermediate z / raw
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _WS[key] = ws
    f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
    rgb, disp, acc = f(B, 3), f(B), f(B)
    ret = {}
    if with_coarse:
        ret = {"rgb_coarse": f(B, 3), "disp_coarse": f(B), "acc_coarse": f(B), "z_vals": f(B, n), "weights": f(B, n)}
    fine = network_fine if network_fine else network_coarse
    if getattr(fine, "precision", 16) != getattr(network_coarse, "precision", 16):
        raise ValueError("render_rays_fused: coarse and fine networks must have the same precision (one nerf_mlp_arch per call)")
    N.check(N.lib().nerf_render_rays_fused(
        C.byref(network_coarse.arch), N.ptr(network_coarse.packed()), N.ptr(fine.packed()), N.ptr(rays), B, n, Nn,
        N.ptr(u) if Nn > 0 else None, 0 if ref_quirks else 1, int(bool(white_bkgd)), N.ptr(ws), N.ptr(rgb), N.ptr(disp),
        N.ptr(acc), N.ptr(ret.get("rgb_coarse")), N.ptr(ret.get("disp_coarse")), N.ptr(ret.get("acc_coarse")),
        N.ptr(ret.get("z_vals")), N.ptr(ret.get("weights")), N.stream()))
    out = {"rgb_map": rgb, "disp_map": disp[:, None], "acc_map": acc[:, None]}
    if with_coarse:
        if Nn == 0:
            ret["disp_coarse"], ret["acc_coarse"] = disp, acc
        out.update({"rgb_coarse": ret["rgb_coarse"], "disp_coarse": ret["disp_coarse"][:, None],
                    "acc_coarse": ret["acc_coarse"][:, None], "z_vals": ret["z_vals"], "weights": ret["weights"][..., None]})
    return out


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, noise=None):
    """`rendering/render.py:20-96`: returns (rgb_map [B,3], disp_map [B,1], acc_map [B,1],
    weights [B,n,1], depth_map [B,1]) -- shapes as upstream."""
    raw, z_vals = N.f32(raw), N.f32(z_vals)
    B = z_vals.shape[0]
    rays = torch.zeros(B, 11, dtype=torch.float32, device=raw.device)
    rays[:, 3:6] = rays_d
    rgb, disp, acc, w, depth = composite(raw, z_vals, rays, float(raw_noise_std), white_bkgd, noise)
    return rgb, disp[:, None], acc[:, None], w[..., None], depth[:, None]


def decompose_ray_batch(rays_batch_linear, is_time_included: bool = False):
    """`rendering/render.py:98-110`."""
    r = rays_batch_linear
    rays_o, rays_d = r[:, 0:3], r[:, 3:6]
    bounds = r[..., 6:8 + int(is_time_included)].reshape(-1, 1, 2 + int(is_time_included))
    near, far = bounds[..., 0], bounds[..., 1]
    frame_time = bounds[..., 2] if is_time_included else None
    return rays_o, rays_d, near, far, r[:, -3:], frame_time


def _query(network_query_fn, rays, z, model):
    if hasattr(network_query_fn, "fused") and hasattr(model, "query"):
        return network_query_fn.fused(rays, z, model)
    o, d, _, _, viewdirs, _ = decompose_ray_batch(rays)
    pos = o[..., None, :] + z[..., :, None] * d[..., None, :]
    return network_query_fn(pos, viewdirs, model)


def _coarse_pass(rays, network_coarse, network_query_fn, n_depth_samples, retraw, lindisp, perturb, white_bkgd,
                 raw_noise_std):
    rays = N.f32(rays)
    z_vals = sampling.sample_coarse(rays, n_depth_samples, lindisp=lindisp, perturb=float(perturb or 0.0))
    raw = _query(network_query_fn, rays, z_vals, network_coarse)
    rgb, disp, acc, weights, depth = composite(raw, z_vals, rays, float(raw_noise_std or 0.0), white_bkgd)
    ret = {}
    if retraw:
        ret["raw"] = raw
    ret.update({"rgb_map": rgb, "disp_map": disp[:, None], "acc_map": acc[:, None], "rgb_coarse": rgb,
                "disp_coarse": disp[:, None], "acc_coarse": acc[:, None], "z_vals": z_vals,
                "weights": weights[..., None]})
    return rays, ret


def render_rays(rays_batch_linear, network_coarse, network_query_fn, n_depth_samples, retraw=False, lindisp=False,
                perturb=0.0, N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0.0, verbose=False,
                pytest=False, **kwargs):
    """Coarse-only pass (`rendering/render.py:112-162`); ignores N_importance / network_fine like upstream."""
    return _coarse_pass(rays_batch_linear, network_coarse, network_query_fn, n_depth_samples, retraw, lindisp, perturb,
                        white_bkgd, raw_noise_std)[1]


def render_rays_eval(rays_batch_linear, network_coarse, network_query_fn, n_depth_samples, retraw=False, lindisp=False,
                     perturb=0.0, N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0.0,
                     verbose=False, pytest=False, u=None, **kwargs):
    """Coarse pass + importance sampling + sort + second pass (`rendering/render.py:164-241`).
    Upstream always runs the second pass (with network_coarse when there is no fine net, Q19);
    with N_importance == 0 that pass would re-evaluate identical samples, so it is skipped."""
    rays, ret = _coarse_pass(rays_batch_linear, network_coarse, network_query_fn, n_depth_samples, retraw, lindisp,
                             perturb, white_bkgd, raw_noise_std)
    if N_importance and N_importance > 0:
        _, z_fine = sampling.importance_sample(ret["z_vals"], ret["weights"], N_importance, u=u)
        run_fn = network_fine if network_fine else network_coarse
        raw = _query(network_query_fn, rays, z_fine, run_fn)
        rgb, disp, acc, _, _ = composite(raw, z_fine, rays, float(raw_noise_std or 0.0), white_bkgd, need_weights=False)
        ret["rgb_map"], ret["disp_map"], ret["acc_map"] = rgb, disp[:, None], acc[:, None]
    return ret


def batchify_rays(rays_linear, chunk=1024 * 32, **kwargs):
    """`rendering/render.py:243-266`."""
    render_rays_func = kwargs["render_rays_func"]
    u_all = kwargs.pop("u", None)
    results: Dict[str, list] = {}
    for i in range(0, rays_linear.shape[0], chunk):
        extra = {} if u_all is None else {"u": u_all[i:i + chunk]}
        out = render_rays_func(rays_linear[i:i + chunk], **kwargs, **extra)
        for k, v in out.items():
            results.setdefault(k, []).append(v)
    return {k: torch.cat(v, dim=0) for k, v in results.items()}


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0.0, far=1.0, use_viewdirs=False,
           c2w_staticcam=None, device="cuda", **kwargs):
    """`rendering/render.py:268-345`: returns [rgb_map, disp_map, acc_map, extras_dict] reshaped to
    the ray grid.  Ray generation runs on the device (no host image of rays)."""
    if c2w is not None:
        packed = ray.gen_rays(H, W, K, c2w, near, far, None, device)
        rays_shape = (H, W, 3)
        if c2w_staticcam is not None:                         # viewdirs from c2w, geometry from the static camera
            st = ray.gen_rays(H, W, K, c2w_staticcam, near, far, None, device)
            st[:, 8:11] = packed[:, 8:11]
            packed = st
    else:
        rays_o, rays_d = rays
        rays_shape = tuple(rays_d.shape)
        rays_o, rays_d = N.f32(rays_o).reshape(-1, 3), N.f32(rays_d).reshape(-1, 3)
        packed = torch.empty(rays_o.shape[0], 11, dtype=torch.float32, device=rays_o.device)
        packed[:, 0:3], packed[:, 3:6] = rays_o, rays_d
        packed[:, 6], packed[:, 7] = near, far
        packed[:, 8:11] = rays_d / torch.linalg.norm(rays_d, dim=-1, keepdim=True)
    if ndc:
        o, d = ray.ndc_rays(H, W, K[0][0], 1.0, packed[:, 0:3].contiguous(), packed[:, 3:6].contiguous())
        packed[:, 0:3], packed[:, 3:6] = o, d
    if not use_viewdirs:
        raise ValueError("only the view-dependent model is a supported volume path (SURVEY Q16)")
    res = batchify_rays(packed, chunk, **kwargs)
    for k, v in res.items():
        res[k] = v.reshape(tuple(rays_shape[:-1]) + tuple(v.shape[1:]))
    keys = ["rgb_map", "disp_map", "acc_map"]
    return [res[k] for k in keys] + [{k: v for k, v in res.items() if k not in keys}]
