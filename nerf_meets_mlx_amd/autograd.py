"""torch.autograd glue: the reference differentiates `render_rays -> raw2outputs -> MSE` with
`nn.value_and_grad(model, loss_fn)` (`entrypoints/__test_nerf.py:47-145`).  These two Functions put the HIP
forward/backward kernels behind `loss.backward()` so that user-written losses keep working:

    params = model.trainable()                      # flat float32 leaf, requires_grad
    out = render_rays_grad(rays, model, n_depth_samples=64, white_bkgd=True)
    loss = ((out["rgb_map"] - y) ** 2).mean()
    loss.backward()                                 # params.grad = dL/dparams  (nerf_mlp_backward et al.)
    optimizer.update(model, params.grad)

The Trainer does not go through autograd (it calls the kernels directly); this module is API parity.
"""
import torch

from . import sampling
from .rendering import render


class FusedQuery(torch.autograd.Function):
    """raw[B,n,4] = MLP(PE(o + z d), PE(viewdirs)); gradient w.r.t. the flat parameter buffer only (the sample
    positions carry no gradient in the reference either: z is detached, rays are data)."""

    @staticmethod
    def forward(ctx, params, rays, z, model, ref_quirks):
        if params.data_ptr() != model.params.data_ptr():
            model.load_flat(params.detach())
        ctx.model = model
        raw = model.query(rays, z, ref_quirks=ref_quirks, train=True)
        ctx.generation = model.activation_generation()     # a second graph on the same model invalidates this one
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        g = ctx.model.backward(d_raw.contiguous(), generation=ctx.generation)
        return g.clone(), None, None, None, None


class Composite(torch.autograd.Function):
    """`raw2outputs` (rendering/render.py:20-96) with gradients for raw from rgb / acc / depth."""

    @staticmethod
    def forward(ctx, raw, z, rays, white_bkgd):
        rgb, disp, acc, weights, depth = render.composite(raw, z, rays, 0.0, white_bkgd)
        ctx.save_for_backward(raw, z, rays)
        ctx.white = white_bkgd
        ctx.mark_non_differentiable(disp, weights)
        return rgb, disp, acc, weights, depth

    @staticmethod
    def backward(ctx, d_rgb, d_disp, d_acc, d_weights, d_depth):
        raw, z, rays = ctx.saved_tensors
        zero = lambda t, shape: torch.zeros(shape, device=raw.device) if t is None else t.contiguous()
        B = z.shape[0]
        d_raw = render.composite_backward(raw, z, rays, zero(d_rgb, (B, 3)), ctx.white, zero(d_acc, (B,)),
                                          zero(d_depth, (B,)))
        return d_raw, None, None, None


def render_rays_grad(rays, model, n_depth_samples=64, white_bkgd=False, ref_quirks=True, z=None):
    """Differentiable coarse pass (what `mlx_mse_coarse` differentiates, __test_nerf.py:47-90); pass `z` for the
    fine pass on importance-sampled depths (`mlx_mse_fine`, :93-126)."""
    if z is None:
        z = sampling.sample_coarse(rays, n_depth_samples)
    raw = FusedQuery.apply(model.trainable(), rays, z, model, ref_quirks)
    rgb, disp, acc, weights, depth = Composite.apply(raw, z, rays, white_bkgd)
    return {"rgb_map": rgb, "disp_map": disp[:, None], "acc_map": acc[:, None], "weights": weights[..., None],
            "depth_map": depth[:, None], "z_vals": z, "raw": raw}
