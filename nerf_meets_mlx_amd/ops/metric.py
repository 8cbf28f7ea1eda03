"""MSE / PSNR / SSIM with the reference's call surface (`mlx_nerf/ops/metric.py:12-64`).
SSIM is unfinished upstream (the body stops at "# TODO" after the five windowed moments); here it is
finished with the formula those moments feed and runs as one HIP kernel (csrc/metric.hip).  LPIPS needs a VGG
checkpoint (not on the path, no network): out of scope."""
import ctypes as C
import math

import torch

from .. import _native as N


class MSE:
    def __call__(self, pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
        pred, gt = N.f32(pred), N.f32(gt)
        loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
        N.check(N.lib().nerf_mse_loss_grad(N.ptr(pred), N.ptr(gt), pred.numel(), 1.0, N.ptr(loss), None, N.stream()))
        return loss[0]


class PSNR:
    def __call__(self, pred, gt):
        return 10.0 * torch.log10(1.0 / MSE()(pred, gt))


def gaussian_window(w_size: int, sigma: float = 1.5, ref_quirks: bool = True):
    """1-D window of `SSIM.gaussian` (ops/metric.py:57-64).  Upstream writes `exp(-(x - c)**2) / (2 sigma**2)`: the
    division sits OUTSIDE the exponential and cancels in the normalisation, so the effective window is exp(-(x-c)^2)
    (sigma_eff = 0.707) whatever `sigma` says -- mirrored under ref_quirks; ref_quirks=False is the usual
    exp(-(x-c)^2 / (2 sigma^2))."""
    c = w_size // 2
    if ref_quirks:
        g = [math.exp(-(x - c) ** 2) / float(2 * sigma ** 2) for x in range(w_size)]
    else:
        g = [math.exp(-(x - c) ** 2 / float(2 * sigma ** 2)) for x in range(w_size)]
    t = sum(g)
    return [v / t for v in g]


class SSIM:
    """`SSIM()(pred, gt, w_size=11, size_average=True, full=False)` on [N,C,H,W] tensors (ops/metric.py:20-55).
    Dynamic range from `pred` like upstream (:24-28): max > 128 -> 255 else 1, min < -0.5 -> -1 else 0."""

    def __init__(self, ref_quirks: bool = True):
        self.ref_quirks = ref_quirks

    def __call__(self, pred: torch.Tensor, gt: torch.Tensor, w_size: int = 11, size_average: bool = True, full: bool = False):
        pred, gt = N.f32(pred), N.f32(gt)
        if pred.dim() != 4 or pred.shape != gt.shape:
            raise ValueError(f"SSIM needs two [N,C,H,W] tensors of one shape, got {tuple(pred.shape)} and {tuple(gt.shape)}")
        n, ch, H, W = pred.shape
        _max = 255 if float(pred.max()) > 128 else 1
        _min = -1 if float(pred.min()) < -0.5 else 0
        L = _max - _min
        c1, c2 = (0.01 * L) ** 2, (0.03 * L) ** 2
        win = (C.c_float * w_size)(*gaussian_window(w_size, 1.5, self.ref_quirks))
        sums = torch.empty(n, 2, dtype=torch.float64, device=pred.device)
        N.check(N.lib().nerf_ssim_sums(N.ptr(pred), N.ptr(gt), n, ch, H, W, win, w_size, c1, c2, N.ptr(sums), N.stream()))
        per_image = (sums / float(ch * (H - w_size + 1) * (W - w_size + 1))).to(torch.float32)
        ret = per_image.mean(0) if size_average else per_image.t()
        return (ret[0], ret[1]) if full else ret[0]


def mse_loss_grad(pred, target, grad_scale: float = 1.0):
    """(loss, d loss / d pred): loss = mean((pred-target)^2)  (entrypoints/__test_nerf.py:88)."""
    loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
    d = torch.empty_like(pred)
    N.check(N.lib().nerf_mse_loss_grad(N.ptr(pred), N.ptr(target), pred.numel(), grad_scale, N.ptr(loss), N.ptr(d),
                                       N.stream()))
    return loss, d
