"""MSE / PSNR with the reference's call surface (`mlx_nerf/ops/metric.py:12-18`).
SSIM / LPIPS are unfinished upstream and out of scope."""
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


def mse_loss_grad(pred, target, grad_scale: float = 1.0):
    """(loss, d loss / d pred): loss = mean((pred-target)^2)  (entrypoints/__test_nerf.py:88)."""
    loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
    d = torch.empty_like(pred)
    N.check(N.lib().nerf_mse_loss_grad(N.ptr(pred), N.ptr(target), pred.numel(), grad_scale, N.ptr(loss), N.ptr(d),
                                       N.stream()))
    return loss, d
