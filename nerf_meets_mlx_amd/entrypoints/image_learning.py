"""Headless 2-D image fitting (BASELINE.json configs[0]; `mlx_nerf/entrypoints/__viser_image_learning.py:186-288`
without the viser GUI / mp4 writer, which are UI and out of scope).

    embed = SinusoidalEncoding(2, 10, 0.0, 8.0, include_input=False)      (:198)   40 channels
    model = NeRF(channel_input=40, channel_input_views=0, channel_output=3, is_use_view_directions=False)   (:203-208)
    Adam(lr 1e-3, betas (0.9, 0.99))                                      (:224-227)
    per epoch: a fresh permutation of all H*W pixels cut into batches of H*W/64 INTEGER (row, col)
    coordinates (:86-124, 271), loss = mean((model(embed(X)) - y)^2) (:210-219), one Adam step per batch.
"""
from typing import Optional

import numpy as np
import torch

from ..encoding import SinusoidalEncoding
from ..models.NeRF import Adam, NeRF
from ..ops import index
from ..ops.metric import mse_loss_grad


def load_image(path: str, size=(400, 400), device="cuda") -> torch.Tensor:
    """[H,W,3] float32 in [0,1]; single-channel images are repeated to 3 channels (:126-162)."""
    from PIL import Image
    img = np.asarray(Image.open(path).resize(size))
    if img.ndim == 2:
        img = np.repeat(img[..., None], 3, axis=-1)
    img = img[..., :3]
    return torch.from_numpy(img.astype(np.float32) / 255.0).to(device)


class ImageFitter:
    def __init__(self, image: torch.Tensor, batch_downsample_factor: int = 64, lr: float = 1e-3, seed: int = 0,
                 precision: int = 22):
        """precision: `NeRF(precision=...)`; the default (22) reproduces the reference's float32 fit (:198-236) at the float32
        tolerance, 32 runs literal float32 operands on the fp32 MFMA, 16 is the declared reduced-precision bf16 mode."""
        self.img = image.contiguous()
        self.H, self.W = image.shape[0], image.shape[1]
        self.dev = image.device
        self.embed = SinusoidalEncoding(2, 10, min_freq_exp=0.0, max_freq_exp=8.0, is_include_input=False)
        self.model = NeRF(channel_input=self.embed.get_out_dim(), channel_input_views=0, channel_output=3,
                          is_use_view_directions=False, device=self.dev, seed=seed, precision=precision)
        self.opt = Adam(learning_rate=lr, betas=(0.9, 0.99), shared_state=False)
        self.batch = self.H * self.W // batch_downsample_factor
        self.seed, self.epoch = seed, 0

    def batch_iterate(self, perm: Optional[torch.Tensor] = None):
        """Yields (X [B,2] integer (row, col) coordinates as float, y [B,3]) like `batch_iterate` (:86-124)."""
        if perm is None:
            perm = index.pixel_permutation(self.H * self.W, self.H * self.W, self.seed * 7919 + self.epoch, 0, self.dev)
        flat = self.img.reshape(-1, 3)
        for s in range(0, self.H * self.W, self.batch):
            sel = perm[s:s + self.batch]
            X = torch.stack([sel // self.W, sel % self.W], dim=-1).to(torch.float32)
            yield X, index.gather_rows(flat, sel)

    def step(self, X: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        pred = self.model.forward(self.embed(X), train=True)
        loss, d_pred = mse_loss_grad(pred, y)
        self.opt.update(self.model, self.model.backward(d_pred))
        return loss

    def train_epoch(self, perm: Optional[torch.Tensor] = None) -> torch.Tensor:
        loss = None
        for X, y in self.batch_iterate(perm):
            loss = self.step(X, y)
        self.epoch += 1
        return loss

    def predict(self) -> torch.Tensor:
        """[H,W,3] prediction at every integer pixel coordinate (:282-288)."""
        rc = torch.stack(torch.meshgrid(torch.arange(self.H, device=self.dev), torch.arange(self.W, device=self.dev),
                                        indexing="ij"), -1).reshape(-1, 2).float()
        return self.model.forward(self.embed(rc)).reshape(self.H, self.W, 3)


def main(path_image: str, batch_downsample_factor: int = 64, max_frames: int = 600, device="cuda", precision: int = 22):
    fit = ImageFitter(load_image(path_image, device=device), batch_downsample_factor, precision=precision)
    for _ in range(max_frames):
        loss = fit.train_epoch()
    mse = torch.mean((fit.predict() - fit.img) ** 2)
    return float(loss), float(10.0 * torch.log10(1.0 / mse))
