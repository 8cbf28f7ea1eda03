"""Deterministic synthetic "Lego-like" scene: there is no Lego data offline.

An analytic teacher field in [-1.5, 1.5]^3 (axis-aligned coloured boxes, sigma in {0, 50})
is rendered with the textbook compositor at 192 uniform samples onto a white background to
make ground-truth images; cameras sit on the upper hemisphere at the Lego radius and the 160
render poses are the reference's (`mlx_nerf/dataset/dataloader.py:68-74`).  This is data
generation in plain torch (any device); it is not part of the measured hot path.
"""
import numpy as np
import torch

from ..ops.pose import pose_spherical

CAMERA_ANGLE_X = 0.6911112070083618     # Lego transforms_*.json
LEGO_RADIUS = 4.031128874

# (centre, half-size, rgb)
_BOXES = [
    ((0.0, 0.0, -0.45), (1.1, 0.7, 0.10), (0.55, 0.55, 0.55)),    # base plate
    ((-0.45, 0.0, -0.10), (0.45, 0.45, 0.25), (0.90, 0.75, 0.10)),  # yellow brick
    ((0.50, 0.15, -0.15), (0.35, 0.30, 0.20), (0.80, 0.10, 0.10)),  # red brick
    ((-0.45, 0.0, 0.35), (0.25, 0.25, 0.20), (0.10, 0.35, 0.80)),   # blue brick on top
    ((0.50, 0.15, 0.15), (0.10, 0.10, 0.10), (0.95, 0.95, 0.95)),   # stud
    ((0.15, -0.45, -0.20), (0.12, 0.12, 0.15), (0.10, 0.65, 0.25)),  # green post
]


def teacher_field(pts: torch.Tensor):
    """pts [...,3] -> (sigma [...], rgb [...,3]); later boxes paint over earlier ones."""
    sigma = torch.zeros(pts.shape[:-1], dtype=pts.dtype, device=pts.device)
    rgb = torch.zeros_like(pts)
    for c, h, col in _BOXES:
        c_t = torch.tensor(c, dtype=pts.dtype, device=pts.device)
        h_t = torch.tensor(h, dtype=pts.dtype, device=pts.device)
        inside = ((pts - c_t).abs() <= h_t).all(dim=-1)
        sigma = torch.where(inside, torch.full_like(sigma, 50.0), sigma)
        rgb = torch.where(inside[..., None], torch.tensor(col, dtype=pts.dtype, device=pts.device).expand_as(rgb), rgb)
    return sigma, rgb


def intrinsics(H: int, W: int):
    f = 0.5 * W / np.tan(0.5 * CAMERA_ANGLE_X)
    return np.array([[f, 0, 0.5 * W], [0, f, 0.5 * H], [0, 0, 1]], dtype=np.float64), f


def train_poses(n: int, seed: int = 0) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    th = rng.uniform(-180.0, 180.0, size=n)
    ph = -np.degrees(np.arcsin(rng.uniform(0.05, 0.95, size=n)))       # upper hemisphere, area-uniform
    return torch.stack([pose_spherical(float(t), float(p), LEGO_RADIUS) for t, p in zip(th, ph)], 0)


def render_poses() -> torch.Tensor:
    """160 poses, theta in linspace(-180,180,161)[:-1], phi=-30, r=4 (dataloader.py:68-74)."""
    return torch.stack([pose_spherical(float(a), -30.0, 4.0) for a in np.linspace(-180, 180, 160 + 1)[:-1]], 0)


@torch.no_grad()
def render_gt(H: int, W: int, c2w: torch.Tensor, near=2.0, far=6.0, n_samples=192, device="cpu", chunk=1 << 16,
              dtype=torch.float32) -> torch.Tensor:
    """[H,W,3] ground-truth image of the teacher field on a white background."""
    K, _ = intrinsics(H, W)
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    dirs = torch.stack([(i - K[0, 2]) / K[0, 0], -(j - K[1, 2]) / K[1, 1], -torch.ones_like(i)], -1).reshape(-1, 3)
    R = c2w[:3, :3].double()
    d = (dirs @ R.T).to(dtype).to(device)
    o = c2w[:3, 3].to(dtype).to(device)
    t = torch.linspace(near, far, n_samples, dtype=dtype, device=device)
    out = torch.empty(H * W, 3, dtype=dtype, device=device)
    for s in range(0, H * W, chunk):
        dd = d[s:s + chunk]
        pts = o + dd[:, None, :] * t[None, :, None]
        sigma, rgb = teacher_field(pts)
        delta = torch.cat([t[1:] - t[:-1], t[-1:] * 0 + 1e10]) * dd.norm(dim=-1, keepdim=True)
        alpha = 1.0 - torch.exp(-sigma * delta)
        T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
        w = alpha * T
        out[s:s + chunk] = (w[..., None] * rgb).sum(1) + (1.0 - w.sum(1, keepdim=True))
    return out.reshape(H, W, 3)


def make_dataset(H: int, W: int, n_train: int, seed: int = 0, device="cpu"):
    """(images [N,H,W,3] on `device`, poses [N,4,4] cpu, render_poses [160,4,4], [H,W,focal], K)."""
    poses = train_poses(n_train, seed)
    imgs = torch.stack([render_gt(H, W, p, device=device) for p in poses], 0)
    K, f = intrinsics(H, W)
    return imgs, poses, render_poses(), [H, W, f], K
