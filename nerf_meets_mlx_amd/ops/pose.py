"""Spherical camera poses (`mlx_nerf/ops/pose.py:7-58`): host-side 4x4 float32 algebra."""
import numpy as np
import torch


def pose_spherical(theta: float, phi: float, radius: float) -> torch.Tensor:
    """c2w = swap . R_y(theta) . R_x(phi) . T_z(radius); angles in degrees."""
    def f32(rows):
        return np.array(rows, dtype=np.float32)
    ph, th = phi / 180.0 * np.pi, theta / 180.0 * np.pi
    t = f32([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]])
    rp = f32([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]])
    rt = f32([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]])
    swap = f32([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
    return torch.from_numpy(swap @ (rt @ (rp @ t)))
