"""Blur kernel generators (deblur_denoise/core/blur.py).  Host-side, k x k, evaluated once per
problem; they must reproduce the reference bit for bit because they are parity inputs."""
from __future__ import annotations

import numpy as np
import torch


def create_motion_blur_kernel(size=15, angle=0) -> torch.Tensor:
    """blur.py:4-49.  A horizontal line of ones; for angle != 0 every cell (i, j) of the k x k grid is
    rotated about the centre with truncating int() and its image set to 1, so the result is a rotated
    filled square rather than a rotated line (156 of 225 non-zeros for 15 @ 45 degrees)."""
    if size % 2 == 0:
        size += 1
    kern = torch.zeros((size, size))
    centre = size // 2
    kern[centre, :] = 1
    if angle != 0:
        theta = np.radians(angle)
        rot = torch.tensor([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])
        for i in range(size):
            dx = i - centre
            for j in range(size):
                dy = j - centre
                ti = int(centre + dx * rot[0, 0] + dy * rot[0, 1])
                tj = int(centre + dx * rot[1, 0] + dy * rot[1, 1])
                if 0 <= ti < size and 0 <= tj < size:
                    kern[ti, tj] = 1
    return kern / kern.sum()


def gaussian_filter(size, sigma):
    """blur.py:52-62: normalised float64 Gaussian on a linspace grid (size = [rows, cols])."""
    gx, gy = np.meshgrid(np.linspace(-(size[0] // 2), size[0] // 2, size[0]),
                         np.linspace(-(size[1] // 2), size[1] // 2, size[1]))
    h = np.exp(-(gx ** 2 + gy ** 2) / (2 * sigma ** 2))
    return h / h.sum()
