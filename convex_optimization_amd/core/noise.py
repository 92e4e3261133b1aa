"""Noise models used to synthesise test inputs (deblur_denoise/core/noise.py:4-54).  Input
synthesis is outside the iteration hot path; these draw from torch's RNG on the tensor's device."""
from __future__ import annotations

import numpy as np
import torch


def _t(image):
    return torch.from_numpy(image) if isinstance(image, np.ndarray) else image


def add_gaussian_noise(image, mean=0.0, std=0.1) -> torch.Tensor:
    image = _t(image)
    return torch.clamp(image + torch.randn_like(image) * std + mean, 0, 1)


def add_salt_pepper_noise(image, salt_prob=0.01, pepper_prob=0.01) -> torch.Tensor:
    image = _t(image)
    out = image.clone()
    r = torch.rand_like(image)
    out[r < salt_prob] = 1.0
    out[(r >= salt_prob) & (r < salt_prob + pepper_prob)] = 0.0
    return out


def add_poisson_noise(image, scale=1.0) -> torch.Tensor:
    image = _t(image)
    return torch.clamp(torch.poisson(image * 255.0 * scale) / (255.0 * scale), 0, 1)


def add_speckle_noise(image, std=0.1) -> torch.Tensor:
    image = _t(image)
    return torch.clamp(image + image * (torch.randn_like(image) * std), 0, 1)
