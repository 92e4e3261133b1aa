"""circular_convolve2d with the reference's signature (deblur_denoise/core/convolution.py:22-52)."""
from __future__ import annotations

import numpy as np
import torch

from ..plan import create_plan, compute_device, real_dtype, stage


def circular_convolve2d(image, kernel, **kwargs):
    """FFT circular convolution of a (C, H, W) image with a k x k kernel; the float32 zero-padded
    kernel is rolled by (-kh//2, -kw//2) exactly as conv:38-40 does (so the blur is displaced by one
    pixel relative to a centred correlation, and is the flip of K for asymmetric kernels)."""
    if isinstance(image, np.ndarray):
        image = torch.from_numpy(image)
    if image.dim() != 3:
        raise ValueError("circular_convolve2d expects an image of shape (C, H, W)")
    src = image.device
    dt = real_dtype(kwargs.get("dtype", image.dtype))
    dev = compute_device(kwargs.get("device", src))
    Cn, H, W = image.shape
    plan = create_plan(H, W, Cn, dt, dev)
    try:
        out = plan.circular_convolve2d(kernel, stage(image, dev, dt))
        torch.cuda.synchronize(dev)
    finally:
        plan.close()
    return out.to(src)
