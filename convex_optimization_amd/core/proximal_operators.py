"""Proximal maps with the reference's signatures (deblur_denoise/core/proximal_operators.py:4-57),
evaluated by the HIP kernel behind ddn_prox.  Inputs may live on the CPU or on the GPU; the result
comes back where the input lives."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib as L
from ..plan import compute_device, prox_call


def _run(kind, x, lam):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    src = x.device
    dt = torch.float64 if x.dtype == torch.float64 else torch.float32
    dev = compute_device(src)
    xd = x.detach().to(device=dev, dtype=dt).contiguous()
    if xd.numel() == 0:
        return x.clone()
    y = prox_call(kind, xd, lam)
    return y.to(src)


def prox_l1(x, lambda_val: float):
    """sign(x) * max(|x| - lambda, 0)   (proximal_operators.py:4-8)"""
    return _run(L.PROX_L1, x, lambda_val)


def prox_l2(x, lambda_val: float):
    """Block soft-threshold with the whole-tensor norm (proximal_operators.py:10-14)."""
    return _run(L.PROX_L2, x, lambda_val)


def prox_l2_squared(x, lambda_val: float):
    """x / (2 lambda + 1)   (proximal_operators.py:16-20)"""
    return _run(L.PROX_L2SQ, x, lambda_val)


def prox_iso(x: torch.Tensor, lambda_val: float) -> torch.Tensor:
    """Per-pixel 2-vector shrink of x = [x_1, x_2]; zero where the norm is <= lambda
    (proximal_operators.py:22-50)."""
    if x.shape[0] != 2:
        raise ValueError("prox_iso expects a tensor of shape (2, H, W)")
    return _run(L.PROX_ISO, x, lambda_val)


def prox_box(x: torch.Tensor, lambda_val: float = None) -> torch.Tensor:
    """Projection onto 0 <= x <= 1; lambda is ignored (proximal_operators.py:52-57)."""
    return _run(L.PROX_BOX, x, 0.0)
