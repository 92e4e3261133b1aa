"""X^T w for the blind-kernel extension (deblur_denoise/op_math/python_code/apply_x_trans.py:4-46)."""
from __future__ import annotations

import torch

from .. import _lib as L
from ..plan import _ptr, compute_device, real_dtype, stage


def apply_x_trans(w: torch.Tensor, x: torch.Tensor, kernel_size: int, **kwargs) -> torch.Tensor:
    """(X^T w)_(ij) = <w, conv2(x, impulse at (i, j), 'same')> for kernel_size x kernel_size w and x.
    The result lives on w's device (axt:26-27).  Even kernel sizes raise, as in the reference, whose
    zero-padded correlation is then (k+1) x (k+1) and cannot be multiplied with w (axt:40-44)."""
    k = int(kernel_size)
    if tuple(w.shape) != (k, k) or tuple(x.shape) != (k, k):
        raise RuntimeError(f"apply_x_trans expects {k} x {k} tensors, got {tuple(w.shape)} and {tuple(x.shape)}")
    if k % 2 == 0:
        raise RuntimeError(f"The size of tensor a ({k}) must match the size of tensor b ({k + 1}) at non-singleton dimension 1")
    src = w.device
    dt = real_dtype(kwargs.get("dtype", torch.float32))       # axt:26: the reference accumulates into a float32 array
    dev = compute_device(kwargs.get("device", src))
    wd, xd = stage(w, dev, dt).contiguous(), stage(x, dev, dt).contiguous()
    out = torch.empty((k, k), dtype=dt, device=dev)
    with torch.cuda.device(dev):
        L.check(L.lib().ddn_apply_x_trans(L.F64 if dt == torch.float64 else L.F32, _ptr(wd), _ptr(xd), k, _ptr(out), None))
        torch.cuda.synchronize(dev)
    return out.to(src)
