"""Loss / stop metrics with the reference's signatures (deblur_denoise/core/loss.py).
l1_loss, l2_loss, mse and psnr are wavefront reductions on the GPU (ddn_reduce); inside the solver
loops the same four values are produced by the fused row kernels.  ssim (loss.py:39-110) is one fused
device pass (ddn_ssim); inside the solver loops it is evaluated between the launches of an iteration
without a host round trip (ddn_plan_set_ssim).  There is no host implementation of any of them."""
from __future__ import annotations

import torch

from .. import _lib as L
from ..plan import compute_device, reduce_call


def _four(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    src = x.device
    dt = torch.float64 if x.dtype == torch.float64 else torch.float32
    dev = compute_device(src)
    xd = x.detach().to(device=dev, dtype=dt)
    yd = y.detach().to(device=dev, dtype=dt)
    xd, yd = torch.broadcast_tensors(xd, yd)
    out = reduce_call(xd.contiguous(), yd.contiguous(), 1)[0]
    return out.to(device=src, dtype=dt)


def l1_loss(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """torch.norm(x - y, p=1)   (loss.py:5-10)"""
    return _four(x, y)[0]


def l2_loss(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """torch.norm(x - y, p=2)   (loss.py:12-17)"""
    return _four(x, y)[1]


def mse(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """mean((x - y)^2)   (loss.py:112-117)"""
    return _four(x, y)[2]


def psnr(x: torch.Tensor, y: torch.Tensor, max_val: float = 1.0) -> torch.Tensor:
    """10 log10(max_val^2 / mse), inf when mse == 0   (loss.py:19-36)"""
    m = _four(x, y)[2]
    if m == 0:
        return torch.tensor(float("inf"))
    return 10 * torch.log10(max_val ** 2 / m)


l1_loss._ddn_kind = L.LOSS_L1
l2_loss._ddn_kind = L.LOSS_L2
mse._ddn_kind = L.LOSS_MSE
psnr._ddn_kind = L.LOSS_PSNR


def ssim_window_taps(window_size: int = 11) -> torch.Tensor:
    """The normalised float32 factors of the reference's Gaussian window (loss.py:75-81), evaluated with the same
    torch expression so that they agree bit for bit."""
    co = torch.arange(window_size, dtype=torch.float32) - window_size // 2
    g = torch.exp(-(co ** 2) / (2 * 1.5 ** 2))
    return (g / g.sum()).contiguous()


def ssim(x: torch.Tensor, y: torch.Tensor, window_size: int = 11, max_val: float = 1.0) -> torch.Tensor:
    """SSIM with a Gaussian (sigma 1.5) window and zero padding (loss.py:39-110): one fused pass over the two images on
    the GPU for the default 11 x 11 window (ddn_ssim), direct evaluation of the taps for any other window_size
    (ddn_ssim_window; an even size gives a map one row and one column larger than the image, as in the reference).
    Inputs on the host are staged to the device; the result returns where x lives."""
    def to4(t):
        return t[None, None] if t.dim() == 2 else (t[None] if t.dim() == 3 else t)
    x, y = to4(x), to4(y)
    if x.shape != y.shape:
        raise ValueError(f"Input tensors must have the same shape, got {x.shape} and {y.shape}")
    window_size = int(window_size)
    if window_size < 1:
        raise ValueError(f"window_size must be positive, got {window_size}")
    import ctypes as C
    from ..plan import _ptr
    src = x.device
    dt = torch.float64 if x.dtype == torch.float64 else torch.float32
    dev = compute_device(src)
    B, nc, H, W = x.shape
    xc = x.detach().to(device=dev, dtype=dt).contiguous()
    yc = y.detach().to(device=dev, dtype=dt).contiguous()
    out = torch.empty((B * nc,), dtype=torch.float64, device=dev)
    g = (C.c_float * window_size)(*ssim_window_taps(window_size).tolist())
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if window_size == 11:
            L.check(L.lib().ddn_ssim(L.F64 if dt == torch.float64 else L.F32, _ptr(xc), _ptr(yc), H, W, B * nc, float(max_val), g, _ptr(out), st))
        else:
            L.check(L.lib().ddn_ssim_window(L.F64 if dt == torch.float64 else L.F32, _ptr(xc), _ptr(yc), H, W, B * nc, float(max_val), g, window_size,
                                            _ptr(out), st))
    return out.view(B, nc).mean(dim=1).to(device=src, dtype=dt)


ssim._ddn_kind = L.LOSS_SSIM
