"""Loss / stop metrics with the reference's signatures (deblur_denoise/core/loss.py).
l1_loss, l2_loss, mse and psnr are wavefront reductions on the GPU (ddn_reduce); inside the solver
loops the same four values are produced by the fused row kernels.  ssim (loss.py:39-110) is not yet
a fused kernel: it is evaluated with torch ops on whatever device its inputs live on."""
from __future__ import annotations

import torch
import torch.nn.functional as F

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


def ssim(x: torch.Tensor, y: torch.Tensor, window_size: int = 11, max_val: float = 1.0) -> torch.Tensor:
    """SSIM with an 11x11 Gaussian (sigma 1.5) window and zero padding (loss.py:39-110)."""
    def to4(t):
        return t[None, None] if t.dim() == 2 else (t[None] if t.dim() == 3 else t)
    x, y = to4(x), to4(y)
    if x.shape != y.shape:
        raise ValueError(f"Input tensors must have the same shape, got {x.shape} and {y.shape}")
    if x.is_cuda and y.is_cuda and window_size == 11 and x.dtype == y.dtype and x.dtype in (torch.float32, torch.float64):
        # fused device kernel (ddn_ssim): one pass over the two images
        import ctypes as C
        from ..plan import _ptr
        B, nc, H, W = x.shape
        xc, yc = x.contiguous(), y.contiguous()
        out = torch.empty((B * nc,), dtype=torch.float64, device=x.device)
        co_ = torch.arange(window_size, dtype=torch.float32) - window_size // 2      # loss.py:75-81, on the host
        g_ = torch.exp(-(co_ ** 2) / (2 * 1.5 ** 2))
        g_ = (g_ / g_.sum()).contiguous()
        g11 = (C.c_float * 11)(*g_.tolist())
        with torch.cuda.device(x.device):
            L.check(L.lib().ddn_ssim(L.F64 if x.dtype == torch.float64 else L.F32, _ptr(xc), _ptr(yc), H, W, B * nc, float(max_val), g11, _ptr(out),
                                     C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return out.view(B, nc).mean(dim=1).to(x.dtype)
    C1, C2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    co = torch.arange(window_size, dtype=torch.float32) - window_size // 2
    g = torch.exp(-(co ** 2) / (2 * 1.5 ** 2))
    g = g / g.sum()
    w = (g[:, None] * g[None, :])[None, None].to(device=x.device, dtype=x.dtype)
    nc, p = x.shape[1], window_size // 2
    mu_x = F.conv2d(x, w, padding=p, groups=nc)
    mu_y = F.conv2d(y, w, padding=p, groups=nc)
    sx = F.conv2d(x * x, w, padding=p, groups=nc) - mu_x ** 2
    sy = F.conv2d(y * y, w, padding=p, groups=nc) - mu_y ** 2
    sxy = F.conv2d(x * y, w, padding=p, groups=nc) - mu_x * mu_y
    m = ((2 * mu_x * mu_y + C1) * (2 * sxy + C2)) / ((mu_x ** 2 + mu_y ** 2 + C1) * (sx + sy + C2))
    return m.mean(dim=(2, 3)).mean(dim=1)
