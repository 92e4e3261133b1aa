"""CPU oracle for the deblur_denoise iteration hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under convex_optimization_amd/ may import this
module; it is used by tests/, by __graft_entry__.smoke() and by the cpu_baseline
leg of bench.py as the checker / the timed CPU baseline, never as the product.

It restates, op for op, the algorithm of rahul3/convex_optimization
(`deblur_denoise/`), using the same torch CPU primitives the reference uses
(complex fft2/ifft2, conv2d), so that on identical inputs it reproduces the
reference's iterates to floating-point roundoff.  Every function cites the
reference file:line it follows (paths relative to the reference root).

Parity pinning: tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which oracle/make_golden.py produced by running the
reference itself.

Two families:
  * `*_faithful` / plain names: same op sequence as the reference (complex c2c
    transforms, every intermediate `.real` where the reference takes it).
  * `fused_*`: the real-FFT restatement the HIP kernels implement (two real
    arrays packed per complex transform, Hermitian-symmetrised multipliers,
    D/D^T as row-local stencils or row-spectral multiplies).  Executable spec.
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- kernels
def gaussian_filter(size: Sequence[int], sigma: float) -> np.ndarray:
    """deblur_denoise/core/blur.py:52-62 -- float64 ndarray, normalised."""
    ax0 = np.linspace(-(size[0] // 2), size[0] // 2, size[0])
    ax1 = np.linspace(-(size[1] // 2), size[1] // 2, size[1])
    x, y = np.meshgrid(ax0, ax1)
    h = np.exp(-(x ** 2 + y ** 2) / (2 * sigma ** 2))
    return h / h.sum()


def create_motion_blur_kernel(size: int = 15, angle: float = 0) -> torch.Tensor:
    """deblur_denoise/core/blur.py:4-49 -- horizontal line, then every (i,j) of the
    grid is rotated with truncating int() and the target set to 1; normalised."""
    if size % 2 == 0:
        size += 1
    k = torch.zeros((size, size))
    k[size // 2, :] = 1
    if angle != 0:
        a = np.radians(angle)
        c = size // 2
        # the reference stores the rotation matrix as a default-dtype torch tensor
        rot = torch.tensor([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        for i in range(size):
            for j in range(size):
                x, y = i - c, j - c
                nx = int(c + x * rot[0, 0] + y * rot[0, 1])
                ny = int(c + x * rot[1, 0] + y * rot[1, 1])
                if 0 <= nx < size and 0 <= ny < size:
                    k[nx, ny] = 1
    return k / k.sum()


# ---------------------------------------------------------------------- eigen arrays
def eig_vals_for_periodic_conv_op(filter_kernel, num_rows: int, num_cols: int) -> torch.Tensor:
    """deblur_denoise/op_math/python_code/eig_vals_for_periodic_conv_op.py:5-61.
    float32 impulse -> circular pad -> conv2d (cross-correlation) -> crop -> fft2."""
    if isinstance(filter_kernel, np.ndarray):
        filter_kernel = torch.from_numpy(filter_kernel)
    a = torch.zeros((num_rows, num_cols))
    a[0, 0] = 1
    if filter_kernel.dim() == 2:
        filter_kernel = filter_kernel[None, None]
    ph, pw = filter_kernel.shape[2] // 2, filter_kernel.shape[3] // 2
    ap = F.pad(a[None, None], (pw, pw, ph, ph), mode="circular")
    ra = F.conv2d(ap.float(), filter_kernel.float(), padding=0).squeeze()
    return torch.fft.fft2(ra[:num_rows, :num_cols])


def apply_periodic_conv_2d(x: torch.Tensor, eig: torch.Tensor) -> torch.Tensor:
    """deblur_denoise/op_math/python_code/apply_periodic_conv_2d.py:28."""
    return torch.fft.ifft2(eig * torch.fft.fft2(x))


class DeblurDenoiseOperators:
    """deblur_denoise/op_math/python_code/multiplying_matrix.py:6-112."""

    def __init__(self, kernel, blurred_image, tprimaldr, s=None, device="cpu"):
        self.kernel = torch.from_numpy(kernel) if isinstance(kernel, np.ndarray) else kernel
        self.b = blurred_image
        self.t = tprimaldr
        self.s = s if s is not None else tprimaldr
        self.num_rows, self.num_cols = blurred_image.shape
        H, W = self.num_rows, self.num_cols
        self.eig_arry_K = eig_vals_for_periodic_conv_op(self.kernel, H, W)          # mm:51
        d = torch.tensor([[-1, 1]])                                                 # mm:55-56 (D1 == D2)
        self.eig_arry_D1 = eig_vals_for_periodic_conv_op(d, H, W)
        self.eig_arry_D2 = eig_vals_for_periodic_conv_op(d, H, W)
        self.eig_arry_KTrans = torch.conj(self.eig_arry_K)                           # mm:62-64
        self.eig_arry_D1Trans = torch.conj(self.eig_arry_D1)
        self.eig_arry_D2Trans = torch.conj(self.eig_arry_D2)
        self.eig_vals_mat = (torch.ones((H, W))                                      # mm:67-72
                             + self.t * self.t * self.eig_arry_KTrans * self.eig_arry_K
                             + self.t * self.t * self.eig_arry_D1Trans * self.eig_arry_D1
                             + self.t * self.s * self.eig_arry_D2Trans * self.eig_arry_D2)

    def apply_K(self, x): return apply_periodic_conv_2d(x, self.eig_arry_K)          # mm:74-76
    def apply_D1(self, x): return apply_periodic_conv_2d(x, self.eig_arry_D1)        # mm:78-80
    def apply_D2(self, x): return apply_periodic_conv_2d(x, self.eig_arry_D2)        # mm:82-84
    def apply_KTrans(self, x): return apply_periodic_conv_2d(x, self.eig_arry_KTrans)  # mm:86-88
    def apply_D1Trans(self, x): return apply_periodic_conv_2d(x, self.eig_arry_D1Trans)
    def apply_D2Trans(self, x): return apply_periodic_conv_2d(x, self.eig_arry_D2Trans)
    def apply_D(self, x): return torch.stack([self.apply_D1(x), self.apply_D2(x)], dim=2)  # mm:98-100
    def apply_DTrans(self, y): return self.apply_D1Trans(y[:, :, 0]) + self.apply_D2Trans(y[:, :, 1])  # mm:102-104
    def apply_Mat(self, x): return x + self.apply_KTrans(self.apply_K(x)) + self.apply_DTrans(self.apply_D(x))  # mm:106-108
    def invert_matrix(self, x): return torch.fft.ifft2(torch.fft.fft2(x) / self.eig_vals_mat)  # mm:110-112


# ------------------------------------------------------------------------------ prox
def prox_l1(x, lam):
    """deblur_denoise/core/proximal_operators.py:4-8."""
    return torch.sign(x) * torch.clamp(torch.abs(x) - lam, min=0)


def prox_l2(x, lam):
    """proximal_operators.py:10-14 (whole-tensor norm; unused by the solvers)."""
    n = torch.norm(x)
    return (1 - lam / n) * x if n > lam else torch.zeros_like(x)


def prox_l2_squared(x, lam):
    """proximal_operators.py:16-20."""
    return x / (2 * lam + 1)


def prox_iso(x, lam):
    """proximal_operators.py:22-50: per-pixel 2-vector shrink, zero where norm <= lam."""
    x1, x2 = x[0], x[1]
    n = torch.sqrt(x1 ** 2 + x2 ** 2)
    o1, o2 = torch.zeros_like(x1), torch.zeros_like(x2)
    c = n > lam
    o1[c] = (1 - lam / n[c]) * x1[c]
    o2[c] = (1 - lam / n[c]) * x2[c]
    return torch.stack((o1, o2))


def prox_box(x, lam=None):
    """proximal_operators.py:52-57 (lambda ignored)."""
    return torch.minimum(torch.ones_like(x), torch.maximum(torch.zeros_like(x), x))


# ------------------------------------------------------------------------------ loss
def l1_loss(x, y): return torch.norm(x - y, p=1)             # core/loss.py:5-10
def l2_loss(x, y): return torch.norm(x - y, p=2)             # core/loss.py:12-17
def mse(x, y): return torch.mean((x - y) ** 2)               # core/loss.py:112-117


def psnr(x, y, max_val: float = 1.0):
    """core/loss.py:19-36."""
    m = torch.mean((x - y) ** 2)
    if m == 0:
        return torch.tensor(float("inf"))
    return 10 * torch.log10(max_val ** 2 / m)


def ssim(x, y, window_size: int = 11, max_val: float = 1.0):
    """core/loss.py:39-110: 11x11 Gaussian (sigma 1.5) window, zero padding."""
    def to4(t):
        return t[None, None] if t.dim() == 2 else (t[None] if t.dim() == 3 else t)
    x, y = to4(x), to4(y)
    C1, C2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    co = torch.arange(window_size, dtype=torch.float32) - window_size // 2
    g = torch.exp(-(co ** 2) / (2 * 1.5 ** 2))
    g = g / g.sum()
    w = (g[:, None] * g[None, :])[None, None].to(x.dtype)
    nc, p = x.shape[1], window_size // 2
    mu_x = F.conv2d(x, w, padding=p, groups=nc)
    mu_y = F.conv2d(y, w, padding=p, groups=nc)
    sx = F.conv2d(x * x, w, padding=p, groups=nc) - mu_x ** 2
    sy = F.conv2d(y * y, w, padding=p, groups=nc) - mu_y ** 2
    sxy = F.conv2d(x * y, w, padding=p, groups=nc) - mu_x * mu_y
    m = ((2 * mu_x * mu_y + C1) * (2 * sxy + C2)) / ((mu_x ** 2 + mu_y ** 2 + C1) * (sx + sy + C2))
    return m.mean(dim=(2, 3)).mean(dim=1)


# ---------------------------------------------------------------------- data generation
def circular_convolve2d(image, kernel):
    """deblur_denoise/core/convolution.py:22-52: FFT circular convolution, float32
    padded kernel rolled by (-kh//2, -kw//2)."""
    if isinstance(image, np.ndarray):
        image = torch.from_numpy(image)
    if isinstance(kernel, np.ndarray):
        kernel = torch.from_numpy(kernel)
    _, H, W = image.shape
    kh, kw = kernel.shape[0], kernel.shape[1]
    pk = torch.zeros((H, W), dtype=torch.float32)
    pk[:kh, :kw] = kernel
    pk = torch.roll(pk, shifts=(-kh // 2, -kw // 2), dims=(0, 1))
    return torch.fft.ifft2(torch.fft.fft2(image) * torch.fft.fft2(pk)).real


def add_gaussian_noise(image, mean=0.0, std=0.1):
    """deblur_denoise/core/noise.py:4-11."""
    return torch.clamp(image + torch.randn_like(image) * std + mean, 0, 1)


# --------------------------------------------------------------------- faithful solvers
def _obj_prox(objective, v, lam):
    if objective == "l1":
        return prox_l1(v, lam)
    if objective == "l2":
        return prox_l2_squared(v, lam)
    raise ValueError("Invalid objective function. Choose 'l1' or 'l2'.")


def chambolle_pock(b, kernel, objective_function="l1", t=0.4, s=0.7, gamma=0.01, max_iter=500,
                   loss_function: Optional[Callable] = None, init_state=None, losses: Optional[list] = None):
    """deblur_denoise/chambolle_pock/algorithm.py:21-143 (loop cp:58-111)."""
    ops = DeblurDenoiseOperators(kernel, b.squeeze(), 1, 1)                          # cp:46
    H, W = b.squeeze().shape
    if init_state is None:                                                            # cp:50-52 draw order
        x, z, y = torch.rand(H, W), torch.rand(H, W), torch.rand(3, H, W)
    else:
        x, z, y = [v.clone() for v in init_state]
    x_next = x.clone()
    for k in range(1, max_iter):
        Kz = ops.apply_K(z.squeeze())                                                 # cp:60-63
        Dz = ops.apply_D(z.squeeze())
        Az = torch.cat((Kz.unsqueeze(0), Dz.permute(2, 0, 1)), dim=0)
        g = torch.real(y + s * Az)                                                    # cp:66
        c1 = g[0] - s * b - s * _obj_prox(objective_function, g[0] / s - b, 1 / s)    # cp:69-74
        c2 = g[1:] - s * prox_iso(g[1:] / s, 1 / s * gamma)                           # cp:75
        y_next = torch.cat((c1, c2), dim=0)
        ATy = ops.apply_KTrans(y_next[0]) + ops.apply_DTrans(y_next[1:].permute(1, 2, 0))  # cp:79
        x_next = prox_box(torch.real(x - t * ATy.unsqueeze(0)), 1)                    # cp:83-86
        z_next = 2 * x_next - x                                                       # cp:87
        x, y, z = x_next, y_next, z_next
        if k > 1 and loss_function is not None:                                       # cp:94-99
            v = loss_function(x_next, b)
            if losses is not None:
                losses.append(float(v))
    return x_next


def admm_solver(b, objective_function="l1", t=18, rho=0.001, gamma=0.5, kernel=None, niters=500,
                tol=1e-4, loss_function: Optional[Callable] = None, init_state=None, losses: Optional[list] = None):
    """deblur_denoise/admm/algorithm.py:20-201 (loop admm:69-157, final admm:160-173)."""
    H, W = b.shape
    if init_state is None:                                                            # admm:49-54
        _x = torch.clamp(torch.randn((H, W), dtype=torch.float32), 0, 1)
        u = torch.clamp(torch.randn((H, W), dtype=torch.float32), 0, 1)
        w = torch.clamp(torch.randn((H, W), dtype=torch.float32), 0, 1)
        y = torch.clamp(torch.randn((3, H, W), dtype=torch.float32), 0, 1)
        z = torch.clamp(torch.randn((3, H, W), dtype=torch.float32), 0, 1)
    else:
        u, w, y, z = [v.clone() for v in init_state]
    ops = DeblurDenoiseOperators(kernel, b, 1, 1)                                    # admm:56-59

    def AT(v):                                                                        # admm:72-76
        return ops.apply_KTrans(v[0]) + ops.apply_DTrans(torch.stack([v[1], v[2]], dim=0).permute(1, 2, 0))

    for i in range(1, niters):
        x = ops.invert_matrix(u + AT(y) - (1 / t) * (w + AT(z)))                      # admm:85 (complex)
        u_next = prox_box(torch.real(rho * x + (1 - rho) * u + (1 / t) * w), 1 / t)   # admm:87-88
        Ax = torch.cat([ops.apply_K(x).unsqueeze(-1), ops.apply_D(x)], dim=-1).permute(2, 0, 1)  # admm:91-96
        pp = torch.real(rho * Ax + (1 - rho) * y + (1 / t) * z)                       # admm:98
        pK = b + _obj_prox(objective_function, pp[0] - b, 1 / t)                      # admm:99-104
        pD = prox_iso(pp[1:], (1 / t) * gamma)                                        # admm:105
        y_next = torch.cat([pK.unsqueeze(0), pD], dim=0)
        w = w + t * (x - u_next)                                                      # admm:108
        z = z + t * (Ax - y_next)                                                     # admm:109
        u, y = u_next, y_next
        if i > 1:                                                                     # admm:118-154
            sol = torch.real(ops.invert_matrix(u + AT(y) - (1 / t) * (w + AT(z))))
            if loss_function is not None:
                d = float(loss_function(sol, b))
                if losses is not None:
                    losses.append(d)
                if d < tol:
                    break
    return torch.real(ops.invert_matrix(u + AT(y) - (1 / t) * (w + AT(z))))          # admm:160-173


def primal_dr_splitting(problem, kernel, b, loss_function: Optional[Callable] = None, niters=500, gamma=0.049,
                        t=2.0, rho=0.1, tol=1e-6, losses: Optional[list] = None, init_state=None):
    """deblur_denoise/primal_dr/algorithm.py:21-187 (loop pdr:94-161)."""
    if init_state is None:
        z1 = b                                                                        # pdr:88-89
        z2 = torch.stack((b, b, b))
    else:
        z1, z2 = [v.clone() for v in init_state]
    ops = DeblurDenoiseOperators(kernel, b, t)                                       # pdr:91
    for j in range(niters):
        x = prox_box(z1, t)                                                           # pdr:96
        iso = prox_iso(z2[[1, 2]], t * gamma)                                         # pdr:103/108
        if problem == "l1":
            nrm = b + prox_l1(z2[0] - b, t)                                           # pdr:105
        else:
            nrm = b + prox_l2_squared(z2[0] - b, t)                                   # pdr:110
        y = torch.stack((nrm, iso[0], iso[1]))
        arg = 2 * y - z2                                                              # pdr:115-117
        ATa = (ops.apply_KTrans(arg[0]) + ops.apply_D1Trans(arg[1]) + ops.apply_D2Trans(arg[2])).real
        u = ops.invert_matrix(2 * x - z1 + ATa).real                                  # pdr:119-120
        v = torch.stack((ops.apply_K(u), ops.apply_D1(u), ops.apply_D2(u))).real      # pdr:123-124
        z1prev = z1
        z1 = z1 + rho * (u - x)                                                       # pdr:128-129
        z2 = z2 + rho * (v - y)
        if j > 1 and loss_function is not None:                                       # pdr:143-161
            lv = float(loss_function(z1, z1prev))
            if losses is not None:
                losses.append(lv)
            if lv < tol:
                break
    return prox_box(z1, t)                                                            # pdr:187


def primal_dual_dr_splitting(b, kernel, niters=1000, t=3, rho=0.4, gamma=0.03, loss_function: Optional[Callable] = None,
                             tol=1e-6, init_state=None, losses: Optional[list] = None):
    """deblur_denoise/primal_dual_dr/algorithm.py:20-139 (loop pddr:65-121)."""
    if b.dim() == 2:
        b = b.unsqueeze(0)                                                            # pddr:52-53
    H, W = b.squeeze().shape
    if init_state is None:                                                            # pddr:57-58
        p, q = torch.rand(H, W), torch.rand(3, H, W)
    else:
        p, q = [v.clone() for v in init_state]
    ops = DeblurDenoiseOperators(kernel, b.squeeze(), t, t)                          # pddr:61
    for k in range(1, niters):
        x = prox_box(p, 1)                                                            # pddr:66
        z1 = q[0] - t * b - t * prox_l1((1 / t) * q[0] - b, 1 / t)                    # pddr:69
        z2 = q[1:] - t * prox_iso((1 / t) * q[1:], (1 / t) * gamma)                   # pddr:70
        z = torch.cat((z1, z2), dim=0)
        i1 = 2 * x - p                                                                # pddr:76-80
        i2 = 2 * z - q
        o = i2.clone()
        i1 = i1 - t * ops.apply_KTrans(i2[0]) - t * ops.apply_DTrans(torch.stack((i2[1], i2[2]), dim=0).permute(1, 2, 0))
        i1 = ops.invert_matrix(i1)                                                    # pddr:86
        i2 = torch.cat(((t * ops.apply_K(i1.squeeze())).unsqueeze(0),                 # pddr:90-93
                        (t * ops.apply_D(i1.squeeze())).permute(2, 0, 1)), dim=0)
        p = torch.real(p + rho * (i1 - x))                                            # pddr:100-101
        q = torch.real(q + rho * (i2 + o - z))
        if k > 1 and loss_function is not None:                                       # pddr:103-121
            lv = float(loss_function(prox_box(p, 1), b))
            if losses is not None:
                losses.append(lv)
            if lv < tol:
                break
    return prox_box(p, 1)                                                             # pddr:123,139


# --------------------------------------------------------- fused real-FFT restatement
def apply_x_trans(w, x, kernel_size):
    """deblur_denoise/op_math/python_code/apply_x_trans.py:4-46: entry (i, j) is the sum of w times x
    correlated with a one-hot kernel at (i, j) under zero padding kernel_size // 2 (axt:29-44)."""
    import torch.nn.functional as F
    k = kernel_size
    xtw = torch.zeros((k, k))                                                          # axt:26-27
    for i in range(k):
        for j in range(k):
            shift = torch.zeros((k, k))
            shift[i, j] = 1                                                            # axt:31-32
            conv = F.conv2d(x[None, None], shift[None, None], padding=(k // 2, k // 2)).squeeze()   # axt:36-41
            xtw[i, j] = torch.sum(w * conv)                                            # axt:44
    return xtw


def herm_half(e: np.ndarray) -> np.ndarray:
    """Hermitian-symmetrise a full H x W multiplier and keep columns 0..W/2:
    e_h[k] = (e[k] + conj(e[-k]))/2 -- what `.real` after ifft2 does implicitly."""
    e = np.asarray(e).astype(np.complex128)   # symmetrise in double: complex64 arithmetic would add 1e-8
    H, W = e.shape
    em = np.conj(np.roll(np.flip(e, (0, 1)), (1, 1), (0, 1)))
    return (0.5 * (e + em))[:, : W // 2 + 1]


class FusedOps:
    """Half-plane multipliers in numpy with dtype `rt` (np.float32 / np.float64)."""

    def __init__(self, eigK, eigD, M, rt):
        ct = np.complex64 if rt == np.float32 else np.complex128
        self.rt, self.ct = rt, ct
        self.eK = herm_half(np.asarray(eigK)).astype(ct)
        self.eD = herm_half(np.asarray(eigD)).astype(ct)
        # M = 1 + t^2 conj(eK) eK + ... is formed in complex64 arithmetic (mm:67-72); its imaginary part
        # is exactly 0 only when t^2 is a power of two, so the resolvent multiplier stays complex.
        self.G = herm_half(1.0 / np.asarray(M).astype(np.complex128)).astype(ct)
        self.W = np.asarray(eigK).shape[1]

    def F(self, x): return np.fft.rfft2(x).astype(self.ct)
    def Fi(self, X): return np.fft.irfft2(X, s=(X.shape[0], self.W)).astype(self.rt)

    @staticmethod
    def D(x): return x - np.roll(x, 1, axis=1)        # (Dx)[i,j] = x[i,j] - x[i,j-1]
    @staticmethod
    def DT(y): return y - np.roll(y, -1, axis=1)      # (D^T y)[i,j] = y[i,j] - y[i,j+1]


def _soft(v, lam): return np.sign(v) * np.maximum(np.abs(v) - lam, 0)


def _iso(a, c, lam):
    n = np.sqrt(a * a + c * c)
    with np.errstate(divide="ignore", invalid="ignore"):
        f = np.where(n > lam, 1 - lam / n, 0).astype(a.dtype)
    return f * a, f * c


def fused_pdr(fo: FusedOps, b, problem, niters, gamma, t, rho, spectral_D=False):
    """primal_dr loop as 2 real->complex + 2 complex->real transforms per iteration."""
    rt = fo.rt
    b = b.astype(rt); z1 = b.copy(); z2 = [b.copy(), b.copy(), b.copy()]
    t, rho, gamma = rt(t), rt(rho), rt(gamma)
    for _ in range(niters):
        x = np.clip(z1, 0, 1)
        y0 = b + (_soft(z2[0] - b, t) if problem == "l1" else (z2[0] - b) / (2 * t + 1))
        y1, y2 = _iso(z2[1], z2[2], t * gamma)
        a0, a1, a2 = 2 * y0 - z2[0], 2 * y1 - z2[1], 2 * y2 - z2[2]
        if spectral_D:
            U = (fo.F(2 * x - z1) + np.conj(fo.eK) * fo.F(a0) + np.conj(fo.eD) * fo.F(a1 + a2)) * fo.G
            u, v0, dv = fo.Fi(U), fo.Fi(fo.eK * U), fo.Fi(fo.eD * U)
        else:
            U = (fo.F(2 * x - z1 + fo.DT(a1 + a2)) + np.conj(fo.eK) * fo.F(a0)) * fo.G
            u, v0 = fo.Fi(U), fo.Fi(fo.eK * U)
            dv = fo.D(u)
        z1 = z1 + rho * (u - x)
        z2 = [z2[0] + rho * (v0 - y0), z2[1] + rho * (dv - y1), z2[2] + rho * (dv - y2)]
    return np.clip(z1, 0, 1)


def fused_pddr(fo: FusedOps, b, p, q, niters, t, rho, gamma, spectral_D=False):
    rt = fo.rt
    b = b.astype(rt); p = p.astype(rt); q = [q[i].astype(rt) for i in range(3)]
    t, rho, gamma = rt(t), rt(rho), rt(gamma)
    it = rt(1) / t
    for _ in range(1, niters):
        x = np.clip(p, 0, 1)
        z0 = q[0] - t * b - t * _soft(it * q[0] - b, it)
        i1, i2 = _iso(it * q[1], it * q[2], it * gamma)
        z1, z2 = q[1] - t * i1, q[2] - t * i2
        j1 = 2 * x - p
        o0, o1, o2 = 2 * z0 - q[0], 2 * z1 - q[1], 2 * z2 - q[2]
        if spectral_D:
            Wh = (fo.F(j1) - t * np.conj(fo.eK) * fo.F(o0) - t * np.conj(fo.eD) * fo.F(o1 + o2)) * fo.G
            w, kw, dw = fo.Fi(Wh), fo.Fi(fo.eK * Wh), fo.Fi(fo.eD * Wh)
        else:
            Wh = (fo.F(j1 - t * fo.DT(o1 + o2)) + np.conj(fo.eK) * fo.F(-t * o0)) * fo.G
            w, kw = fo.Fi(Wh), fo.Fi(fo.eK * Wh)
            dw = fo.D(w)
        p = p + rho * (w - x)
        q = [q[0] + rho * (t * kw + o0 - z0), q[1] + rho * (t * dw + o1 - z1), q[2] + rho * (t * dw + o2 - z2)]
    return np.clip(p, 0, 1)


def fused_admm(fo: FusedOps, b, u, w, y, z, objective, niters, t, rho, gamma, spectral_D=False):
    rt = fo.rt
    b = b.astype(rt); u = u.astype(rt); w = w.astype(rt)
    y = [y[i].astype(rt) for i in range(3)]; z = [z[i].astype(rt) for i in range(3)]
    t, rho, gamma = rt(t), rt(rho), rt(gamma)
    it = rt(1) / t

    def xhat():
        if spectral_D:
            return (fo.F(u - it * w) + np.conj(fo.eK) * fo.F(y[0] - it * z[0])
                    + np.conj(fo.eD) * fo.F(y[1] + y[2] - it * (z[1] + z[2]))) * fo.G
        return (fo.F(u - it * w + fo.DT(y[1] + y[2] - it * (z[1] + z[2]))) + np.conj(fo.eK) * fo.F(y[0] - it * z[0])) * fo.G

    for _ in range(1, niters):
        X = xhat()
        x, kx = fo.Fi(X), fo.Fi(fo.eK * X)
        dx = fo.Fi(fo.eD * X) if spectral_D else fo.D(x)
        un = np.clip(rho * x + (1 - rho) * u + it * w, 0, 1)
        pp0 = rho * kx + (1 - rho) * y[0] + it * z[0]
        pp1 = rho * dx + (1 - rho) * y[1] + it * z[1]
        pp2 = rho * dx + (1 - rho) * y[2] + it * z[2]
        y0n = b + (_soft(pp0 - b, it) if objective == "l1" else (pp0 - b) / (2 * it + 1))
        y1n, y2n = _iso(pp1, pp2, it * gamma)
        w = w + t * (x - un)
        z = [z[0] + t * (kx - y0n), z[1] + t * (dx - y1n), z[2] + t * (dx - y2n)]
        u, y = un, [y0n, y1n, y2n]
    return fo.Fi(xhat())


def fused_cp(fo: FusedOps, b, x, z, y, objective, max_iter, t, s, gamma, spectral_D=False):
    rt = fo.rt
    b = b.astype(rt); x = x.astype(rt); z = z.astype(rt); y = [y[i].astype(rt) for i in range(3)]
    t, s, gamma = rt(t), rt(s), rt(gamma)
    for _ in range(1, max_iter):
        Z = fo.F(z)
        Kz = fo.Fi(fo.eK * Z)
        Dz = fo.Fi(fo.eD * Z) if spectral_D else fo.D(z)
        g0, g1, g2 = y[0] + s * Kz, y[1] + s * Dz, y[2] + s * Dz
        pr = _soft(g0 / s - b, 1 / s) if objective == "l1" else (g0 / s - b) / (2 / s + 1)
        y0 = g0 - s * b - s * pr
        i1, i2 = _iso(g1 / s, g2 / s, gamma / s)
        y = [y0, g1 - s * i1, g2 - s * i2]
        if spectral_D:
            aty = fo.Fi(np.conj(fo.eK) * fo.F(y[0]) + np.conj(fo.eD) * fo.F(y[1] + y[2]))
        else:
            aty = fo.Fi(np.conj(fo.eK) * fo.F(y[0])) + fo.DT(y[1] + y[2])
        xn = np.clip(x - t * aty, 0, 1)
        z = 2 * xn - x
        x = xn
    return x


def rel_l2(a, b) -> float:
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
