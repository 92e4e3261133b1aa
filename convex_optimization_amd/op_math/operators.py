"""Spectral operator algebra with the reference's interface
(deblur_denoise/op_math/python_code/{multiplying_matrix,eig_vals_for_periodic_conv_op,apply_periodic_conv_2d}.py).

Everything is evaluated by libddn on the GPU: eigenvalue arrays come from the library's own 2-D
transform of the impulse response, operator applications are real-to-complex / complex-to-real
round trips with Hermitian-symmetrised multipliers.  The operator class returns complex tensors with a
zero imaginary part where the reference returns complex tensors whose imaginary parts are roundoff;
apply_periodic_conv_2d itself returns the full complex result for any multiplier array.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .. import _lib as L
from ..plan import Plan, clear_plan_cache, create_plan, acquire_plan, compute_device, plan_key, real_dtype, release_plan, stage


def _ctype(dt):
    return torch.complex128 if dt == torch.float64 else torch.complex64


def eig_vals_for_periodic_conv_op(filter_kernel, num_rows: int, num_cols: int, **kwargs) -> torch.Tensor:
    """Eigenvalues (2-D DFT of the impulse response) of the periodic correlation with
    `filter_kernel`, centred at k//2; the kernel is rounded to float32 first
    (eig_vals_for_periodic_conv_op.py:5-61).  Returns a num_rows x num_cols complex tensor."""
    dt = real_dtype(kwargs.get("dtype", torch.float32))
    dev = compute_device(kwargs.get("device"))
    out_dev = kwargs.get("device", "cpu")
    plan = create_plan(num_rows, num_cols, 1, dt, dev)
    try:
        plan.set_kernel(filter_kernel)
        e = plan.get_eigK()
        torch.cuda.synchronize(dev)
    finally:
        plan.close()
    return e.to(out_dev)


def apply_periodic_conv_2d(x: torch.Tensor, eig_val_arr: torch.Tensor, **kwargs) -> torch.Tensor:
    """ifft2(eig_val_arr * fft2(x)) as a complex m x n tensor, for a real or complex image x and ANY complex
    multiplier array (apply_periodic_conv_2d.py:3-28).

    The library transforms real images with Hermitian multipliers.  A general multiplier e splits into two Hermitian
    ones, e = eh + i ea with eh[k] = (e[k] + conj(e[-k])) / 2 and ea = the same part of -i e, so for a real image
    ifft2(e fft2(x)) = A_eh x + i A_ea x, two real applications (the library symmetrises what it is handed, so the
    second one is simply the multiplier -i e).  A complex image x = xr + i xi takes four:
    (A_eh xr - A_ea xi) + i (A_ea xr + A_eh xi).  For the eigenvalue array of a real kernel ea is roundoff, as is the
    imaginary part the reference returns."""
    src = x.device
    dt = real_dtype(x.real.dtype if x.is_complex() else x.dtype)
    dev = compute_device(kwargs.get("device", src))
    H, W = x.shape
    e = torch.as_tensor(eig_val_arr)
    if not e.is_complex():
        e = torch.complex(e, torch.zeros_like(e))
    e_rot = e * torch.tensor(-1j, dtype=e.dtype, device=e.device)
    plan = create_plan(H, W, 1, dt, dev)
    try:
        def both(v):
            vd = stage(v, dev, dt).reshape(1, H, W)
            return plan.apply_multiplier(e, vd)[0], plan.apply_multiplier(e_rot, vd)[0]
        if x.is_complex():
            hr, ar = both(x.real)
            hi, ai = both(x.imag)
            y = torch.complex(hr - ai, ar + hi)
        else:
            h, a = both(x)
            y = torch.complex(h, a)
    finally:
        plan.close()
    return y.to(src)


class DeblurDenoiseOperators:
    """Operators required for deblurring and denoising (multiplying_matrix.py:6-112).

    kernel: k x k tensor / ndarray; blurred_image: (H, W) tensor (only its shape and dtype are used);
    tprimaldr, s: the step sizes entering eig_vals_mat = 1 + t^2 |eK|^2 + t^2 |eD1|^2 + t s |eD2|^2.
    Extensions through keyword arguments: batch (number of images a solver will drive through this
    plan), dtype, eig_arrays = {"K":..., "D":..., "M":...} to consume externally supplied eigenvalue
    arrays bit for bit (needed for fp64 parity), tune = {"cols":..,"split":..} column tiling knobs.
    """

    def __init__(self, kernel, blurred_image, tprimaldr, s=None, device="cpu", **kwargs):
        self.kernel = kernel
        self.b = blurred_image
        self.t = tprimaldr
        self.s = s if s is not None else tprimaldr
        self.num_rows, self.num_cols = blurred_image.shape[-2], blurred_image.shape[-1]
        self.device = device
        self._dt = real_dtype(kwargs.get("dtype", blurred_image.dtype if isinstance(blurred_image, torch.Tensor) else torch.float32))
        self._dev = compute_device(device if device != "cpu" else (blurred_image.device if isinstance(blurred_image, torch.Tensor) else None))
        self._batch = int(kwargs.get("batch", 1))
        self._plan1: Optional[Plan] = None
        self._eig_arrays = kwargs.get("eig_arrays")
        self._cache = {}
        # plans built from a blur kernel (not from injected eigen arrays) are reused across calls: see plan.py
        key = None
        if self._eig_arrays is None:
            cK = float(self.t) * float(self.t)
            key = plan_key(self.num_rows, self.num_cols, self._batch, self._dt, self._dev, kernel, cK, cK + float(self.t) * float(self.s), kwargs.get("tune"))
        self.plan = acquire_plan(key) if key is not None else None
        if self.plan is None:
            def build():
                self.plan = create_plan(self.num_rows, self.num_cols, self._batch, self._dt, self._dev, tune=kwargs.get("tune"))
                self.plan.cache_key = key
                self.setup_operators()
            try:
                build()
            except L.DdnError as e:
                # out of device memory while plans released earlier are still parked: give their memory back, once
                if "memory" not in str(e).lower():
                    raise
                if self.plan is not None:
                    self.plan.close()
                    self.plan = None
                clear_plan_cache()
                torch.cuda.empty_cache()
                build()

    # mm:48-72
    def setup_operators(self):
        cK = float(self.t) * float(self.t)
        cD = float(self.t) * float(self.t) + float(self.t) * float(self.s)
        ea = self._eig_arrays
        for pl in self._plans():
            if ea is not None:
                pl.set_eig(ea["K"], ea.get("D"))
                pl.set_resolvent(cK, cD, ea["M"])
            else:
                pl.set_kernel(self.kernel)
                pl.set_resolvent(cK, cD)

    def _plans(self):
        return [self.plan]

    def _single(self) -> Plan:
        """Plan for one image, used by the apply_* methods when the solver plan is batched."""
        if self._batch == 1:
            return self.plan
        if self._plan1 is None:
            self._plan1 = create_plan(self.num_rows, self.num_cols, 1, self._dt, self._dev)
            ea = self._eig_arrays
            cK = float(self.t) ** 2
            cD = cK + float(self.t) * float(self.s)
            if ea is not None:
                self._plan1.set_eig(ea["K"], ea.get("D"))
                self._plan1.set_resolvent(cK, cD, ea["M"])
            else:
                self._plan1.set_kernel(self.kernel)
                self._plan1.set_resolvent(cK, cD)
        return self._plan1

    # ---- eigenvalue arrays in the reference's layout (mm:51-72), materialised on demand
    @property
    def eig_arry_K(self):
        if "K" not in self._cache:
            self._cache["K"] = self._single().get_eigK().to(self.device)
        return self._cache["K"]

    @property
    def eig_arry_D1(self):
        if "D" not in self._cache:
            if self._eig_arrays is not None and self._eig_arrays.get("D") is not None:
                d = torch.as_tensor(self._eig_arrays["D"])
            else:
                n = torch.arange(self.num_cols, dtype=torch.float64)
                row = 1 - torch.exp(-2j * np.pi * n / self.num_cols)
                d = row.to(_ctype(self._dt))[None, :].expand(self.num_rows, -1).contiguous()
            self._cache["D"] = d.to(self.device)
        return self._cache["D"]

    eig_arry_D2 = eig_arry_D1

    @property
    def eig_arry_KTrans(self): return torch.conj(self.eig_arry_K)
    @property
    def eig_arry_D1Trans(self): return torch.conj(self.eig_arry_D1)
    @property
    def eig_arry_D2Trans(self): return torch.conj(self.eig_arry_D2)

    @property
    def eig_vals_mat(self):
        if self._eig_arrays is not None:
            return torch.as_tensor(self._eig_arrays["M"]).to(self.device)
        K, D = self.eig_arry_K, self.eig_arry_D1
        return (torch.ones((self.num_rows, self.num_cols), device=K.device) + self.t * self.t * torch.conj(K) * K
                + self.t * self.t * torch.conj(D) * D + self.t * self.s * torch.conj(D) * D)

    # ---- applications (mm:74-112)
    def _apply(self, op, x):
        src = x.device
        xr = x.real if x.is_complex() else x
        xd = stage(xr, self._dev, self._dt).reshape(1, self.num_rows, self.num_cols)
        y = self._single().apply(op, xd)[0]
        if x.is_complex():
            yi = self._single().apply(op, stage(x.imag, self._dev, self._dt).reshape(1, self.num_rows, self.num_cols))[0]
            return torch.complex(y, yi).to(src)
        return torch.complex(y, torch.zeros_like(y)).to(src)

    def apply_K(self, x): return self._apply(L.OP_K, x)
    def apply_D1(self, x): return self._apply(L.OP_D, x)
    def apply_D2(self, x): return self._apply(L.OP_D, x)
    def apply_KTrans(self, x): return self._apply(L.OP_KT, x)
    def apply_D1Trans(self, x): return self._apply(L.OP_DT, x)
    def apply_D2Trans(self, x): return self._apply(L.OP_DT, x)

    def apply_D(self, x):
        return torch.stack([self.apply_D1(x), self.apply_D2(x)], dim=2)

    def apply_DTrans(self, y):
        return self.apply_D1Trans(y[:, :, 0]) + self.apply_D2Trans(y[:, :, 1])

    def apply_Mat(self, x): return self._apply(L.OP_MAT, x)
    def invert_matrix(self, x): return self._apply(L.OP_RESOLVENT, x)

    def close(self):
        if self.plan is not None:
            release_plan(self.plan)
            self.plan = None
        if self._plan1 is not None:
            self._plan1.close()
            self._plan1 = None
