"""Thin owner of a libddn plan: device selection, tensor staging, pointer plumbing."""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict
from typing import Optional

import numpy as np
import torch

from . import _lib as L


def compute_device(device=None) -> torch.device:
    """The HIP device work runs on.  `device` may be a cuda device; 'cpu'/None select cuda:current."""
    if not torch.cuda.is_available():
        raise L.DdnError("no MI355X/HIP device visible: this package has no CPU path")
    if device is not None:
        d = torch.device(device)
        if d.type == "cuda":
            return d if d.index is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cuda", torch.cuda.current_device())


def real_dtype(dtype) -> torch.dtype:
    if dtype in (torch.float64, torch.complex128, "float64", "f64"):
        return torch.float64
    return torch.float32


def stage(x, dev: torch.device, dtype: torch.dtype) -> torch.Tensor:
    """Contiguous copy of x on the compute device (never aliases caller memory: callers keep ownership)."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.detach().to(device=dev, dtype=dtype, copy=True).contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def taps_array(kernel) -> np.ndarray:
    if isinstance(kernel, torch.Tensor):
        k = kernel.detach().cpu().numpy()
    else:
        k = np.asarray(kernel)
    if k.ndim != 2:
        k = k.reshape(k.shape[-2], k.shape[-1])
    return np.ascontiguousarray(k, dtype=np.float64)


class Plan:
    def __init__(self, H: int, W: int, batch: int, dtype: torch.dtype, dev: torch.device, tune=None, slab: bool = False):
        """slab: a plan for the row-slab solvers that holds the multipliers of a tile range only (restrict_tiles)."""
        self.H, self.W, self.batch, self.dtype, self.dev = int(H), int(W), int(batch), dtype, dev
        self._h = C.c_void_p(0)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            dt = L.F64 if dtype == torch.float64 else L.F32
            if slab:
                if self.batch != 1 or tune:
                    raise ValueError("a slab plan holds one image and takes no tuning")
                L.check(L.lib().ddn_plan_create_slab(C.byref(self._h), self.H, self.W, dt, C.c_void_p(stream)))
            else:
                L.check(L.lib().ddn_plan_create(C.byref(self._h), self.H, self.W, self.batch, dt, C.c_void_p(stream)))
            if tune:
                L.check(L.lib().ddn_plan_tune(self._h, int(tune.get("cols", 0)), int(tune.get("split", 0)),
                                              int(tune.get("col_threads", tune.get("row_threads", 0)))))
        self._keep = []
        self.cache_key = None

    def close(self):
        if self._h and self._h.value:
            L.lib().ddn_plan_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def restrict_tiles(self, tile_lo: int, n_tiles: int):
        """Slab plans: allocate the multipliers of spectral tiles [tile_lo, tile_lo + n_tiles) (before set_kernel)."""
        with torch.cuda.device(self.dev):
            L.check(L.lib().ddn_slab_restrict(self._h, int(tile_lo), int(n_tiles)))

    def sync_stream(self):
        L.check(L.lib().ddn_plan_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))

    # ---- operator set-up
    def set_kernel(self, kernel):
        k = taps_array(kernel)
        with torch.cuda.device(self.dev):
            L.check(L.lib().ddn_plan_set_kernel(self._h, k.ctypes.data_as(C.POINTER(C.c_double)), k.shape[0], k.shape[1]))

    def _cplx(self, a):
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(a)
        if not a.is_complex():
            a = a.to(torch.complex128 if a.dtype == torch.float64 else torch.complex64)
        a = a.detach().to(self.dev).contiguous()
        return a, (L.C128 if a.dtype == torch.complex128 else L.C64)

    def set_eig(self, eigK, eigD=None):
        K, ck = self._cplx(eigK)
        D = None
        if eigD is not None:
            D, cd = self._cplx(eigD)
            if cd != ck:
                D = D.to(K.dtype)
        with torch.cuda.device(self.dev):
            L.check(L.lib().ddn_plan_set_eig(self._h, _ptr(K), _ptr(D), ck))
            torch.cuda.synchronize(self.dev)

    def set_resolvent(self, cK: float, cD: float, M=None):
        with torch.cuda.device(self.dev):
            if M is not None:
                Mt, cm = self._cplx(M)
                L.check(L.lib().ddn_plan_set_resolvent(self._h, float(cK), float(cD), _ptr(Mt), cm))
                torch.cuda.synchronize(self.dev)
            else:
                L.check(L.lib().ddn_plan_set_resolvent(self._h, float(cK), float(cD), C.c_void_p(0), 0))

    def set_ssim(self, ring: Optional[torch.Tensor], max_val: float = 1.0):
        """ssim as the in-loop loss of the following *_run calls: ring = float64 device tensor [iters][batch][2], or None = off."""
        from .core.loss import ssim_window_taps
        if ring is None:
            L.check(L.lib().ddn_plan_set_ssim(self._h, None, 0.0, C.c_void_p(0)))
            return
        g11 = (C.c_float * 11)(*ssim_window_taps().tolist())
        L.check(L.lib().ddn_plan_set_ssim(self._h, g11, float(max_val), _ptr(ring)))

    def get_eigK(self) -> torch.Tensor:
        ct = torch.complex128 if self.dtype == torch.float64 else torch.complex64
        out = torch.empty((self.H, self.W), dtype=ct, device=self.dev)
        with torch.cuda.device(self.dev):
            L.check(L.lib().ddn_plan_get_eigK(self._h, _ptr(out)))
        return out

    # ---- applications
    def apply(self, op: int, x: torch.Tensor) -> torch.Tensor:
        y = torch.empty_like(x)
        with torch.cuda.device(self.dev):
            self.sync_stream()
            L.check(L.lib().ddn_apply(self._h, op, _ptr(x), _ptr(y)))
        return y

    def apply_multiplier(self, mult, x: torch.Tensor, conj=False, recip=False) -> torch.Tensor:
        m, cm = self._cplx(mult)
        y = torch.empty_like(x)
        with torch.cuda.device(self.dev):
            self.sync_stream()
            L.check(L.lib().ddn_apply_multiplier(self._h, _ptr(m), cm, (1 if conj else 0) | (2 if recip else 0), _ptr(x), _ptr(y)))
            torch.cuda.synchronize(self.dev)
        return y

    def circular_convolve2d(self, kernel, image: torch.Tensor) -> torch.Tensor:
        k = taps_array(kernel)
        out = torch.empty_like(image)
        with torch.cuda.device(self.dev):
            self.sync_stream()
            L.check(L.lib().ddn_circular_convolve2d(self._h, k.ctypes.data_as(C.POINTER(C.c_double)), k.shape[0], k.shape[1], _ptr(image), _ptr(out)))
        return out

    def enable_timing(self, on=True):
        L.check(L.lib().ddn_plan_enable_timing(self._h, int(on)))

    def last_timing(self):
        r, c = C.c_double(0), C.c_double(0)
        nr, nc = C.c_int(0), C.c_int(0)
        L.check(L.lib().ddn_plan_last_timing(self._h, C.byref(r), C.byref(c), C.byref(nr), C.byref(nc)))
        return {"row_ms": r.value, "col_ms": c.value, "n_row": nr.value, "n_col": nc.value}


def prox_call(kind: int, x: torch.Tensor, lam: float) -> torch.Tensor:
    y = torch.empty_like(x)
    n = x.numel() // 2 if kind == L.PROX_ISO else x.numel()
    with torch.cuda.device(x.device):
        L.check(L.lib().ddn_prox(kind, L.F64 if x.dtype == torch.float64 else L.F32, _ptr(x), _ptr(y), n, float(lam),
                                 C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
    return y


def reduce_call(x: torch.Tensor, y: torch.Tensor, batch: int = 1) -> torch.Tensor:
    out = torch.empty((batch, 4), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        L.check(L.lib().ddn_reduce(L.F64 if x.dtype == torch.float64 else L.F32, _ptr(x), _ptr(y), x.numel() // batch, batch, _ptr(out),
                                   C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
    return out


# ---- plan cache ---------------------------------------------------------------------------------
# A plan owns the spectral scratch, the twiddle tables and the multipliers of one (size, batch, dtype, blur kernel,
# resolvent coefficients) combination: about 0.6 GB and a few milliseconds of set-up at 4096 x 4096.  Solver calls
# in a loop (deblur_image per frame, parameter sweeps) repeat that combination, so plans that a caller releases are
# parked here and handed to the next caller with the same key instead of being rebuilt.  A parked plan is owned by
# nobody; acquire removes it from the cache, so two live users never share one.  DDN_PLAN_CACHE=<n> sets the number
# of parked plans (default 2, 0 = off).
_PLAN_CACHE: "OrderedDict[tuple, Plan]" = OrderedDict()
_CACHE_STATS = {"hits": 0, "misses": 0}


def _cache_capacity() -> int:
    try:
        return max(0, int(os.environ.get("DDN_PLAN_CACHE", "2")))
    except ValueError:
        return 2


# Environment switches that decide how the library builds a plan (it reads them once, at plan creation): a plan built
# under one setting must not be handed to a caller that runs under another, so they are part of the key.
_BUILD_SWITCHES = ("DDN_FORCE_LONG_ROWS", "DDN_NO_WAVE_COLS", "DDN_NO_WAVE_ROWS", "DDN_FORCE_WAVE_ROWS", "DDN_NO_WHOLE_ROWS", "DDN_NO_ILV",
                   "DDN_NO_NYQ_DENSE", "DDN_NO_NYQ_FIR", "DDN_NO_PDR_TWIN", "DDN_NO_TAPS", "DDN_NO_FIXED_COLS", "DDN_TILE_PAD_ROWS",
                   "DDN_DEBUG_MODE", "DDN_DEBUG_STAMPS", "DDN_COL_STAGGER_US", "DDN_COLN_NP")


def plan_key(H, W, batch, dtype, dev, kernel, cK, cD, tune):
    k = taps_array(kernel)
    tk = tuple(sorted((tune or {}).items()))
    env = tuple((n, os.environ[n]) for n in _BUILD_SWITCHES if n in os.environ)
    return (int(H), int(W), int(batch), str(dtype), str(dev), k.shape, k.tobytes(), float(cK), float(cD), tk, env)


def _parked_bytes_limit() -> int:
    """Device memory the parked plans of one process may hold together (DDN_PLAN_CACHE_GIB, default 8 GiB): a plan of a
    16384 x 16384 or 32768 x 32768 image holds tens of GiB and is not worth keeping for a caller that may never return."""
    try:
        return int(float(os.environ.get("DDN_PLAN_CACHE_GIB", "8")) * (1 << 30))
    except ValueError:
        return 8 << 30


def plan_device_bytes(plan: "Plan") -> int:
    """Estimate of the device memory a whole-plane plan owns: two half-spectra, the tiled and column-major multipliers,
    the resolvent and scratch multipliers -- about 10 real words per pixel and image (split-row scratch not included)."""
    word = 8 if plan.dtype == torch.float64 else 4
    return int(10 * word * plan.H * plan.W * max(1, plan.batch))


def acquire_plan(key):
    """A parked plan with this key (its kernel and resolvent are already set), or None."""
    pl = _PLAN_CACHE.pop(key, None)
    _CACHE_STATS["hits" if pl is not None else "misses"] += 1
    return pl


def release_plan(plan: "Plan"):
    """Parks a plan for reuse (or destroys it when it has no key or the cache is off)."""
    cap = _cache_capacity()
    if plan.cache_key is None or cap == 0 or not (plan._h and plan._h.value):
        plan.close()
        return
    old = _PLAN_CACHE.pop(plan.cache_key, None)
    if old is not None and old is not plan:
        old.close()
    limit = _parked_bytes_limit()
    if plan_device_bytes(plan) > limit:
        plan.close()
        return
    _PLAN_CACHE[plan.cache_key] = plan
    while len(_PLAN_CACHE) > cap or sum(plan_device_bytes(q) for q in _PLAN_CACHE.values()) > limit:
        _, victim = _PLAN_CACHE.popitem(last=False)
        victim.close()


def create_plan(*args, **kwargs) -> "Plan":
    """Plan(...) that, when the device is out of memory, frees the parked plans and tries once more."""
    try:
        return Plan(*args, **kwargs)
    except L.DdnError as e:
        if not _PLAN_CACHE or "memory" not in str(e).lower():
            raise
        clear_plan_cache()
        torch.cuda.empty_cache()
        return Plan(*args, **kwargs)


def clear_plan_cache():
    while _PLAN_CACHE:
        _, pl = _PLAN_CACHE.popitem()
        pl.close()


def plan_cache_stats():
    return dict(_CACHE_STATS, parked=len(_PLAN_CACHE))
