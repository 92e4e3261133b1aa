"""The four TV-deblurring solvers with the reference's public signatures; each drives the fused
HIP iteration loop of libddn (ddn_*_run).

Keyword extensions common to all four (all optional, carried by **kwargs as in the reference):
  device        compute device (default: current cuda device); results return where `b` lives
  dtype         torch.float32 (default: dtype of b) or torch.float64
  init_state    tuple of initial iterates (skips the RNG draws); shapes as in the reference
  return_state  also return the final iterates
  return_losses also return the per-iteration loss list
  eig_arrays    {"K","D","M"}: externally supplied eigenvalue arrays (fp64 parity)
  tune          column tiling knobs, see DeblurDenoiseOperators
Images may carry a leading batch dimension (B, H, W) or (B, C, H, W): every plane is solved independently,
exactly as B (B*C) calls of the reference would, inside one launch sequence; the result has the same leading
shape, `return_losses` then gives one list per plane (a plane that stopped early has a shorter list), and the
progress log lines describe plane 0.
Loss functions: l1_loss, l2_loss, mse, psnr and ssim of this package are evaluated on the device inside the fused
loop (no host round trip per iteration, early stop included), and with save_loss=True the five metric columns of the
reference's CSV come from the same device records (one ddn_*_run call for the whole solve); any other callable takes
the per-iteration host loop the reference has.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Callable, Optional

import numpy as np
import torch

from . import _lib as L
from .core.loss import l1_loss, l2_loss, mse, psnr, ssim
from .core.proximal_operators import prox_box
from .op_math.operators import DeblurDenoiseOperators
from .plan import _ptr, compute_device, real_dtype, stage
from .utils.logging_utils import log_execution_time, logger, save_loss_data

_KIND_INDEX = {L.LOSS_L1: 0, L.LOSS_L2: 1, L.LOSS_MSE: 2, L.LOSS_PSNR: 3}
_OWN_LOSSES = (l1_loss, l2_loss, mse, psnr, ssim)


def _as_batch(b: torch.Tensor):
    """(H,W) -> (1,H,W); (B,H,W) as is; (B,C,H,W) -> (B*C,H,W): every plane is an independent image for the
    reference (multiplying_matrix.py:31).  Returns (tensor3d, had_batch, lead) with lead = (B, C) for 4-D input."""
    if isinstance(b, np.ndarray):
        b = torch.from_numpy(b)
    if b.dim() == 2:
        return b.unsqueeze(0), False, None
    if b.dim() == 3:
        return b, True, None
    if b.dim() == 4:
        return b.reshape(b.shape[0] * b.shape[1], b.shape[2], b.shape[3]), True, (b.shape[0], b.shape[1])
    raise ValueError(f"expected an image of shape (H, W), (B, H, W) or (B, C, H, W), got {tuple(b.shape)}")


def _shape_out(t: torch.Tensor, had_batch: bool, lead):
    """(B', H, W) result -> the caller's leading shape."""
    if lead is not None:
        return t.reshape(*lead, t.shape[-2], t.shape[-1])
    return t if had_batch else t[0]


def _eval_loss(fn, a: torch.Tensor, b: torch.Tensor, src):
    """loss_function(iterate, b) on the host path.  The package's own losses run on the device tensors (only the scalar
    comes back); any other callable gets tensors on the device its inputs came from, as the reference would pass them."""
    if any(fn is f for f in _OWN_LOSSES):
        v = fn(a, b)
    else:
        v = fn(a.to(src), b.to(src))
    return v.item() if isinstance(v, torch.Tensor) and v.numel() == 1 else v


def _loss_lists(vals: torch.Tensor, done, lo: int, had_batch: bool):
    """vals: [iters][batch] on the host; image i's list covers ring slots lo .. done[i] - 1 (an image that stopped early
    has no later entries).  One list for a single image, a list per image for a batch."""
    per = [[float(v) for v in vals[lo:int(done[i]), i]] for i in range(vals.shape[1])]
    return per if had_batch else per[0]


class _SsimRing:
    """The device record of the in-loop ssim (ddn_plan_set_ssim) for one fused run; switched off again on exit.
    It is on when ssim is the loss, and when the metric history is wanted (record=True: save_loss); b_every is the
    cadence of primal DR's ssim(z1, b) (1 = every iteration, the saved history; None = the library's default)."""

    def __init__(self, plan, kind, n, B, dev, record=False, b_every=None):
        self.plan, self.kind, self.on = plan, kind, (kind == L.LOSS_SSIM or record)
        self.b_every = b_every
        self.ring = torch.full((max(n, 1), B, 2), float("nan"), dtype=torch.float64, device=dev) if self.on else None

    def __enter__(self):
        if self.on:
            self.plan.set_ssim(self.ring)
            if self.b_every is not None:
                L.check(L.lib().ddn_plan_set_ssim_cadence(self.plan._h, int(self.b_every)))
        return self

    def __exit__(self, *exc):
        if self.on:
            self.plan.set_ssim(None)
            if self.b_every is not None:
                L.check(L.lib().ddn_plan_set_ssim_cadence(self.plan._h, 50))
        return False

    def values(self, ring, kind, pair=0):
        if kind == L.LOSS_SSIM:
            return self.ring[:, :, pair].cpu()
        return _ring_values(ring, kind, pair)

    def metric_lists(self, ring, pair, ssim_entry, lo, hi):
        """The five metric columns of the reference's saved history (psnr, ssim, mse, l1_loss, l2_loss of (iterate, b),
        image 0; cp:101-107, admm:142-147, pdr:146-154, pddr:108-113) for ring rows lo .. hi - 1."""
        r = ring[lo:hi, 0].cpu()
        sv = self.ring[lo:hi, 0, ssim_entry].cpu()
        cols = {"psnr": r[:, 4 * pair + 3], "ssim": sv, "mse": r[:, 4 * pair + 2], "l1_loss": r[:, 4 * pair + 0], "l2_loss": r[:, 4 * pair + 1]}
        return {k: [float(v) for v in cols[k]] for k in ("psnr", "ssim", "mse", "l1_loss", "l2_loss")}


def _loss_kind(loss_function) -> int:
    return int(getattr(loss_function, "_ddn_kind", 0)) if loss_function is not None else 0


def _loss_name(loss_function) -> str:
    return getattr(loss_function, "__name__", str(loss_function))


def _ring_values(ring: torch.Tensor, kind: int, pair: int = 0):
    """ring: [iters][batch][8] -> python list (batch 1) of the chosen loss kind per iteration."""
    col = 4 * pair + _KIND_INDEX[kind]
    r = ring[:, :, col].cpu()
    return r


# =============================================================================== Chambolle-Pock
@log_execution_time(logger)
def chambolle_pock(b: torch.Tensor, kernel: torch.Tensor, objective_function: str = "l1", t: float = 0.4, s: float = 0.7,
                   gamma: float = 0.01, max_iter: int = 500, loss_function: Callable = psnr, save_loss: bool = False, **kwargs) -> torch.Tensor:
    """Chambolle-Pock primal-dual iteration (chambolle_pock/algorithm.py:21-143).  b: (1, H, W) (or
    (H, W) / (B, H, W)); returns the same leading shape.  Runs max_iter - 1 iterations, no early stop."""
    if objective_function not in ("l1", "l2"):
        raise ValueError("Invalid objective function. Choose 'l1' or 'l2'.")
    start_time = time.time()
    b3, _, lead = _as_batch(b)
    src = b3.device
    dt = real_dtype(kwargs.get("dtype", b3.dtype))
    dev = compute_device(kwargs.get("device", src))
    B, H, W = b3.shape
    logger.info(f"Running Chambolle-Pock with values: t={t}, s={s}, gamma={gamma}")
    ops = DeblurDenoiseOperators(kernel, b3[0], 1, 1, device=dev, batch=B, dtype=dt,
                                 eig_arrays=kwargs.get("eig_arrays"), tune=kwargs.get("tune"))          # cp:46
    try:
        init = kwargs.get("init_state")
        if init is None:                                                                            # cp:50-52 draw order
            xs, zs, ys = [], [], []
            for _ in range(B):
                xs.append(torch.rand(size=(H, W), dtype=dt)); zs.append(torch.rand(size=(H, W), dtype=dt)); ys.append(torch.rand(size=(3, H, W), dtype=dt))
            x0, z0, y0 = torch.stack(xs), torch.stack(zs), torch.stack(ys, dim=1)
        else:
            x0, z0, y0 = [torch.as_tensor(v) for v in init]
            x0 = x0.reshape(B, H, W); z0 = z0.reshape(B, H, W); y0 = y0.reshape(3, B, H, W)
        x, z, y = stage(x0, dev, dt), stage(z0, dev, dt), stage(y0, dev, dt)
        bd = stage(b3, dev, dt)
        n = max(int(max_iter) - 1, 0)
        kind = _loss_kind(loss_function)
        losses = []
        extra = {}
        with torch.cuda.device(dev):
            ops.plan.sync_stream()
            if kind:
                ring = torch.zeros((max(n, 1), B, 8), dtype=torch.float64, device=dev)
                with _SsimRing(ops.plan, kind, n, B, dev, record=save_loss) as sr:
                    L.check(L.lib().ddn_cp_run(ops.plan._h, _ptr(x), _ptr(z), _ptr(y), _ptr(bd), float(t), float(s), float(gamma),
                                               L.OBJ[objective_function], n, _ptr(ring)))
                    vals = sr.values(ring, kind)
                    if save_loss:
                        extra = sr.metric_lists(ring, 0, 0, 1, n)                                    # (x_next, b), cp:101-107
                losses = _loss_lists(vals, [n] * B, 1, B > 1)                                        # loss for k > 1 (cp:94)
                first = losses[0] if B > 1 else losses
                for k in range(2, n + 1):
                    if k % 50 == 0 or k == max_iter - 1:
                        logger.info(f"Iteration {k} completed. {_loss_name(loss_function)} : {first[k - 2]}")
            else:
                lists = {m: [] for m in ("psnr", "ssim", "mse", "l1_loss", "l2_loss")}
                for k in range(1, n + 1):
                    L.check(L.lib().ddn_cp_run(ops.plan._h, _ptr(x), _ptr(z), _ptr(y), _ptr(bd), float(t), float(s), float(gamma),
                                               L.OBJ[objective_function], 1, C.c_void_p(0)))
                    if k > 1:
                        lv = _eval_loss(loss_function, x if B > 1 else x[0:1], bd, src) if loss_function is not None else 0.0
                        losses.append(lv)
                        if save_loss:
                            for nm, fn in (("psnr", psnr), ("ssim", ssim), ("mse", mse), ("l1_loss", l1_loss), ("l2_loss", l2_loss)):
                                v = fn(x[0], bd[0])
                                lists[nm].append(float(v.reshape(-1)[0]) if isinstance(v, torch.Tensor) else float(v))
                        if k % 50 == 0 or k == max_iter - 1:
                            logger.info(f"Iteration {k} completed. {_loss_name(loss_function)} : {lv}")
                extra = lists
            torch.cuda.synchronize(dev)
        if save_loss and losses:
            params = {"t": t, "gamma": gamma, "s": s, "niters": max_iter, "objective_function": objective_function,
                      "loss_function": _loss_name(loss_function)}
            params.update(extra)
            save_loss_data(losses[0] if (kind and B > 1) else losses, "chambolle_pock", _loss_name(loss_function), params, start_time, kwargs.get("results_dir"))
        out = x.to(src)                                                                             # the reference broadcasts (H,W) b to (1,H,W) at cp:83
        if lead is not None:
            out = _shape_out(out, True, lead)
        ret = [out]
        if kwargs.get("return_state"):
            ret.append((x.to(src), z.to(src), y.to(src)))
        if kwargs.get("return_losses"):
            ret.append(losses)
        return ret[0] if len(ret) == 1 else tuple(ret)
    finally:
        ops.close()


# ========================================================================================= ADMM
@log_execution_time(logger)
def admm_solver(b: torch.Tensor, objective_function: str = "l1", t: float = 18, rho: float = 0.001, gamma: float = 0.5,
                kernel: torch.Tensor = None, niters: int = 500, tol: float = 1e-4, loss_function: Callable = psnr,
                save_loss: bool = False, **kwargs):
    """ADMM / dual Douglas-Rachford (admm/algorithm.py:20-201).  b: (H, W) (or (B, H, W)).
    Runs up to niters - 1 iterations; stops when loss_function(sol, b) < tol (admm:139-154);
    returns the solution of the final solve (admm:160-173)."""
    if objective_function not in ("l1", "l2"):
        raise ValueError("Invalid objective function. Choose 'l1' or 'l2'.")
    start_time = time.time()
    b3, had_batch, lead = _as_batch(b)
    src = b3.device
    dt = real_dtype(kwargs.get("dtype", b3.dtype))
    dev = compute_device(kwargs.get("device", src))
    B, H, W = b3.shape
    logger.info(f"Starting ADMM solver with parameters: t={t}, rho={rho}, gamma={gamma}, niters={niters}")
    init = kwargs.get("init_state")
    if init is None:                                                                                # admm:49-54 draw order; x is drawn and unused
        us, ws, ys, zs = [], [], [], []
        for _ in range(B):
            _x = torch.clamp(torch.randn((H, W), dtype=torch.float32), min=0, max=1)
            us.append(torch.clamp(torch.randn((H, W), dtype=torch.float32), min=0, max=1))
            ws.append(torch.clamp(torch.randn((H, W), dtype=torch.float32), min=0, max=1))
            ys.append(torch.clamp(torch.randn((3, H, W), dtype=torch.float32), min=0, max=1))
            zs.append(torch.clamp(torch.randn((3, H, W), dtype=torch.float32), min=0, max=1))
        u0, w0, y0, z0 = torch.stack(us), torch.stack(ws), torch.stack(ys, dim=1), torch.stack(zs, dim=1)
    else:
        u0, w0, y0, z0 = [torch.as_tensor(v) for v in init]
        u0 = u0.reshape(B, H, W); w0 = w0.reshape(B, H, W); y0 = y0.reshape(3, B, H, W); z0 = z0.reshape(3, B, H, W)
    ops = DeblurDenoiseOperators(kernel, b3[0], 1, 1, device=dev, batch=B, dtype=dt,
                                 eig_arrays=kwargs.get("eig_arrays"), tune=kwargs.get("tune"))          # admm:56-59
    try:
        u, w, y, z = stage(u0, dev, dt), stage(w0, dev, dt), stage(y0, dev, dt), stage(z0, dev, dt)
        bd = stage(b3, dev, dt)
        sol = torch.zeros_like(bd)
        n = max(int(niters) - 1, 0)
        kind = _loss_kind(loss_function)
        losses, extra = [], {}
        obj = L.OBJ[objective_function]
        with torch.cuda.device(dev):
            ops.plan.sync_stream()
            if kind or (loss_function is None and not save_loss):
                ring = torch.zeros((max(n, 1), B, 8), dtype=torch.float64, device=dev)
                done = torch.zeros((B,), dtype=torch.int32, device=dev)
                with _SsimRing(ops.plan, kind, n, B, dev, record=save_loss) as sr:
                    L.check(L.lib().ddn_admm_run(ops.plan._h, _ptr(u), _ptr(w), _ptr(y), _ptr(z), _ptr(bd), float(t), float(rho), float(gamma), obj, n,
                                                 kind, float(tol), 2, _ptr(ring), _ptr(done), _ptr(sol)))
                    if kind:
                        vals = sr.values(ring, kind)
                        if save_loss:
                            extra = sr.metric_lists(ring, 0, 0, 1, int(done[0]))                     # (sol, b), admm:142-147
                if kind:
                    dn = done.cpu().tolist()
                    losses = _loss_lists(vals, dn, 1, had_batch)                                     # i > 1 (admm:118)
                    first = losses[0] if had_batch else losses
                    for i in range(2, dn[0] + 1):
                        if i % 50 == 0 or i == niters - 1:
                            logger.info(f"Iteration {i} completed. {_loss_name(loss_function)} : {first[i - 2]}")
            else:
                lists = {m: [] for m in ("psnr", "ssim", "mse", "l1_loss", "l2_loss")}
                if n == 0:
                    L.check(L.lib().ddn_admm_run(ops.plan._h, _ptr(u), _ptr(w), _ptr(y), _ptr(z), _ptr(bd), float(t), float(rho), float(gamma), obj, 0,
                                                 0, 0.0, 2, C.c_void_p(0), C.c_void_p(0), _ptr(sol)))
                for i in range(1, n + 1):
                    L.check(L.lib().ddn_admm_run(ops.plan._h, _ptr(u), _ptr(w), _ptr(y), _ptr(z), _ptr(bd), float(t), float(rho), float(gamma), obj, 1,
                                                 0, 0.0, 2, C.c_void_p(0), C.c_void_p(0), _ptr(sol)))
                    if i > 1:
                        lv = _eval_loss(loss_function, sol if had_batch else sol[0], bd if had_batch else bd[0], src)
                        losses.append(lv)
                        if save_loss:
                            for nm, fn in (("psnr", psnr), ("ssim", ssim), ("mse", mse), ("l1_loss", l1_loss), ("l2_loss", l2_loss)):
                                v = fn(sol[0], bd[0])
                                lists[nm].append(float(v.reshape(-1)[0]) if isinstance(v, torch.Tensor) else float(v))
                        if lv < tol:
                            logger.info(f"Converged at iteration {i} with relative difference {lv:.6f}")
                            break
                        if i % 50 == 0 or i == niters - 1:
                            logger.info(f"Iteration {i} completed. {_loss_name(loss_function)} : {lv}")
                extra = lists
            torch.cuda.synchronize(dev)
        if save_loss and losses:
            params = {"t": t, "rho": rho, "gamma": gamma, "niters": niters, "objective_function": objective_function,
                      "loss_function": _loss_name(loss_function)}
            params.update(extra)
            save_loss_data(losses[0] if (kind and had_batch) else losses, "admm", _loss_name(loss_function), params, start_time, kwargs.get("results_dir"))
        out = _shape_out(sol, had_batch, lead).to(src)
        ret = [out]
        if kwargs.get("return_state"):
            ret.append((u.to(src), w.to(src), y.to(src), z.to(src)))
        if kwargs.get("return_losses"):
            ret.append(losses)
        return ret[0] if len(ret) == 1 else tuple(ret)
    finally:
        ops.close()


# ============================================================================ primal Douglas-Rachford
@log_execution_time(logger)
def primal_dr_splitting(problem: str, kernel: torch.Tensor, b: torch.Tensor, image_path: str = None, shape: tuple = (100, 100),
                        loss_function: Callable = ssim, niters: int = 500, gamma: float = 0.049, t: float = 2.0, rho: float = 0.1,
                        tol: float = 10 ** -6, display: bool = False, save_loss: bool = False, **kwargs) -> torch.Tensor:
    """Primal Douglas-Rachford (Spingarn) splitting (primal_dr/algorithm.py:21-187).  problem 'l1'
    selects the l1 data term, anything else the squared-l2 one (pdr:101-110).  Runs up to niters
    iterations; stops when loss_function(z1, z1prev) < tol for j > 1 (pdr:143-161); returns clip(z1)."""
    start_time = time.time()
    b3, had_batch, lead = _as_batch(b)
    src = b3.device
    dt = real_dtype(kwargs.get("dtype", b3.dtype))
    dev = compute_device(kwargs.get("device", src))
    B, H, W = b3.shape
    logger.info(f"Running Primal Douglas-Rachford Splitting with parameters: t={t} gamma={gamma} rho={rho} tol={tol} niters={niters}")
    ops = DeblurDenoiseOperators(kernel, b3[0], t, device=dev, batch=B, dtype=dt,
                                 eig_arrays=kwargs.get("eig_arrays"), tune=kwargs.get("tune"))          # pdr:91
    try:
        bd = stage(b3, dev, dt)
        init = kwargs.get("init_state")
        if init is None:                                                                            # pdr:88-89
            z1 = bd.clone()
            z2 = torch.stack((bd, bd, bd)).contiguous()
        else:
            z1 = stage(torch.as_tensor(init[0]).reshape(B, H, W), dev, dt)
            z2 = stage(torch.as_tensor(init[1]).reshape(3, B, H, W), dev, dt)
        n = max(int(niters), 0)
        kind = _loss_kind(loss_function)
        obj = 0 if problem == "l1" else 1
        # the reference starts both difference duals from b (pdr:88-89) and D1 == D2 keeps them equal: the library is
        # told so and streams one array for the two (ddn.h, DDN_OBJ_DUALS_EQUAL)
        twin = L.OBJ_DUALS_EQUAL if (init is None or bool(torch.equal(z2[1], z2[2]))) else 0
        losses, extra = [], {}
        with torch.cuda.device(dev):
            ops.plan.sync_stream()
            if kind or (loss_function is None and not save_loss):
                ring = torch.zeros((max(n, 1), B, 8), dtype=torch.float64, device=dev)
                done = torch.zeros((B,), dtype=torch.int32, device=dev)
                with _SsimRing(ops.plan, kind, n, B, dev, record=save_loss, b_every=1 if save_loss else None) as sr:
                    L.check(L.lib().ddn_pdr_run(ops.plan._h, _ptr(z1), _ptr(z2), _ptr(bd), float(t), float(rho), float(gamma), obj | twin, n,
                                                kind, float(tol), 3, _ptr(ring), _ptr(done)))
                    if kind:
                        vals = sr.values(ring, kind)
                        vb = sr.values(ring, kind, pair=1)
                        if save_loss:
                            extra = sr.metric_lists(ring, 1, 1, 2, int(done[0]))                     # (z1, b), pdr:146-154
                if kind:
                    dn = done.cpu().tolist()
                    losses = _loss_lists(vals, dn, 2, had_batch)                                     # j > 1 (pdr:143)
                    for j in range(0, dn[0], 50):
                        logger.info(f"Iteration {j} completed. {_loss_name(loss_function)} loss : {float(vb[j, 0])}")
                    if max(dn) == n:
                        logger.info(f"Warning: maxiter reached ({niters}), primal_dr did not converge")
            else:
                lists = {m: [] for m in ("psnr", "ssim", "mse", "l1_loss", "l2_loss")}
                broke = False
                for j in range(n):
                    z1prev = z1.clone()
                    L.check(L.lib().ddn_pdr_run(ops.plan._h, _ptr(z1), _ptr(z2), _ptr(bd), float(t), float(rho), float(gamma), obj, 1,
                                                0, 0.0, 3, C.c_void_p(0), C.c_void_p(0)))
                    zv = (z1 if had_batch else z1[0])
                    if j % 50 == 0:
                        lv = _eval_loss(loss_function, zv, bd if had_batch else bd[0], src)
                        logger.info(f"Iteration {j} completed. {_loss_name(loss_function)} loss : {lv}")
                    if j > 1:
                        lv = _eval_loss(loss_function, zv, z1prev if had_batch else z1prev[0], src)
                        losses.append(lv)
                        if save_loss:
                            for nm, fn in (("psnr", psnr), ("ssim", ssim), ("mse", mse), ("l1_loss", l1_loss), ("l2_loss", l2_loss)):
                                v = fn(z1[0], bd[0])
                                lists[nm].append(float(v.reshape(-1)[0]) if isinstance(v, torch.Tensor) else float(v))
                        if lv < tol:
                            broke = True
                            break
                if not broke:
                    logger.info(f"Warning: maxiter reached ({niters}), primal_dr did not converge")
                extra = lists
            torch.cuda.synchronize(dev)
        if save_loss and losses:
            params = {"t": t, "gamma": gamma, "rho": rho, "niters": niters, "tol": tol}
            params.update(extra)
            save_loss_data(losses[0] if (kind and had_batch) else losses, "primal_dr_splitting", _loss_name(loss_function), params, start_time, kwargs.get("results_dir"))
        solution = prox_box(z1, t)                                                                  # pdr:182-187
        out = _shape_out(solution, had_batch, lead).to(src)
        ret = [out]
        if kwargs.get("return_state"):
            ret.append((z1.to(src), z2.to(src)))
        if kwargs.get("return_losses"):
            ret.append(losses)
        return ret[0] if len(ret) == 1 else tuple(ret)
    finally:
        ops.close()


# ====================================================================== primal-dual Douglas-Rachford
@log_execution_time(logger)
def primal_dual_dr_splitting(b: torch.Tensor, kernel: torch.Tensor, niters: int = 1000, t: float = 3, rho: float = 0.4, gamma: float = 0.03,
                             loss_function: Callable = l2_loss, tol: float = 1e-6, save_loss: bool = False, **kwargs) -> torch.Tensor:
    """Primal-dual Douglas-Rachford splitting, l1 data term (primal_dual_dr/algorithm.py:20-139).
    Runs up to niters - 1 iterations; stops when loss_function(clip(p), b) < tol for k > 1
    (pddr:103-121); returns clip(p)."""
    start_time = time.time()
    if isinstance(b, torch.Tensor) and b.dim() == 3 and b.shape[0] == 1:
        b3, had_batch, lead = b, False, None                                                        # pddr:52-56: (1,H,W) is squeezed
    else:
        b3, had_batch, lead = _as_batch(b)
    src = b3.device
    dt = real_dtype(kwargs.get("dtype", b3.dtype))
    dev = compute_device(kwargs.get("device", src))
    B, H, W = b3.shape
    init = kwargs.get("init_state")
    if init is None:                                                                                # pddr:57-58
        ps, qs = [], []
        for _ in range(B):
            ps.append(torch.rand(size=(H, W), dtype=dt)); qs.append(torch.rand(size=(3, H, W), dtype=dt))
        p0, q0 = torch.stack(ps), torch.stack(qs, dim=1)
    else:
        p0, q0 = [torch.as_tensor(v) for v in init]
        p0 = p0.reshape(B, H, W); q0 = q0.reshape(3, B, H, W)
    ops = DeblurDenoiseOperators(kernel, b3[0], t, t, device=dev, batch=B, dtype=dt,
                                 eig_arrays=kwargs.get("eig_arrays"), tune=kwargs.get("tune"))          # pddr:61
    try:
        p, q = stage(p0, dev, dt), stage(q0, dev, dt)
        bd = stage(b3, dev, dt)
        n = max(int(niters) - 1, 0)
        kind = _loss_kind(loss_function)
        losses, extra = [], {}
        with torch.cuda.device(dev):
            ops.plan.sync_stream()
            if kind or (loss_function is None and not save_loss):
                ring = torch.zeros((max(n, 1), B, 8), dtype=torch.float64, device=dev)
                done = torch.zeros((B,), dtype=torch.int32, device=dev)
                with _SsimRing(ops.plan, kind, n, B, dev, record=save_loss) as sr:
                    L.check(L.lib().ddn_pddr_run(ops.plan._h, _ptr(p), _ptr(q), _ptr(bd), float(t), float(rho), float(gamma), n,
                                                 kind, float(tol), 2, _ptr(ring), _ptr(done)))
                    if kind:
                        vals = sr.values(ring, kind)
                        if save_loss:
                            extra = sr.metric_lists(ring, 0, 0, 1, int(done[0]))                     # (clip(p), b), pddr:108-113
                if kind:
                    dn = done.cpu().tolist()
                    losses = _loss_lists(vals, dn, 1, had_batch)                                     # k > 1 (pddr:103)
                    first = losses[0] if had_batch else losses
                    for k in range(2, dn[0] + 1):
                        if k % 50 == 0:
                            logger.info(f"Iteration {k} completed. {_loss_name(loss_function)} loss : {first[k - 2]}")
            else:
                lists = {m: [] for m in ("psnr", "ssim", "mse", "l1_loss", "l2_loss")}
                for k in range(1, n + 1):
                    L.check(L.lib().ddn_pddr_run(ops.plan._h, _ptr(p), _ptr(q), _ptr(bd), float(t), float(rho), float(gamma), 1,
                                                 0, 0.0, 2, C.c_void_p(0), C.c_void_p(0)))
                    if k > 1:
                        sol = prox_box(p, 1)
                        lv = _eval_loss(loss_function, sol if B > 1 else sol[0], bd, src)
                        losses.append(lv)
                        if save_loss:
                            for nm, fn in (("psnr", psnr), ("ssim", ssim), ("mse", mse), ("l1_loss", l1_loss), ("l2_loss", l2_loss)):
                                v = fn(sol[0], bd[0])
                                lists[nm].append(float(v.reshape(-1)[0]) if isinstance(v, torch.Tensor) else float(v))
                        if k % 50 == 0:
                            logger.info(f"Iteration {k} completed. {_loss_name(loss_function)} loss : {lv}")
                        if lv < tol:
                            logger.info(f"Converged at iteration {k} with relative difference {lv:.6f}")
                            break
                extra = lists
            torch.cuda.synchronize(dev)
        if save_loss:
            params = {"t": t, "rho": rho, "gamma": gamma, "niters": niters, "tol": tol}
            params.update(extra)
            save_loss_data(losses[0] if (kind and had_batch) else losses, "primal_dual_dr_splitting", _loss_name(loss_function), params, start_time, kwargs.get("results_dir"))
        x_sol = prox_box(p, 1)                                                                      # pddr:123
        out = _shape_out(x_sol, had_batch, lead).to(src)
        ret = [out]
        if kwargs.get("return_state"):
            ret.append((p.to(src), q.to(src)))
        if kwargs.get("return_losses"):
            ret.append(losses)
        return ret[0] if len(ret) == 1 else tuple(ret)
    finally:
        ops.close()
