"""Rows wider than 8192 pixels (the row transform split W = R x M over three kernels, csrc/ddn_long.hpp) against the CPU
oracle: the 16 x 32768 and 32768 x 32 cases, all four solvers, and the same code path forced onto small images
(DDN_FORCE_LONG_ROWS) so that float64 and odd splits are covered at sizes the oracle finishes in a moment."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ddn_oracle as O  # checker only


@pytest.fixture(scope="module")
def pkg():
    import convex_optimization_amd as pkg
    from convex_optimization_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return pkg


@pytest.fixture(autouse=True)
def _fresh_plans():
    from convex_optimization_amd import plan as P
    P.clear_plan_cache()      # the cache key does not know about DDN_FORCE_LONG_ROWS
    yield
    P.clear_plan_cache()


def rel(a, b):
    return O.rel_l2(np.asarray(a), np.asarray(b))


def _problem(golden_dir, H, W, kname, dt, seed=0):
    k = np.load(os.path.join(golden_dir, "kernels.npz"))
    kern = torch.from_numpy(k[kname])
    torch.manual_seed(seed)
    img = torch.rand(1, H, W, dtype=dt)
    b = O.add_gaussian_noise(O.circular_convolve2d(img, kern), 0.0, 0.05)[0].to(dt)
    stC = (torch.rand(H, W, dtype=dt), torch.rand(H, W, dtype=dt), torch.rand(3, H, W, dtype=dt))
    stA = tuple(torch.clamp(torch.randn(s, dtype=torch.float32), 0, 1).to(dt) for s in [(H, W), (H, W), (3, H, W), (3, H, W)])
    stP = (torch.rand(H, W, dtype=dt), torch.rand(3, H, W, dtype=dt))
    return kern, b, stC, stA, stP


def _all_four(pkg, kern, b, stC, stA, stP, N, eig=None, cp=True):
    ea = (lambda t, s: None) if eig is None else eig
    out = {}
    if cp:
        out["cp"] = (pkg.chambolle_pock_solver(b[None], kern, t=0.3, s=0.3, max_iter=N + 1, init_state=stC, eig_arrays=ea(1, 1)),
                     O.chambolle_pock(b[None], kern, t=0.3, s=0.3, max_iter=N + 1, init_state=stC))
    for obj in ("l1", "l2"):
        out["admm_" + obj] = (pkg.admm_solver(b, obj, kernel=kern, niters=N + 1, tol=-1.0, init_state=stA, eig_arrays=ea(1, 1)),
                              O.admm_solver(b, obj, kernel=kern, niters=N + 1, tol=-1.0, init_state=stA))
    out["pdr"] = (pkg.primal_dr_splitting("l1", kern, b, niters=N, loss_function=pkg.core.mse, tol=-1.0, eig_arrays=ea(2.0, None)),
                  O.primal_dr_splitting("l1", kern, b, niters=N))
    out["pddr"] = (pkg.primal_dual_dr_splitting(b, kern, niters=N + 1, init_state=stP, tol=-1.0, eig_arrays=ea(3, 3)),
                   O.primal_dual_dr_splitting(b, kern, niters=N + 1, init_state=stP))
    return {k: rel(g.numpy(), r.numpy()) for k, (g, r) in out.items()}


@pytest.mark.parametrize("H,W,kname,R", [(64, 64, "gauss9", 1), (48, 80, "motion15_45", 2), (24, 96, "gauss5", 4), (32, 128, "gauss9", 8), (16, 256, "gauss9", 16),
                                         (16, 8192, "motion15_45", 8), (16, 8192, "gauss9", 2)])
def test_forced_split_rows_fp32_vs_oracle(pkg, golden_dir, monkeypatch, H, W, kname, R):
    """The split row path forced onto small images, W = R x M for every supported R (1 = the smallest that fits):
    all four solvers, N = 10 and 40."""
    monkeypatch.setenv("DDN_FORCE_LONG_ROWS", str(R))
    kern, b, stC, stA, stP = _problem(golden_dir, H, W, kname, torch.float32)
    for N in (10, 40):
        res = _all_four(pkg, kern, b, stC, stA, stP, N)
        bad = {k: v for k, v in res.items() if not v < 1e-5}
        assert not bad, (N, res)
    # operator applications through the plain split transforms
    x = torch.rand(H, W)
    ops, oo = pkg.DeblurDenoiseOperators(kern, b, 2.0, 0.5), O.DeblurDenoiseOperators(kern, b, 2.0, 0.5)
    for nm in ("apply_K", "apply_KTrans", "invert_matrix", "apply_Mat"):
        got, want = getattr(ops, nm)(x).real.numpy(), getattr(oo, nm)(x).real.numpy()
        assert np.abs(got - want).max() < 5e-6, nm
    ops.close()


@pytest.mark.parametrize("H,W,R", [(64, 64, 1), (24, 96, 4), (16, 256, 16)])
def test_forced_split_rows_fp64_vs_oracle(pkg, golden_dir, monkeypatch, H, W, R):
    """float64 with the reference's complex64-derived eigen arrays handed in: 1e-11, all four solvers.  Chambolle-Pock
    then carries D z and D^T(y1 + y2) across the split row transform as a second spectrum (the reference's D is
    spectral: eig_vals_for_periodic_conv_op.py:49-59, chambolle_pock/algorithm.py:60-79)."""
    monkeypatch.setenv("DDN_FORCE_LONG_ROWS", str(R))
    kern, b, stC, stA, stP = _problem(golden_dir, H, W, "gauss9", torch.float64)

    def eig(t, s):
        ops = O.DeblurDenoiseOperators(kern, b, t, s)
        return {"K": ops.eig_arry_K, "D": ops.eig_arry_D1, "M": ops.eig_vals_mat}
    res = _all_four(pkg, kern, b, stC, stA, stP, 20, eig=eig)
    bad = {k: v for k, v in res.items() if not v < 1e-11}
    assert not bad, res


@pytest.mark.parametrize("H,W,dt,tol", [(16, 16384, torch.float32, 1e-5), (16, 8192, torch.float64, 1e-11)])
def test_chambolle_pock_with_injected_eigen_arrays_on_long_rows(pkg, golden_dir, H, W, dt, tol):
    """Rows beyond what one workgroup's LDS holds (16384 pixels in float32, 8192 in float64) with the reference's eigen
    arrays injected: Chambolle-Pock on the split row path with the row-spectral D."""
    kern, b, stC, stA, stP = _problem(golden_dir, H, W, "gauss9", dt)
    ops = O.DeblurDenoiseOperators(kern, b, 1, 1)
    ea = {"K": ops.eig_arry_K, "D": ops.eig_arry_D1, "M": ops.eig_vals_mat}
    got = pkg.chambolle_pock_solver(b[None], kern, t=0.3, s=0.3, max_iter=11, init_state=stC, eig_arrays=ea)
    want = O.chambolle_pock(b[None], kern, t=0.3, s=0.3, max_iter=11, init_state=stC)
    assert rel(got.numpy(), want.numpy()) < tol


@pytest.mark.parametrize("H,W,rows", [(16, 32768, "whole"), (16, 32768, "split"), (32768, 32, ""), (16, 16384, "whole"), (16, 16384, "split"),
                                        (16, 8192, "whole"), (16, 8192, "split")])
def test_rows_and_columns_of_32768_vs_oracle(pkg, golden_dir, monkeypatch, H, W, rows):
    """The line lengths of the 32768 x 32768 configuration: 32768-row columns take the split column transform;
    32768-, 16384- and 8192-pixel rows run with the whole row in one workgroup's LDS (csrc/ddn_wholerow.hpp: the
    policy kernel of the three resolvent solvers and the Chambolle-Pock kernel); with DDN_NO_WHOLE_ROWS they take the
    split row transform (8 x 4096, 4 x 4096; at 8192 pixels the workgroup-per-row LDS kernel with the full-length
    complex transform).  All four solvers, N = 10, fp32 <= 1e-5."""
    if rows == "split":
        monkeypatch.setenv("DDN_NO_WHOLE_ROWS", "1")
    kern, b, stC, stA, stP = _problem(golden_dir, H, W, "motion15_45", torch.float32)
    res = _all_four(pkg, kern, b, stC, stA, stP, 10)
    bad = {k: v for k, v in res.items() if not v < 1e-5}
    assert not bad, res


def test_early_stop_and_batch_on_whole_row_kernels(pkg, golden_dir):
    """The per-image stop flag on the whole-row kernels (rows of 8192 pixels): primal DR stops at the oracle's
    iteration with the oracle's iterate, and in a batch of two images each freezes at its own iteration, bit-identical
    to the single-image runs; ADMM's look-ahead solution (admm:118-154) the same way."""
    H, W = 16, 8192
    k = np.load(os.path.join(golden_dir, "kernels.npz"))
    kern = torch.from_numpy(k["gauss9"])
    torch.manual_seed(0)
    img = torch.rand(2, H, W)
    b2 = O.add_gaussian_noise(O.circular_convolve2d(img, kern), 0.0, 0.05)
    b2[1] = 0.1 * b2[1] + 0.45      # a flatter second image: its loss runs below the first one's
    trajs = []
    for i in range(2):
        trajs.append([])
        O.primal_dr_splitting("l1", kern, b2[i], niters=14, loss_function=O.mse, tol=-1.0, losses=trajs[i])
    tol = 0.5 * (float(trajs[0][7]) + float(trajs[1][7]))
    assert float(trajs[1][7]) < 0.97 * float(trajs[0][7])
    counts, singles = [], []
    for i in range(2):
        ol = []
        xo = O.primal_dr_splitting("l1", kern, b2[i], niters=14, loss_function=O.mse, tol=tol, losses=ol)
        x, losses = pkg.primal_dr_splitting("l1", kern, b2[i], niters=14, loss_function=pkg.core.mse, tol=tol, return_losses=True)
        assert len(losses) == len(ol)
        assert rel(x.numpy(), xo.numpy()) < 1e-5
        counts.append(len(ol)); singles.append(x)
    assert counts[0] < 14 and counts[0] != counts[1]
    xb = pkg.primal_dr_splitting("l1", kern, b2, niters=14, loss_function=pkg.core.mse, tol=tol)
    for i in range(2):
        assert np.array_equal(xb[i].numpy(), singles[i].numpy())
    stA = tuple(torch.clamp(torch.randn(s), 0, 1) for s in [(H, W), (H, W), (3, H, W), (3, H, W)])
    ol = []
    O.admm_solver(b2[0], "l1", kernel=kern, niters=30, tol=-1.0, loss_function=O.mse, init_state=stA, losses=ol)
    tolA = 0.5 * (float(ol[9]) + float(ol[10])) if ol[10] < ol[9] else None
    if tolA is not None:
        ol2 = []
        xo = O.admm_solver(b2[0], "l1", kernel=kern, niters=30, tol=tolA, loss_function=O.mse, init_state=stA, losses=ol2)
        x, losses = pkg.admm_solver(b2[0], "l1", kernel=kern, niters=30, tol=tolA, loss_function=pkg.core.mse, init_state=stA, return_losses=True)
        assert len(losses) == len(ol2) and len(ol2) < 29
        assert rel(x.numpy(), xo.numpy()) < 1e-5
