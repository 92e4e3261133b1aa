"""CPU-only checks of the drop-in boundary: the shared library loads, exports every entry point
declared in include/ddn.h, and the host package refuses to run without a HIP device (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ddn.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ddn_[A-Za-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib_path = os.path.join(ROOT, "convex_optimization_amd", "libddn.so")
    assert os.path.exists(lib_path), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(lib_path)
    names = declared_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.ddn_version.restype = ctypes.c_int
    assert lib.ddn_version() >= 100
    from convex_optimization_amd import _lib
    assert set(_lib.EXPORTED) <= set(names)


def test_signatures_match_reference():
    """The public Python signatures are the reference's: same names, order, defaults."""
    import inspect
    import convex_optimization_amd as pkg
    from convex_optimization_amd.core import loss
    sig = inspect.signature(pkg.chambolle_pock.chambolle_pock)
    assert list(sig.parameters)[:9] == ["b", "kernel", "objective_function", "t", "s", "gamma", "max_iter", "loss_function", "save_loss"]
    assert (sig.parameters["t"].default, sig.parameters["s"].default, sig.parameters["gamma"].default, sig.parameters["max_iter"].default) == (0.4, 0.7, 0.01, 500)
    assert sig.parameters["loss_function"].default is loss.psnr
    sig = inspect.signature(pkg.admm.admm_solver)
    assert list(sig.parameters)[:10] == ["b", "objective_function", "t", "rho", "gamma", "kernel", "niters", "tol", "loss_function", "save_loss"]
    assert (sig.parameters["t"].default, sig.parameters["rho"].default, sig.parameters["gamma"].default, sig.parameters["tol"].default) == (18, 0.001, 0.5, 1e-4)
    sig = inspect.signature(pkg.primal_dr.primal_dr_splitting)
    assert list(sig.parameters)[:13] == ["problem", "kernel", "b", "image_path", "shape", "loss_function", "niters", "gamma", "t", "rho", "tol", "display", "save_loss"]
    assert sig.parameters["loss_function"].default is loss.ssim and sig.parameters["gamma"].default == 0.049
    sig = inspect.signature(pkg.primal_dual_dr.primal_dual_dr_splitting)
    assert list(sig.parameters)[:9] == ["b", "kernel", "niters", "t", "rho", "gamma", "loss_function", "tol", "save_loss"]
    assert (sig.parameters["niters"].default, sig.parameters["t"].default, sig.parameters["rho"].default, sig.parameters["gamma"].default) == (1000, 3, 0.4, 0.03)
    assert sig.parameters["loss_function"].default is loss.l2_loss
    sig = inspect.signature(pkg.DeblurDenoiseOperators.__init__)
    assert list(sig.parameters)[1:6] == ["kernel", "blurred_image", "tprimaldr", "s", "device"]
    for name in ["apply_K", "apply_KTrans", "apply_D1", "apply_D2", "apply_D1Trans", "apply_D2Trans", "apply_D", "apply_DTrans", "apply_Mat", "invert_matrix"]:
        assert hasattr(pkg.DeblurDenoiseOperators, name)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    """Without a HIP device the product path raises; it never routes through a CPU implementation."""
    import convex_optimization_amd as pkg
    from convex_optimization_amd._lib import DdnError
    b = torch.rand(32, 32)
    k = pkg.core.gaussian_filter([5, 5], 0.8)
    with pytest.raises(DdnError):
        pkg.primal_dr_splitting("l1", k, b, niters=2)
    with pytest.raises(DdnError):
        pkg.core.prox_l1(b, 0.1)
    with pytest.raises(DdnError):
        pkg.DeblurDenoiseOperators(k, b, 1.0)
    with pytest.raises(DdnError):
        pkg.core.ssim(b, b)


def test_host_side_kernels_match_golden(golden_dir):
    """The host-side blur kernel generators are parity inputs: bit for bit equal to the reference's."""
    import numpy as np
    from convex_optimization_amd.core import create_motion_blur_kernel, gaussian_filter
    g = np.load(os.path.join(golden_dir, "kernels.npz"))
    assert np.array_equal(gaussian_filter([5, 5], 0.8), g["gauss5"])
    assert np.array_equal(gaussian_filter([9, 9], 0.8), g["gauss9"])
    assert np.array_equal(gaussian_filter([10, 10], 0.8), g["gauss10"])
    assert np.array_equal(create_motion_blur_kernel(15, 45).numpy(), g["motion15_45"])
    assert np.array_equal(create_motion_blur_kernel(15, 0).numpy(), g["motion15_0"])
    assert np.array_equal(create_motion_blur_kernel(5, 30).numpy(), g["motion5_30"])
    assert np.array_equal(create_motion_blur_kernel(8, 120).numpy(), g["motion8_120"])


@pytest.mark.gpu
def test_ssim_matches_golden(golden_dir):
    import numpy as np
    from convex_optimization_amd.core.loss import ssim
    g = np.load(os.path.join(golden_dir, "loss.npz"))
    v = ssim(torch.from_numpy(g["a"]), torch.from_numpy(g["b"]))
    np.testing.assert_allclose(v.numpy().reshape(-1), g["ssim"], rtol=1e-6)


@pytest.mark.gpu
def test_ssim_window_sizes_match_golden(golden_dir):
    """ssim(x, y, window_size, max_val) of core/loss.py:39 for windows other than the default 11 (ddn_ssim_window), odd and
    even, float32 and float64, against values the reference itself produced."""
    import numpy as np
    from convex_optimization_amd.core.loss import ssim
    g = np.load(os.path.join(golden_dir, "ssim_windows.npz"))
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    a4, b4 = torch.from_numpy(g["a4"]), torch.from_numpy(g["b4"])
    for ws in (1, 3, 7, 8, 15, 16):
        np.testing.assert_allclose(ssim(a, b, window_size=ws).numpy().reshape(-1), g[f"w{ws}"], rtol=2e-5)
        np.testing.assert_allclose(ssim(a * 255, b * 255, window_size=ws, max_val=255.0).numpy().reshape(-1), g[f"w{ws}_max255"], rtol=2e-5)
        np.testing.assert_allclose(ssim(a4, b4, window_size=ws).numpy().reshape(-1), g[f"w{ws}_batch"], rtol=2e-5)
        np.testing.assert_allclose(ssim(a.double(), b.double(), window_size=ws).numpy().reshape(-1), g[f"w{ws}"], rtol=2e-5)
    with pytest.raises(ValueError):
        ssim(a, b, window_size=0)


@pytest.mark.gpu
def test_ssim_record_is_filled_whatever_the_loss_kind(golden_dir):
    """ddn_plan_set_ssim: every *_run fills the caller's ssim record while one is set, also when the stop test is made
    on another loss kind (tol_kind = MSE here); primal DR's entry 1 (ssim(z1, b)) follows ddn_plan_set_ssim_cadence, its
    entry 0 (the pair (z1, z1prev)) exists only with tol_kind = SSIM.  Called through the C ABI."""
    import ctypes as C
    import numpy as np
    import convex_optimization_amd as pkg
    from convex_optimization_amd import _lib as L
    from convex_optimization_amd.core.loss import ssim
    from convex_optimization_amd.plan import _ptr
    from convex_optimization_amd.core.blur import gaussian_filter
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    H = W = 64
    n = 6
    kern = torch.from_numpy(gaussian_filter([5, 5], 0.8))
    b = torch.rand(1, H, W, device=dev)
    lib = L.lib()

    def ring2():
        return torch.full((n, 1, 2), float("nan"), dtype=torch.float64, device=dev)
    done = torch.zeros((1,), dtype=torch.int32, device=dev)
    ring = torch.zeros((n, 1, 8), dtype=torch.float64, device=dev)
    # primal-dual DR: entry 0 = ssim(clip(p), b) at every iteration
    ops = pkg.DeblurDenoiseOperators(kern, b[0], 3, 3, device=dev, batch=1)
    p0, q0 = torch.rand(1, H, W, device=dev), torch.rand(3, 1, H, W, device=dev)
    sr = ring2()
    ops.plan.set_ssim(sr)
    L.check(lib.ddn_pddr_run(ops.plan._h, _ptr(p0), _ptr(q0), _ptr(b), 3.0, 0.4, 0.03, n, L.LOSS_MSE, -1.0, 2, _ptr(ring), _ptr(done)))
    torch.cuda.synchronize()
    ops.plan.set_ssim(None)
    v = sr.cpu().numpy()
    assert np.isfinite(v[:, 0, 0]).all() and np.isnan(v[:, 0, 1]).all()
    np.testing.assert_allclose(v[n - 1, 0, 0], float(ssim(p0[0].clamp(0, 1), b[0])), rtol=1e-6)
    ops.close()
    # ADMM: entry 0 = ssim(sol, b), the look-ahead solution of every iteration
    ops = pkg.DeblurDenoiseOperators(kern, b[0], 1, 1, device=dev, batch=1)
    st = [torch.rand(1, H, W, device=dev), torch.rand(1, H, W, device=dev), torch.rand(3, 1, H, W, device=dev), torch.rand(3, 1, H, W, device=dev)]
    sol = torch.zeros(1, H, W, device=dev)
    sr = ring2()
    ops.plan.set_ssim(sr)
    L.check(lib.ddn_admm_run(ops.plan._h, _ptr(st[0]), _ptr(st[1]), _ptr(st[2]), _ptr(st[3]), _ptr(b), 18.0, 0.001, 0.5, 0, n, L.LOSS_NONE, -1.0, 2,
                             C.c_void_p(0), _ptr(done), _ptr(sol)))
    torch.cuda.synchronize()
    ops.plan.set_ssim(None)
    v = sr.cpu().numpy()
    assert np.isfinite(v[:, 0, 0]).all()
    np.testing.assert_allclose(v[n - 1, 0, 0], float(ssim(sol[0], b[0])), rtol=1e-6)
    ops.close()
    # primal DR: entry 1 = ssim(z1, b) at the set cadence; entry 0 stays empty unless ssim is the stop criterion
    ops = pkg.DeblurDenoiseOperators(kern, b[0], 2.0, device=dev, batch=1)
    for every, rows in ((1, list(range(n))), (4, [0, 4])):
        z1, z2 = b.clone(), torch.stack((b, b, b)).contiguous()
        sr = ring2()
        ops.plan.set_ssim(sr)
        L.check(lib.ddn_plan_set_ssim_cadence(ops.plan._h, every))
        L.check(lib.ddn_pdr_run(ops.plan._h, _ptr(z1), _ptr(z2), _ptr(b), 2.0, 0.1, 0.049, L.OBJ_DUALS_EQUAL, n, L.LOSS_MSE, -1.0, 3, _ptr(ring), _ptr(done)))
        torch.cuda.synchronize()
        ops.plan.set_ssim(None)
        v = sr.cpu().numpy()
        assert np.isnan(v[:, 0, 0]).all()
        assert [i for i in range(n) if np.isfinite(v[i, 0, 1])] == rows, (every, v[:, 0, 1])
        if every == 1:
            np.testing.assert_allclose(v[n - 1, 0, 1], float(ssim(z1[0], b[0])), rtol=1e-6)
        assert torch.equal(z2[1], z2[2])      # the equal-duals run hands both arrays back
    L.check(lib.ddn_plan_set_ssim_cadence(ops.plan._h, 50))
    ops.close()

