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
