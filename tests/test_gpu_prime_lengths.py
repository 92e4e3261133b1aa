"""Image sizes with prime factors above 13 (the reference's torch.fft takes any length: 2-D transforms of 34 x 38 or
97 x 101 images just work there).  Those factors run in a runtime-radix pass (csrc/ddn_fft.hpp, prime_pass); all four
solvers against the CPU oracle, fp32 and fp64, and the documented limit: a prime factor above 8191 is refused."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ddn_oracle as O  # checker only
from test_gpu_long_rows import _all_four, _problem, rel, pkg, _fresh_plans  # noqa: F401  (fixtures)


@pytest.mark.parametrize("H,W,kname", [(34, 38, "gauss9"), (51, 46, "motion15_45"), (97, 101, "gauss5"), (64, 2 * 127, "gauss9"), (17 * 19, 32, "gauss9")])
def test_sizes_with_large_prime_factors_fp32(pkg, golden_dir, H, W, kname):
    kern, b, stC, stA, stP = _problem(golden_dir, H, W, kname, torch.float32)
    for N in (10, 40):
        res = _all_four(pkg, kern, b, stC, stA, stP, N)
        bad = {k: v for k, v in res.items() if not v < 1e-5}
        assert not bad, (N, res)


def test_sizes_with_large_prime_factors_fp64(pkg, golden_dir):
    """fp64 with the oracle's own eigen arrays injected (the route of the other fp64 parity tests): <= 1e-11."""
    H, W = 46, 34
    kern, b, stC, stA, stP = _problem(golden_dir, H, W, "gauss9", torch.float64)

    def eig(t, s):
        ops = O.DeblurDenoiseOperators(kern, b, t, s)
        return {"K": ops.eig_arry_K, "D": ops.eig_arry_D1, "M": ops.eig_vals_mat}

    res = _all_four(pkg, kern, b, stC, stA, stP, 20, eig=eig)
    bad = {k: v for k, v in res.items() if not v < 1e-11}
    assert not bad, res


# The runtime-radix pass gives every thread at most two outputs of a round, and a round is at least one butterfly: a
# radix above 2 x 64 needs more than the smallest workgroup.  Widths and heights whose launch shapes used to come
# with 64 threads: W = 2 * 251 and W = 509 (scalar-access rows), W = 4 * 257 (16-byte-access rows), H = 263 (column
# tiles narrowed to one column), and radices beyond 509 (1021 rows, 2 * 523 columns).
@pytest.mark.parametrize("H,W", [(16, 2 * 251), (16, 509), (16, 4 * 257), (263, 16), (16, 1021), (2 * 523, 16)])
def test_radices_above_two_wavefronts_fp32(pkg, golden_dir, H, W):
    kern, b, stC, stA, stP = _problem(golden_dir, H, W, "gauss5", torch.float32)
    res = _all_four(pkg, kern, b, stC, stA, stP, 6)
    bad = {k: v for k, v in res.items() if not v < 1e-5}
    assert not bad, res


@pytest.mark.parametrize("H,W", [(16, 2 * 251), (263, 16), (16, 4 * 257)])
def test_radices_above_two_wavefronts_fp64(pkg, golden_dir, H, W):
    """float64 with the oracle's eigen arrays injected.  The 1e-11 contract needs the reference's D eigen array to have
    identical rows (this package applies D row by row from ONE row of it): that holds whenever torch's float32 transform
    along the image columns is exact for an impulse, which it is not at heights with a large prime factor (a chirp-z
    transform there: the rows differ in the last float32 bits, and the solvers then agree to that, 1e-6).  The column
    transform itself is held to 1e-11 through K, K^T and the resolvent, which involve no D."""
    kern, b, stC, stA, stP = _problem(golden_dir, H, W, "gauss5", torch.float64)
    oo = O.DeblurDenoiseOperators(kern, b, 2.0, 0.5)
    ea = {"K": oo.eig_arry_K, "D": oo.eig_arry_D1, "M": oo.eig_vals_mat}
    ops = pkg.DeblurDenoiseOperators(kern, b, 2.0, 0.5, dtype=torch.float64, eig_arrays=ea)
    x = torch.rand(H, W, dtype=torch.float64)
    for nm in ("apply_K", "apply_KTrans", "invert_matrix"):
        got, want = getattr(ops, nm)(x).real.numpy(), getattr(oo, nm)(x).real.numpy()
        assert np.abs(got - want).max() < 1e-11, nm
    ops.close()
    D = oo.eig_arry_D1
    rows_identical = bool(torch.equal(D, D[0:1].expand_as(D)))
    assert rows_identical == (H != 263)

    def eig(t, s):
        o2 = O.DeblurDenoiseOperators(kern, b, t, s)
        return {"K": o2.eig_arry_K, "D": o2.eig_arry_D1, "M": o2.eig_vals_mat}

    res = _all_four(pkg, kern, b, stC, stA, stP, 6, eig=eig)
    bad = {k: v for k, v in res.items() if not v < (1e-11 if rows_identical else 1e-6)}
    assert not bad, res


# Radices above 2 x 1024: eight outputs per thread and round, float sums accumulated in double.  16 x 4099 and 16 x 2053
# are rows of prime length (scalar-access row kernels with 576 / 1024 threads), 2063 x 16 a prime height that only fits
# the LDS with one-column tiles, 8191 the largest radix there is.
@pytest.mark.parametrize("H,W", [(16, 4099), (16, 2053), (2063, 16), (16, 8191)])
def test_radices_above_two_thousand_fp32(pkg, golden_dir, H, W):
    kern, b, stC, stA, stP = _problem(golden_dir, H, W, "gauss5", torch.float32)
    res = _all_four(pkg, kern, b, stC, stA, stP, 4)
    bad = {k: v for k, v in res.items() if not v < 1e-5}
    assert not bad, res


def test_radix_above_two_thousand_fp64(pkg, golden_dir):
    H, W = 16, 2053
    kern, b, stC, stA, stP = _problem(golden_dir, H, W, "gauss5", torch.float64)

    def eig(t, s):
        o2 = O.DeblurDenoiseOperators(kern, b, t, s)
        return {"K": o2.eig_arry_K, "D": o2.eig_arry_D1, "M": o2.eig_vals_mat}

    res = _all_four(pkg, kern, b, stC, stA, stP, 4, eig=eig)
    bad = {k: v for k, v in res.items() if not v < 1e-11}
    assert not bad, res


def test_prime_factor_above_the_limit_is_refused(pkg):
    """8209 is the first prime above the limit: plan creation fails with a message, nothing silently falls back."""
    from convex_optimization_amd._lib import DdnError
    b = torch.rand(8209, 16)
    with pytest.raises(DdnError, match="prime factor"):
        pkg.primal_dr_splitting("l1", torch.ones(3, 3) / 9, b, niters=2)
