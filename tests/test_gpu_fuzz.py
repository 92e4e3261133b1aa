"""A short fixed-seed run of the randomised parity sweep (tools/tools_fuzz_parity.py): random shapes from every
kernel family's territory, random blur kernels and objectives, all four solvers for a few iterations against the CPU
oracle, fp32 rel-L2 <= 1e-5."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed,mode", [(11, "f32"), (12, "f32"), (13, "f64")])
def test_random_shapes_all_solvers_vs_oracle(seed, mode):
    """mode f64: float64 images with the reference's complex64-derived eigenvalue arrays handed in, rel-L2 <= 1e-11."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tools_fuzz_parity.py"), "16", str(seed), mode], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 failing case(s)" in r.stdout
