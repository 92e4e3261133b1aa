"""Diagnostic: float64 parity (injected eigen arrays) at heights with / without large prime factors, odd and even."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import convex_optimization_amd as pkg
from oracle import ddn_oracle as O
from test_gpu_long_rows import _problem, _all_four
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for H, W in [(263, 16), (127, 16), (131, 16), (262, 16), (46, 16), (45, 16), (16, 263), (16, 127), (127, 32), (255, 16), (257, 16)]:
    kern, b, stC, stA, stP = _problem(g, H, W, "gauss5", torch.float64)
    def eig(t, s):
        ops = O.DeblurDenoiseOperators(kern, b, t, s)
        return {"K": ops.eig_arry_K, "D": ops.eig_arry_D1, "M": ops.eig_vals_mat}
    res = _all_four(pkg, kern, b, stC, stA, stP, 6, eig=eig)
    print(H, W, {k: f"{v:.1e}" for k, v in res.items()}, flush=True)
