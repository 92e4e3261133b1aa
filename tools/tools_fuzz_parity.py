"""Randomised parity sweep: random image shapes from every kernel family's territory, random blur kernels, all four
solvers for a few iterations against the CPU oracle (fp32, rel-L2 <= 1e-5).  Prints one line per case and a summary;
exit status 1 when a case fails.  usage: tools_fuzz_parity.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import convex_optimization_amd as pkg
from oracle import ddn_oracle as O   # checker only

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
small = [16, 24, 32, 34, 48, 51, 64, 80, 96, 100, 128, 200, 256, 300, 500, 512, 1000, 1024]
large = [2048, 4096, 8192, 16384, 32768]
fails = []
for case in range(ncases):
    if rng.random() < 0.45:
        a, b_ = int(rng.choice(large)), int(rng.choice([16, 32, 48, 64]))
        H, W = (a, b_) if rng.random() < 0.5 else (b_, a)
    else:
        H, W = int(rng.choice(small)), int(rng.choice(small))
    if H * W > (1 << 20):
        continue
    ktype = rng.choice(["gauss", "motion"])
    if ktype == "gauss":
        k = int(rng.choice([3, 5, 9, 10]))
        kern = torch.from_numpy(O.gaussian_filter([k, k], float(rng.choice([0.8, 1.5]))))
    else:
        kern = O.create_motion_blur_kernel(int(rng.choice([5, 9, 15])), float(rng.choice([0, 30, 45])))
    if kern.shape[0] > min(H, W):
        continue
    torch.manual_seed(int(rng.integers(1 << 30)))
    img = torch.rand(1, H, W)
    b = O.add_gaussian_noise(O.circular_convolve2d(img, kern), 0.0, 0.05)[0]
    N = int(rng.choice([3, 6, 12]))
    obj = str(rng.choice(["l1", "l2"]))
    stC = (torch.rand(H, W), torch.rand(H, W), torch.rand(3, H, W))
    stA = tuple(torch.clamp(torch.randn(s), 0, 1) for s in [(H, W), (H, W), (3, H, W), (3, H, W)])
    stP = (torch.rand(H, W), torch.rand(3, H, W))
    t0 = time.time()
    try:
        res = {
            "cp": (pkg.chambolle_pock_solver(b[None], kern, obj, t=0.3, s=0.3, max_iter=N + 1, init_state=stC), O.chambolle_pock(b[None], kern, obj, t=0.3, s=0.3, max_iter=N + 1, init_state=stC)),
            "admm": (pkg.admm_solver(b, obj, kernel=kern, niters=N + 1, tol=-1.0, init_state=stA), O.admm_solver(b, obj, kernel=kern, niters=N + 1, tol=-1.0, init_state=stA)),
            "pdr": (pkg.primal_dr_splitting(obj, kern, b, niters=N, loss_function=pkg.core.mse, tol=-1.0), O.primal_dr_splitting(obj, kern, b, niters=N)),
            "pddr": (pkg.primal_dual_dr_splitting(b, kern, niters=N + 1, init_state=stP, tol=-1.0), O.primal_dual_dr_splitting(b, kern, niters=N + 1, init_state=stP)),
        }
        errs = {k_: O.rel_l2(g.cpu().numpy(), r.numpy()) for k_, (g, r) in res.items()}
        bad = {k_: v for k_, v in errs.items() if not v < 1e-5}
    except Exception as e:      # a refused size is a finding too
        errs, bad = {"error": repr(e)[:200]}, {"error": 1}
    line = f"case {case:3d} {H:5d} x {W:5d} {ktype}{tuple(kern.shape)} {obj} N={N:2d} " + " ".join(f"{k_} {v:.1e}" if isinstance(v, float) else f"{k_} {v}" for k_, v in errs.items()) + f"  [{time.time() - t0:.1f}s]"
    print(line + ("   <-- FAIL" if bad else ""), flush=True)
    if bad:
        fails.append(line)
print(f"{len(fails)} failing case(s)")
sys.exit(1 if fails else 0)
