"""Randomised parity sweep: random image shapes from every kernel family's territory, random blur kernels, all four
solvers for a few iterations against the CPU oracle (fp32, rel-L2 <= 1e-5).  Prints one line per case and a summary;
exit status 1 when a case fails.  With a third argument "f64": float64 images with the eigenvalue arrays the reference
would hold (float32-derived complex64) handed in, rel-L2 <= 1e-11.  usage: tools_fuzz_parity.py [cases] [seed] [f64|f32] [mid]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import convex_optimization_amd as pkg
from oracle import ddn_oracle as O   # checker only

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f64 = len(sys.argv) > 3 and sys.argv[3] == "f64"
dt = torch.float64 if f64 else torch.float32
TOL = 1e-11 if f64 else 1e-5
rng = np.random.default_rng(seed)
small = [16, 24, 32, 34, 48, 51, 56, 64, 77, 80, 91, 96, 100, 104, 128, 143, 200, 256, 300, 500, 512, 1000, 1024]   # (56 ... 143: radices 7, 11, 13)
large = [2048, 4096] if f64 else [2048, 4096, 8192, 16384, 32768]
# common photo and video sizes: compile-time mixed-radix rows and columns, split columns from 2560 rows on; paired with
# each other (images with several rounds of row workgroups) when the fourth argument is "mid"
mid = [480, 600, 640, 720, 768, 800, 900, 960, 1000, 1080, 1200, 1280, 1440, 1536, 1600, 1920, 2000, 2160, 2400, 2560, 2880, 3072, 3200, 3600, 3840,
       3000, 3024, 4000, 4032, 4320, 4608, 5120, 5184, 5472, 6000, 6144, 7680, 8000]
mid_mode = len(sys.argv) > 4 and sys.argv[4] == "mid"
fails = []
for case in range(ncases):
    u = rng.random()
    if mid_mode:
        H, W = int(rng.choice(mid)), int(rng.choice(mid))
        if u < 0.3:
            H = int(rng.choice([16, 48, 300, 500]))
        elif u < 0.6:
            W = int(rng.choice([32, 64, 300, 500]))
    elif u < 0.45:
        a, b_ = int(rng.choice(large)), int(rng.choice([16, 32, 48, 64]))
        H, W = (a, b_) if rng.random() < 0.5 else (b_, a)
    else:
        H, W = int(rng.choice(small)), int(rng.choice(small))
    if H * W > ((9 << 19) if mid_mode else (1 << 20)):
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
    img = torch.rand(1, H, W, dtype=dt)
    b = O.add_gaussian_noise(O.circular_convolve2d(img, kern), 0.0, 0.05)[0].to(dt)
    N = int(rng.choice([3, 6, 12]))
    obj = str(rng.choice(["l1", "l2"]))
    stC = (torch.rand(H, W, dtype=dt), torch.rand(H, W, dtype=dt), torch.rand(3, H, W, dtype=dt))
    stA = tuple(torch.clamp(torch.randn(s), 0, 1).to(dt) for s in [(H, W), (H, W), (3, H, W), (3, H, W)])
    stP = (torch.rand(H, W, dtype=dt), torch.rand(3, H, W, dtype=dt))

    def eig(t, s_):
        if not f64:
            return None
        ops = O.DeblurDenoiseOperators(kern, b, t, s_)
        return {"K": ops.eig_arry_K, "D": ops.eig_arry_D1, "M": ops.eig_vals_mat}
    t0 = time.time()
    try:
        res = {
            "cp": (pkg.chambolle_pock_solver(b[None], kern, obj, t=0.3, s=0.3, max_iter=N + 1, init_state=stC, eig_arrays=eig(1, 1)), O.chambolle_pock(b[None], kern, obj, t=0.3, s=0.3, max_iter=N + 1, init_state=stC)),
            "admm": (pkg.admm_solver(b, obj, kernel=kern, niters=N + 1, tol=-1.0, init_state=stA, eig_arrays=eig(1, 1)), O.admm_solver(b, obj, kernel=kern, niters=N + 1, tol=-1.0, init_state=stA)),
            "pdr": (pkg.primal_dr_splitting(obj, kern, b, niters=N, loss_function=pkg.core.mse, tol=-1.0, eig_arrays=eig(2.0, None)), O.primal_dr_splitting(obj, kern, b, niters=N)),
            "pddr": (pkg.primal_dual_dr_splitting(b, kern, niters=N + 1, init_state=stP, tol=-1.0, eig_arrays=eig(3, 3)), O.primal_dual_dr_splitting(b, kern, niters=N + 1, init_state=stP)),
        }
        errs = {k_: O.rel_l2(g.cpu().numpy(), r.numpy()) for k_, (g, r) in res.items()}
        bad = {k_: v for k_, v in errs.items() if not v < TOL}
    except Exception as e:      # a refused size is a finding too
        errs, bad = {"error": repr(e)[:200]}, {"error": 1}
    line = f"case {case:3d} {H:5d} x {W:5d} {ktype}{tuple(kern.shape)} {obj} N={N:2d} " + " ".join(f"{k_} {v:.1e}" if isinstance(v, float) else f"{k_} {v}" for k_, v in errs.items()) + f"  [{time.time() - t0:.1f}s]"
    print(line + ("   <-- FAIL" if bad else ""), flush=True)
    if bad:
        fails.append(line)
print(f"{len(fails)} failing case(s)")
sys.exit(1 if fails else 0)
