#!/usr/bin/env python3
"""Generate golden vectors by running the reference implementation itself.

TEST INFRASTRUCTURE ONLY.  Runs in the build container where /root/reference is
mounted; the GPU box never sees the reference, only the .npz files this script
writes under tests/golden/.

How the reference is imported (nothing from it is copied into this repository):
  * the package directory is copied to a scratch dir under /tmp because
    deblur_denoise/utils/logging_utils.py:10-15 creates `logs/` and `results/`
    next to itself at import time and /root/reference is read-only;
  * `torchvision` is absent from this image and is imported at module top by
    deblur_denoise/utils/conv_utils.py:10-13; stub modules are injected (only
    read_image uses them, and read_image is never called here).

fp64 runs (torch default dtype float64, float64 `b`) need two harness patches:
  * primal_dr/algorithm.py:146 always calls ssim whose window is hard-wired
    float32 (core/loss.py:75) -> the module-level `ssim` name is replaced by a
    stub returning 0;
  * admm/algorithm.py:49-54 hard-wires float32 init -> the module's `torch.randn`
    is wrapped so the same float32 draws are promoted to float64.

Usage:  python oracle/make_golden.py [--out tests/golden]
"""
import argparse
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"


def import_reference():
    scratch = tempfile.mkdtemp(prefix="refcopy_")
    shutil.copytree(os.path.join(REF, "deblur_denoise"), os.path.join(scratch, "deblur_denoise"))
    os.environ.setdefault("MPLBACKEND", "Agg")
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvio = types.ModuleType("torchvision.io")
    tvt.ToTensor = object
    tvt.functional = tvf
    tv.transforms = tvt
    tv.io = tvio
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvf, "torchvision.io": tvio})
    sys.path.insert(0, scratch)
    import logging
    import deblur_denoise  # noqa: F401
    logging.getLogger().setLevel(logging.ERROR)
    for name in list(logging.root.manager.loggerDict):
        lg = logging.getLogger(name)
        lg.setLevel(logging.ERROR)
        for h in lg.handlers:
            h.setLevel(logging.ERROR)
    return scratch


def to_np(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


class LossRecorder:
    """Wraps a reference loss function and records every value it returns."""

    def __init__(self, fn):
        self.fn = fn
        self.__name__ = fn.__name__
        self.values = []

    def __call__(self, x, y):
        v = self.fn(x, y)
        self.values.append(float(v))
        return v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default="", help="write only this fixture file (e.g. xtrans) and leave the others untouched")
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    scratch = import_reference()

    from deblur_denoise.core.blur import create_motion_blur_kernel, gaussian_filter
    from deblur_denoise.core.convolution import circular_convolve2d
    from deblur_denoise.core import proximal_operators as prox
    from deblur_denoise.core import loss as refloss
    from deblur_denoise.core.noise import add_gaussian_noise
    from deblur_denoise.op_math.python_code.multiplying_matrix import DeblurDenoiseOperators
    from deblur_denoise.op_math.python_code.eig_vals_for_periodic_conv_op import eig_vals_for_periodic_conv_op
    from deblur_denoise.chambolle_pock import algorithm as cp_mod
    from deblur_denoise.admm import algorithm as admm_mod
    from deblur_denoise.primal_dr import algorithm as pdr_mod
    from deblur_denoise.primal_dual_dr import algorithm as pddr_mod

    if args.only == "ssimwin":
        # ---------------------------------------------------------------- ssim with other window sizes (core/loss.py:39)
        sw = {}
        torch.manual_seed(9)
        a = torch.rand(48, 80)
        bb = torch.rand(48, 80)
        torch.manual_seed(10)
        a4 = torch.rand(2, 1, 20, 33)    # (the reference window is (1, 1, ws, ws) with groups = C: one channel only)
        b4 = (a4 + 0.1 * torch.randn(2, 1, 20, 33)).clamp(0, 1)
        sw["a"], sw["b"], sw["a4"], sw["b4"] = to_np(a), to_np(bb), to_np(a4), to_np(b4)
        for ws in (1, 3, 7, 8, 15, 16):
            sw[f"w{ws}"] = np.asarray(to_np(refloss.ssim(a, bb, window_size=ws)), dtype=np.float64).reshape(-1)
            sw[f"w{ws}_max255"] = np.asarray(to_np(refloss.ssim(a * 255, bb * 255, window_size=ws, max_val=255.0)), dtype=np.float64).reshape(-1)
            sw[f"w{ws}_batch"] = np.asarray(to_np(refloss.ssim(a4, b4, window_size=ws)), dtype=np.float64).reshape(-1)
        np.savez_compressed(os.path.join(out, "ssim_windows.npz"), **sw)
        shutil.rmtree(scratch, ignore_errors=True)
        print("wrote ssim_windows.npz to", out)
        return

    if args.only == "xtrans":
        # ---------------------------------------------------------------- apply_x_trans
        from deblur_denoise.op_math.python_code.apply_x_trans import apply_x_trans
        xt = {}
        for k in (3, 5, 9, 15):
            torch.manual_seed(40 + k)
            w, x = torch.rand(k, k), torch.rand(k, k)
            xt[f"w{k}"], xt[f"x{k}"] = to_np(w), to_np(x)
            xt[f"out{k}"] = to_np(apply_x_trans(w, x, k))
        np.savez_compressed(os.path.join(out, "xtrans.npz"), **xt)
        shutil.rmtree(scratch, ignore_errors=True)
        return

    # ------------------------------------------------------------------ kernels
    kernels = {
        "gauss5": gaussian_filter([5, 5], 0.8),
        "gauss9": gaussian_filter([9, 9], 0.8),
        "gauss9_s15": gaussian_filter([9, 9], 1.5),
        "gauss10": gaussian_filter([10, 10], 0.8),
        "motion15_45": create_motion_blur_kernel(15, 45),
        "motion15_0": create_motion_blur_kernel(15, 0),
        "motion5_30": create_motion_blur_kernel(5, 30),
        "motion8_120": create_motion_blur_kernel(8, 120),
    }
    np.savez_compressed(os.path.join(out, "kernels.npz"), **{k: to_np(v) for k, v in kernels.items()})

    # --------------------------------------------------------------- eigen arrays
    eig = {}
    for kname in ["gauss5", "gauss9", "gauss10", "motion15_45", "motion5_30"]:
        for (H, W) in [(32, 32), (64, 64), (48, 80)]:
            eig[f"{kname}_{H}x{W}_K"] = to_np(eig_vals_for_periodic_conv_op(kernels[kname], H, W))
    for (H, W) in [(32, 32), (64, 64), (48, 80)]:
        b = torch.zeros(H, W)
        for (t, s) in [(1, 1), (2, None), (3, 3)]:
            op = DeblurDenoiseOperators(kernels["gauss9"], b, t, s)
            eig[f"gauss9_{H}x{W}_t{t}_s{s}_M"] = to_np(op.eig_vals_mat)
        eig[f"{H}x{W}_D1"] = to_np(op.eig_arry_D1)
        eig[f"{H}x{W}_D2"] = to_np(op.eig_arry_D2)
    np.savez_compressed(os.path.join(out, "eigen.npz"), **eig)

    # ------------------------------------------------------- operator applications
    ops = {}
    for kname in ["gauss9", "motion15_45", "gauss10"]:
        for (H, W) in [(64, 64), (48, 80)]:
            torch.manual_seed(11)
            x = torch.rand(H, W)
            y2 = torch.rand(H, W, 2)
            op = DeblurDenoiseOperators(kernels[kname], x, 2.0, 0.5)
            tag = f"{kname}_{H}x{W}"
            ops[tag + "_x"] = to_np(x)
            ops[tag + "_y2"] = to_np(y2)
            ops[tag + "_K"] = to_np(op.apply_K(x))
            ops[tag + "_KT"] = to_np(op.apply_KTrans(x))
            ops[tag + "_D1"] = to_np(op.apply_D1(x))
            ops[tag + "_D1T"] = to_np(op.apply_D1Trans(x))
            ops[tag + "_D"] = to_np(op.apply_D(x))
            ops[tag + "_DT"] = to_np(op.apply_DTrans(y2))
            ops[tag + "_inv"] = to_np(op.invert_matrix(x))
            ops[tag + "_Mat"] = to_np(op.apply_Mat(x))
            ops[tag + "_eigM"] = to_np(op.eig_vals_mat)
    np.savez_compressed(os.path.join(out, "operators.npz"), **ops)

    # ----------------------------------------------------------------------- prox
    px = {}
    torch.manual_seed(5)
    x = torch.randn(3, 40, 24)
    x2 = torch.randn(2, 40, 24)
    px["x"] = to_np(x)
    px["x2"] = to_np(x2)
    for i, lam in enumerate([0.05, 1.0 / 18.0, 2.0]):
        px[f"lam{i}"] = np.float64(lam)
        px[f"l1_{i}"] = to_np(prox.prox_l1(x, lam))
        px[f"l2_{i}"] = to_np(prox.prox_l2(x, lam))
        px[f"l2sq_{i}"] = to_np(prox.prox_l2_squared(x, lam))
        px[f"iso_{i}"] = to_np(prox.prox_iso(x2, lam))
        px[f"box_{i}"] = to_np(prox.prox_box(x, lam))
    # edge cases: zeros, |x| == lambda exactly, norm == lambda exactly, outside [0,1]
    e = torch.tensor([[0.0, 0.5, -0.5, 2.0, -2.0, 1.0, 0.25, -0.25]])
    px["edge"] = to_np(e)
    px["edge_l1"] = to_np(prox.prox_l1(e, 0.5))
    px["edge_box"] = to_np(prox.prox_box(e, 0.5))
    px["edge_l2sq"] = to_np(prox.prox_l2_squared(e, 0.5))
    e2 = torch.tensor([[[0.0, 3.0, 0.3, 0.0, 5.0]], [[0.0, 4.0, 0.4, 5.0, 12.0]]])  # norms 0,5,0.5,5,13
    px["edge2"] = to_np(e2)
    px["edge2_iso5"] = to_np(prox.prox_iso(e2, 5.0))
    px["edge2_iso05"] = to_np(prox.prox_iso(e2, 0.5))
    big = torch.tensor([3.0, 4.0])
    px["edge_l2_eq"] = to_np(prox.prox_l2(big, 5.0))
    px["edge_l2_gt"] = to_np(prox.prox_l2(big, 2.5))
    np.savez_compressed(os.path.join(out, "prox.npz"), **px)

    # --------------------------------------------------------- circular_convolve2d
    cc = {}
    for kname in ["gauss9", "gauss10", "motion15_45", "motion5_30"]:
        for (H, W) in [(32, 32), (48, 80)]:
            delta = torch.zeros(1, H, W)
            delta[0, 3, 5] = 1.0
            torch.manual_seed(3)
            img = torch.rand(2, H, W)
            tag = f"{kname}_{H}x{W}"
            cc[tag + "_delta"] = to_np(circular_convolve2d(delta, kernels[kname]))
            cc[tag + "_img"] = to_np(img)
            cc[tag + "_out"] = to_np(circular_convolve2d(img, kernels[kname]))
    np.savez_compressed(os.path.join(out, "circconv.npz"), **cc)

    # ----------------------------------------------------------------------- loss
    ls = {}
    torch.manual_seed(9)
    a = torch.rand(48, 80)
    bb = torch.rand(48, 80)
    ls["a"], ls["b"] = to_np(a), to_np(bb)
    for fn in ["l1_loss", "l2_loss", "mse", "psnr", "ssim"]:
        ls[fn] = np.asarray(to_np(getattr(refloss, fn)(a, bb)), dtype=np.float64).reshape(-1)
    ls["psnr_same"] = np.asarray([float(refloss.psnr(a, a))])
    np.savez_compressed(os.path.join(out, "loss.npz"), **ls)

    # ---------------------------------------------------------------- trajectories
    def make_problem(H, W, kname, seed, noise):
        torch.manual_seed(seed)
        img = torch.rand(1, H, W)
        b = circular_convolve2d(img, kernels[kname])
        if noise > 0:
            b = add_gaussian_noise(b, 0.0, noise)
        return img, b.to(torch.get_default_dtype())

    SEED = 7
    real_randn = torch.randn

    def run_all(dtype_name, H, W, kname, Ns_long, Ns_cp):
        res = {}
        kern = kernels[kname]
        img, b1 = make_problem(H, W, kname, 0, 0.05)
        b = b1.squeeze(0)
        res["b"] = to_np(b)
        res["kernel"] = to_np(kern)
        op1 = DeblurDenoiseOperators(kern, b, 1, 1)
        res["eigK"] = to_np(op1.eig_arry_K)
        res["eigD"] = to_np(op1.eig_arry_D1)
        res["M_t1_s1"] = to_np(op1.eig_vals_mat)

        # Chambolle-Pock: state drawn as rand(H,W), rand(H,W), rand(3,H,W)  (cp:50-52)
        torch.manual_seed(SEED)
        res["cp_x0"] = to_np(torch.rand(H, W)); res["cp_z0"] = to_np(torch.rand(H, W)); res["cp_y0"] = to_np(torch.rand(3, H, W))
        for label, kw in [("cpdef_l1", dict()), ("cpstab_l1", dict(t=0.3, s=0.3)), ("cpstab_l2", dict(t=0.3, s=0.3, objective_function="l2"))]:
            for N in (Ns_cp if label.startswith("cpdef") else Ns_long):
                torch.manual_seed(SEED)
                rec = LossRecorder(refloss.mse)
                x = cp_mod.chambolle_pock(b1, kern, max_iter=N + 1, loss_function=rec, **kw)
                res[f"{label}_N{N}"] = to_np(x)
                if N == 20:
                    res[f"{label}_N{N}_mse"] = np.asarray(rec.values)

        # ADMM: clamp(randn) x5: x,u,w (H,W); y,z (3,H,W)  (admm:49-54)
        torch.manual_seed(SEED)
        draws = [real_randn((H, W), dtype=torch.float32) for _ in range(3)] + [real_randn((3, H, W), dtype=torch.float32) for _ in range(2)]
        for nm, d in zip(["x0", "u0", "w0", "y0", "z0"], draws):
            res["admm_" + nm] = to_np(torch.clamp(d, 0, 1))
        for obj in ["l1", "l2"]:
            for N in Ns_long:
                torch.manual_seed(SEED)
                rec = LossRecorder(refloss.mse)
                x = admm_mod.admm_solver(b, objective_function=obj, kernel=kern, niters=N + 1, tol=-1.0, loss_function=rec)
                res[f"admm_{obj}_N{N}"] = to_np(x)
                if N == 20:
                    res[f"admm_{obj}_N{N}_mse"] = np.asarray(rec.values)
        res["M_pdr_t2"] = to_np(DeblurDenoiseOperators(kern, b, 2.0).eig_vals_mat)
        res["M_pddr_t3"] = to_np(DeblurDenoiseOperators(kern, b, 3, 3).eig_vals_mat)

        # primal DR (deterministic init)
        for prob in ["l1", "l2"]:
            for N in Ns_long:
                rec = LossRecorder(refloss.mse)
                x = pdr_mod.primal_dr_splitting(prob, kern, b, niters=N, loss_function=rec, tol=-1.0)
                res[f"pdr_{prob}_N{N}"] = to_np(x)
                if N == 20:
                    res[f"pdr_{prob}_N{N}_mse"] = np.asarray(rec.values)

        # primal-dual DR: p = rand(H,W), q = rand(3,H,W) (pddr:57-58)
        torch.manual_seed(SEED)
        res["pddr_p0"] = to_np(torch.rand(H, W)); res["pddr_q0"] = to_np(torch.rand(3, H, W))
        for N in Ns_long:
            torch.manual_seed(SEED)
            rec = LossRecorder(refloss.mse)
            x = pddr_mod.primal_dual_dr_splitting(b, kern, niters=N + 1, loss_function=rec, tol=-1.0)
            res[f"pddr_N{N}"] = to_np(x)
            if N == 20:
                res[f"pddr_N{N}_mse"] = np.asarray(rec.values)
        return res

    Ns_long = [1, 2, 5, 20, 100, 500]
    Ns_cp = [1, 2, 5, 20, 100]
    # fp32
    torch.set_default_dtype(torch.float32)
    for kname in ["gauss9", "motion15_45"]:
        r = run_all("f32", 64, 64, kname, Ns_long, Ns_cp)
        np.savez_compressed(os.path.join(out, f"traj_f32_{kname}_64x64.npz"), **r)
    r = run_all("f32", 48, 80, "gauss5", [1, 5, 20], [1, 5, 20])
    np.savez_compressed(os.path.join(out, "traj_f32_gauss5_48x80.npz"), **r)

    # fp64 (with the two harness patches described in the module docstring)
    torch.set_default_dtype(torch.float64)
    pdr_mod.ssim = lambda x, y, *a, **k: torch.tensor(0.0)

    class _TorchProxy:
        def __getattr__(self, n):
            return getattr(torch, n)

        @staticmethod
        def randn(*a, **k):
            return real_randn(*a, **k).double()

    admm_mod.torch = _TorchProxy()
    for kname in ["gauss9", "motion15_45"]:
        r = run_all("f64", 64, 64, kname, Ns_long, Ns_cp)
        np.savez_compressed(os.path.join(out, f"traj_f64_{kname}_64x64.npz"), **r)
    torch.set_default_dtype(torch.float32)
    admm_mod.torch = torch

    shutil.rmtree(scratch, ignore_errors=True)
    tot = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
    print(f"wrote golden vectors to {out}: {tot/1e6:.2f} MB")


if __name__ == "__main__":
    main()
