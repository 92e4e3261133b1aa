"""ctypes binding of libddn (include/ddn.h).  There is no CPU fallback: importing any
compute entry point without the compiled HIP library raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DDN_LIB_PATH: another build of the same library (A/B measurements of kernel variants)
_LIB_PATH = os.environ.get("DDN_LIB_PATH") or os.path.join(_HERE, "libddn.so")

F32, F64 = 0, 1
C64, C128 = 0, 1
OBJ = {"l1": 0, "l2": 1}
OBJ_DUALS_EQUAL = 0x100   # ddn_pdr_run: the two difference duals start equal (ddn.h)
LOSS_NONE, LOSS_L1, LOSS_L2, LOSS_MSE, LOSS_PSNR, LOSS_SSIM = 0, 1, 2, 3, 4, 5
OP_K, OP_KT, OP_D, OP_DT, OP_RESOLVENT, OP_MAT = 0, 1, 2, 3, 4, 5
PROX_L1, PROX_L2SQ, PROX_ISO, PROX_BOX, PROX_L2 = 0, 1, 2, 3, 4

_lib = None


class DdnError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' at the repository root). "
            "There is no CPU fallback.")
    L = C.CDLL(_LIB_PATH)
    vp, i, d, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
    L.ddn_version.restype = i
    L.ddn_last_error.restype = C.c_char_p
    sigs = {
        "ddn_plan_create": [C.POINTER(vp), i, i, i, i, vp],
        "ddn_plan_destroy": [vp],
        "ddn_plan_set_stream": [vp, vp],
        "ddn_plan_tune": [vp, i, i, i],
        "ddn_plan_set_kernel": [vp, C.POINTER(d), i, i],
        "ddn_plan_set_eig": [vp, vp, vp, i],
        "ddn_plan_set_resolvent": [vp, d, d, vp, i],
        "ddn_plan_get_eigK": [vp, vp],
        "ddn_plan_set_ssim": [vp, C.POINTER(C.c_float), d, vp],
        "ddn_plan_set_ssim_cadence": [vp, i],
        "ddn_apply": [vp, i, vp, vp],
        "ddn_apply_multiplier": [vp, vp, i, i, vp, vp],
        "ddn_circular_convolve2d": [vp, C.POINTER(d), i, i, vp, vp],
        "ddn_prox": [i, i, vp, vp, sz, d, vp],
        "ddn_reduce": [i, vp, vp, sz, i, vp, vp],
        "ddn_reduce_static": [i, vp, vp, sz, i, vp, vp],
        "ddn_ssim": [i, vp, vp, i, i, i, d, C.POINTER(C.c_float), vp, vp],
        "ddn_ssim_window": [i, vp, vp, i, i, i, d, C.POINTER(C.c_float), i, vp, vp],
        "ddn_apply_x_trans": [i, vp, vp, i, vp, vp],
        "ddn_cp_run": [vp, vp, vp, vp, vp, d, d, d, i, i, vp],
        "ddn_admm_run": [vp, vp, vp, vp, vp, vp, d, d, d, i, i, i, d, i, vp, vp, vp],
        "ddn_pdr_run": [vp, vp, vp, vp, d, d, d, i, i, i, d, i, vp, vp],
        "ddn_pddr_run": [vp, vp, vp, vp, d, d, d, i, i, d, i, vp, vp],
        "ddn_plan_enable_timing": [vp, i],
        "ddn_slab_layout": [vp, C.POINTER(i), C.POINTER(i)],
        "ddn_plan_create_slab": [C.POINTER(vp), i, i, i, vp],
        "ddn_slab_cols_part": [vp, vp, i, i, i, i, i, i],
        "ddn_slab_restrict": [vp, i, i],
        "ddn_slab_rows": [vp, i, i, vp, vp, vp, vp, vp, vp, d, d, d, i, i, i, vp, i],
        "ddn_slab_cols": [vp, vp, i, i, i],
        "ddn_slab_loss": [vp, i, vp],
        "ddn_slab_begin": [vp],
        "ddn_slab_loss_sums": [vp, i, vp],
        "ddn_slab_stop_test": [vp, vp, i, d, i, i, vp],
        "ddn_slab_iters_done": [vp, vp],
        "ddn_slab_rows_cp": [vp, i, vp, vp, vp, vp, vp, vp, d, d, d, i, i, i, i],
        "ddn_slab_cols_multiply": [vp, vp, i, i, i, i],
        "ddn_plan_last_timing": [vp, C.POINTER(d), C.POINTER(d), C.POINTER(i), C.POINTER(i)],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        msg = lib().ddn_last_error().decode("utf-8", "replace")
        if "Invalid objective function" in msg:
            raise ValueError(msg)
        raise DdnError(f"libddn error {rc}: {msg}")


EXPORTED = ["ddn_version", "ddn_last_error", "ddn_plan_create", "ddn_plan_destroy", "ddn_plan_set_stream", "ddn_plan_tune",
            "ddn_plan_set_kernel", "ddn_plan_set_eig", "ddn_plan_set_resolvent", "ddn_plan_set_ssim", "ddn_plan_set_ssim_cadence", "ddn_plan_get_eigK", "ddn_apply",
            "ddn_apply_multiplier", "ddn_circular_convolve2d", "ddn_prox", "ddn_reduce", "ddn_reduce_static", "ddn_ssim", "ddn_ssim_window", "ddn_apply_x_trans", "ddn_cp_run", "ddn_admm_run",
            "ddn_pdr_run", "ddn_pddr_run", "ddn_plan_last_timing", "ddn_plan_enable_timing", "ddn_plan_debug_stamps",
            "ddn_slab_layout", "ddn_slab_cols_part", "ddn_plan_create_slab", "ddn_slab_restrict", "ddn_slab_rows", "ddn_slab_cols", "ddn_slab_loss", "ddn_slab_begin", "ddn_slab_loss_sums", "ddn_slab_stop_test", "ddn_slab_iters_done", "ddn_slab_rows_cp", "ddn_slab_cols_multiply"]
