import os, sys, ctypes as C
os.environ["DDN_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import convex_optimization_amd as pkg
from convex_optimization_amd import _lib as L
from convex_optimization_amd.plan import _ptr
from convex_optimization_amd.core.blur import create_motion_blur_kernel
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
kern = create_motion_blur_kernel(15, 45)
dt = torch.float64 if os.environ.get("DDN_STAMPS_DTYPE") == "f64" else torch.float32
b = torch.rand(1, H, W, device=dev, dtype=dt)
ops = pkg.DeblurDenoiseOperators(kern, b[0], 2.0, device=dev, batch=1, dtype=dt)
st = [b.clone(), torch.stack((b, b, b)).contiguous()]
lib = L.lib()
lib.ddn_plan_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
L.check(lib.ddn_pdr_run(ops.plan._h, _ptr(st[0]), _ptr(st[1]), _ptr(b), 2.0, 0.1, 0.049, 0, 4, 0, -1.0, 3, C.c_void_p(0), C.c_void_p(0)))
torch.cuda.synchronize()
n = 1024
buf = np.zeros((n, 16), dtype=np.int64)
L.check(lib.ddn_plan_debug_stamps(ops.plan._h, buf.ctypes.data_as(C.c_void_p), n))
d = np.diff(buf[:, :9], axis=1).astype(np.float64)
ok = buf[:, 8] > 0
print("workgroups with stamps:", ok.sum())
names = ["pack", "ifft", "uprev", "elementwise", "reduce", "fixup", "fft", "unpack"]
for i, nm in enumerate(names):
    print(f"{nm:12s} median {np.median(d[ok, i]):10.0f} cycles  mean {d[ok, i].mean():10.0f}")
print("total median", np.median(buf[ok, 8] - buf[ok, 0]))
