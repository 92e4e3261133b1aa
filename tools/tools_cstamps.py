import os, sys, ctypes as C
os.environ["DDN_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import convex_optimization_amd as pkg
from convex_optimization_amd import _lib as L
from convex_optimization_amd.plan import _ptr
from convex_optimization_amd.core.blur import create_motion_blur_kernel
H = W = 4096
dev = torch.device("cuda", 0)
kern = create_motion_blur_kernel(15, 45)
b = torch.rand(1, H, W, device=dev)
ops = pkg.DeblurDenoiseOperators(kern, b[0], 2.0, device=dev, batch=1)
st = [b.clone(), torch.stack((b, b, b)).contiguous()]
lib = L.lib()
lib.ddn_plan_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
if os.environ.get("DDN_STAMPS_SOLVER") == "cp":
    # multiplier mode (Chambolle-Pock): four-wave workgroups, two per CU, one round
    ops.close()
    ops = pkg.DeblurDenoiseOperators(kern, b[0], 1, 1, device=dev, batch=1)
    stc = [torch.rand(1, H, W, device=dev), torch.rand(1, H, W, device=dev), torch.rand(3, 1, H, W, device=dev)]
    ring = torch.zeros((8, 1, 8), dtype=torch.float64, device=dev)
    L.check(lib.ddn_cp_run(ops.plan._h, _ptr(stc[0]), _ptr(stc[1]), _ptr(stc[2]), _ptr(b), 0.4, 0.7, 0.01, 0, 4, _ptr(ring)))
else:
    L.check(lib.ddn_pdr_run(ops.plan._h, _ptr(st[0]), _ptr(st[1]), _ptr(b), 2.0, 0.1, 0.049, 0, 4, 0, -1.0, 3, C.c_void_p(0), C.c_void_p(0)))
torch.cuda.synchronize()
buf = np.zeros((8192, 16), dtype=np.int64)
L.check(lib.ddn_plan_debug_stamps(ops.plan._h, buf.ctypes.data_as(C.c_void_p), 8192))
c = buf[4096:4096 + 544, :6]  # (513 blocks per image)
w = buf[4096:4096 + 544, 6:8]
ok = c[:, 5] > 0
dn = (w[:, 1] > 0) & ~ok   # the dense column-W/2 job stamps its wall-clock start and end only
print("column workgroups with stamps:", ok.sum())
t0 = c[ok, 0].min()
d = np.diff(c[ok], axis=1).astype(np.float64)
for i, nm in enumerate(["load", "fft", "pointwise", "ifft", "store"]):
    print(f"{nm:10s} median {np.median(d[:, i]):9.0f}  mean {d[:, i].mean():9.0f}  max {d[:, i].max():9.0f}")
st_ = c[ok, 0] - t0; en = c[ok, 5] - t0
print("start: min/median/max", st_.min(), np.median(st_), st_.max())
print("end:   min/median/max", en.min(), np.median(en), en.max())
print("per-WG total median", np.median(en - st_))
order = np.argsort(st_)
print("starts (sorted, every 32nd):", st_[order][::32])

wk = w[ok]
w0 = wk[:, 0].min()
ws = (wk[:, 0] - w0) / 100.0
we = (wk[:, 1] - w0) / 100.0
print("wall-clock (us, 100 MHz counter): kernel span", we.max())
for i in np.nonzero(dn)[0]:
    print("dense column job: block", int(i), "start", (w[i, 0] - w0) / 100.0, "end", (w[i, 1] - w0) / 100.0)
o = np.argsort(ws)
print("start times sorted, every 32nd:", np.round(ws[o][::32], 1))
print("end times sorted, every 32nd:  ", np.round(np.sort(we)[::32], 1))
idx = np.nonzero(ok)[0]
print("slowest WGs (block, start, end):", [(int(idx[i]), round(ws[i], 1), round(we[i], 1)) for i in np.argsort(-we)[:6]])
print("block 0 (packed pair):", round(ws[0], 1), round(we[0], 1))
print("WGs starting <5us:", int((ws < 5).sum()), " 5..100us:", int(((ws >= 5) & (ws < 100)).sum()), " >=100us:", int((ws >= 100).sum()))
print("second-round starts histogram (us):", np.histogram(ws[(ws >= 5)], bins=[5, 60, 70, 72, 74, 76, 80, 100, 120, 200])[0])
print("block 0 phases (ticks):", np.diff(c[0]).tolist())
print("block 8 phases (ticks):", np.diff(c[8]).tolist())
print("last block phases (ticks):", np.diff(c[idx[np.argmax(we)]]).tolist())
# workgroups that start out of step with the rest (the CU of the short column-W/2 job takes its next jobs early):
# their phase durations show what a job costs when the other CUs are not in the same phase
odd = [i for i in range(len(ws)) if (5 < ws[i] < 38) or ws[i] > 56]
for i in odd[:12]:
    print("out-of-step WG block", int(idx[i]), "start", round(ws[i], 1), "end", round(we[i], 1), "phases (us):", np.round(np.diff(c[idx[i]]) / 2100.0, 1).tolist())
# phase durations by start time: first-round workgroups that started on time, late ones, second round
for nm, sel in (("first round, on time", ws < 2), ("first round, late", (ws >= 2) & (ws < 30)), ("second round", ws >= 30)):
    if sel.sum():
        ph = np.diff(c[idx[sel]], axis=1) / 2100.0
        print(f"{nm:22s} n={int(sel.sum()):4d} start {ws[sel].mean():6.1f} end {we[sel].mean():6.1f}  phases (us) load {ph[:,0].mean():5.1f} fft {ph[:,1].mean():5.1f} pointwise {ph[:,2].mean():5.1f} ifft {ph[:,3].mean():5.1f} store {ph[:,4].mean():5.1f}")
