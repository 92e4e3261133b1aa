#!/usr/bin/env python3
"""Column stage alone, launched back to back on one caller-owned pair of half-spectra (ddn_slab_cols over
the whole tile range, one row block).  Between two launches nothing else touches memory, so the second
and later launches find the spectra and the multipliers wherever the previous one left them (last-level
cache for a 4096 x 4096 image): the difference to the in-loop time of the same kernel is what the
iterate streams of the row kernel cost the column stage."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--n", type=int, default=20)
ap.add_argument("--flush-mb", type=int, default=0, help="write this many MB to an unrelated buffer before every launch")
args = ap.parse_args()
dev = torch.device("cuda", 0)
import convex_optimization_amd.slab as S
from convex_optimization_amd.plan import _ptr
from convex_optimization_amd.core.blur import create_motion_blur_kernel
H = W = args.size
sv = S.SlabSolver(create_motion_blur_kernel(15, 45), (H, W), 1, 1, dtype=torch.float32, device=dev)
spec = sv.xch.spec_cols
spec.normal_()
junk = torch.empty(max(args.flush_mb, 1) * 1024 * 1024 // 4, device=dev)
lib = sv.L.lib()
def go():
    sv.L.check(lib.ddn_slab_cols(sv.plan._h, _ptr(spec), sv.lay.tile_lo, sv.lay.local_tiles, sv.xch.n_row_blocks))
sv.plan.sync_stream()
for _ in range(3): go()
torch.cuda.synchronize()
ts = []
for _ in range(args.n):
    if args.flush_mb: junk.add_(1.0)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
    spec.mul_(0.0).add_(0.01)  # keep the values bounded (the resolvent stage is not a contraction on arbitrary data)
    if not args.flush_mb: go(); torch.cuda.synchronize()
ts.sort()
print(f"size {H} flush {args.flush_mb} MB: column stage ms min {ts[0]:.4f} median {ts[len(ts)//2]:.4f} max {ts[-1]:.4f}")
