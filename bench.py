#!/usr/bin/env python3
"""Benchmark of the deblur_denoise iteration hot path on MI355X.

Workload (BASELINE.json configs[2]): 4096 x 4096 grayscale image, 15-tap motion blur at 45 degrees,
noise-free b, fp32; all four solvers (chambolle_pock, primal_dr_splitting, primal_dual_dr_splitting,
admm_solver) with the reference's default parameters.  A "step" is one iteration of every solver
(four solver iterations); `value` is whole-job Mpixel-iterations/s = (N_gpus * 4 * H*W * steps) /
seconds / 1e6 with all inputs resident in HBM when the timed region starts.

With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) every rank deblurs its own
independent image (weak scaling); there is no data-path collective, the per-iteration loss records
are all-reduced over RCCL once per solver run, inside the timed region.

Other workloads (--config):
  batch1024rgb  BASELINE.json configs[3]: 1024 RGB images of 1024 x 1024 = 3072 independent planes, split over the
                ranks with shard_range (strong scaling: the 3072 planes are fixed); every rank runs its planes in
                chunks of --batch planes; the loss records of every `--check-every` iterations are summed over the
                ranks with one small all-reduce (what a global stop decision needs), inside the timed region.
  slab32768     BASELINE.json configs[4]: ONE 32768 x 32768 image, ADMM, rows split over the ranks, the column stage
                on the rank's spectral tiles, two all-to-all exchanges per iteration, stop test (all-reduce of four
                doubles) every iteration.  --size overrides the edge length (rehearsals).  --force-exchange makes a
                one-rank run go through the collective calls as well.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic HBM words (4 B in fp32) per pixel-iteration: iterate arrays read once + written once,
# each transformed array written once and read once on each side of the column stage, the blur
# eigenvalues read once per column stage.  (Where the column kernel rebuilds the eigenvalues from the blur taps --
# 4096 rows, fp32 -- that word is not moved at all; the algorithmic count keeps it, `bytes_moved_model` leaves it out.)
WORDS_ITER = {"chambolle_pock": 23, "primal_dr_splitting": 18, "primal_dual_dr_splitting": 18, "admm_solver": 26}
# the same count restricted to the fused row kernel (per launch): iterate reads + writes + 2 words of
# half-spectra in + 2 out; Chambolle-Pock launches two row kernels per iteration (dual 10 + primal 7).
WORDS_ROW = {"chambolle_pock": 17 / 2.0, "primal_dr_splitting": 13, "primal_dual_dr_splitting": 13, "admm_solver": 21}
# the same for one launch of the column stage (both half-spectra in and out + the blur eigenvalues;
# Chambolle-Pock: one half-spectrum in and out + the eigenvalues, two launches per iteration)
WORDS_COL = {"chambolle_pock": 3, "primal_dr_splitting": 5, "primal_dual_dr_splitting": 5, "admm_solver": 5}
ROW_KERNEL = {"chambolle_pock": "k_wave_cp", "primal_dr_splitting": "k_wave_row<pdr>", "primal_dual_dr_splitting": "k_wave_row<pddr>",
              "admm_solver": "k_wave_row<admm>"}
COL_KERNEL = {"chambolle_pock": "k_wave_col<multiplier>", "primal_dr_splitting": "k_wave_col<resolvent>",
              "primal_dual_dr_splitting": "k_wave_col<resolvent>", "admm_solver": "k_wave_col<resolvent>"}


def words_without_eigenvalue_reads(H, f64):
    """The word tables for the case that the column kernels rebuild the blur eigenvalues from the taps instead of
    reading them (the register-resident 4096-row kernels, blur kernels of up to 31 rows): one word less per column
    launch, two for Chambolle-Pock's iteration (it has two column launches)."""
    if f64 or H != 4096 or os.environ.get("DDN_NO_TAPS"):
        return WORDS_ITER, WORDS_COL
    it = {k: v - (2 if k == "chambolle_pock" else 1) for k, v in WORDS_ITER.items()}
    col = {k: v - 1 for k, v in WORDS_COL.items()}
    return it, col


def kernel_name(table, solver, n, f64):
    """Name of the kernel family that runs a line length n: register-resident for 4096 (fp32), several lines per
    wavefront for 512-2048 (fp32), else the LDS kernels."""
    base = table[solver]
    if f64 or n not in (512, 1024, 2048, 4096):
        return base.replace("k_wave_cp", "k_row_cp").replace("k_wave_row", "k_row").replace("k_wave_col", "k_col")
    if n != 4096:
        return base.replace("k_wave_cp", "k_wave_cpN").replace("k_wave_row<", "k_wave_rowN<").replace("k_wave_col<", "k_wave_colN<")
    return base
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def measured_traffic(lib_version, H, W, B):
    """HBM bytes per full launch of each kernel from the committed rocprofv3 PMC passes (profiles/*_pmc.json, written by
    tools/tools_pmc_summary.py: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction applied).  Only figures measured on
    the library version that is running, at this problem size, are reported; otherwise null."""
    import glob
    best, seen = None, []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        if rec.get("size") == [H, W] and rec.get("batch") == B:
            seen.append(rec.get("ddn_version"))
            if rec.get("ddn_version") == lib_version:
                best = rec
    if best is None:
        # no silent null: say why there is no figure
        why = (f"no PMC record for library version {lib_version} at {H}x{W}, batch {B} under profiles/ (records of this size exist for versions "
               f"{sorted(set(v for v in seen if v is not None))}); run tools/tools_pmc.sh on this version") if seen else \
              f"no PMC record at {H}x{W}, batch {B} under profiles/; run tools/tools_pmc.sh"
        return {"kernels": {}, "source": None, "missing": why}
    return best


def _sync_all(dev, distributed):
    torch.cuda.synchronize(dev)
    if distributed:
        import torch.distributed as dist
        dist.barrier()
        torch.cuda.synchronize(dev)


def run_slab(args, dev, dt, rank, world, distributed, allreduce):
    """One image of size x size, ADMM, rows split over the ranks (slab.py): W warm-up iterations, then K timed ones."""
    import torch.distributed as dist
    from convex_optimization_amd import _lib as L
    from convex_optimization_amd import slab as S
    from convex_optimization_amd.core.blur import create_motion_blur_kernel
    H = W = args.size
    K, Wm = args.steps, args.warmup
    if H % world:
        raise SystemExit(f"--size {H} is not a multiple of the {world} ranks")
    rows = H // world
    kern = create_motion_blur_kernel(15, 45)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    free0, total_mem = torch.cuda.mem_get_info(dev)
    b = torch.rand((rows, W), device=dev, dtype=dt, generator=gen)   # rank-local rows of the blurred image
    st = tuple(torch.clamp(torch.randn(sh, device=dev, dtype=dt, generator=gen), 0, 1) for sh in ((rows, W), (rows, W), (3, rows, W), (3, rows, W)))
    sol = torch.zeros_like(b)
    sv = S.SlabSolver(kern, (H, W), 1, 1, dtype=dt, device=dev, force_exchange=args.force_exchange, exchange_groups=args.exchange_groups)
    ring = torch.zeros((max(K, Wm, 1), 8), dtype=torch.float64, device=dev)
    t, rho, gamma, obj = 18.0, 0.001, 0.5, L.OBJ["l1"]
    stages = {"rows": 0.0, "column_stage": 0.0, "stop_test": 0.0}
    ev = []

    def iterations(n, timed):
        # steady-state iterations of admm_solver_slab (slab.py): column stage (exchange, column kernels, exchange),
        # fused row kernel (consume + produce, look-ahead solution kept), stop test on the all-reduced loss sums
        for c in range(1, n + 1):
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed and c % 5 == 0 else None
            if marks: marks[0].record()
            sv.column_stage()           # exchange, column kernels, exchange: pipelined over tile groups when there is an exchange
            if marks: marks[1].record()
            sv.rows(S.SOLVER_ADMM, st, b, t, rho, gamma, obj, True, True, sol=sol)
            if marks: marks[2].record()
            sv.stop_test(L.LOSS_PSNR, -1e300, c, 2, ring[c - 1])    # armed, never true
            if marks:
                marks[3].record()
                ev.append(marks)
            if c % args.check_every == 0:
                assert sv.stopped_at() < 0

    with torch.cuda.device(dev):
        sv.plan.sync_stream()
        sv.rows(S.SOLVER_ADMM, st, b, t, rho, gamma, obj, False, True)
        iterations(max(Wm, 1), False)
        _sync_all(dev, distributed)
        t0 = time.perf_counter()
        iterations(K, True)
        _sync_all(dev, distributed)
        t1 = time.perf_counter()
    free1, _ = torch.cuda.mem_get_info(dev)
    for m in ev:
        stages["column_stage"] += m[0].elapsed_time(m[1])
        stages["rows"] += m[1].elapsed_time(m[2])
        stages["stop_test"] += m[2].elapsed_time(m[3])
    stages = {k: (v / len(ev) if ev else None) for k, v in stages.items()}
    el = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        allreduce(el, dist.ReduceOp.MAX)
    sec = float(el.item())
    esz = 8 if dt == torch.float64 else 4
    # bytes a rank sends per exchange: both half-spectra of its rows, minus the share it keeps
    a2a_bytes = 2 * rows * (W // 2 + 1) * 2 * esz * (world - 1) / world
    px = float(H) * W
    out = {"metric": "Mpixel-iterations/sec, ADMM on one image split into row slabs", "value": px * K / sec / 1e6, "unit": "Mpixel-iterations/s",
           "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": sec / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"one {H}x{W} image, ADMM (reference defaults), rows split over {world} rank(s), slab-decomposed 2-D transform, "
                                  "two all-to-all exchanges and one 4-double all-reduce per iteration",
                      "rows_per_rank": rows, "step": "one ADMM iteration of the whole image", "ddn_version": int(L.lib().ddn_version()),
                      "exchange_forced": bool(args.force_exchange and world == 1), "backend": (dist.get_backend() if distributed else None),
                      "all_to_all_bytes_sent_per_rank_per_exchange": a2a_bytes, "exchanges_per_iteration": 2, "exchange_tile_groups": sv.xch.groups},
           "stage_ms": stages, "device_memory_in_use_gib": round((total_mem - free1) / 2**30, 2),
           "iter_hbm_frac": WORDS_ITER["admm_solver"] * esz * px / world * K / sec / 1e9 / HBM_PEAK_GBS,
           "final_loss": [float(v) for v in ring[K - 1, :4].cpu()], "solution_finite": bool(torch.isfinite(sol).all().item()),
           "mse_per_iteration": [float(v) for v in ring[:K, 2].cpu()],
           "roofline": {"bound": "hbm", "achieved": WORDS_ITER["admm_solver"] * esz * px / world * K / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": WORDS_ITER["admm_solver"] * esz * px / world * K / sec / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "kernel": "whole iteration (per GPU): the split-row kernels, the two-pass column kernels and the exchanges"}}
    sv.close()
    if world == 1 and not args.no_slab_check:
        # the same problem on the single-plan path (ddn_admm_run on a whole-plane plan), same seed: the whole-image mse of
        # the first timed iterations must agree.  Timed launch c is consuming launch W' + c of the run, which records
        # the look-ahead solution of iteration W' + c - 1: row W' + c - 2 of ddn_admm_run's record.
        del st, sol, b, sv
        torch.cuda.empty_cache()
        free2, _ = torch.cuda.mem_get_info(dev)
        nchk, wm = min(3, K), max(Wm, 1)
        if 26.0 * esz * px < free2:
            import convex_optimization_amd as pkg
            from convex_optimization_amd.plan import _ptr
            gen.manual_seed(1000 + rank)
            b1 = torch.rand((1, H, W), device=dev, dtype=dt, generator=gen)
            s1 = tuple(torch.clamp(torch.randn(sh, device=dev, dtype=dt, generator=gen), 0, 1) for sh in ((1, H, W), (1, H, W), (3, 1, H, W), (3, 1, H, W)))
            sol1 = torch.zeros_like(b1)
            ops = pkg.DeblurDenoiseOperators(kern, b1[0], 1, 1, device=dev, batch=1, dtype=dt)
            n1 = wm + nchk - 1
            ring1 = torch.zeros((n1, 1, 8), dtype=torch.float64, device=dev)
            done1 = torch.zeros((1,), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                ops.plan.sync_stream()
                L.check(L.lib().ddn_admm_run(ops.plan._h, _ptr(s1[0]), _ptr(s1[1]), _ptr(s1[2]), _ptr(s1[3]), _ptr(b1), t, rho, gamma, obj, n1,
                                             L.LOSS_PSNR, -1e300, 2, _ptr(ring1), _ptr(done1), _ptr(sol1)))
                torch.cuda.synchronize(dev)
            single = [float(v) for v in ring1[wm - 1:wm - 1 + nchk, 0, 2].cpu()]
            slab = out["mse_per_iteration"][:nchk]
            diff = max(abs(a - c) / max(abs(c), 1e-300) for a, c in zip(slab, single))
            out["single_plan_check"] = {"what": "whole-image mse of (look-ahead solution, b) of the first timed iterations: slab entry points vs ddn_admm_run on one whole-plane plan, same seed",
                                        "mse_slab": slab, "mse_single_plan": single, "max_rel_diff": diff, "agrees": bool(diff < 1e-4)}
            ops.close()
            assert diff < 1e-4, ("slab path and single-plan path disagree", slab, single)
        else:
            out["single_plan_check"] = {"skipped": f"needs about {26.0 * esz * px / 2**30:.0f} GiB of free device memory, {free2 / 2**30:.0f} GiB free"}
    return out


def run_batch(args, dev, dt, rank, world, distributed, allreduce):
    """--planes independent planes of size x size split over the ranks; all four solvers; W warm-up, K timed iterations."""
    import torch.distributed as dist
    import convex_optimization_amd as pkg
    from convex_optimization_amd import _lib as L
    from convex_optimization_amd.plan import _ptr, Plan
    from convex_optimization_amd.distributed import shard_range
    from convex_optimization_amd.core.blur import create_motion_blur_kernel
    H = W = args.size
    K, Wm, ce = args.steps, args.warmup, max(1, args.check_every)
    lo, hi = shard_range(args.planes, rank, world)
    nloc = hi - lo
    chunk = max(1, min(args.batch, nloc))
    kern = create_motion_blur_kernel(15, 45)
    lib = L.lib()
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    solvers = [s for s in args.solvers.split(",") if s]
    comm = torch.cuda.Stream(device=dev) if distributed else None
    per_solver, total_time = {}, 0.0
    esz = 8 if dt == torch.float64 else 4

    def rnd(*shape):
        return torch.rand(shape, device=dev, dtype=dt, generator=gen)

    # the rank's planes, blurred, resident in HBM: chunks of `chunk` planes (the last one may be shorter)
    chunks = []
    for c0 in range(0, nloc, chunk):
        n = min(chunk, nloc - c0)
        pl = Plan(H, W, n, dt, dev)
        chunks.append(pl.circular_convolve2d(kern, rnd(n, H, W)))
        pl.close()
    torch.cuda.synchronize(dev)

    for name in solvers:
        # one operator set (plan) per chunk size; iterate arrays of every chunk resident before timing
        ops_by_n, states = {}, []
        for b in chunks:
            n = b.shape[0]
            if n not in ops_by_n:
                tr = {"chambolle_pock": (1, 1), "primal_dr_splitting": (2.0,), "primal_dual_dr_splitting": (3, 3), "admm_solver": (1, 1)}[name]
                ops_by_n[n] = pkg.DeblurDenoiseOperators(kern, b[0], *tr, device=dev, batch=n, dtype=dt)
            if name == "chambolle_pock":
                states.append([rnd(n, H, W), rnd(n, H, W), rnd(3, n, H, W)])
            elif name == "primal_dr_splitting":
                states.append([b.clone(), torch.stack((b, b, b)).contiguous()])
            elif name == "primal_dual_dr_splitting":
                states.append([rnd(n, H, W), rnd(3, n, H, W)])
            else:
                states.append([torch.clamp(torch.randn((n, H, W), device=dev, dtype=dt, generator=gen), 0, 1) for _ in range(2)]
                              + [torch.clamp(torch.randn((3, n, H, W), device=dev, dtype=dt, generator=gen), 0, 1) for _ in range(2)]
                              + [torch.empty((n, H, W), device=dev, dtype=dt)])
        done = torch.zeros((chunk,), dtype=torch.int32, device=dev)
        rings = [torch.zeros((ce, b.shape[0], 8), dtype=torch.float64, device=dev) for b in chunks]
        pending = []

        def segment(ci, n_it):
            b, st, ring = chunks[ci], states[ci], rings[ci]
            h = ops_by_n[b.shape[0]].plan._h
            if name == "chambolle_pock":
                L.check(lib.ddn_cp_run(h, _ptr(st[0]), _ptr(st[1]), _ptr(st[2]), _ptr(b), 0.4, 0.7, 0.01, 0, n_it, _ptr(ring)))
            elif name == "primal_dr_splitting":
                L.check(lib.ddn_pdr_run(h, _ptr(st[0]), _ptr(st[1]), _ptr(b), 2.0, 0.1, 0.049, L.OBJ_DUALS_EQUAL, n_it, L.LOSS_MSE, -1.0, 3, _ptr(ring), _ptr(done)))
            elif name == "primal_dual_dr_splitting":
                L.check(lib.ddn_pddr_run(h, _ptr(st[0]), _ptr(st[1]), _ptr(b), 3.0, 0.4, 0.03, n_it, L.LOSS_L2, -1.0, 2, _ptr(ring), _ptr(done)))
            else:
                L.check(lib.ddn_admm_run(h, _ptr(st[0]), _ptr(st[1]), _ptr(st[2]), _ptr(st[3]), _ptr(b), 18.0, 0.001, 0.5, 0, n_it,
                                         L.LOSS_PSNR, -1.0, 2, _ptr(ring), _ptr(done), _ptr(st[4])))

        def sweep(n_it):
            # n_it iterations of every local plane, `ce` at a time; after each group of `ce` iterations the loss records,
            # summed over the rank's planes, join an all-reduce over the ranks on a side stream
            it = 0
            while it < n_it:
                m = min(ce, n_it - it)
                acc = torch.zeros((ce, 8), dtype=torch.float64, device=dev)
                for ci in range(len(chunks)):
                    segment(ci, m)
                    acc[:m] += rings[ci][:m].sum(dim=1)
                if distributed:
                    if dist.get_backend() == "nccl":
                        comm.wait_stream(torch.cuda.current_stream(dev))
                        with torch.cuda.stream(comm):
                            pending.append((acc, dist.all_reduce(acc, op=dist.ReduceOp.SUM, async_op=True)))
                    else:
                        allreduce(acc, dist.ReduceOp.SUM)
                it += m
            for acc, work in pending:
                work.wait()
            pending.clear()

        for o in ops_by_n.values():
            o.plan.sync_stream()
        if Wm > 0:
            sweep(Wm)
        _sync_all(dev, distributed)
        for o in ops_by_n.values():
            o.plan.enable_timing(0 if args.no_kernel_timing else 5)
        t0 = time.perf_counter()
        sweep(K)
        _sync_all(dev, distributed)
        t1 = time.perf_counter()
        tms = []
        for o in ops_by_n.values():
            o.plan.enable_timing(False)
            tms.append((o.plan.batch, o.plan.last_timing()))
        el = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
        if distributed:
            allreduce(el, dist.ReduceOp.MAX)
        sec = float(el.item())
        total_time += sec
        px_all = float(args.planes) * H * W
        nb, tm = max(tms, key=lambda x: x[0])     # kernel durations of the full-size chunk
        pxc = float(nb) * H * W
        per_solver[name] = {"mpix_it_per_s": px_all * K / sec / 1e6, "ms_per_iter": sec / K * 1e3,
                            "iter_hbm_frac": WORDS_ITER[name] * esz * (px_all / world) * K / sec / 1e9 / HBM_PEAK_GBS,
                            "row_kernel_ms": tm["row_ms"], "col_kernel_ms": tm["col_ms"], "planes_per_launch": nb,
                            "row_kernel_gbs": (WORDS_ROW[name] * esz * pxc / (tm["row_ms"] * 1e-3) / 1e9) if tm["row_ms"] > 0 else None,
                            "col_kernel_gbs": (WORDS_COL[name] * esz * pxc / (tm["col_ms"] * 1e-3) / 1e9) if tm["col_ms"] > 0 else None}
        for o in ops_by_n.values():
            o.close()
        del states, rings
        torch.cuda.empty_cache()

    px_all = float(args.planes) * H * W
    nsol = len(solvers)
    value = px_all * K * nsol / total_time / 1e6
    kernels = {}
    for sname, ps in per_solver.items():
        lpi = 2 if sname == "chambolle_pock" else 1
        for kind, words, ms, gbs in (("row", WORDS_ROW[sname], ps["row_kernel_ms"], ps["row_kernel_gbs"]), ("col", WORDS_COL[sname], ps["col_kernel_ms"], ps["col_kernel_gbs"])):
            if ms and ms > 0:
                kernels[f"{kind} kernel of {sname}"] = {"ms": ms, "launches_per_iteration_per_chunk": lpi, "algorithmic_bytes": words * esz * ps["planes_per_launch"] * H * W,
                                                        "achieved_gbs": gbs, "frac": gbs / HBM_PEAK_GBS, "share_ms": ms * lpi}
    roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "kernel": None}
    if kernels:
        dom = max(kernels, key=lambda k: kernels[k]["share_ms"])
        roof.update({"achieved": kernels[dom]["achieved_gbs"], "frac": kernels[dom]["frac"], "kernel": dom,
                     "algorithmic_bytes_per_launch": kernels[dom]["algorithmic_bytes"], "kernel_ms": kernels[dom]["ms"]})
    roof["kernels"] = kernels
    roof["whole_iteration_frac"] = {s_: per_solver[s_]["iter_hbm_frac"] for s_ in per_solver}
    return {"metric": "Mpixel-iterations/sec per solver, batch of independent planes", "value": value, "unit": "Mpixel-iterations/s", "n_gpus": world,
            "steps": K, "warmup": Wm, "ms_per_step": total_time / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.planes} independent {H}x{W} planes ({args.planes // 3} RGB images), 15-tap motion blur, all four solvers, reference defaults; "
                                   f"planes split over the ranks (shard_range), {chunk} planes per launch, loss records all-reduced every {ce} iterations",
                       "planes_per_gpu": nloc, "solvers": solvers, "step": "one iteration of each solver on every plane", "ddn_version": int(lib.ddn_version())},
            "per_solver": per_solver, "roofline": roof}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--height", type=int, default=0, help="image4096 with another shape: rows of the image (default: --size); --size is then the width")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--solvers", default="chambolle_pock,primal_dr_splitting,primal_dual_dr_splitting,admm_solver")
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--split", type=int, default=0)
    ap.add_argument("--col-threads", "--row-threads", dest="col_threads", type=int, default=0, help="threads per workgroup of the tile-generic column kernels")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-kernel HIP events (roofline fields become null)")
    ap.add_argument("--config", default="image4096", choices=["image4096", "batch1024rgb", "slab32768"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--check-every", type=int, default=10)
    ap.add_argument("--planes", type=int, default=3072, help="batch1024rgb: planes in all (1024 RGB images)")
    ap.add_argument("--force-exchange", action="store_true")
    ap.add_argument("--no-slab-check", action="store_true", help="slab32768 on one rank: skip the comparison with the single-plan path")
    ap.add_argument("--exchange-groups", type=int, default=0, help="slab32768: tile groups of the pipelined exchange (0 = automatic, 1 = one exchange)")
    args = ap.parse_args()
    if args.config == "batch1024rgb" and "--size" not in sys.argv:
        args.size = 1024
    if args.config == "batch1024rgb" and "--batch" not in sys.argv:
        args.batch = 256
    if args.config == "slab32768" and "--size" not in sys.argv:
        args.size = 32768
    if args.config == "slab32768" and "--steps" not in sys.argv:
        args.steps = 20

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or (args.force_exchange and args.config == "slab32768")
    # one rank per GPU; the modulo only matters in a rehearsal of the multi-rank path on a box with fewer GPUs than
    # ranks (DDN_BENCH_BACKEND=gloo: RCCL itself refuses two ranks on one device)
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if distributed:
        import torch.distributed as dist
        backend = os.environ.get("DDN_BENCH_BACKEND", "nccl")
        if world == 1:      # a one-rank group of its own (--force-exchange): no launcher has set the rendezvous
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
            import faulthandler
            faulthandler.dump_traceback_later(150, exit=True)   # a rehearsal that hangs names the place and ends

    def allreduce(t, op):
        # RCCL reduces device tensors in place; the gloo rehearsal goes through host memory
        if dist.get_backend() == "nccl":
            dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    import convex_optimization_amd as pkg
    from convex_optimization_amd import _lib as L
    from convex_optimization_amd.plan import _ptr
    from convex_optimization_amd.core.blur import create_motion_blur_kernel

    H = W = args.size
    if args.height > 0 and args.config == "image4096":
        H = args.height
    B = args.batch
    K, Wm = args.steps, args.warmup
    dt = torch.float64 if args.dtype == "f64" else torch.float32
    if args.config == "slab32768":
        out = run_slab(args, dev, dt, rank, world, distributed, allreduce if distributed else None)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if distributed:
            dist.destroy_process_group()
        return
    if args.config == "batch1024rgb":
        out = run_batch(args, dev, dt, rank, world, distributed, allreduce if distributed else None)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if distributed:
            dist.destroy_process_group()
        return
    f64 = dt == torch.float64
    esz = 8.0 if f64 else 4.0
    words_iter, words_col = words_without_eigenvalue_reads(H, f64)
    kern = create_motion_blur_kernel(15, 45)
    tune = {"cols": args.cols, "split": args.split, "col_threads": args.col_threads}

    # synthetic input, generated rank-locally on the device, resident in HBM before timing
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    img = torch.rand((B, H, W), device=dev, dtype=dt, generator=gen)
    from convex_optimization_amd.plan import Plan
    pl = Plan(H, W, B, dt, dev)
    b = pl.circular_convolve2d(kern, img)
    torch.cuda.synchronize(dev)
    pl.close()
    del img

    lib = L.lib()
    # iterations applied per image in the last run: the early-stop test stays armed (negative
    # tolerance, never true) and the bench asserts that every timed iteration really ran
    done = torch.zeros((B,), dtype=torch.int32, device=dev)
    solvers = [s for s in args.solvers.split(",") if s]
    per_solver = {}
    total_time = 0.0

    def barrier():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def rnd(*shape):
        return torch.rand(shape, device=dev, dtype=dt, generator=gen)

    for name in solvers:
        if name == "chambolle_pock":
            ops = pkg.DeblurDenoiseOperators(kern, b[0], 1, 1, device=dev, batch=B, dtype=dt, tune=tune)
            st = [rnd(B, H, W), rnd(B, H, W), rnd(3, B, H, W)]
            def run(n, ring, st=st, ops=ops):
                L.check(lib.ddn_cp_run(ops.plan._h, _ptr(st[0]), _ptr(st[1]), _ptr(st[2]), _ptr(b), 0.4, 0.7, 0.01, 0, n, _ptr(ring)))
        elif name == "primal_dr_splitting":
            ops = pkg.DeblurDenoiseOperators(kern, b[0], 2.0, device=dev, batch=B, dtype=dt, tune=tune)
            st = [b.clone(), torch.stack((b, b, b)).contiguous()]
            def run(n, ring, st=st, ops=ops):
                L.check(lib.ddn_pdr_run(ops.plan._h, _ptr(st[0]), _ptr(st[1]), _ptr(b), 2.0, 0.1, 0.049, L.OBJ_DUALS_EQUAL, n, L.LOSS_MSE, -1.0, 3, _ptr(ring), _ptr(done)))
        elif name == "primal_dual_dr_splitting":
            ops = pkg.DeblurDenoiseOperators(kern, b[0], 3, 3, device=dev, batch=B, dtype=dt, tune=tune)
            st = [rnd(B, H, W), rnd(3, B, H, W)]
            def run(n, ring, st=st, ops=ops):
                L.check(lib.ddn_pddr_run(ops.plan._h, _ptr(st[0]), _ptr(st[1]), _ptr(b), 3.0, 0.4, 0.03, n, L.LOSS_L2, -1.0, 2, _ptr(ring), _ptr(done)))
        elif name == "admm_solver":
            ops = pkg.DeblurDenoiseOperators(kern, b[0], 1, 1, device=dev, batch=B, dtype=dt, tune=tune)
            st = [torch.clamp(torch.randn((B, H, W), device=dev, dtype=dt, generator=gen), 0, 1), torch.clamp(torch.randn((B, H, W), device=dev, dtype=dt, generator=gen), 0, 1),
                  torch.clamp(torch.randn((3, B, H, W), device=dev, dtype=dt, generator=gen), 0, 1), torch.clamp(torch.randn((3, B, H, W), device=dev, dtype=dt, generator=gen), 0, 1),
                  torch.empty((B, H, W), device=dev, dtype=dt)]
            def run(n, ring, st=st, ops=ops):
                L.check(lib.ddn_admm_run(ops.plan._h, _ptr(st[0]), _ptr(st[1]), _ptr(st[2]), _ptr(st[3]), _ptr(b), 18.0, 0.001, 0.5, 0, n,
                                         L.LOSS_PSNR, -1.0, 2, _ptr(ring), _ptr(done), _ptr(st[4])))
        else:
            raise SystemExit(f"unknown solver {name}")
        ring = torch.zeros((max(K, Wm, 1), B, 8), dtype=torch.float64, device=dev)
        done.fill_(K)
        ops.plan.sync_stream()
        if Wm > 0:
            run(Wm, ring)
        barrier()
        # kernel durations are sampled: every 5th row launch and every 5th column launch carries HIP events (each pair
        # costs a few microseconds of an iteration that lasts a few hundred)
        # (an odd stride, so that both alternating Chambolle-Pock kernels are sampled)
        ops.plan.enable_timing(0 if args.no_kernel_timing else 5)
        t0 = time.perf_counter()
        run(K, ring)
        if distributed:
            red = ring[:K].clone()
            allreduce(red, dist.ReduceOp.SUM)
        barrier()
        t1 = time.perf_counter()
        ops.plan.enable_timing(False)
        tm = ops.plan.last_timing()
        assert int(done.min().item()) == K, f"{name}: only {int(done.min().item())} of {K} iterations ran"
        el = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
        if distributed:
            allreduce(el, dist.ReduceOp.MAX)
        sec = float(el.item())
        total_time += sec
        px = float(B) * H * W
        row_ms = tm["row_ms"]
        per_solver[name] = {
            "mpix_it_per_s": world * px * K / sec / 1e6,
            "ms_per_iter": sec / K * 1e3,
            "iter_hbm_frac": WORDS_ITER[name] * esz * px * K / sec / 1e9 / HBM_PEAK_GBS,
            "row_kernel_ms": row_ms, "col_kernel_ms": tm["col_ms"], "n_row": tm["n_row"], "n_col": tm["n_col"],
            "row_kernel_gbs": (WORDS_ROW[name] * esz * px / (row_ms * 1e-3) / 1e9) if row_ms > 0 else None,
            "col_kernel_gbs": (WORDS_COL[name] * esz * px / (tm["col_ms"] * 1e-3) / 1e9) if tm["col_ms"] > 0 else None,
            "final_loss": [float(v) for v in ring[K - 1, 0, :4].cpu()],
        }
        ops.close()
        del st, ring
        torch.cuda.empty_cache()

    nsol = len(solvers)
    px = float(B) * H * W
    value = world * px * K * nsol / total_time / 1e6
    # per-kernel table: mean launch duration (HIP events on the plan's stream), launches per step, algorithmic bytes
    # per launch, achieved rate.  The dominant kernel is the one with the largest share of a step, row or column.
    lib_version = int(lib.ddn_version())
    pmc = measured_traffic(lib_version, H, W, B)
    kernels = {}
    for sname, ps in per_solver.items():
        lpi = 2 if sname == "chambolle_pock" else 1
        for kind, kname, words, ms in (("row", kernel_name(ROW_KERNEL, sname, W, f64), WORDS_ROW[sname], ps["row_kernel_ms"]),
                                       ("col", kernel_name(COL_KERNEL, sname, H, f64), WORDS_COL[sname], ps["col_kernel_ms"])):
            if not ms or ms <= 0:
                continue
            e = kernels.setdefault(kname, {"ms_per_step": 0.0, "launches_per_step": 0, "algorithmic_bytes": words * esz * px, "_bytes": 0.0})
            e["ms_per_step"] += ms * lpi
            e["launches_per_step"] += lpi
            e["_bytes"] += words * esz * px * lpi
    for kname, e in kernels.items():
        e["ms"] = e["ms_per_step"] / e["launches_per_step"]
        e["achieved_gbs"] = e.pop("_bytes") / (e["ms_per_step"] * 1e-3) / 1e9
        e["frac"] = e["achieved_gbs"] / HBM_PEAK_GBS
        e["traffic"] = (pmc or {}).get("kernels", {}).get(kname, {}).get("hbm_bytes_per_launch")
        if kname.startswith("k_wave_col<") and words_col is not WORDS_COL:
            # these launches rebuild the blur eigenvalues from the taps: one word per element less is actually moved
            e["bytes_moved_model"] = e["algorithmic_bytes"] - esz * px
    if kernels:
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        domk = kernels[dom]
        roof = {"bound": "hbm", "achieved": domk["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": domk["frac"],
                "traffic": domk["traffic"], "traffic_source": (pmc or {}).get("source") or (pmc or {}).get("missing"), "kernel": dom,
                "algorithmic_bytes_per_launch": domk["algorithmic_bytes"], "kernel_ms": domk["ms"]}
    else:
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "kernel": None}
    roof["kernels"] = kernels
    roof["whole_iteration_frac"] = {s_: per_solver[s_]["iter_hbm_frac"] for s_ in per_solver}
    out = {
        "metric": "Mpixel-iterations/sec per solver at 4096^2 + % HBM roofline",
        "value": value, "unit": "Mpixel-iterations/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": total_time / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{H}x{W} grayscale motion-blur (15 taps @45deg) deconvolution, {args.dtype}, all four solvers, reference default parameters"
                               " (Chambolle-Pock's defaults t=0.4, s=0.7 are outside its stable range; its iterates are pinned against the reference for N <= 20 only)",
                   "images_per_gpu": B, "solvers": solvers, "step": "one iteration of each solver", "ddn_version": lib_version},
        "per_solver": per_solver,
        "roofline": roof,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle (a port of the reference's op sequence to the same torch CPU ops),
        # same workload, a bounded number of iterations per solver, all host cores torch uses.
        from oracle import ddn_oracle as O
        bc = b[0].cpu()
        kc = kern
        n = max(2, args.cpu_iters)
        cores = torch.get_num_threads()

        def cpu_run(name, n_it):
            torch.manual_seed(7)
            t0 = time.perf_counter()
            if name == "chambolle_pock":
                O.chambolle_pock(bc[None], kc, max_iter=n_it + 1, loss_function=O.psnr)
            elif name == "primal_dr_splitting":
                O.primal_dr_splitting("l1", kc, bc, niters=n_it, loss_function=O.mse)
            elif name == "primal_dual_dr_splitting":
                O.primal_dual_dr_splitting(bc, kc, niters=n_it + 1, loss_function=O.l2_loss)
            else:
                O.admm_solver(bc, kernel=kc, niters=n_it + 1, loss_function=O.psnr)
            return time.perf_counter() - t0

        # each solver once with 1 iteration and once with n: the difference is n - 1 iterations without the operator
        # set-up (three conv2d + fft2 of the image size) and, for ADMM, without the final solve, which the GPU leg
        # does not time either
        t_one = sum(cpu_run(name, 1) for name in solvers)
        t_n = sum(cpu_run(name, n) for name in solvers)
        its = (n - 1) * len(solvers)
        out["cpu_baseline"] = {"value": H * W * its / max(t_n - t_one, 1e-9) / 1e6, "unit": "Mpixel-iterations/s", "cores": cores, "kind": "port",
                               "value_with_setup": H * W * n * len(solvers) / t_n / 1e6,
                               "sample": f"{n - 1} iteration(s) of each of the {len(solvers)} solvers at {H}x{W}: time of an {n}-iteration run minus "
                                         f"time of a 1-iteration run (operator set-up excluded, as in the GPU leg); value_with_setup: the {n}-iteration runs whole"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
