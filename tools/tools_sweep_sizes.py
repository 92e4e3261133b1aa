#!/usr/bin/env python3
"""bench.py over a list of image sizes (one image each, every run in its own process): one line per configuration with
the aggregate rate and, per solver, ms per iteration, row launch, column launch (x launches per iteration) and the
whole-iteration fraction of the HBM roofline.  usage: tools_sweep_sizes.py out.txt "SIZE [bench args]" ...  (for H x W: "W --height H")"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = open(sys.argv[1], "w")
for cfg in sys.argv[2:]:
    words = cfg.split()
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--size", words[0], "--steps", "50", "--warmup", "5"] + words[1:]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=200)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        line = f"{cfg:18s} value {d['value']:9.1f} " + " ".join(
            f"{k[:4]}: it {v['ms_per_iter']:.4f} r {v['row_kernel_ms']:.4f} c {v['col_kernel_ms']:.4f} x{v['n_col'] // max(1, v['n_row'])} frac {v['iter_hbm_frac']:.3f}"
            for k, v in d["per_solver"].items())
    except Exception as e:  # noqa: BLE001
        line = f"{cfg:18s} failed: {e}"
    print(line, flush=True)
    out.write(line + "\n")
    out.flush()
