#!/bin/bash
# Column-stage first-round delay sweep: bench.py per delay value (own process: the library reads the variable once)
out=gpurun_out/stagger_sweep.txt
: > $out
for us in ${STAGGER_LIST:-0 3 5 6 7 8 10 12}; do
  DDN_COL_STAGGER_US=$us timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/_sw.json 2>/dev/null || exit 1
  python - "$us" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/_sw.json"))
ps = d["per_solver"]
print("stagger_us", sys.argv[1], "value", round(d["value"]), " ".join(f"{k[:6]}: it {v['ms_per_iter']:.4f} row {v['row_kernel_ms']:.4f} col {v['col_kernel_ms']:.4f}" for k, v in ps.items()))
PY
done
cat $out
