#!/bin/bash
# ms per iteration of the four solvers on one small image (the reference's demo sizes): launch-bound or kernel-bound?
out=gpurun_out/small_sizes.txt
: > $out
for n in ${SIZES:-64 100 128 200 256 300 500 512}; do
  timeout -k 10 120 python bench.py --size $n --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/_ss.json 2>gpurun_out/_ss.err || { echo "size $n failed" >> $out; tail -2 gpurun_out/_ss.err >> $out; continue; }
  python - $n >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/_ss.json"))
print("size", sys.argv[1], "value", round(d["value"], 1), " ".join(f"{k[:4]}: it {v['ms_per_iter']*1e3:.1f}us row {v['row_kernel_ms']*1e3:.1f} col {v['col_kernel_ms']*1e3:.1f}" for k, v in d["per_solver"].items()))
PY
done
cat $out
