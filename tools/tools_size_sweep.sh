#!/bin/bash
# bench.py over the other image sizes / precisions of the design document's table: one line per configuration
out=gpurun_out/size_sweep.txt
: > $out
run() {  # label, bench args...
  local label=$1; shift
  timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/_sz.json 2>gpurun_out/_sz.err || { echo "$label failed" >> $out; tail -2 gpurun_out/_sz.err >> $out; return; }
  cp gpurun_out/_sz.json gpurun_out/sweep_$label.json
  python - "$label" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/_sz.json"))
ps = d["per_solver"]
print(f"{sys.argv[1]:12s} value {d['value']:9.1f}", " ".join(f"{k[:4]}: it {v['ms_per_iter']:.4f} r {v['row_kernel_ms']:.4f} c {v['col_kernel_ms']:.4f} frac {v['iter_hbm_frac']:.3f}" for k, v in ps.items()))
PY
}
run s2048 --size 2048 --steps 100 --warmup 10
run s1024 --size 1024 --steps 200 --warmup 20
run s512 --size 512 --steps 200 --warmup 20
run s1000 --size 1000 --steps 100 --warmup 10
run s500 --size 500 --steps 200 --warmup 20
run s300 --size 300 --steps 200 --warmup 20
run f64s512 --dtype f64 --size 512 --steps 200 --warmup 20
run f64s4096 --dtype f64 --steps 20 --warmup 3
run s8192 --size 8192 --steps 20 --warmup 3
cat $out
