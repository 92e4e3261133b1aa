#!/bin/bash
# A/B runs of bench.py over builds of the library (LIBS: paths, "-" = the in-tree build) and environment settings
# (ENVS: "VAR=value" words, "-" = none); every run in its own process.  REPS repeats the whole list.
out=gpurun_out/ab.txt
: > $out
for rep in $(seq 1 ${REPS:-2}); do
for lib in ${LIBS:--}; do
for ev in ${ENVS:--}; do
  ( [ "$lib" != "-" ] && export DDN_LIB_PATH=$PWD/$lib; [ "$ev" != "-" ] && export $ev
    timeout -k 10 120 python bench.py --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/_ab.json 2>gpurun_out/_ab.err ) || { echo "FAILED $lib $ev"; tail -3 gpurun_out/_ab.err; exit 1; }
  python - "$lib" "$ev" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/_ab.json"))
ps = d["per_solver"]
print(f"{sys.argv[1]:28s} {sys.argv[2]:24s} value {round(d['value'])}", " ".join(f"{k[:4]}: {v['ms_per_iter']:.4f} r {v['row_kernel_ms']:.4f} c {v['col_kernel_ms']:.4f}" for k, v in ps.items()))
PY
done; done; done
cat $out
