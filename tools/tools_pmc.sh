#!/bin/bash
# usage: tools_pmc.sh <tag> <bench args...>
# Kernel trace + the two HBM traffic passes (FETCH_SIZE and WRITE_SIZE do not fit one pass) of bench.py, each in a run
# of its own, then profiles-ready summaries: gpurun_out/<tag>/<tag>_pmc.json, <tag>_kernel_stats.csv, <tag>_bench.json.
set -e
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --no-cpu-baseline "$@" > $OUT/${TAG}_bench.json 2> $OUT/kt.err
echo "kernel trace done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-kernel-timing "$@" > $OUT/pmc_fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu-baseline --no-kernel-timing "$@" > $OUT/pmc_write.log 2>&1
echo "write pass done"
python3 tools/tools_pmc_summary.py $OUT $TAG
