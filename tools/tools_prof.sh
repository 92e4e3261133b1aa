#!/bin/bash
# usage: tools_prof.sh <tag> <bench args...>  -- kernel trace + three PMC passes, outputs under gpurun_out/<tag>
set -e
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --no-cpu-baseline "$@" > $OUT/kt.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -- python3 bench.py --no-cpu-baseline "$@" > $OUT/pmc1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -- python3 bench.py --no-cpu-baseline "$@" > $OUT/pmc2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc3 -- python3 bench.py --no-cpu-baseline "$@" > $OUT/pmc3.log 2>&1
find $OUT -name "*.csv" | head -20
