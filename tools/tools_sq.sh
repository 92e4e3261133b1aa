#!/bin/bash
# usage: tools_sq.sh <tag> <bench args...>
# SQ counters of every kernel of a bench.py run, two passes of at most 8 SQ counters each (one rocprofv3 run per pass),
# summarised per kernel family as means per launch: gpurun_out/<tag>/<tag>_sq_counters.txt
set -e -o pipefail
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq1 -- python3 bench.py --no-cpu-baseline --no-kernel-timing "$@" > $OUT/sq1.log 2>&1
echo "sq pass 1 done"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/sq2 -- python3 bench.py --no-cpu-baseline --no-kernel-timing "$@" > $OUT/sq2.log 2>&1 || echo "sq pass 2 FAILED (see sq2.log)"
echo "sq pass 2 done"
python3 tools/tools_sq_summary.py $OUT $TAG "$@"
