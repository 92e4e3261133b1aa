#!/bin/bash
# usage: tools_prof2.sh <tag> "<counters>" <bench args...>
TAG=$1; shift
CTR=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc $CTR --output-format csv -d $OUT/pmc1 -- python3 bench.py --no-cpu-baseline "$@" > $OUT/pmc1.log 2>&1
python3 tools/tools_prof_summary.py $OUT
