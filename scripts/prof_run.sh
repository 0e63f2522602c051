#!/bin/bash
# usage: scripts/prof_run.sh tag [bench args...]  -- rocprofv3 kernel trace of one bench run + per-kernel summary
tag=$1; shift
export TMPDIR=/tmp
R=$PWD
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
f=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
echo "== $tag"
python scripts/prof_summary.py $f | sed -n 3,7p
