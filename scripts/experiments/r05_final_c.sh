#!/bin/bash
# round 5, files of record, part C: the default line again on the last build (offline traffic per active launch), and the
# default-run kernel trace with the robust active-launch summary
export TMPDIR=/tmp
O=gpurun_out/r05_final; R=$PWD; mkdir -p $O
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail $O/bench_default.err; exit 1; }
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway --no-cohmix-line --configs no > /dev/null 2>&1 || exit 1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 scripts/prof_summary.py $f > $O/kernel_trace_summary.md
g=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$g" ] && head -14 $g > $O/kernel_stats_head.csv
head -6 $O/kernel_trace_summary.md
rm -rf $O/prof
python3 - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("%.4f Gs/s" % d["value"], {k: (round(v["roofline"]["traffic"] / v["roofline"]["algorithmic_bytes_per_launch"], 3)) for k, v in d["configs"].items()})
PY
