#!/bin/bash
# dev tool: everything the round-end driver runs, plus the profiles that get copied into profiles/
# usage: scripts/final_round.sh <tag>     (outputs under gpurun_out/<tag>/)
tag=${1:-final}
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail $O/bench_default.err; exit 1; }
timeout -k 10 600 python bench.py --no-overlap --no-cpu-baseline --mc-rounds 0 > $O/bench_no_overlap.json 2>/dev/null || exit 1
timeout -k 10 600 python bench.py --frontend cohmix --no-cpu-baseline --mc-rounds 0 > $O/bench_cohmix.json 2>/dev/null || exit 1
timeout -k 10 600 python bench.py --power-ladder --no-cpu-baseline --mc-rounds 0 > $O/bench_ladder.json 2>/dev/null || exit 1
timeout -k 10 600 python bench.py --mc --no-cpu-baseline --mc-rounds 4 --mc-frames 512 > $O/bench_mc.json 2>/dev/null || exit 1
timeout -k 10 600 python bench.py --nsymb 16384 --frames 16 --steps 2 --warmup 1 --variants 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 > $O/bench_c4_frame.json 2>/dev/null || exit 1
timeout -k 10 900 python bench.py --nsymb 16384 --frames 8 --spans 40 --power-ladder --steps 1 --warmup 0 --variants 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 > $O/bench_c4_40spans.json 2>/dev/null || exit 1
cat $O/bench_default.json $O/bench_no_overlap.json $O/bench_cohmix.json $O/bench_ladder.json $O/bench_mc.json $O/bench_c4_frame.json $O/bench_c4_40spans.json > $O/bench.jsonl
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 > /dev/null 2>&1 || exit 1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python scripts/prof_summary.py $f > $O/kernel_trace_summary.md
g=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$g" ] && head -12 $g > $O/kernel_stats_head.csv
head -8 $O/kernel_trace_summary.md
rm -rf $O/prof
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-overlap > /dev/null 2>&1 || exit 1
f=$(find $O/prof2 -name "*kernel_trace.csv" | head -1)
python scripts/prof_summary.py $f > $O/kernel_trace_no_overlap_summary.md
head -6 $O/kernel_trace_no_overlap_summary.md
rm -rf $O/prof2
scripts/traffic_pmc.sh 256 > $O/traffic.log 2>&1 && cp gpurun_out/traffic/traffic.json $O/traffic.json
tail -3 $O/traffic.log
python - <<PY
import json
for l in open("$O/bench.jsonl"):
    d = json.loads(l); c = d["config"]; r = d["roofline"]
    print("%.4f Gs/s %.1f ms/step fibre %.1f rx %.1f | %s | group %.3f | mc %s" % (d["value"], d["ms_per_step"], c["fibre_ms_per_step"], c["rxdsp_ms_per_step"],
          {k: (round(v["avg_launch_us"]), round(v["frac_of_8TBs"], 3)) for k, v in r["kernels"].items()}, r["step_group"]["frac_of_8TBs"],
          d["mc"] and round(d["mc"]["realisations_per_s"])))
PY
