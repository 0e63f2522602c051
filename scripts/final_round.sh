#!/bin/bash
# dev tool: the bench lines of record and the profiles that get copied into profiles/ (three GPU calls: gpurun allows 20 min each)
# usage: scripts/final_round.sh lines|trace|traffic <tag>     (outputs under gpurun_out/<tag>/)
part=${1:-lines}; tag=${2:-final}
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
if [ $part = lines ]; then
  timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail $O/bench_default.err; exit 1; }
  timeout -k 10 300 python3 bench.py --no-overlap --no-cpu-baseline --mc-rounds 0 --no-gateway --no-cohmix-line --configs no > $O/bench_no_overlap.json 2>/dev/null || exit 1
  timeout -k 10 300 python3 bench.py --frontend cohmix --no-cpu-baseline --mc-rounds 0 --no-gateway > $O/bench_cohmix.json 2>/dev/null || exit 1
  timeout -k 10 300 python3 bench.py --power-ladder --no-cpu-baseline --mc-rounds 0 --no-gateway > $O/bench_ladder.json 2>/dev/null || exit 1
  timeout -k 10 300 python3 bench.py --mc --no-cpu-baseline --no-gateway > $O/bench_mc.json 2>/dev/null || exit 1
  timeout -k 10 300 python3 bench.py --nsymb 16384 --frames 16 --steps 4 --warmup 1 --variants 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway > $O/bench_c4_frame.json 2>/dev/null || exit 1
  timeout -k 10 300 python3 bench.py --nsymb 16384 --frames 16 --steps 4 --warmup 1 --variants 1 --share-device no --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway > $O/bench_c4_frame_fused.json 2>/dev/null || exit 1
  timeout -k 10 600 python3 bench.py --nsymb 16384 --frames 8 --spans 40 --power-ladder --ladder-world 8 --steps 2 --warmup 1 --variants 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway > $O/bench_c4_40spans.json 2>/dev/null || exit 1
  timeout -k 10 600 python3 bench.py --nch 16 --spans 10 --nf 5 --frames 32 --steps 3 --warmup 1 --mc-rounds 0 --no-gateway > $O/bench_c2.json 2>/dev/null || exit 1
  timeout -k 10 300 python3 bench.py --nsymb 4096 --frames 256 --steps 4 --warmup 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway > $O/bench_2pow18.json 2>/dev/null || exit 1
  timeout -k 10 300 python3 bench.py --nsymb 16384 --flag gps- --frames 16 --steps 3 --warmup 1 --variants 1 --share-device no --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway > $O/bench_2pow20_pmd.json 2>/dev/null || exit 1
  cat $O/bench_default.json $O/bench_no_overlap.json $O/bench_cohmix.json $O/bench_ladder.json $O/bench_mc.json $O/bench_c4_frame.json $O/bench_c4_frame_fused.json $O/bench_c4_40spans.json $O/bench_c2.json $O/bench_2pow18.json $O/bench_2pow20_pmd.json > $O/bench.jsonl
  python3 - <<PY
import json
for l in open("$O/bench.jsonl"):
    d = json.loads(l); c = d["config"]; r = d["roofline"]
    print("%.4f Gs/s %.1f ms/step fibre %.1f rx %.1f | %s | group %.3f | mc %s | %s" % (d["value"], d["ms_per_step"], c["fibre_ms_per_step"], c["rxdsp_ms_per_step"],
          {k: (round(v["avg_launch_us"]), round(v["frac_of_8TBs"], 3)) for k, v in r["kernels"].items()}, r["step_group"]["frac_of_8TBs"],
          d["mc"] and round(d["mc"]["realisations_per_s"]), c["workload"][:60]))
PY
fi
if [ $part = trace ]; then
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway --no-cohmix-line --configs no > /dev/null 2>&1 || exit 1
  f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
  python3 scripts/prof_summary.py $f > $O/kernel_trace_summary.md
  g=$(find $O/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$g" ] && head -14 $g > $O/kernel_stats_head.csv
  head -8 $O/kernel_trace_summary.md
  rm -rf $O/prof
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-overlap --no-gateway --no-cohmix-line --configs no > /dev/null 2>&1 || exit 1
  f=$(find $O/prof2 -name "*kernel_trace.csv" | head -1)
  python3 scripts/prof_summary.py $f > $O/kernel_trace_no_overlap_summary.md
  g=$(find $O/prof2 -name "*kernel_stats.csv" | head -1)
  [ -n "$g" ] && head -14 $g > $O/kernel_stats_no_overlap_head.csv
  head -6 $O/kernel_trace_no_overlap_summary.md
  rm -rf $O/prof2
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof3 -- python3 bench.py --nsymb 16384 --frames 16 --steps 2 --warmup 1 --variants 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-overlap --no-gateway > /dev/null 2>&1 || exit 1
  f=$(find $O/prof3 -name "*kernel_trace.csv" | head -1)
  python3 scripts/prof_summary.py $f > $O/kernel_trace_2pow20_summary.md
  head -6 $O/kernel_trace_2pow20_summary.md
  rm -rf $O/prof3
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof4 -- python3 bench.py --nch 16 --spans 2 --nf 5 --frames 32 --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-overlap --no-gateway > /dev/null 2>&1 || exit 1
  f=$(find $O/prof4 -name "*kernel_trace.csv" | head -1)
  python3 scripts/prof_summary.py $f > $O/kernel_trace_wdm16_summary.md
  head -6 $O/kernel_trace_wdm16_summary.md
  rm -rf $O/prof4
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof5 -- python3 bench.py --nsymb 4096 --frames 256 --steps 2 --warmup 1 --variants 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-overlap --no-gateway > /dev/null 2>&1 || exit 1
  f=$(find $O/prof5 -name "*kernel_trace.csv" | head -1)
  python3 scripts/prof_summary.py $f > $O/kernel_trace_2pow18_summary.md
  head -6 $O/kernel_trace_2pow18_summary.md
  rm -rf $O/prof5
fi
if [ $part = traffic ]; then
  scripts/traffic_pmc.sh 256 > $O/traffic.log 2>&1 && cp gpurun_out/traffic/traffic.json $O/traffic.json
  tail -3 $O/traffic.log
fi
