#!/bin/bash
# dev tool: everything the round-end driver runs, plus the profiles that get copied into profiles/
mkdir -p gpurun_out/final
export TMPDIR=/tmp
R=$PWD
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final/gpu_tests.log 2>&1 || { tail -20 gpurun_out/final/gpu_tests.log; exit 1; }
tail -2 gpurun_out/final/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1 || { tail -20 gpurun_out/final/smoke.log; exit 1; }
tail -1 gpurun_out/final/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err || { tail gpurun_out/final/bench_default.err; exit 1; }
tail -c 600 gpurun_out/final/bench_default.json; echo
timeout -k 10 600 python bench.py --frontend cohmix --no-cpu-baseline > gpurun_out/final/bench_cohmix.json 2>/dev/null || exit 1
timeout -k 10 600 python bench.py --mc --no-cpu-baseline > gpurun_out/final/bench_mc.json 2>/dev/null || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame > /dev/null 2>&1 || exit 1
f=$(find gpurun_out/final/prof -name "*kernel_trace.csv" | head -1)
python scripts/prof_summary.py $f > gpurun_out/final/kernel_trace_summary.md
g=$(find gpurun_out/final/prof -name "*kernel_stats.csv" | head -1)
[ -n "$g" ] && cp $g gpurun_out/final/kernel_stats.csv
head -8 gpurun_out/final/kernel_trace_summary.md
rm -rf gpurun_out/final/prof
