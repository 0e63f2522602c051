#!/bin/bash
# dev tool: kernel trace of a no-overlap pass and the idle gaps between kernels
export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/gapprof
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gapprof -- python3 bench.py --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-overlap --no-single-frame > gpurun_out/gap_bench.json 2>/dev/null
f=$(find gpurun_out/gapprof -name "*kernel_trace.csv" | head -1)
python scripts/gap_analysis.py $f
python -c "import json; d=json.loads(open('gpurun_out/gap_bench.json').read().strip().splitlines()[-1]); print('fibre ms', d['config']['fibre_ms_per_step'], 'rx ms', d['config']['rxdsp_ms_per_step'], 'step', d['ms_per_step'])"
rm -rf gpurun_out/gapprof
