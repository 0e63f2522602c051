#!/bin/bash
# dev tool: kernel trace of an overlapped and a no-overlap pass, gaps between the kernels of the fibre's step loop
export TMPDIR=/tmp
R=$PWD
for mode in "" "--no-overlap"; do
rm -rf gpurun_out/gapprof
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gapprof -- python3 bench.py --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 $mode > gpurun_out/gap_bench.json 2>/dev/null
f=$(find gpurun_out/gapprof -name "*kernel_trace.csv" | head -1)
echo "== bench.py $mode"
python scripts/experiments/fibre_gaps.py $f
python -c "import json; d=json.loads(open('gpurun_out/gap_bench.json').read().strip().splitlines()[-1]); print('fibre ms', d['config']['fibre_ms_per_step'], 'rx ms', d['config']['rxdsp_ms_per_step'], 'step', d['ms_per_step'], {k: round(v['avg_launch_us']) for k, v in d['roofline']['kernels'].items()})"
done
rm -rf gpurun_out/gapprof
