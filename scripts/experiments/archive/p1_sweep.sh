#!/bin/bash
# dev tool: plain three-sweep step with 256-row (p1=8) against 128-row (p1=7) column tiles; per-kernel averages from the trace
export TMPDIR=/tmp
F=${1:-256}
for p1 in 8 7; do
for ct in 256 512; do
  export PLX_SSFM_NO_FUSE=1 PLX_SSFM_P1=$p1 PLX_SSFM_COL_THREADS=$ct
  timeout -k 10 200 python bench.py --frames $F --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --no-single-frame 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('p1=$p1 colthr=$ct F=$F fibre ms %.2f'%d['config']['fibre_ms_per_step'], d['config']['bit_errors_xy'])"
done; done
export PLX_SSFM_P1=7 PLX_SSFM_COL_THREADS=256
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/p1prof -- python3 bench.py --frames $F --steps 2 --warmup 1 --no-cpu-baseline --no-overlap --no-single-frame > /dev/null 2>&1
g=$(find gpurun_out/p1prof -name "*kernel_stats.csv" | head -1); head -5 $g | cut -c1-150
rm -rf gpurun_out/p1prof
