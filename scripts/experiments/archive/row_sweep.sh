#!/bin/bash
# dev tool: rows per workgroup x workgroup size of k_row (fused default path), fibre ms per pass
F=${1:-256}
for R in 1 2 4 8; do for T in 64 128 256 512; do
  PLX_SSFM_ROWS=$R PLX_SSFM_ROW_THREADS=$T timeout -k 10 120 python bench.py --frames $F --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --no-single-frame 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('R=$R T=$T F=$F fibre ms %.2f'%d['config']['fibre_ms_per_step'], d['config']['bit_errors_xy'])" || echo "R=$R T=$T failed"
done; done
