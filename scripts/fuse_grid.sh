#!/bin/bash
# dev tool: resident-grid size of the fused column sweep
F=${1:-512}
for g in 512 480 448 384 256; do
  PLX_SSFM_FUSE_GRID=$g timeout -k 10 300 python bench.py --frames $F --steps 3 --warmup 1 --no-cpu-baseline --no-overlap 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('F=$F grid=$g', 'fibre ms %.2f'%d['config']['fibre_ms_per_step'], 'frac %.3f'%d['roofline']['frac'])"
done
