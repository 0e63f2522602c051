#!/bin/bash
# dev tool: repeat the fibre pass a few times (A/B by alternating builds or env between calls)
F=${1:-512}
for v in 1 2 3; do
  timeout -k 10 200 python bench.py --frames $F --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --no-single-frame 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('F=$F run $v', 'fibre ms %.2f'%d['config']['fibre_ms_per_step'], d['config']['bit_errors_xy'])"
done
