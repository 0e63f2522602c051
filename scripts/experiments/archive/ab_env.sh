#!/bin/bash
# dev tool: A/B one environment switch on one box: scripts/ab_env.sh VAR [frames]
V=$1; F=${2:-512}
for r in 1 2; do for on in 0 1; do
  if [ $on = 1 ]; then export $V=1; else unset $V; fi
  for i in 1 2 3; do
  timeout -k 10 200 python bench.py --frames $F --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --no-single-frame 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V=$on', 'fibre ms %.2f'%d['config']['fibre_ms_per_step'], d['config']['bit_errors_xy'])"
  done
done; done
