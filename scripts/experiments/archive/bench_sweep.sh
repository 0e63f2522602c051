#!/bin/bash
# usage: scripts/bench_sweep.sh "P1:LOGW ..." frames  -- geometry sweep of the SSFM tile (dev tool)
for cfg in $1; do
  p1=${cfg%%:*}; lw=${cfg##*:}
  PLX_SSFM_P1=$p1 PLX_SSFM_LOGW=$lw timeout -k 10 200 python bench.py --frames $2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('p1=$p1 logW=$lw', 'Gs/s %.4f'%d['value'], 'fibre ms %.2f'%d['config']['fibre_ms_per_step'], 'rx ms %.2f'%d['config']['rxdsp_ms_per_step'], 'frac %.3f'%d['roofline']['frac'])"
done
