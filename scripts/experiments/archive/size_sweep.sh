#!/bin/bash
# dev tool: bench across frame sizes at constant total samples per step
for cfg in "16384 64 16" "4096 64 64" "1024 64 256" "256 64 1024"; do set -- $cfg
  timeout -k 10 300 python bench.py --nsymb $1 --nt $2 --frames $3 --steps 2 --warmup 1 --no-cpu-baseline --no-overlap 2>/dev/null | tail -1 | \
  python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('nsymb $1 F $3', 'Gs/s %.4f'%d['value'], 'fibre ms %.2f'%c['fibre_ms_per_step'], 'rx ms %.2f'%c['rxdsp_ms_per_step'], 'steps %.1f'%c['ssfm_steps_per_frame'], 'frac %.3f'%d['roofline']['frac'], 'errs', c['bit_errors_xy'])"
done
