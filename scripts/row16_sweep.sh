#!/bin/bash
# dev tool: register-blocked row kernel vs the general one at the bench geometry
mkdir -p gpurun_out
one() {
  env "$@" timeout -k 10 200 python bench.py --frames 256 --steps 3 --warmup 1 --no-cpu-baseline --no-overlap 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'Gs/s %.4f'%d['value'], 'fibre ms %.2f'%d['config']['fibre_ms_per_step'], 'frac %.3f'%d['roofline']['frac'], 'errs', d['config']['bit_errors_xy'])"
}
one PLX_X=0 && one PLX_SSFM_ROW16=1
