#!/bin/bash
# dev tool: fused column sweep vs plain sweeps at the bench geometry.  usage: fuse_sweep.sh FRAMES
F=${1:-256}
one() {
  env "$@" timeout -k 10 300 python bench.py --frames $F --steps 3 --warmup 1 --no-cpu-baseline --no-overlap 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('F=$F $*', 'fibre ms %.2f'%d['config']['fibre_ms_per_step'], 'frac %.3f'%d['roofline']['frac'], 'errs', d['config']['bit_errors_xy'])"
}
one PLX_X=0 && one PLX_SSFM_FUSE=1 && one PLX_X=0 && one PLX_SSFM_FUSE=1
