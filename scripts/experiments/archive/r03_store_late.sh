#!/bin/bash
# round 3: k_colx16 with a tile's stores held back until the next tile has landed (PLX_SSFM_STORE_LATE), same box
O=gpurun_out/r03sl; mkdir -p $O
line() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', 'fibre ms/step %.2f' % d['config']['fibre_ms_per_step'], {k:(round(v['avg_launch_us'],1)) for k,v in r['kernels'].items()})"; }
for rep in 1 2; do for m in 0 1; do
  PLX_SSFM_STORE_LATE=$m timeout -k 10 300 python3 bench.py --nsymb 16384 --frames 16 --steps 2 --warmup 1 --variants 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway --no-overlap > $O/c4_$m.json 2>/dev/null && line $O/c4_$m.json "2^20 x16  late=$m"
  PLX_SSFM_STORE_LATE=$m timeout -k 10 300 python3 bench.py --nsymb 16384 --frames 8 --spans 10 --power-ladder --steps 1 --warmup 0 --variants 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway > $O/l_$m.json 2>/dev/null && line $O/l_$m.json "ladder 8x10 late=$m"
  PLX_SSFM_STORE_LATE=$m timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway --no-overlap --no-cohmix-line > $O/c1_$m.json 2>/dev/null && line $O/c1_$m.json "C1 x1024  late=$m"
done; done
