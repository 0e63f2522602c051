#!/bin/bash
# round 3: row pass over the listed frames in descending order (PLX_SSFM_ROW_REV=1): 2^20 frames, the ladder, PMD, default bench
O=gpurun_out/r03rr; mkdir -p $O
line() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', 'ms/step %.2f fibre %.2f' % (d['ms_per_step'], d['config']['fibre_ms_per_step']), {k:(round(v['avg_launch_us'],1)) for k,v in r['kernels'].items()})"; }
C="--no-cpu-baseline --no-single-frame --no-gateway --no-cohmix-line --mc-rounds 0"
for rep in 1 2; do for m in 0 1; do
  PLX_SSFM_ROW_REV=$m timeout -k 10 300 python3 bench.py --nsymb 16384 --frames 16 --steps 2 --warmup 1 --variants 1 $C --no-overlap > $O/b_$m.json 2>/dev/null && line $O/b_$m.json "2^20 x16       rev=$m"
  PLX_SSFM_ROW_REV=$m timeout -k 10 300 python3 bench.py --nsymb 16384 --frames 8 --spans 10 --power-ladder --steps 1 --warmup 0 --variants 1 $C > $O/c_$m.json 2>/dev/null && line $O/c_$m.json "2^20 ladder8x10 rev=$m"
  PLX_SSFM_ROW_REV=$m timeout -k 10 300 python3 bench.py --mc --steps 3 --warmup 1 $C > $O/d_$m.json 2>/dev/null && line $O/d_$m.json "PMD x1024      rev=$m"
  PLX_SSFM_ROW_REV=$m timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 $C > $O/e_$m.json 2>/dev/null && line $O/e_$m.json "C1 default     rev=$m"
  PLX_SSFM_ROW_REV=$m timeout -k 10 300 python3 bench.py --power-ladder --steps 3 --warmup 1 $C > $O/f_$m.json 2>/dev/null && line $O/f_$m.json "C1 ladder      rev=$m"
done; done
