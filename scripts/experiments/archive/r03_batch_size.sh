#!/bin/bash
# round 3: the bench step at other batch sizes (does the Infinity Cache share of a smaller batch pay for its tails?)
O=gpurun_out/r03bs; mkdir -p $O
line() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', 'Gs/s %.4f ms/step %.2f fibre %.2f' % (d['value'], d['ms_per_step'], d['config']['fibre_ms_per_step']), {k:(round(v['avg_launch_us'],1), round(v['frac_of_8TBs'],3)) for k,v in r['kernels'].items()}, 'group %.3f' % r['step_group']['frac_of_8TBs'])"; }
C="--no-cpu-baseline --no-single-frame --no-gateway --no-cohmix-line --mc-rounds 0"
for F in 1024 768 512 384 256 128; do
  S=$((4096 / F)); [ $S -lt 4 ] && S=4
  timeout -k 10 300 python3 bench.py --frames $F --steps $S --warmup 2 $C > $O/d_$F.json 2>/dev/null && line $O/d_$F.json "default F=$F"
  timeout -k 10 300 python3 bench.py --frames $F --steps $S --warmup 2 $C --no-overlap > $O/n_$F.json 2>/dev/null && line $O/n_$F.json "alone   F=$F"
done
