#!/bin/bash
# same-box A/B of the k_row4k workgroup map on config[4]'s ladder line (8 frames of 2^20 samples, 40 spans) and on 8 / 16 plain frames
O=gpurun_out/r03map2; mkdir -p $O
line() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', 'fibre ms/step %.1f' % d['config']['fibre_ms_per_step'], {k:(round(v['avg_launch_us'],1)) for k,v in r['kernels'].items()})"; }
for rep in 1 2; do for n in old base; do
  timeout -k 10 300 python3 scripts/experiments/bench_with_lib.py $n --nsymb 16384 --frames 8 --spans 10 --power-ladder --steps 1 --warmup 0 --variants 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway > $O/l_$n.json 2>/dev/null && line $O/l_$n.json "ladder8x10 $n"
  timeout -k 10 300 python3 scripts/experiments/bench_with_lib.py $n --nsymb 16384 --frames 8 --steps 2 --warmup 1 --variants 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway > $O/p_$n.json 2>/dev/null && line $O/p_$n.json "plain8 $n"
done; done
