#!/bin/bash
# round 3: late stores in k_colx16, multi-team launches: C1 at several batch sizes, beside the receiver, PMD / Monte-Carlo line
O=gpurun_out/r03sl; mkdir -p $O
line() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', 'ms/step %.2f fibre %.2f' % (d['ms_per_step'], d['config']['fibre_ms_per_step']), {k:(round(v['avg_launch_us'],1)) for k,v in r['kernels'].items()}, d['mc'] and round(d['mc']['realisations_per_s']))"; }
C="--no-cpu-baseline --no-single-frame --no-gateway --no-cohmix-line"
for rep in 1 2; do for m in 0 1; do
  PLX_SSFM_STORE_LATE=$m timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 $C --mc-rounds 0 --no-overlap > $O/a_$m.json 2>/dev/null && line $O/a_$m.json "C1 x1024 alone    late=$m"
  PLX_SSFM_STORE_LATE=$m timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 $C --mc-rounds 0 > $O/b_$m.json 2>/dev/null && line $O/b_$m.json "C1 x1024 default  late=$m"
  PLX_SSFM_STORE_LATE=$m timeout -k 10 300 python3 bench.py --frames 128 --steps 6 --warmup 2 $C --mc-rounds 0 --no-overlap > $O/c_$m.json 2>/dev/null && line $O/c_$m.json "C1 x128 alone     late=$m"
  PLX_SSFM_STORE_LATE=$m timeout -k 10 300 python3 bench.py --mc --steps 3 --warmup 1 $C > $O/d_$m.json 2>/dev/null && line $O/d_$m.json "PMD x1024 + MC    late=$m"
  PLX_SSFM_STORE_LATE=$m timeout -k 10 300 python3 bench.py --power-ladder --steps 3 --warmup 1 $C --mc-rounds 0 > $O/e_$m.json 2>/dev/null && line $O/e_$m.json "ladder x1024      late=$m"
done; done
