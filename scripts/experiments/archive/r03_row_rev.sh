#!/bin/bash
# round 3: k_row256r over the listed frames in descending order (PLX_SSFM_ROW_REV=1), same box A/B, with FETCH_SIZE
export TMPDIR=/tmp
R=$PWD; O=gpurun_out/r03rr; mkdir -p $O
line() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', 'ms/step %.2f fibre %.2f' % (d['ms_per_step'], d['config']['fibre_ms_per_step']), {k:(round(v['avg_launch_us'],1)) for k,v in r['kernels'].items()})"; }
C="--no-cpu-baseline --no-single-frame --no-gateway --no-cohmix-line --mc-rounds 0"
for rep in 1 2 3; do for m in 0 1; do
  PLX_SSFM_ROW_REV=$m timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 $C --no-overlap > $O/a_$m.json 2>/dev/null && line $O/a_$m.json "C1 x1024 alone rev=$m"
done; done
for m in 0 1; do
  PLX_SSFM_ROW_REV=$m timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/p_$m -- python3 bench.py --frames 1024 --steps 1 --warmup 0 --variants 1 $C --no-overlap > /dev/null 2>&1
  f=$(find $O/p_$m -name "*counter_collection.csv" | head -1)
  echo "rev=$m"; python scripts/pmc_summary.py $f | grep -E "k_colx16|k_row"
  rm -rf $O/p_$m
done
