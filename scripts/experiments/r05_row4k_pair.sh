#!/bin/bash
# round 5: k_row4k<pair> with only the table form of the trunk loop (default) against the two-form kernel, alternating
O=gpurun_out/r05_row4k_pair; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_configs.py -q -x -m gpu -k "pmd_2pow20" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
A="--nsymb 16384 --flag gps- --frames 16 --steps 3 --warmup 1 --variants 1 --share-device no --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway --configs no"
for rep in 1 2; do
  for v in 1 0; do
    python scripts/experiments/bench_tuned.py row4k_pair_tab=$v -- $A 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('pair_tab=$v', '%.4f Gs/s fibre %.1f ms' % (d['value'], d['config']['fibre_ms_per_step']), {k:(round(v['avg_launch_us'],1), round(v['frac_of_8TBs'],3)) for k,v in r['kernels'].items()})"
  done
done | tee $O/ab.txt
