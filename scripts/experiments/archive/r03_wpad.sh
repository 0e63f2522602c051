#!/bin/bash
# round 3: plan-private working copy with a padded row pitch for 2^20-sample frames: parity first, then A/B on one box
mkdir -p gpurun_out/r03ev
timeout -k 10 500 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -k "2pow20 or c4" > gpurun_out/r03ev/wpad_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03ev/wpad_tests.txt
run() { # label -- bench args
  local label=$1; shift
  timeout -k 10 150 python3 bench.py "$@" --steps 3 --warmup 1 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame 2> gpurun_out/r03ev/err_$label.txt | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['config']['fibre_ms_per_step']; k=d['roofline']['kernels']; g=d['roofline']['step_group']; print('$label fibre ms %.2f  group frac %.3f  '%(f, g['frac_of_8TBs']) + '  '.join('%s %.1f us x%d'%(n, v['avg_launch_us'], v['active_launches']) for n, v in k.items()))" || tail -5 gpurun_out/r03ev/err_$label.txt
}
for rep in 1 2; do
for W in -1 0 8 264 72 2056; do PLX_SSFM_WPAD=$W run c4_wpad$W --nsymb 16384 --frames 16; done
done
