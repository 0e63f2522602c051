#!/bin/bash
# round 3: does a padded row pitch (plan-private working copy) even out the slow tiles (bx = 3 mod 8) of the C1 column sweep?
mkdir -p gpurun_out/r03ev
PLX_SSFM_WPAD=8 timeout -k 10 500 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -k "c1_ or ladder_batch or matrix_ssfm_gateway or busy_stream or sentinel" > gpurun_out/r03ev/wpadc1_tests.txt 2>&1; echo "tests (WPAD=8) rc=$?"; tail -3 gpurun_out/r03ev/wpadc1_tests.txt
run() { local label=$1; shift
  timeout -k 10 150 python3 bench.py "$@" --steps 3 --warmup 1 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame --no-gateway --no-cohmix-line 2> gpurun_out/r03ev/err_$label.txt | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['config']['fibre_ms_per_step']; k=d['roofline']['kernels']; g=d['roofline']['step_group']; print('$label fibre ms %.2f  group frac %.3f  '%(f, g['frac_of_8TBs']) + '  '.join('%s %.1f us x%d'%(n, v['avg_launch_us'], v['active_launches']) for n, v in k.items()))" || tail -5 gpurun_out/r03ev/err_$label.txt
}
for rep in 1 2; do for W in -1 0 8 24 72 264; do PLX_SSFM_WPAD=$W run c1_wpad$W --frames 1024; done; done
PLX_SSFM_WPAD=8 bash scripts/experiments/stamps.sh run 1024 no 1024 2>&1 | grep -v amdgpu.ids | grep -E "F=|sum|all slots|by workgroup id|by tile"
