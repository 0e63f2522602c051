#!/bin/bash
# round 3: trunk phasor tables for PMD plans: parity, then A/B on one box
mkdir -p gpurun_out/r03ev
timeout -k 10 700 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_mc_sharding.py -x -q -k "pmd or c3 or wdm_16ch or plates or campaign or gps or ex24 or inverse" > gpurun_out/r03ev/pmdtab_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03ev/pmdtab_tests.txt
run() { local label=$1; shift
  timeout -k 10 200 python3 bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-single-frame --no-gateway --no-cohmix-line 2> gpurun_out/r03ev/err_$label.txt | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['config']['fibre_ms_per_step']; k=d['roofline']['kernels']; g=d['roofline']['step_group']; m=d['mc']; print('$label value %.4f fibre ms %.2f  group frac %.3f  '%(d['value'], f, g['frac_of_8TBs']) + '  '.join('%s %.1f us x%d'%(n, v['avg_launch_us'], v['active_launches']) for n, v in k.items()) + ('  mc %.0f/s' % m['realisations_per_s'] if m else ''))" || tail -5 gpurun_out/r03ev/err_$label.txt
}
for rep in 1 2; do
  PLX_SSFM_NO_PMD_TAB=1 run mc_exp --mc --frames 1024
  run mc_tab --mc --frames 1024
done
