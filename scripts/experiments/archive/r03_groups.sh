#!/bin/bash
# round 3: (1) why did small batches stall in the sweep?  (2) cache-resident frame groups, A/B on one box
mkdir -p gpurun_out/r03ev
for F in 32 48; do
  timeout -k 10 90 python3 scripts/experiments/diag_small.py $F > gpurun_out/r03ev/diag_$F.txt 2>&1; echo "diag F=$F rc=$?"; tail -12 gpurun_out/r03ev/diag_$F.txt
done
run() { # label env... -- bench args
  local label=$1; shift
  timeout -k 10 150 python3 bench.py "$@" --steps 3 --warmup 1 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame 2> gpurun_out/r03ev/err_$label.txt | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['config']['fibre_ms_per_step']; k=d['roofline']['kernels']; g=d['roofline']['step_group']; print('$label fibre ms %.2f  group frac %.3f  '%(f, g['frac_of_8TBs']) + '  '.join('%s %.1f us x%d'%(n, v['avg_launch_us'], v['active_launches']) for n, v in k.items()))" || tail -5 gpurun_out/r03ev/err_$label.txt
}
for G in 0 64 96 128 192; do PLX_SSFM_GROUP_MIB=$G run c1_g$G --frames 1024; done
for G in 0 64 128 192; do PLX_SSFM_GROUP_MIB=$G run c4_g$G --nsymb 16384 --frames 16; done
for G in 0 128; do PLX_SSFM_GROUP_MIB=$G run ladder_g$G --frames 1024 --power-ladder; done
