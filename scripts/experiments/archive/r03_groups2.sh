#!/bin/bash
# round 3, after the register form of the row pass: cache-resident frame groups again (the row pass now runs at the streaming
# rate of HBM -- does it run faster from the Infinity Cache?), and a lone frame's fibre time
mkdir -p gpurun_out/r03f
run() { # label env... -- bench args
  local label=$1; shift
  timeout -k 10 150 python3 bench.py "$@" --steps 3 --warmup 1 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame --no-gateway --no-cohmix-line 2> gpurun_out/r03f/err_$label.txt | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['config']['fibre_ms_per_step']; k=d['roofline']['kernels']; g=d['roofline']['step_group']; print('$label fibre ms %.2f  group frac %.3f  '%(f, g['frac_of_8TBs']) + '  '.join('%s %.1f us x%d'%(n, v['avg_launch_us'], v['active_launches']) for n, v in k.items()))" || tail -5 gpurun_out/r03f/err_$label.txt
}
for rep in 1 2; do for G in 0 128 192 224; do PLX_SSFM_GROUP_MIB=$G run c1_g$G --frames 1024; done; done
for F in 1 8; do timeout -k 10 90 python3 scripts/experiments/diag_small.py $F 2>&1 | grep -E "fibre|receive" | tail -4; done
