#!/bin/bash
# round 3: out-of-place sweeps.  A second working copy: the column sweep reads W and writes V, the row pass reads V
# and writes W.  Parity subset first (results must be bit-equal to the in-place sweeps), then a same-box A/B.
O=gpurun_out/r03ev; mkdir -p $O
PLX_SSFM_WPAD=0 PLX_SSFM_PINGPONG=1 timeout -k 10 500 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -k "c1_ or ladder_batch or matrix_ssfm_gateway or busy_stream or sentinel or 2pow20" > $O/pp_tests.txt 2>&1; rc=$?; echo "tests (ping-pong) rc=$rc"; tail -3 $O/pp_tests.txt
[ $rc -eq 0 ] || exit 1
run() { local label=$1; shift
  timeout -k 10 150 python3 bench.py "$@" --steps 3 --warmup 1 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame --no-gateway --no-cohmix-line 2> $O/err_$label.txt | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['config']['fibre_ms_per_step']; k=d['roofline']['kernels']; g=d['roofline']['step_group']; print('$label fibre ms %.2f  group frac %.3f  '%(f, g['frac_of_8TBs']) + '  '.join('%s %.1f us x%d'%(n, v['avg_launch_us'], v['active_launches']) for n, v in k.items()))" || tail -5 $O/err_$label.txt
}
for rep in 1 2; do
  PLX_SSFM_WPAD=-1 run c1_inplace --frames 1024
  PLX_SSFM_WPAD=0 run c1_work --frames 1024
  PLX_SSFM_WPAD=0 PLX_SSFM_PINGPONG=1 run c1_pingpong --frames 1024
done
for rep in 1 2; do
  PLX_SSFM_WPAD=-1 run m20_inplace --nsymb 16384 --frames 16
  PLX_SSFM_WPAD=0 run m20_work --nsymb 16384 --frames 16
  PLX_SSFM_WPAD=0 PLX_SSFM_PINGPONG=1 run m20_pingpong --nsymb 16384 --frames 16
done
