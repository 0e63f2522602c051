#!/bin/bash
# round 3: the one-team fused sweep (k_colx16_solo): parity at 2^20 first, then A/B against k_colx16 on one box
mkdir -p gpurun_out/r03ev
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_mex_shims.py -x -q -k "2pow20 or c4 or wdm_16ch or gateway_state or cma_shim" > gpurun_out/r03ev/solo_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03ev/solo_tests.txt
run() { # label -- bench args
  local label=$1; shift
  timeout -k 10 200 python3 bench.py "$@" --warmup 1 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame --no-gateway 2> gpurun_out/r03ev/err_$label.txt | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['config']['fibre_ms_per_step']; k=d['roofline']['kernels']; g=d['roofline']['step_group']; print('$label fibre ms %.2f  group frac %.3f  '%(f, g['frac_of_8TBs']) + '  '.join('%s %.1f us x%d'%(n, v['avg_launch_us'], v['active_launches']) for n, v in k.items()))" || tail -5 gpurun_out/r03ev/err_$label.txt
}
for rep in 1 2; do
  PLX_SSFM_NO_SOLO=1 run c4_team --nsymb 16384 --frames 16 --steps 3
  run c4_solo --nsymb 16384 --frames 16 --steps 3
done
PLX_SSFM_NO_SOLO=1 run c4_team_8x4 --nsymb 16384 --frames 8 --spans 4 --power-ladder --steps 1
run c4_solo_8x4 --nsymb 16384 --frames 8 --spans 4 --power-ladder --steps 1
timeout -k 10 100 python3 - <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd())
import bench, torch
from polmux_amd import pipeline
cfg = pipeline.HotPathConfig()
hp = pipeline.HotPath(cfg, max_frames=1)
g = bench.gateway_bench(cfg, hp)
print({k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk != "what"} for k, v in g.items() if k.startswith("plx_")})
PY
