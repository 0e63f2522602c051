#!/bin/bash
mkdir -p gpurun_out/r03ev
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_mex_shims.py -x -q -k "graph_replay or small_ladder or ladder_batch or c0 or c1_ or gateway or shim or wdm_16ch or adaptive or scalar" > gpurun_out/r03ev/graph_tests.txt 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r03ev/graph_tests.txt
for e in "" "PLX_SSFM_NO_GRAPH=1"; do
  env $e timeout -k 10 90 python3 scripts/experiments/diag_small.py 1 2>&1 | grep -E "fibre" | tail -2 | sed "s/^/[$e] /"
  env $e timeout -k 10 90 python3 scripts/experiments/diag_small.py 8 2>&1 | grep -E "fibre" | tail -1 | sed "s/^/[$e] /"
  env $e timeout -k 10 90 python3 scripts/experiments/diag_small.py 32 2>&1 | grep -E "fibre" | tail -1 | sed "s/^/[$e] /"
done
timeout -k 10 100 python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import bench, torch
from polmux_amd import pipeline
cfg = pipeline.HotPathConfig()
hp = pipeline.HotPath(cfg, max_frames=1)
g = bench.gateway_bench(cfg, hp)
print({k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk != "what"} for k, v in g.items() if k.startswith("plx_")})
PY
