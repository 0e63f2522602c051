#!/bin/bash
# dev experiment: core clock / power while the fibre runs (is the column sweep power-capped?)
# usage: gpurun -- bash scripts/experiments/clock_watch.sh
mkdir -p gpurun_out
rocm-smi --showclocks --showpower --showmaxpower > gpurun_out/clock_idle.txt 2>&1
( for i in $(seq 1 40); do echo "--- $i"; rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|fclk\|power"; sleep 0.5; done ) > gpurun_out/clock_watch.txt 2>&1 &
W=$!
timeout -k 10 200 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from polmux_amd import pipeline
F = 1024
hp = pipeline.HotPath(pipeline.HotPathConfig(flag="g-s-"), max_frames=F)
hp.profile(True)
for r in range(40):
    ux, uy = hp.make_batch(F)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hp.fibre(ux, uy)
    torch.cuda.synchronize(); print("pass %d %.2f ms" % (r, (time.perf_counter() - t0) * 1e3), flush=True)
hp.close()
PY
wait $W
