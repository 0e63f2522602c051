#!/bin/bash
# round 5: core clock and socket power (rocm-smi, every 0.3 s) under (a) the FP64 FMA microbenchmark, (b) the memory-only tile streamer,
# (c) the fibre loop alone (k_colx16 + k_row256r), (d) the default bench step (receiver beside)
O=gpurun_out/r05_power; mkdir -p $O
watch() {  # tag, command...
  tag=$1; shift
  ( while true; do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Package Power" | tr '\n' ' '; echo; sleep 0.3; done ) > $O/$tag.txt 2>&1 &
  W=$!
  "$@" > $O/$tag.out 2>&1
  kill $W; wait $W 2>/dev/null
  python3 - <<PY
import re
t = open("$O/$tag.txt").read()
rows = [(int(a), float(b)) for a, b in re.findall(r"sclk clock level: \w+: \((\d+)Mhz\).*?Package Power \(W\): ([\d.]+)", t)]
busy = [r for r in rows if r[1] > 500][2:]      # (skip the ramp)
if busy:
    print("%-12s %3d samples: sclk %4.0f MHz (min %d, max %d), power %4.0f W (max %.0f)" % ("$tag", len(busy), sum(r[0] for r in busy) / len(busy), min(r[0] for r in busy), max(r[0] for r in busy), sum(r[1] for r in busy) / len(busy), max(r[1] for r in busy)))
else:
    print("$tag: no busy samples", rows[:5])
PY
}
hipcc --offload-arch=gfx950 -O3 scripts/experiments/micro/fp64_peak.hip -o /tmp/fp64_peak 2>/dev/null
hipcc --offload-arch=gfx950 -O3 scripts/experiments/micro/xcd_speed.hip -o /tmp/xcd_speed 2>/dev/null
watch fp64_fma /tmp/fp64_peak 60
watch mem_tiles /tmp/xcd_speed 6000
watch fibre python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from polmux_amd import pipeline
F = 1024
hp = pipeline.HotPath(pipeline.HotPathConfig(flag="g-s-"), max_frames=F)
for r in range(100):
    ux, uy = hp.make_batch(F)
    hp.fibre(ux, uy)
torch.cuda.synchronize()
hp.close()
PY
watch bench python3 bench.py --steps 60 --warmup 2 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway --no-cohmix-line --configs no
