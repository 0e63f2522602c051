#!/bin/bash
# dev experiment: what does the Kerr arithmetic cost k_colx16?  CNLSE (two sincos + rotation, 704 FP64 instructions per lane and tile)
# against Manakov (one sincos, 288), with the core clock and socket power sampled beside each.
mkdir -p gpurun_out
for MK in no yes; do
( for i in $(seq 1 16); do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Power (W)" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/clock_$MK.txt 2>&1 &
W=$!
MK=$MK timeout -k 10 200 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from polmux_amd import pipeline
F = 1024
hp = pipeline.HotPath(pipeline.HotPathConfig(flag="g-s-", manakov=os.environ["MK"]), max_frames=F)
hp.profile(True)
ts = []
for r in range(30):
    ux, uy = hp.make_batch(F)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hp.fibre(ux, uy)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
ms, n = hp.kernel_times()
print("manakov=%s: fibre best %.2f median %.2f ms  col %.1f us x%d  row %.1f us x%d  ncycle %d" % (os.environ["MK"], min(ts[2:]), sorted(ts[2:])[14],
      ms[0] / max(n[0], 1) * 1e3, n[0], ms[1] / max(n[1], 1) * 1e3, n[1], hp.last_ncycle(F)[0]), flush=True)
hp.close()
PY
wait $W
done
