"""dev tool: fibre-only time of the C1 batch (no receiver), best of N, optional env variants.
usage: python scripts/experiments/fibre_time.py [frames] [flag]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from polmux_amd import pipeline

F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
flag = sys.argv[2] if len(sys.argv) > 2 else "g-s-"
hp = pipeline.HotPath(pipeline.HotPathConfig(flag=flag), max_frames=F)
hp.profile(True)
ts = []
for r in range(4):
    ux, uy = hp.make_batch(F)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hp.fibre(ux, uy)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ms, n = hp.kernel_times()
print("F=%d %s fused=%s: fibre %.2f ms (best of 3)  col %.1f us x%d  row %.1f us x%d  ncycle %d" % (
    F, flag, hp.fused(), min(ts[1:]) * 1e3, ms[0] / max(n[0], 1) * 1e3, n[0], ms[1] / max(n[1], 1) * 1e3, n[1], hp.last_ncycle(F)[0]))
hp.close()
