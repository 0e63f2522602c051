"""dev: a lone frame's step, kernel by kernel (HIP events between the launches: adds ~10 us per step to the total)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from polmux_amd import pipeline
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
hp = pipeline.HotPath(pipeline.HotPathConfig(variants=1), max_frames=F)
for prof in (False, True):
    hp.profile(prof)
    for it in range(3):
        ux, uy = hp.make_batch(F)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hp.fibre(ux, uy)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print("F=%d profile=%s fibre %.3f ms" % (F, prof, (t1 - t0) * 1e3), flush=True)
ms, k = hp.kernel_times()
for i, n in enumerate(("column sweep", hp.row_kernel(), "k_col_inv", "control")):
    if k[i]:
        print("  %-14s %7.2f us x %d" % (n, ms[i] / k[i] * 1e3, k[i]))
hp.close()
