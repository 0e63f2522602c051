"""dev: where does the time between two fibre calls of the bench loop go?  host clock around the calls, HIP-event time inside"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from polmux_amd import pipeline
F = 1024
hp = pipeline.HotPath(pipeline.HotPathConfig(variants=16), max_frames=F)
hp.profile(len(sys.argv) > 1 and sys.argv[1] == "profile")
bufs = [hp.make_batch(F) for _ in range(6)]
rxs = torch.cuda.Stream() if (len(sys.argv) > 2 and sys.argv[2] == "rx") else None
torch.cuda.synchronize()
host, gpu, gaps = [], [], []
tprev = None
t00 = time.perf_counter()
for i in range(6):
    ux, uy = bufs[i]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ta = time.perf_counter()
    if tprev is not None:
        gaps.append(ta - tprev)
    e0.record()
    hp.fibre(ux, uy)
    e1.record()
    tb = time.perf_counter()
    if rxs is not None:
        hp.receive(ux, uy, noise_sigma=0.05, noise_seed=i, side_stream=rxs)
    tprev = time.perf_counter() if rxs is None else tb
    host.append(tb - ta); gpu.append((e0, e1))
    if rxs is not None:
        gaps.append(0.0)
torch.cuda.synchronize()
tot = time.perf_counter() - t00
print("profile=%s rx=%s: per step wall %.2f ms; fibre call host %.2f ms, HIP events %.2f ms; host gap between calls %.3f ms" % (
    hp._profiling, rxs is not None, tot / 6 * 1e3, np.mean(host[1:]) * 1e3, np.mean([a.elapsed_time(b) for a, b in gpu[1:]]), np.mean(gaps) * 1e3 if gaps else 0))
hp.close()
