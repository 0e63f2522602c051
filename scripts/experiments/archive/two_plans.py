"""dev experiment (round 3): two plans of 512 frames propagating side by side on two streams, each fused sweep sized for ONE
workgroup per CU (PLX_SSFM_FUSED_PER_CU=1), so that a CU holds one column workgroup of one half and row waves of the other --
against one plan of 1024 frames.  usage: two_plans.py [per_cu]"""
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
per_cu = sys.argv[1] if len(sys.argv) > 1 else "1"
import torch
from polmux_amd import pipeline

def run_one(F, reps=3):
    hp = pipeline.HotPath(pipeline.HotPathConfig(), max_frames=F)
    ts = []
    for r in range(reps):
        ux, uy = hp.make_batch(F)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hp.fibre(ux, uy)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    chk = float(torch.view_as_real(ux).abs().sum().item())
    hp.close()
    return min(ts), chk

def run_two(F, reps=3):
    os.environ["PLX_SSFM_FUSED_PER_CU"] = per_cu
    hps = [pipeline.HotPath(pipeline.HotPathConfig(), max_frames=F // 2) for _ in range(2)]
    del os.environ["PLX_SSFM_FUSED_PER_CU"]
    streams = [torch.cuda.Stream() for _ in range(2)]
    ts = []
    for r in range(reps):
        bufs = [hp.make_batch(F // 2) for hp in hps]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        def work(i):
            with torch.cuda.stream(streams[i]):
                hps[i].fibre(*bufs[i])
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    chk = sum(float(torch.view_as_real(b[0]).abs().sum().item()) for b in bufs)
    for hp in hps: hp.close()
    return min(ts), chk

F = 1024
for rep in range(2):
    a = run_one(F)
    b = run_two(F)
    print("one plan x%d: %.2f ms   two plans x%d side by side (per_cu=%s): %.2f ms   checksums %.10g %.10g" % (F, a[0], F // 2, per_cu, b[0], a[1], b[1]), flush=True)
