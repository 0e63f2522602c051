"""Experiment: the batch as TWO half-batches on two HIP streams (two plans, two host threads), fused grid 1 WG per CU each,
so that one half's k_row can fill the CUs while the other half's k_colx16 sits in its frame barrier.
usage: python scripts/experiments/two_stream_fused.py [frames]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from polmux_amd import pipeline

F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = 3

def run_single(wg):
    if wg: os.environ["PLX_SSFM_FUSE_WG_PER_CU"] = str(wg)
    else: os.environ.pop("PLX_SSFM_FUSE_WG_PER_CU", None)
    hp = pipeline.HotPath(pipeline.HotPathConfig(), max_frames=F)
    ts = []
    for r in range(reps + 1):
        ux, uy = hp.make_batch(F)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hp.fibre(ux, uy)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    hp.close()
    return min(ts[1:]) * 1e3

def run_two(wg):
    os.environ["PLX_SSFM_FUSE_WG_PER_CU"] = str(wg)
    hps = [pipeline.HotPath(pipeline.HotPathConfig(), max_frames=F // 2) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    ts = []
    for r in range(reps + 1):
        bs = [hp.make_batch(F // 2) for hp in hps]
        torch.cuda.synchronize()
        def work(k):
            with torch.cuda.stream(streams[k]):
                hps[k].fibre(*bs[k])
        th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    for hp in hps: hp.close()
    return min(ts[1:]) * 1e3

print("single plan, 2 WG/CU: %.2f ms" % run_single(0))
print("single plan, 1 WG/CU: %.2f ms" % run_single(1))
print("two plans on two streams, 1 WG/CU each: %.2f ms" % run_two(1))
print("two plans on two streams, 2 WG/CU each: %.2f ms" % run_two(2))
os.environ["PLX_SSFM_EXP"] = "1"
print("single plan, 2 WG/CU, frame barrier NOT waited for (timing only): %.2f ms" % run_single(0))
