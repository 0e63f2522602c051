# dev: does the fibre of 1024 C1 frames finish sooner as TWO half-batches on two streams (two plans, two host threads)?  The column
# sweep is latency-bound and the row pass bandwidth-bound: their tails and phases could fill each other.
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import torch
from polmux_amd import pipeline
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = pipeline.HotPathConfig()
one = pipeline.HotPath(cfg, max_frames=F)
ts = []
for r in range(4):
    ux, uy = one.make_batch(F)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    one.fibre(ux, uy)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("one plan, %d frames: %.2f ms" % (F, min(ts[1:]) * 1e3))
one.close()
halves = [pipeline.HotPath(cfg, max_frames=F // 2) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
def run(i, bufs):
    with torch.cuda.stream(streams[i]):
        halves[i].fibre(*bufs)
ts = []
for r in range(4):
    bufs = []
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            bufs.append(halves[i].make_batch(F // 2))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i, bufs[i])) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("two plans of %d frames on two streams: %.2f ms" % (F // 2, min(ts[1:]) * 1e3))
for h in halves: h.close()
