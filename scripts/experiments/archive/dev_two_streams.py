# dev tool: does running two half-batches of the fibre on two HIP streams (two host threads) beat one batch?
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from polmux_amd import pipeline
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = pipeline.HotPathConfig()
os.environ.setdefault("PLX_SSFM_NO_FUSE", "1")   # concurrent propagate calls: the fused sweep assumes it owns the chip
def run(nstreams):
    hps = [pipeline.HotPath(cfg, F // nstreams) for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    batches = [[hp.make_batch(F // nstreams) for _ in range(2 * S)] for hp in hps]   # fibre works in place: fresh fields per pass
    torch.cuda.synchronize()
    def work(k, off):
        with torch.cuda.stream(streams[k]):
            for i in range(S):
                hps[k].fibre(*batches[k][off + i])
    for off in (0, S):
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k, off)) for k in range(nstreams)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    for hp in hps: hp.close()
    return dt / S * 1e3
for n in (1, 2, 4):
    print("streams", n, "fibre ms per %d-frame pass: %.2f" % (F, run(n)))
