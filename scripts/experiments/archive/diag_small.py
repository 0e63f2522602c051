"""dev: where does a small batch (F < 64) spend its time?  prints a line per stage; dumps the Python stack if a stage stalls"""
import faulthandler, os, sys, time
sys.path.insert(0, os.getcwd())
faulthandler.dump_traceback_later(45, exit=True)
import torch
from polmux_amd import pipeline
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
t0 = time.perf_counter()
hp = pipeline.HotPath(pipeline.HotPathConfig(variants=1), max_frames=F)
print("plan %.2f s" % (time.perf_counter() - t0), flush=True)
hp.profile(len(sys.argv) > 2)
for it in range(3):
    ux, uy = hp.make_batch(F)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hp.fibre(ux, uy)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("F=%d fibre %.2f ms, ncycle %s" % (F, (t1 - t0) * 1e3, hp.last_ncycle(F)[:4]), flush=True)
    hp.receive(ux, uy, noise_sigma=0.05, noise_seed=it)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("F=%d receive %.2f ms" % (F, (t2 - t1) * 1e3), flush=True)
hp.close()
print("done", flush=True)
