"""dev tool: where a Monte-Carlo round of the bench's MC leg spends its time (host clock around launch / collect)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from polmux_amd import pipeline
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = pipeline.HotPathConfig(flag="gps-", frontend="cohmix", rx_amp=True, span_nf_db=31.0)
camp = pipeline.McCampaign(cfg, frames_per_call=F)
hp = camp.hp
camp.simulate(list(range(F)))
torch.cuda.synchronize()
# sequential pieces
t = time.perf_counter(); hp.set_random_pmd(list(range(F))); ux, uy = hp.make_batch(F); torch.cuda.synchronize(); t_host = time.perf_counter() - t
t = time.perf_counter(); hp.fibre(ux, uy, span_keys=list(range(F))); torch.cuda.synchronize(); t_fib = time.perf_counter() - t
t = time.perf_counter(); hp.receive(ux, uy, 0.0, 1, None, list(range(F))); torch.cuda.synchronize(); t_rx = time.perf_counter() - t
t = time.perf_counter(); e = hp.errors_resolved(F); v = hp.evm(F); torch.cuda.synchronize(); t_err = time.perf_counter() - t
print("F=%d sequential: host %.1f ms, fibre %.1f, receive %.1f, errors+evm %.1f" % (F, t_host * 1e3, t_fib * 1e3, t_rx * 1e3, t_err * 1e3))
# pipelined rounds
hs = camp.launch(list(range(F)))
t0 = time.perf_counter()
for r in range(1, 6):
    t1 = time.perf_counter(); h2 = camp.launch(list(range(r * F, (r + 1) * F))); t2 = time.perf_counter()
    camp.collect(hs); t3 = time.perf_counter()
    print("round %d: launch(next) %.1f ms, collect %.1f ms" % (r, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    hs = h2
camp.collect(hs)
print("5 pipelined rounds: %.1f ms per round" % ((time.perf_counter() - t0) * 1e3 / 5))
camp.close()
