#!/bin/bash
# dev experiment: persistent row pass (k_row256p, PLX_SSFM_ROWR=1) against k_row; LIBV: which build of the library
mkdir -p gpurun_out/r03ev
for rep in 1 2; do
for cfg in ${ROWP_CFGS:-"base:0" "base:1" "rowp2:1"}; do
set -- ${cfg%%:*} ${cfg##*:}
ABN=$1 PLX_SSFM_ROWR=$2 timeout -k 10 200 python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from polmux_amd import _abi
n = os.environ["ABN"]
if n != "base": _abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), "libpolmux_hip_%s.so" % n)
from polmux_amd import pipeline
F = 1024
hp = pipeline.HotPath(pipeline.HotPathConfig(), max_frames=F)
hp.profile(True)
t = []
chk = None
for r in range(3):
    ux, uy = hp.make_batch(F)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hp.fibre(ux, uy)
    torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    if r == 0: chk = float(torch.view_as_real(ux).double().abs().sum().item()), float(torch.view_as_real(uy)[5].double().abs().sum().item())
ms, k = hp.kernel_times()
print("%-5s ROWR=%s fibre %.2f ms  col %.1f us  row %.1f us (x%d)  checksum %.17g %.17g" % (n, os.environ["PLX_SSFM_ROWR"], min(t), ms[0] / max(k[0], 1) * 1e3, ms[1] / max(k[1], 1) * 1e3, k[1], chk[0], chk[1]), flush=True)
hp.close()
PY
done
done
