#!/bin/bash
# dev experiment: where a k_colx16 workgroup's tile time goes (instrumented copy of the library, thread-0 wall-clock stamps).
# usage: scripts/experiments/stamps.sh build   (here: instruments COPIES of ssfm_colx.hip / ssfm_plan.hip with stamps_patch.py and cross-compiles polmux_amd/lib/libpolmux_hip_stamps.so)
#        gpurun -- bash scripts/experiments/stamps.sh run [frames]
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  mkdir -p build_stamps
  rm -f build_stamps/*.hip build_stamps/*.o
  python3 scripts/experiments/stamps_patch.py > /dev/null
  OBJS=""
  for f in polmux_amd/csrc/*.hip; do
    o=build_stamps/$(basename ${f%.hip}).o
    [ -f build_stamps/$(basename $f) ] && f=build_stamps/$(basename $f)
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I polmux_amd/csrc $PLX_EXTRA_HIPCC_FLAGS -c $f -o $o &
    OBJS="$OBJS $o"
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o polmux_amd/lib/libpolmux_hip_stamps.so
  exit 0
fi
F=${2:-1024} MK=${3:-no} NSYMB=${4:-1024} NCH=${5:-1} FLAG=${6:-g-s-} timeout -k 10 200 python - <<'PY'
import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
from polmux_amd import _abi
_abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), "libpolmux_hip_stamps.so")
from polmux_amd import pipeline
F = int(os.environ["F"])
hp = pipeline.HotPath(pipeline.HotPathConfig(flag=os.environ["FLAG"], manakov=os.environ["MK"], nsymb=int(os.environ["NSYMB"]), nch=int(os.environ["NCH"])), max_frames=F)
lib = _abi.get().lib
out = (C.c_longlong * (32 + 3072))()
hp.profile(True)
for r in range(4):
    ux, uy = hp.make_batch(F)
    lib.plx_ssfm_stamps(out, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hp.fibre(ux, uy)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) * 1e3
lib.plx_ssfm_stamps(out, 1)
ms, n = hp.kernel_times()
tiles = out[9]
names8 = "slots of the frame seen (part of the frame barrier wait; the rest is the controller)"
names = ["loop top (list / control loads)", "wait for the staged tile", "r16_dit + exchange write + barrier", "exchange read + lvl2_dit + scale + max + barrier",
         "frame barrier", "Kerr", "lvl2_dif + exchange + barrier", "stage issue + r16_dif + stores issued"]
print("F=%d manakov=%s fibre %.2f ms, col %.1f us x%d; %d workgroup-tiles" % (F, os.environ["MK"], t, ms[0] / max(n[0], 1) * 1e3, n[0], tiles))
tot = 0
for i, nm in enumerate(names):
    us = out[i] * 0.01 / tiles
    tot += us
    sd = max(out[16 + i] * 1e-4 / tiles - us * us, 0.0) ** 0.5
    print("  %-50s %6.2f us per tile  (std %.2f)" % (nm, us, sd))
print("  %-50s %6.2f us per tile" % ("sum", tot + out[8] * 0.01 / tiles))
u8 = out[8] * 0.01 / tiles
print("  %-50s %6.2f us per tile  (std %.2f) (inside the frame barrier: the remainder is the step controller + workgroup barrier)" % ("  slot store -> all slots seen", u8, max(out[24] * 1e-4 / tiles - u8 * u8, 0.0) ** 0.5))
print("  polls per tile %.2f" % (out[10] / tiles))
import numpy as np
w = np.array(out[32:32 + 512], dtype=np.float64) * 0.01 / (tiles / 512.0)
print("  barrier wait per tile by workgroup: min %.2f  median %.2f  max %.2f us" % (w.min(), np.median(w), w.max()))
print("  by workgroup id mod 8:", " ".join("%.2f" % w[x::8].mean() for x in range(8)))
xcc = np.array(out[32 + 2048:32 + 2048 + 512], dtype=np.int64)
print("  by HW_REG_XCC_ID:     ", " ".join("%.2f" % (w[xcc == x].mean() if (xcc == x).any() else float("nan")) for x in range(8)), " (workgroups per XCC:", " ".join(str(int((xcc == x).sum())) for x in range(8)), ")")
print("  XCC of workgroups 0..15:", xcc[:16].tolist())
print("  by tile of the frame (id mod 32):", " ".join("%.1f" % w[x::32].mean() for x in range(32)))
print("  first / second workgroup of a CU (id < 256 / >= 256): %.2f / %.2f" % (w[:256].mean(), w[256:].mean()))
e = np.array(out[32 + 1024:32 + 1024 + 512], dtype=np.float64) * 0.01
e -= e.max()
print("  exit of the last launch, us before the last workgroup, by XCD:", " ".join("%.0f" % e[x::8].mean() for x in range(8)))
srt = np.argsort(w)
print("  least waiting workgroups:", " ".join("%d(%.1f)" % (i, w[i]) for i in srt[:16]))
hp.close()
PY
