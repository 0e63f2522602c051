#!/bin/bash
# dev experiment: wave priorities in the fused column sweep.  A team's critical path is the pre-barrier half of its slowest tile;
# the co-resident workgroup of another team is, half of the time, in its post-barrier half.  Variant 1: s_setprio 3 from the loop
# top to the slot store, 0 after the frame barrier.  Variant 2: the reverse.  Variant 3: high everywhere except while polling.
# usage: scripts/experiments/prio_ab.sh build   (here)      gpurun -- bash scripts/experiments/prio_ab.sh run
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  mkdir -p build_abl
  python3 - <<'PY'
import os
src = open("polmux_amd/csrc/plx_ssfm.hip").read()
def rep(old, new):
    global src
    assert src.count(old) == 1, old
    src = src.replace(old, new)
rep("        // [phase 0] loop top\n", "        // [phase 0] loop top\n        PRIO_PRE();\n")
rep("        lds_barrier();\n        // [phase 4] frame barrier\n", "        lds_barrier();\n        PRIO_POST();\n        // [phase 4] frame barrier\n")
rep("            const long long t0 = plx_clock();\n            for (;;) {\n                bool all = true;", "            const long long t0 = plx_clock();\n            PRIO_POLL();\n            for (;;) {\n                bool all = true;")
src = src.replace("namespace {\n", """namespace {
#if PLX_PRIO == 1
#define PRIO_PRE() __builtin_amdgcn_s_setprio(3)
#define PRIO_POST() __builtin_amdgcn_s_setprio(0)
#define PRIO_POLL() ((void)0)
#elif PLX_PRIO == 2
#define PRIO_PRE() __builtin_amdgcn_s_setprio(0)
#define PRIO_POST() __builtin_amdgcn_s_setprio(3)
#define PRIO_POLL() ((void)0)
#else
#define PRIO_PRE() __builtin_amdgcn_s_setprio(3)
#define PRIO_POST() __builtin_amdgcn_s_setprio(3)
#define PRIO_POLL() __builtin_amdgcn_s_setprio(0)
#endif
""", 1)
root = os.getcwd()
src = src.replace('#include "../../include/polmux_hip.h"', '#include "%s/include/polmux_hip.h"' % root)
for h in ("plx_fft.h", "plx_internal.h", "plx_gateway.h"):
    src = src.replace('#include "%s"' % h, '#include "%s/polmux_amd/csrc/%s"' % (root, h))
open("build_abl/plx_ssfm_prio.hip", "w").write(src)
PY
  grep -c PRIO_ build_abl/plx_ssfm_prio.hip
  for v in 1 2 3; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I polmux_amd/csrc -DPLX_PRIO=$v -c build_abl/plx_ssfm_prio.hip -o build_abl/plx_ssfm_prio$v.o &
  done
  wait
  for v in 1 2 3; do
    OBJS=""
    for f in polmux_amd/csrc/*.hip; do
      o=build/$(basename ${f%.hip}).o
      [ $(basename $f) = plx_ssfm.hip ] && o=build_abl/plx_ssfm_prio$v.o
      OBJS="$OBJS $o"
    done
    hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o polmux_amd/lib/libpolmux_hip_prio$v.so
  done
  exit 0
fi
for rep in 1 2; do
for n in base prio1 prio2 prio3; do
ABN=$n timeout -k 10 200 python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from polmux_amd import _abi
n = os.environ["ABN"]
if n != "base": _abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), "libpolmux_hip_%s.so" % n)
from polmux_amd import pipeline
for F, kw in ((1024, {}),):
    hp = pipeline.HotPath(pipeline.HotPathConfig(**kw), max_frames=F)
    hp.profile(True)
    t = []
    for r in range(3):
        ux, uy = hp.make_batch(F)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hp.fibre(ux, uy)
        torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    ms, k = hp.kernel_times()
    print("%-5s F=%d fibre %.2f ms  col %.1f us  row %.1f us (x%d)" % (n, F, min(t), ms[0] / max(k[0], 1) * 1e3, ms[1] / max(k[1], 1) * 1e3, k[1]), flush=True)
    hp.close()
PY
done
done
