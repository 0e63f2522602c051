#!/bin/bash
# dev experiment: what is k_row's time made of?  TIMING ONLY (wrong results): copies of plx_ssfm.hip with parts of the row pass
# compiled out -- bit 0: no row transforms (forward and inverse), bit 1: no spectral multiplier.
# usage: scripts/experiments/row_ablation.sh build   (here)      gpurun -- bash scripts/experiments/row_ablation.sh run
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  mkdir -p build_abl
  python3 - <<'PY'
import os, re
src = open("polmux_amd/csrc/plx_ssfm.hip").read()
a = src.index("__global__ __launch_bounds__(1024) void k_row(SsfmArgs a)")
b = src.index("// ------------------------------------------------- pass 2 for 4096-point rows")
body = src[a:b]
body = body.replace("    row_fft_dif(s, a.p2, a.logR + (a.dual ? 1 : 0), tw, tid, nthr);", "#if !(PLX_ROW_ABL & 1)\n    row_fft_dif(s, a.p2, a.logR + (a.dual ? 1 : 0), tw, tid, nthr);\n#endif")
body = body.replace("    row_fft_dit(s, a.p2, a.logR + (a.dual ? 1 : 0), tw, tid, nthr);", "#if !(PLX_ROW_ABL & 1)\n    row_fft_dit(s, a.p2, a.logR + (a.dual ? 1 : 0), tw, tid, nthr);\n#endif")
i = body.index("    if (!a.dual) {\n        for (int el = tid; el < nel; el += nthr) {\n            const int e = row_lane_point(el, N2); // Hf")
j = body.index("    __syncthreads();\n#if !(PLX_ROW_ABL & 1)\n    row_fft_dit")
body = body[:i] + "#if !(PLX_ROW_ABL & 2)\n" + body[i:j] + "#endif\n" + body[j:]
src = src[:a] + body + src[b:]
root = os.getcwd()
src = src.replace('#include "../../include/polmux_hip.h"', '#include "%s/include/polmux_hip.h"' % root)
for h in ("plx_fft.h", "plx_internal.h", "plx_gateway.h"):
    src = src.replace('#include "%s"' % h, '#include "%s/polmux_amd/csrc/%s"' % (root, h))
open("build_abl/plx_ssfm.hip", "w").write(src)
PY
  for v in 1 2 3; do
    OBJS=""
    for f in polmux_amd/csrc/*.hip; do
      o=build_abl/$(basename ${f%.hip})_$v.o
      [ $(basename $f) = plx_ssfm.hip ] && f=build_abl/plx_ssfm.hip
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I polmux_amd/csrc -DPLX_ROW_ABL=$v -c $f -o $o &
      OBJS="$OBJS $o"
    done
    wait
    hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o polmux_amd/lib/libpolmux_hip_abl$v.so
  done
  exit 0
fi
for n in base abl2 abl1 abl3; do
ABN=$n timeout -k 10 200 python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from polmux_amd import _abi
n = os.environ["ABN"]
if n != "base": _abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), "libpolmux_hip_%s.so" % n)
from polmux_amd import pipeline
F = 1024
hp = pipeline.HotPath(pipeline.HotPathConfig(), max_frames=F)
hp.profile(True)
for r in range(3):
    ux, uy = hp.make_batch(F)
    try:
        hp.fibre(ux, uy)
    except Exception as e:          # (ablated physics may run into the step limit: the kernel times up to there still count)
        print("  (%s)" % str(e)[:60])
    torch.cuda.synchronize()
ms, k = hp.kernel_times()
what = {"base": "full row pass", "abl1": "no row transforms", "abl2": "no spectral multiplier", "abl3": "neither: load, twiddle, LDS round trip, store"}[n]
print("%-5s %-48s col %.1f us  row %.1f us (x%d)" % (n, what, ms[0] / max(k[0], 1) * 1e3, ms[1] / max(k[1], 1) * 1e3, k[1]), flush=True)
hp.close()
PY
done
