#!/bin/bash
# dev tool: A/B timing of library builds on ONE box (boxes differ by a few per cent).
# usage: scripts/experiments/ab.sh build <name> [extra hipcc flags]   (here; builds the current sources as polmux_amd/lib/libpolmux_hip_<name>.so)
#        gpurun -- bash scripts/experiments/ab.sh run <nameA> <nameB> ... [-- frames flag nsymb]   ("base" = the library of record)
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  name=$2; shift 2
  mkdir -p build_ab_$name
  OBJS=""
  for f in polmux_amd/csrc/*.hip; do
    o=build_ab_$name/$(basename ${f%.hip}).o
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w "$@" -c $f -o $o &
    OBJS="$OBJS $o"
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o polmux_amd/lib/libpolmux_hip_$name.so
  rm -rf build_ab_$name
  exit 0
fi
shift
names=""; while [ $# -gt 0 ] && [ "$1" != "--" ]; do names="$names $1"; shift; done
[ "$1" = "--" ] && shift
F=${1:-1024}; FLAG=${2:-g-s-}; NSYMB=${3:-1024}
for rep in 1 2; do for n in $names; do
ABN=$n F=$F FLAG=$FLAG NSYMB=$NSYMB timeout -k 10 200 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from polmux_amd import _abi
n = os.environ["ABN"]
if n != "base": _abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), "libpolmux_hip_%s.so" % n)
from polmux_amd import pipeline
F = int(os.environ["F"])
hp = pipeline.HotPath(pipeline.HotPathConfig(flag=os.environ["FLAG"], nsymb=int(os.environ["NSYMB"]), nch=int(os.environ.get("NCH", "1"))), max_frames=F)
hp.profile(True)
ts = []
for r in range(5):
    ux, uy = hp.make_batch(F)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hp.fibre(ux, uy)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ms, k = hp.kernel_times()
import zlib; print("%-12s fibre %.2f ms  col %.1f us  row %.1f us  crc %08x" % (n, min(ts[1:]) * 1e3, ms[0] / max(k[0], 1) * 1e3, ms[1] / max(k[1], 1) * 1e3, zlib.crc32(ux.cpu().numpy().tobytes())), flush=True)
hp.close()
PY
done; done
