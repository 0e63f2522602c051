#!/bin/bash
# dev tool: A/B environment settings on ONE box.  usage: ab_envs.sh "<frames> <flag> <nsymb> [nch]" "VAR=a" "VAR=b" ...   ("-" = nothing set)
set -e
cd "$(dirname "$0")/../.."
read F FLAG NSYMB NCH <<< "$1"; shift
for rep in 1 2; do for setting in "$@"; do
(
[ "$setting" = "-" ] || export $setting   # (a setting may hold several VAR=value words)
ABN="$setting" F=$F FLAG=$FLAG NSYMB=$NSYMB NCH=${NCH:-1} timeout -k 10 200 python - <<'PY'
import os, sys, time, zlib
sys.path.insert(0, os.getcwd())
import torch
from polmux_amd import pipeline
F = int(os.environ["F"])
hp = pipeline.HotPath(pipeline.HotPathConfig(flag=os.environ["FLAG"], nsymb=int(os.environ["NSYMB"]), nch=int(os.environ["NCH"])), max_frames=F)
hp.profile(True)
ts = []
for r in range(5):
    ux, uy = hp.make_batch(F)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hp.fibre(ux, uy)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ms, k = hp.kernel_times()
chk = zlib.crc32(ux.cpu().numpy().tobytes())
print("%-28s fibre %.2f ms  col %.1f us  row %.1f us  crc %08x" % (os.environ["ABN"], min(ts[1:]) * 1e3, ms[0] / max(k[0], 1) * 1e3, ms[1] / max(k[1], 1) * 1e3, chk), flush=True)
hp.close()
PY
)
done; done
