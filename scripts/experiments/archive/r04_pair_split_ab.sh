for cfg in "1024 128 256" "4096 64 128" "4096 128 64"; do set -- $cfg; for v in 0 1; do PLX_SSFM_ROWG_PAIR_SPLIT=$v python - <<PY
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from polmux_amd import pipeline
hp = pipeline.HotPath(pipeline.HotPathConfig(flag="gps-", nsymb=$1, nt=$2), max_frames=$3)
hp.profile(True)
ts = []
for r in range(4):
    ux, uy = hp.make_batch($3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hp.fibre(ux, uy)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ms, k = hp.kernel_times()
import zlib
print("pairsplit=$v %d x %d x $3: fibre %.2f ms  col %.1f us  row %.1f us crc %08x" % ($1, $2, min(ts[1:]) * 1e3, ms[0] / max(k[0], 1) * 1e3, ms[1] / max(k[1], 1) * 1e3, zlib.crc32(ux.cpu().numpy().tobytes())), flush=True)
hp.close()
PY
done; done
