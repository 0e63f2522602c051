#!/bin/bash
# 2^20-sample frames: the receiver beside the next fibre (three-sweep plan), CMA waves at normal / raised priority; C1 default too
O=gpurun_out/r04_rxb
mkdir -p $O
for lib in base cmaprio; do
  timeout -k 10 300 python scripts/experiments/bench_with_lib.py $lib --nsymb 16384 --frames 16 --spans 2 --steps 3 --warmup 1 --mc-rounds 0 --no-cpu-baseline --no-gateway --no-single-frame > $O/c4_$lib.json 2> $O/c4_$lib.err || { tail $O/c4_$lib.err; exit 1; }
  timeout -k 10 300 python scripts/experiments/bench_with_lib.py $lib --steps 6 --warmup 2 --mc-rounds 2 --mc-total 0 --no-cpu-baseline --no-gateway --no-single-frame --no-cohmix-line > $O/c1_$lib.json 2> $O/c1_$lib.err || { tail $O/c1_$lib.err; exit 1; }
done
python - <<PY
import json
for t in ("c4_base", "c4_cmaprio", "c1_base", "c1_cmaprio"):
    d = json.loads(open("$O/%s.json" % t).read().strip().splitlines()[-1])
    print(t, "Gs/s %.4f  ms/step %.1f  fibre %.1f  rx %.1f | mc %s | %s" % (d["value"], d["ms_per_step"], d["config"]["fibre_ms_per_step"], d["config"]["rxdsp_ms_per_step"], (d.get("mc") or {}).get("realisations_per_s"), d["config"]["fibre_step"][:40]))
PY
