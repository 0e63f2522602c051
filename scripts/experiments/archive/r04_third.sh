#!/bin/bash
O=gpurun_out/r04_third
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_tests.log 2>&1; rc=$?; tail -14 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
hipcc --offload-arch=gfx950 -O3 scripts/experiments/micro/cma_chain.hip -o /tmp/cma_chain && timeout -k 10 120 /tmp/cma_chain > $O/cma_chain.txt 2>&1; cat $O/cma_chain.txt
timeout -k 10 400 python bench.py --nsymb 16384 --frames 16 --spans 2 --steps 3 --warmup 1 --mc-rounds 0 --no-cpu-baseline > $O/c4.json 2> $O/c4.err || { tail -20 $O/c4.err; exit 1; }
timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/c1.json 2> $O/c1.err || { tail -20 $O/c1.err; exit 1; }
python - <<PY
import json
for t in ("c4", "c1"):
    d = json.loads(open("$O/%s.json" % t).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(t, "Gs/s %.4f  ms/step %.1f  fibre %.1f  rx %.1f  %s" % (d["value"], d["ms_per_step"], d["config"]["fibre_ms_per_step"], d["config"]["rxdsp_ms_per_step"], d["config"]["fibre_step"]))
    for k, v in r["kernels"].items():
        print("  %-10s %8.1f us  %6.0f GB/s  frac %.3f  (%d launches)" % (k, v["avg_launch_us"], v["achieved_GBs"], v["frac_of_8TBs"], v["active_launches"]))
    print("  group %.0f GB/s frac %.3f" % (r["step_group"]["achieved_GBs"], r["step_group"]["frac_of_8TBs"]))
    print("  gateway", json.dumps(d.get("gateway", {}).get("plx_cmapolardemux")))
    print("  mc", json.dumps({k: v for k, v in (d.get("mc") or {}).items() if k in ("realisations_per_s", "strong_scaling")}))
PY
