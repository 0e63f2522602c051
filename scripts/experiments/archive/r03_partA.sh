#!/bin/bash
O=gpurun_out/r03final; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -6 $O/gpu_tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
python3 - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
print("%.4f Gs/s %.1f ms/step fibre %.1f rx %.1f | %s | group %.3f | mc %s" % (d["value"], d["ms_per_step"], c["fibre_ms_per_step"], c["rxdsp_ms_per_step"],
      {k: (round(v["avg_launch_us"]), round(v["frac_of_8TBs"], 3)) for k, v in r["kernels"].items()}, r["step_group"]["frac_of_8TBs"], d["mc"] and round(d["mc"]["realisations_per_s"])))
print("single", c["single_frame"])
print("gateway", json.dumps(d["gateway"], indent=0)[:1500])
print("mc cpu", d["mc"].get("cpu_baseline"))
print("cpu", d.get("cpu_baseline"), d.get("cpu_baseline_all_cores"))
PY
