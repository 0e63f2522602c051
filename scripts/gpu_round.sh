#!/bin/bash
# dev tool: GPU test suite + smoke + bench line + kernel trace, logs under gpurun_out/<tag>/
tag=${1:-round}
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_tests.log 2>&1; rc=$?; tail -15 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail $O/bench_default.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("Gs/s %.4f  ms/step %.1f  fibre %.1f  rx %.1f" % (d["value"], d["ms_per_step"], d["config"]["fibre_ms_per_step"], d["config"]["rxdsp_ms_per_step"]))
for k, v in r["kernels"].items():
    print("  %-10s %8.1f us  %6.0f GB/s  frac %.3f  (%d launches)" % (k, v["avg_launch_us"], v["achieved_GBs"], v["frac_of_8TBs"], v["active_launches"]))
print("  group %.0f GB/s frac %.3f  sample-steps/s %.3g" % (r["step_group"]["achieved_GBs"], r["step_group"]["frac_of_8TBs"], r["step_group"]["sample_steps_per_s"]))
print("  mc", d.get("mc"))
print("  cpu", d.get("cpu_baseline"))
PY
