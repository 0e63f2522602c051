#!/bin/bash
# dev tool: GPU test suite + smoke + bench lines, logs under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?; tail -5 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
timeout -k 10 300 python bench.py --frames 256 --steps 3 --no-cpu-baseline > gpurun_out/bench_F256_pick.json 2>gpurun_out/bench_err.log || { tail gpurun_out/bench_err.log; exit 1; }
timeout -k 10 300 python bench.py --frames 256 --steps 3 --no-cpu-baseline --frontend cohmix > gpurun_out/bench_F256_cohmix.json 2>gpurun_out/bench_err.log || { tail gpurun_out/bench_err.log; exit 1; }
python - <<'PY'
import json
for n in ("pick", "cohmix"):
    d = json.loads(open("gpurun_out/bench_F256_%s.json" % n).read().strip().splitlines()[-1])
    print(n, "Gs/s %.4f" % d["value"], "fibre %.2f ms" % d["config"]["fibre_ms_per_step"], "rx %.2f ms" % d["config"]["rxdsp_ms_per_step"],
          "frac %.3f" % d["roofline"]["frac"], "errs", d["config"]["bit_errors_xy"])
PY
