#!/bin/bash
# round 4, first GPU call: BASELINE config[2] timed for the first time (bench --nch), the 2^20 line beside it, pipeline tests
O=gpurun_out/r04_first
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python bench.py --nch 16 --spans 10 --nf 5 --frames 32 --steps 3 --warmup 1 --mc-rounds 0 --no-gateway > $O/c2.json 2> $O/c2.err || { tail -20 $O/c2.err; exit 1; }
timeout -k 10 300 python bench.py --nsymb 16384 --frames 16 --spans 2 --steps 3 --warmup 1 --mc-rounds 0 --no-gateway --no-cpu-baseline > $O/c4.json 2> $O/c4.err || { tail -20 $O/c4.err; exit 1; }
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --mc-rounds 0 --no-gateway --no-cpu-baseline > $O/c1.json 2> $O/c1.err || { tail -20 $O/c1.err; exit 1; }
python - <<PY
import json
for t in ("c2", "c4", "c1"):
    d = json.loads(open("$O/%s.json" % t).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(t, "Gs/s %.4f  ms/step %.1f  fibre %.1f  rx %.1f  steps/frame %.1f" % (d["value"], d["ms_per_step"], d["config"]["fibre_ms_per_step"], d["config"]["rxdsp_ms_per_step"], d["config"]["ssfm_steps_per_frame"]))
    for k, v in r["kernels"].items():
        print("  %-10s %8.1f us  %6.0f GB/s  frac %.3f  (%d launches)" % (k, v["avg_launch_us"], v["achieved_GBs"], v["frac_of_8TBs"], v["active_launches"]))
    print("  group %.0f GB/s frac %.3f" % (r["step_group"]["achieved_GBs"], r["step_group"]["frac_of_8TBs"]), " cpu", d.get("cpu_baseline"))
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "hot_path or monte or c1 or c3 or smoke" > $O/tests.log 2>&1; tail -3 $O/tests.log
