#!/bin/bash
mkdir -p gpurun_out/r05_bench1
timeout -k 10 300 python -m pytest tests/test_gpu_configs.py -q -x -m gpu -k "another_kernel_holds" > gpurun_out/r05_bench1/pytest.log 2>&1; tail -5 gpurun_out/r05_bench1/pytest.log
( time timeout -k 10 600 python bench.py > gpurun_out/r05_bench1/bench.json 2> gpurun_out/r05_bench1/bench.err ) 2>&1 | grep real
tail -3 gpurun_out/r05_bench1/bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r05_bench1/bench.json").read().strip().splitlines()[-1])
print("headline %.4f Gs/s, %.1f ms/step" % (d["value"], d["ms_per_step"]))
for k, v in d.get("configs", {}).items():
    r = v["roofline"]
    print(k, "%.4f Gs/s %.1f ms/step leg %.1f s" % (v["value"], v["ms_per_step"], v["leg_seconds"]), {kk: (round(vv["avg_launch_us"]), round(vv["frac_of_8TBs"], 3)) for kk, vv in r["kernels"].items()},
          "group %.3f" % r["step_group"]["frac_of_8TBs"], "steps", v["config"]["ssfm_steps_min_max"], v["config"]["fibre_step"][:40])
    print("   cpu", v.get("cpu_baseline"))
PY
