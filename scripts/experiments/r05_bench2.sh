#!/bin/bash
mkdir -p gpurun_out/r05_bench2
( time timeout -k 10 600 python bench.py > gpurun_out/r05_bench2/bench.json 2> gpurun_out/r05_bench2/bench.err ) 2>&1 | grep real
tail -3 gpurun_out/r05_bench2/bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r05_bench2/bench.json").read().strip().splitlines()[-1])
def show(tag, v):
    r = v["roofline"]
    print(tag, "%.4f Gs/s %.1f ms/step" % (v["value"], v["ms_per_step"]), "bound", r["bound"], "|", r["bound_note"])
    print("   hbm", {kk: (round(vv["avg_launch_us"]), round(vv["frac_of_8TBs"], 3)) for kk, vv in r["kernels"].items()}, "group %.3f" % r["step_group"]["frac_of_8TBs"])
    if r.get("fp64"): print("   fp64", {kk: (round(vv["achieved_TFLOPs"], 1), round(vv["frac"], 3)) for kk, vv in r["fp64"]["kernels"].items()})
    print("   cpu", (v.get("cpu_baseline") or {}).get("value"))
show("c1", d)
for k, v in d.get("configs", {}).items(): show(k, v)
m = d["mc"]
print("mc", round(m["realisations_per_s"]), "strong", round(m["strong_scaling"]["realisations_per_s"]))
print("share", json.dumps(m.get("strong_scaling_rank_share"), indent=1))
PY
