set -e
timeout -k 10 400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -k "c4 or 2pow20 or front_end or inverse_pmd or twin or mat" > gpurun_out/t2.log 2>&1 || { tail -30 gpurun_out/t2.log; exit 1; }
tail -3 gpurun_out/t2.log
timeout -k 10 300 python bench.py --nsymb 16384 --frames 16 --steps 2 --warmup 1 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-single-frame --no-overlap > gpurun_out/c4_fused.json 2>gpurun_out/c4.err
PLX_SSFM_NO_FUSE=1 timeout -k 10 300 python bench.py --nsymb 16384 --frames 16 --steps 2 --warmup 1 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-single-frame --no-overlap > gpurun_out/c4_plain.json 2>gpurun_out/c4.err
python - <<'PY'
import json
for n in ("fused", "plain"):
    d = json.loads(open("gpurun_out/c4_%s.json" % n).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(n, "fibre %.2f ms/step" % d["config"]["fibre_ms_per_step"], "steps/frame %.0f" % d["config"]["ssfm_steps_per_frame"], {k: round(v["avg_launch_us"], 1) for k, v in r["kernels"].items()}, "group frac %.3f" % r["step_group"]["frac_of_8TBs"], "sample-steps/s %.3g" % r["step_group"]["sample_steps_per_s"])
PY
