set -e
for mode in "" "--power-ladder"; do
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --mc-rounds 0 --no-cpu-baseline --no-single-frame --no-overlap $mode > gpurun_out/lad.json 2>gpurun_out/lad.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/lad.json").read().strip().splitlines()[-1])
r = d["roofline"]; c = d["config"]
print("ladder" if c["power_ladder"] else "homog ", "fibre %.2f ms/step" % c["fibre_ms_per_step"], "steps/frame %.1f" % c["ssfm_steps_per_frame"], "min/max", c["ssfm_steps_min_max"],
      "util list %.3f launched %.3f none %.3f" % (c["active_frame_utilisation"], c["launched_slot_utilisation"], c["utilisation_without_compaction"]),
      {k: round(v["avg_launch_us"], 1) for k, v in r["kernels"].items()}, "sample-steps/s %.4g" % r["step_group"]["sample_steps_per_s"])
PY
done
for mf in 128 512; do
timeout -k 10 300 python bench.py --steps 1 --warmup 0 --frames 64 --mc-rounds 4 --mc-frames $mf --no-cpu-baseline --no-single-frame > gpurun_out/mcx.json 2>gpurun_out/mcx.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/mcx.json").read().strip().splitlines()[-1])
m = d["mc"]; print("mc", m["per_gpu_per_round"], "per round: %.0f realisations/s" % m["realisations_per_s"], "ber %.2e" % m["avgber"], "%.3f s" % m["seconds"])
PY
done
