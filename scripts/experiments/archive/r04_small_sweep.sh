O=gpurun_out/r04_sizes_small; mkdir -p $O
for cfg in "64 64 16384" "256 32 8192" "256 64 4096" "256 128 2048" "1024 32 2048"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py --nsymb $1 --nt $2 --frames $3 --steps 3 --warmup 1 --variants 1 --no-overlap --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway --no-cohmix-line > $O/s_$1_$2.json 2> $O/s_$1_$2.err || { echo "FAILED $cfg"; tail -n 3 $O/s_$1_$2.err; }
done
python3 - <<PY
import json
for cfg in "64 64 16384|256 32 8192|256 64 4096|256 128 2048|1024 32 2048".split("|"):
    a, b, f = cfg.split()
    try:
        d = json.loads(open("$O/s_%s_%s.json" % (a, b)).read().strip().split("\n")[-1]); c = d["config"]; r = d["roofline"]
        print("N=2^%d x %s frames: fibre %.1f ms | %s | group %.3f" % ((int(a) * int(b)).bit_length() - 1, f, c["fibre_ms_per_step"],
          {k: (round(v["avg_launch_us"]), round(v["frac_of_8TBs"], 3)) for k, v in r["kernels"].items()}, r["step_group"]["frac_of_8TBs"]))
    except Exception as e: print(cfg, "ERR", e)
PY
