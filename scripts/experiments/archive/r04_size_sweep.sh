#!/bin/bash
# dev: the fibre's kernels across frame sizes at a constant batch of 2^26 samples ('g-s-', C1 physics): which shapes are off the pace?
O=gpurun_out/r04_sizes; mkdir -p $O
for cfg in "256 64 4096" "1024 16 4096" "1024 64 1024" "1024 128 512" "4096 64 256" "4096 128 128" "16384 64 64"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py --nsymb $1 --nt $2 --frames $3 --steps 3 --warmup 1 --variants 1 --no-overlap --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway --no-cohmix-line > $O/s_$1_$2.json 2> $O/s_$1_$2.err || { echo "FAILED $cfg"; tail -n 3 $O/s_$1_$2.err; exit 1; }
done
python3 - <<PY
import json, glob
for cfg in "256 64 4096|1024 16 4096|1024 64 1024|1024 128 512|4096 64 256|4096 128 128|16384 64 64".split("|"):
    a, b, f = cfg.split()
    d = json.loads(open("$O/s_%s_%s.json" % (a, b)).read().strip().split("\n")[-1]); c = d["config"]; r = d["roofline"]
    print("N=2^%d x %s frames: fibre %.1f ms | %s | group %.3f | info %s" % ((int(a) * int(b)).bit_length() - 1, f, c["fibre_ms_per_step"],
          {k: (round(v["avg_launch_us"]), round(v["frac_of_8TBs"], 3)) for k, v in r["kernels"].items()}, r["step_group"]["frac_of_8TBs"], c.get("fibre_step")))
PY
