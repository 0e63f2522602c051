#!/bin/bash
# dev: PMD plans ('gps-', 100 waveplates) across frame sizes at a constant batch of 2^25 samples, register-form rows against k_row (PLX_SSFM_ROWR=0)
O=gpurun_out/r04_pmd_sizes; mkdir -p $O
for rowr in 1 0; do
for cfg in "1024 64 512" "1024 128 256" "4096 64 128" "4096 128 64"; do
  set -- $cfg
  PLX_SSFM_ROWR=$rowr timeout -k 10 300 python3 bench.py --nsymb $1 --nt $2 --frames $3 --flag gps- --steps 3 --warmup 1 --variants 1 --no-overlap --share-device no --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway --no-cohmix-line > $O/p${rowr}_$1_$2.json 2> $O/p${rowr}_$1_$2.err || { echo "FAILED $rowr $cfg"; tail -n 3 $O/p${rowr}_$1_$2.err; exit 1; }
done; done
python3 - <<PY
import json
for rowr in (1, 0):
  for cfg in "1024 64 512|1024 128 256|4096 64 128|4096 128 64".split("|"):
    a, b, f = cfg.split()
    d = json.loads(open("$O/p%d_%s_%s.json" % (rowr, a, b)).read().strip().split("\n")[-1]); c = d["config"]; r = d["roofline"]
    print("ROWR=%d N=2^%d x %s frames: fibre %.1f ms | %s | group %.3f" % (rowr, (int(a) * int(b)).bit_length() - 1, f, c["fibre_ms_per_step"],
          {k: (round(v["avg_launch_us"]), round(v["frac_of_8TBs"], 3)) for k, v in r["kernels"].items()}, r["step_group"]["frac_of_8TBs"]))
PY
