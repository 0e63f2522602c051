#!/bin/bash
# round 5: the library built with -mllvm -amdgpu-sched-strategy=max-ilp against the default build, alternating on one box
O=gpurun_out/r05_sched; mkdir -p $O
A="--steps 8 --warmup 2 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway --no-cohmix-line --configs no --no-overlap"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1], "%.4f Gs/s fibre %.1f ms" % (d["value"], d["config"]["fibre_ms_per_step"]), {k:(round(v["avg_launch_us"],1), round(v["frac_of_8TBs"],3)) for k,v in r["kernels"].items()})'
for rep in 1 2; do
  for v in base max-ilp; do
    python scripts/experiments/bench_with_lib.py $v $A 2>/dev/null | python -c "$show" "$v C1      "
  done
done | tee $O/ab.txt
for v in base max-ilp; do
  python scripts/experiments/bench_with_lib.py $v $A --mc 2>/dev/null | python -c "$show" "$v --mc    "
  python scripts/experiments/bench_with_lib.py $v $A --nsymb 16384 --frames 16 --steps 4 --warmup 1 --variants 1 --share-device no 2>/dev/null | python -c "$show" "$v 2^20    "
  python scripts/experiments/bench_with_lib.py $v $A --nsymb 4096 --frames 256 --steps 4 --warmup 1 2>/dev/null | python -c "$show" "$v 2^18    "
done | tee -a $O/ab.txt
