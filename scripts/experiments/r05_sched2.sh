#!/bin/bash
# round 5: max-ilp scheduling for single objects (the 256-point row pass, the receiver kernels) against the default build
O=gpurun_out/r05_sched2; mkdir -p $O
A="--steps 8 --warmup 2 --no-cpu-baseline --mc-rounds 0 --no-gateway --no-cohmix-line --configs no"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]; s=c.get("single_frame") or {}
print(sys.argv[1], "%.4f Gs/s %.1f ms/step fibre %.1f rx %.1f" % (d["value"], d["ms_per_step"], c["fibre_ms_per_step"], c["rxdsp_ms_per_step"]), {k:(round(v["avg_launch_us"],1), round(v["frac_of_8TBs"],3)) for k,v in r["kernels"].items()}, "single rx %.2f ms" % s.get("rx_ms", 0))'
for rep in 1 2; do
  for v in base r256 rx both; do
    python scripts/experiments/bench_with_lib.py $v $A 2>/dev/null | python -c "$show" "$v overlap   "
    python scripts/experiments/bench_with_lib.py $v $A --no-overlap 2>/dev/null | python -c "$show" "$v no-overlap"
  done
done | tee $O/ab.txt
for v in base both; do
  python scripts/experiments/bench_with_lib.py $v $A --mc --no-single-frame 2>/dev/null | python -c "$show" "$v --mc      "
  python scripts/experiments/bench_with_lib.py $v $A --nch 16 --frames 32 --steps 3 --warmup 1 --no-single-frame 2>/dev/null | python -c "$show" "$v c2 frame  "
done | tee -a $O/ab.txt
