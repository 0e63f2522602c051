#!/bin/bash
# round 5: early poll of the frame barrier's slots (teams of <= 64 tiles) against the standard sweep, alternating
O=gpurun_out/r05_early; mkdir -p $O
timeout -k 10 500 python scripts/crc_set.py early_poll=1 > $O/crc.txt 2> $O/crc.err || { tail -5 $O/crc.err; exit 1; }
if diff profiles/r05_crc_before_split.txt $O/crc.txt > $O/crc.diff; then echo "crc set with early_poll=1 identical: $(wc -l < $O/crc.txt) lines"; else echo "CRC DIFFERENCES"; head -10 $O/crc.diff; fi
A="--steps 8 --warmup 2 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway --no-cohmix-line --configs no"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1], "%.4f Gs/s %.1f ms/step fibre %.1f" % (d["value"], d["ms_per_step"], d["config"]["fibre_ms_per_step"]), {k:(round(v["avg_launch_us"],1), round(v["frac_of_8TBs"],3)) for k,v in r["kernels"].items()})'
for rep in 1 2 3; do
  for v in 0 1; do
    python scripts/experiments/bench_tuned.py early_poll=$v -- $A 2>/dev/null | python -c "$show" "early=$v overlap   "
    python scripts/experiments/bench_tuned.py early_poll=$v -- $A --no-overlap 2>/dev/null | python -c "$show" "early=$v no-overlap"
  done
done | tee $O/ab.txt
for v in 0 1; do
  python scripts/experiments/bench_tuned.py early_poll=$v -- $A --nsymb 2048 --frames 512 --steps 4 --warmup 1 2>/dev/null | python -c "$show" "early=$v 2^17      "
  python scripts/experiments/bench_tuned.py early_poll=$v -- $A --mc 2>/dev/null | python -c "$show" "early=$v --mc      "
done | tee -a $O/ab.txt
