#!/bin/bash
# round 5: the lite form of the fused sweep (k_colx16l, three workgroups per CU) against the standard one
O=gpurun_out/r05_lite; mkdir -p $O
timeout -k 10 500 python scripts/crc_set.py colx_lite=1 > $O/crc.txt 2> $O/crc.err || { tail -5 $O/crc.err; exit 1; }
if diff profiles/r05_crc_before_split.txt $O/crc.txt > $O/crc.diff; then echo "crc set with colx_lite=1 identical: $(wc -l < $O/crc.txt) lines"; else echo "CRC DIFFERENCES"; head -10 $O/crc.diff; fi
A="--steps 8 --warmup 2 --no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway --no-cohmix-line --configs no"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1], "%.4f Gs/s %.1f ms/step fibre %.1f" % (d["value"], d["ms_per_step"], d["config"]["fibre_ms_per_step"]), {k:(round(v["avg_launch_us"],1), round(v["frac_of_8TBs"],3)) for k,v in r["kernels"].items()}, "grid", d["config"]["fused_grid_workgroups"])'
for rep in 1 2; do
  for v in 0 1; do
    python scripts/experiments/bench_tuned.py colx_lite=$v -- $A 2>/dev/null | python -c "$show" "lite=$v overlap   "
    python scripts/experiments/bench_tuned.py colx_lite=$v -- $A --no-overlap 2>/dev/null | python -c "$show" "lite=$v no-overlap"
  done
done | tee $O/ab.txt
for v in 0 1; do
  python scripts/experiments/bench_tuned.py colx_lite=$v -- $A --nsymb 4096 --frames 256 --steps 4 --warmup 1 2>/dev/null | python -c "$show" "lite=$v 2^18      "
  python scripts/experiments/bench_tuned.py colx_lite=$v -- $A --mc 2>/dev/null | python -c "$show" "lite=$v --mc      "
done | tee -a $O/ab.txt
