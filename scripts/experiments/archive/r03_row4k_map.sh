#!/bin/bash
# round 3: k_row4k with a row's users dealt to one XCD -- time (A/B against libpolmux_hip_old.so) and FETCH_SIZE
export TMPDIR=/tmp
R=$PWD; O=gpurun_out/r03map; mkdir -p $O
bash scripts/experiments/ab.sh run old base -- 16 g-s- 16384 2>&1 | grep fibre
bash scripts/experiments/ab.sh run old base -- 64 g-s- 16384 2>&1 | grep fibre
B="python3 bench.py --nsymb 16384 --frames 16 --steps 1 --warmup 0 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-single-frame --no-overlap --no-gateway"
for c in FETCH_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/p_$c -- $B > /dev/null 2> $O/p_$c.err || { echo "pass failed"; tail -2 $O/p_$c.err; continue; }
  f=$(find $O/p_$c -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py $f | grep -E "k_colx16|k_row"
  rm -rf $O/p_$c
done
