#!/bin/bash
# round 3: LDS counters of the step kernels at 2^20 (16 frames) after k_row4k's padded level-2 twiddles
export TMPDIR=/tmp
R=$PWD; O=gpurun_out/r03pmc4; mkdir -p $O
B="python3 bench.py --nsymb 16384 --frames 16 --steps 1 --warmup 0 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-single-frame --no-overlap --no-gateway"
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-48)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/p_$tag -- $B > /dev/null 2> $O/p_$tag.err || { echo "pass failed: $set"; tail -2 $O/p_$tag.err; continue; }
  f=$(find $O/p_$tag -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py $f > $O/$tag.txt
  echo "== pass: $set"; grep -E "k_colx16|k_row" $O/$tag.txt
  rm -rf $O/p_$tag $O/p_$tag.err
done
