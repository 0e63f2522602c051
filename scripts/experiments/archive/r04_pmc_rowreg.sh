#!/bin/bash
# dev: where k_rowreg's wave-cycles go (2^18-sample frames, 64 per launch): issue / wait / LDS conflict counters, own rocprofv3 passes, kernel trace only
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/pmc_rowreg
rm -rf $O; mkdir -p $O
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/p_$tag -- python3 bench.py --nsymb 4096 --frames 64 --steps 1 --warmup 0 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame --no-gateway > /dev/null 2>&1 || { echo "pass failed: $set"; continue; }
  f=$(find $O/p_$tag -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py $f > $O/$tag.txt
  grep -E "k_colx16|k_rowreg" $O/$tag.txt
  rm -rf $O/p_$tag
done
