#!/bin/bash
# dev: instruction mix and LDS conflict counters of the step kernels (rocprofv3 --pmc, own passes, kernel trace only)
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/pmcmix
rm -rf $O; mkdir -p $O
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/p_$tag -- python3 bench.py --frames 256 --steps 1 --warmup 0 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame > /dev/null 2>&1 || { echo "pass failed: $set"; continue; }
  f=$(find $O/p_$tag -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py $f > $O/$tag.txt
  grep -E "k_colx16|k_row|counter|kernel" $O/$tag.txt | head -8
  rm -rf $O/p_$tag
done
