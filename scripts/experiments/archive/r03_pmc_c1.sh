#!/bin/bash
# round 3: counters of the step kernels at C1 (256 frames, fibre alone) after the register form of the row pass:
# LDS conflicts, wait / issue split, instruction mix, L2 traffic.  One counter set per rocprofv3 pass, --kernel-trace only.
export TMPDIR=/tmp
R=$PWD; O=gpurun_out/r03pmc; mkdir -p $O
B="python3 bench.py --frames 256 --steps 1 --warmup 0 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-single-frame --no-overlap --no-gateway --no-cohmix-line"
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-48)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/p_$tag -- $B > /dev/null 2> $O/p_$tag.err || { echo "pass failed: $set"; tail -2 $O/p_$tag.err; continue; }
  f=$(find $O/p_$tag -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py $f > $O/$tag.txt
  echo "== pass: $set"; grep -E "k_colx16|k_row" $O/$tag.txt
  rm -rf $O/p_$tag $O/p_$tag.err
done
