#!/bin/bash
# dev tool: k_row ablations (PLX_SSFM_DBG bits: 2 skip row FFTs, 8 skip exp(-i beta dz)); results are NOT valid physics
export TMPDIR=/tmp
R=$PWD
for dbg in 0 2 8 10; do
  rm -rf gpurun_out/abl_$dbg
  PLX_SSFM_DBG=$dbg timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abl_$dbg -- python3 bench.py --frames 256 --steps 1 --warmup 0 --no-cpu-baseline --no-overlap > /dev/null 2>&1
  f=$(find gpurun_out/abl_$dbg -name "*kernel_trace.csv" | head -1)
  echo "dbg=$dbg $(python scripts/prof_summary.py $f | grep -E 'k_row|k_col_fwd|k_col_inv' | awk -F'|' '{printf "%s %s us; ", $2, $6}')"
  rm -rf gpurun_out/abl_$dbg
done
