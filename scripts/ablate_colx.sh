#!/bin/bash
# dev tool: k_colx16 ablations (PLX_SSFM_DBG bits: 1 skip column FFTs, 4 skip Kerr math, 16 skip the frame-barrier wait);
# results are NOT valid physics (and with 16 the step sizes are wrong) -- timing only
export TMPDIR=/tmp
R=$PWD
for dbg in 0 1 4 5 16 21; do
  rm -rf gpurun_out/abl_$dbg
  PLX_SSFM_DBG=$dbg timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abl_$dbg -- python3 bench.py --frames 256 --steps 1 --warmup 0 --no-cpu-baseline --no-overlap > /dev/null 2>&1
  f=$(find gpurun_out/abl_$dbg -name "*kernel_trace.csv" | head -1)
  echo "dbg=$dbg $(python scripts/prof_summary.py $f | grep -E 'k_row|k_colx16' | awk -F'|' '{printf "%s %s launches %s us; ", $2, $5, $6}')"
  rm -rf gpurun_out/abl_$dbg
done
