#!/bin/bash
# HBM traffic of the SSFM step kernels from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes, with --kernel-trace only.  Writes gpurun_out/traffic/{fetch,write}.txt
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/traffic
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/traffic/pmc_$c
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/traffic/pmc_$c -- python3 bench.py --frames 256 --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-single-frame > /dev/null 2>&1
  f=$(find gpurun_out/traffic/pmc_$c -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py $f > gpurun_out/traffic/$c.txt
  cat gpurun_out/traffic/$c.txt | grep -E "k_colx16|k_row|k_col_fwd|k_col_inv"
  rm -rf gpurun_out/traffic/pmc_$c
done
