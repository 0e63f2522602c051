#!/bin/bash
# HBM traffic of the SSFM step kernels from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes, with --kernel-trace only.  Fused (default) and three-sweep (PLX_SSFM_NO_FUSE=1) step.
# Writes gpurun_out/traffic/{fused,plain}_{FETCH_SIZE,WRITE_SIZE}.txt and gpurun_out/traffic/traffic.json
# (copy the latter to profiles/rNN_traffic.json).   usage: scripts/traffic_pmc.sh [frames]
export TMPDIR=/tmp
R=$PWD
F=${1:-256}
mkdir -p gpurun_out/traffic
for mode in fused plain; do
  if [ $mode = plain ]; then export PLX_SSFM_NO_FUSE=1; else unset PLX_SSFM_NO_FUSE; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/traffic/pmc_$c
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/traffic/pmc_$c -- python3 bench.py --frames $F --steps 1 --warmup 0 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame --no-gateway --no-cohmix-line --configs no > /dev/null 2>&1 || exit 1
    f=$(find gpurun_out/traffic/pmc_$c -name "*counter_collection.csv" | head -1)
    python scripts/pmc_summary.py $f > gpurun_out/traffic/${mode}_$c.txt
    grep -E "k_colx16|k_row|k_col_fwd|k_col_inv" gpurun_out/traffic/${mode}_$c.txt || true
    rm -rf gpurun_out/traffic/pmc_$c
  done
done
unset PLX_SSFM_NO_FUSE
# 2^20-sample frames (BASELINE config[4]'s frame), fused step: k_colx16 + k_row4k, 16 frames per launch
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/traffic/pmc_$c
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/traffic/pmc_$c -- python3 bench.py --nsymb 16384 --frames 16 --steps 1 --warmup 0 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame --no-gateway --no-cohmix-line --configs no > /dev/null 2>&1 || exit 1
  f=$(find gpurun_out/traffic/pmc_$c -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py $f > gpurun_out/traffic/big_$c.txt
  grep -E "k_colx16|k_row" gpurun_out/traffic/big_$c.txt
  rm -rf gpurun_out/traffic/pmc_$c
done
# 16-channel 'sepfields' WDM frames (BASELINE config[2]'s frame, 'gps-'), fused step: k_colx16 + k_row256r<PMD>, 32 frames per launch
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/traffic/pmc_$c
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/traffic/pmc_$c -- python3 bench.py --nch 16 --frames 32 --steps 1 --warmup 0 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame --no-gateway --no-cohmix-line --configs no > /dev/null 2>&1 || exit 1
  f=$(find gpurun_out/traffic/pmc_$c -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py $f > gpurun_out/traffic/wdm_$c.txt
  grep -E "k_colx16|k_row" gpurun_out/traffic/wdm_$c.txt
  rm -rf gpurun_out/traffic/pmc_$c
done
# 2^18-sample frames (the size Run_my_PDM_QPSK.m ships with), fused step: k_colx16 + k_rowreg<10>, 64 frames per launch
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/traffic/pmc_$c
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/traffic/pmc_$c -- python3 bench.py --nsymb 4096 --frames 64 --steps 1 --warmup 0 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame --no-gateway --no-cohmix-line --configs no > /dev/null 2>&1 || exit 1
  f=$(find gpurun_out/traffic/pmc_$c -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py $f > gpurun_out/traffic/mid_$c.txt
  grep -E "k_colx16|k_row" gpurun_out/traffic/mid_$c.txt
  rm -rf gpurun_out/traffic/pmc_$c
done
python scripts/traffic_json.py gpurun_out/traffic $F > gpurun_out/traffic/traffic.json && cat gpurun_out/traffic/traffic.json
