#!/bin/bash
# round 3: lane rotation on padded rows (k_row / k_row4k): same-box A/B, LDS conflict counters, single-frame latency, then the GPU suite
export TMPDIR=/tmp
R=$PWD; O=gpurun_out/r03ev; mkdir -p $O
bash scripts/experiments/ab.sh run old base -- 1024 g-s- 1024 2>&1 | grep -v amdgpu.ids
bash scripts/experiments/ab.sh run old base -- 16 g-s- 16384 2>&1 | grep -v amdgpu.ids
bash scripts/experiments/ab.sh run old base -- 128 gps- 1024 2>&1 | grep -v amdgpu.ids
for cfg in "c1:--frames 256" "c4:--nsymb 16384 --frames 16"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/$O/p_lds_$tag -- python3 bench.py $args --steps 1 --warmup 0 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame --no-gateway > /dev/null 2> $O/p_lds_$tag.err || { echo "pmc pass failed"; tail -3 $O/p_lds_$tag.err; }
  f=$(find $O/p_lds_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 scripts/pmc_summary.py $f | grep -E "k_colx16|k_row" | sed "s/^/$tag /"
  rm -rf $O/p_lds_$tag
done
timeout -k 10 90 python3 scripts/experiments/diag_small.py 1 2>&1 | grep -v amdgpu.ids | tail -7
timeout -k 10 90 python3 scripts/experiments/diag_small.py 1 profile 2>&1 | grep -v amdgpu.ids | tail -3
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gpu_tests.txt
