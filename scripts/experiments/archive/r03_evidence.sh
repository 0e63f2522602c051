#!/bin/bash
# round 3, first GPU call: the memory side alone (sweep_modes), the counter list, and counters / stamps for 2^20-sample frames.
export TMPDIR=/tmp
R=$PWD; O=gpurun_out/r03ev; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/sweep_modes scripts/experiments/micro/sweep_modes.hip && timeout -k 10 300 /tmp/sweep_modes > $O/sweep_modes.txt 2>&1
echo "sweep_modes done: $?"
rocprofv3 -L > $O/counters.txt 2>&1
B4="python3 bench.py --nsymb 16384 --frames 16 --steps 1 --warmup 0 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-single-frame --no-overlap"
timeout -k 10 300 $B4 > $O/c4_bench.json 2> $O/c4_bench.err || { echo "c4 bench failed"; tail -5 $O/c4_bench.err; exit 1; }
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-48)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/p_$tag -- $B4 > /dev/null 2> $O/p_$tag.err || { echo "pass failed: $set"; tail -2 $O/p_$tag.err; continue; }
  f=$(find $O/p_$tag -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py $f > $O/$tag.txt
  echo "== $set"; grep -E "k_colx16|k_row" $O/$tag.txt
  rm -rf $O/p_$tag $O/p_$tag.err
done
bash scripts/experiments/stamps.sh run 16 no 16384 > $O/stamps_c4.txt 2>&1; tail -30 $O/stamps_c4.txt
bash scripts/experiments/stamps.sh run 1024 no 1024 > $O/stamps_c1.txt 2>&1; tail -30 $O/stamps_c1.txt
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
