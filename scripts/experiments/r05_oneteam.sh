#!/bin/bash
# round 5: the one-team column sweep -- stamps on this box (which XCD is late?), the XCD probe, and one PMC pass per frame shape
export TMPDIR=/tmp
O=gpurun_out/r05_oneteam; mkdir -p $O
R=$PWD
hipcc --offload-arch=gfx950 -O3 scripts/experiments/micro/xcd_speed.hip -o /tmp/xcd_speed 2>/dev/null && timeout -k 5 120 /tmp/xcd_speed > $O/xcd_speed.txt; cat $O/xcd_speed.txt
for spec in "32 no 1024 16 gps-" "16 no 16384 1 g-s-" "1024 no 1024 1 g-s-"; do
  bash scripts/experiments/stamps.sh run $spec >> $O/stamps.txt 2>&1 || { tail -5 $O/stamps.txt; exit 1; }
done
grep -E "^F=|by HW_REG|sum|slot store|first / second" $O/stamps.txt
COMMON="--steps 1 --warmup 0 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame --no-gateway --no-cohmix-line --configs no"
for tag in c1 c2frame big; do
  case $tag in c1) A="--frames 256";; c2frame) A="--nch 16 --frames 32";; big) A="--nsymb 16384 --frames 16 --share-device no";; esac
  echo "== $tag" >> $O/pmc.txt
  for C in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum"; do
    rm -rf $O/pmc
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$O/pmc -- python3 bench.py $COMMON $A > /dev/null 2>&1 || { echo "pmc pass failed: $tag $C" >> $O/pmc.txt; continue; }
    f=$(find $O/pmc -name "*counter_collection.csv" | head -1)
    echo "-- pass: $C" >> $O/pmc.txt
    python3 scripts/pmc_summary.py $f | grep -E "^k_colx16|^k_row" >> $O/pmc.txt
    rm -rf $O/pmc
  done
done
cat $O/pmc.txt
