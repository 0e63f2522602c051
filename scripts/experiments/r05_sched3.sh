#!/bin/bash
# round 5: k_row256r under max-ilp scheduling (only ssfm_row256.o differs), eight alternating pairs on one box
O=gpurun_out/r05_sched3; mkdir -p $O
timeout -k 10 500 python scripts/crc_set.py lib=r256 > $O/crc.txt 2> $O/crc.err || { tail -3 $O/crc.err; exit 1; }
if diff profiles/r05_crc_before_split.txt $O/crc.txt > $O/crc.diff; then echo "crc set of the max-ilp row pass identical: $(wc -l < $O/crc.txt) lines"; else echo "CRC DIFFERENCES: $(grep -c '^<' $O/crc.diff) lines"; fi
A="--steps 10 --warmup 2 --no-cpu-baseline --mc-rounds 0 --no-gateway --no-cohmix-line --configs no --no-single-frame --no-overlap"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]
print(sys.argv[1], "%.4f Gs/s fibre %.2f" % (d["value"], c["fibre_ms_per_step"]), {k:round(v["avg_launch_us"],1) for k,v in r["kernels"].items()})'
for rep in 1 2 3 4 5 6 7 8; do
  for v in base r256; do
    python scripts/experiments/bench_with_lib.py $v $A 2>/dev/null | python -c "$show" "$v"
  done
done | tee $O/ab.txt
python - <<'PY'
import re
rows = [l.split() for l in open("gpurun_out/r05_sched3/ab.txt")]
import statistics as st
for v in ("base", "r256"):
    t = [float(re.search(r"k_row256r': ([\d.]+)", " ".join(r)).group(1)) for r in rows if r[0] == v]
    f = [float(r[4]) for r in rows if r[0] == v]
    print(v, "k_row256r us: mean %.1f median %.1f min %.1f max %.1f | fibre ms: mean %.2f median %.2f" % (st.mean(t), st.median(t), min(t), max(t), st.mean(f), st.median(f)))
PY
