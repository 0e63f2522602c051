#!/bin/bash
# dev experiment: distribution of k_colx16 launch durations in the default (overlapped) bench -- do launches beside the receiver double?
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/colhist
rm -rf $O; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame --mc-rounds 0 > /dev/null 2>&1 || exit 1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
col = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_colx16" in r["Kernel_Name"]]
cma = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_cma16" in r["Kernel_Name"]]
d = [(e - s) / 1e3 for s, e in col]
act = [x for x in d if x > 0.5 * max(d) * 0.4]
h = collections.Counter(int(x // 100) * 100 for x in act)
print("k_colx16 active launches %d; histogram of durations (us):" % len(act))
for k in sorted(h): print("  %5d-%5d: %d" % (k, k + 99, h[k]))
inside = [((e - s) / 1e3) for s, e in col if any(cs < e and s < ce for cs, ce in cma) and (e - s) / 1e3 > 400]
outside = [((e - s) / 1e3) for s, e in col if not any(cs < e and s < ce for cs, ce in cma) and (e - s) / 1e3 > 400]
print("beside k_cma16: %d launches, mean %.0f us; otherwise: %d launches, mean %.0f us" % (len(inside), sum(inside) / max(len(inside), 1), len(outside), sum(outside) / max(len(outside), 1)))
print("k_cma16 launches (ms):", [round((e - s) / 1e6, 1) for s, e in cma])
PY
rm -rf $O/prof
