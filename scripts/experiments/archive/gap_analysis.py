"""dev tool: idle gaps between consecutive kernels of a rocprofv3 kernel_trace.csv (all streams merged)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]) for r in rows))
big = []
tot_gap = 0
busy_end = ev[0][1]
for i in range(1, len(ev)):
    s, e, n = ev[i]
    if s > busy_end:
        g = s - busy_end
        tot_gap += g
        big.append((g / 1e3, ev[i - 1][2], n))
    busy_end = max(busy_end, e)
span = (max(e for s, e, n in ev) - ev[0][0]) / 1e6
print("span %.1f ms, idle %.1f ms in %d gaps" % (span, tot_gap / 1e6, len(big)))
import collections
c = collections.Counter()
for g, a, b in big:
    c[(a, b)] += g
for k, v in c.most_common(8):
    n = sum(1 for g, a, b in big if (a, b) == k)
    print("%-20s -> %-20s total %.2f ms over %d gaps (avg %.1f us)" % (k[0], k[1], v / 1e3, n, v / n))
