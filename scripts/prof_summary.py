"""Summarise a rocprofv3 kernel_trace.csv: per-kernel totals, and for the SSFM kernels the average over
ACTIVE launches (at least half as long as the kernel's 90th-percentile launch: the chunked step loop also issues no-op
launches, and a few launches that start beside the receiver's CMA waves run much longer than the rest)."""
import csv, sys, collections, re


def kname(raw):
    """kernel name without namespace, return type, template and argument lists: `void (anonymous namespace)::k_row_t<false>(...)` -> k_row"""
    n = raw.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void\s+", "", n).split("(")[0]
    n = re.sub(r"<.*$", "", n)
    return "k_row" if n == "k_row_t" else n[:48]


rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = kname(r["Kernel_Name"])
    agg[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print("| kernel | launches | total ms | active launches | avg active us | % of GPU time |")
print("|---|---|---|---|---|---|")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    ref = sorted(v)[min(len(v) - 1, int(0.9 * len(v)))]
    act = [x for x in v if x >= 0.5 * ref] or v        # active launches: at least half as long as the 90th percentile (the chunked
                                                       # step loop also issues launches after every frame has finished)
    print("| `%s` | %d | %.2f | %d | %.1f | %.1f |" % (k, len(v), sum(v) / 1e3, len(act), sum(act) / len(act), 100 * sum(v) / tot))
