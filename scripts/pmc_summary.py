"""Summarise rocprofv3 --pmc counter_collection.csv: mean per ACTIVE dispatch of each counter, per kernel."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:40]
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if not k.startswith("k_"): continue
    out = []
    for c, v in sorted(cs.items()):
        top = [x for x in v if x >= 0.5 * max(v)] or v   # active launches: at least half of the largest (the chunked loop
                                                         # also issues launches that return at once)
        out.append("%s=%.4g" % (c, sum(top) / len(top)))
    print(k, " ".join(out))
