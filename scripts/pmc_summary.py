"""Summarise rocprofv3 --pmc counter_collection.csv: mean per ACTIVE dispatch of each counter, per kernel."""
import csv, sys, collections, re


def kname(raw):
    """kernel name without namespace, return type, template and argument lists: `void (anonymous namespace)::k_row_t<false>(...)` -> k_row"""
    n = raw.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void\s+", "", n).split("(")[0]
    n = re.sub(r"<.*$", "", n)
    return "k_row" if n == "k_row_t" else n[:48]


rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = kname(r["Kernel_Name"])
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if not k.startswith("k_"): continue
    out = []
    for c, v in sorted(cs.items()):
        top = [x for x in v if x >= 0.5 * max(v)] or v   # active launches: at least half of the largest (the chunked loop
                                                         # also issues launches that return at once)
        out.append("%s=%.4g" % (c, sum(top) / len(top)))
    print(k, " ".join(out))
