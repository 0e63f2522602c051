"""dev tool: for the kernels of the fibre's step loop (k_compact, k_colx16, k_row) in a rocprofv3 kernel_trace.csv:
idle time between the end of the previous one and the start of the next, per kernel, and their durations."""
import csv, sys, re, collections
def kname(raw):
    n = raw.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void\s+", "", n).split("(")[0]
    n = re.sub(r"<.*$", "", n)
    return "k_row" if n == "k_row_t" else n
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kname(r["Kernel_Name"])) for r in rows)
fib = [e for e in ev if e[2] in ("k_compact", "k_colx16", "k_row")]
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
for a, b in zip(fib, fib[1:]):
    if b[1] - b[0] > 20000 or b[2] == "k_compact":      # active launches
        gap[b[2]].append((b[0] - a[1]) / 1e3)
        dur[b[2]].append((b[1] - b[0]) / 1e3)
for k in gap:
    g, d = sorted(gap[k]), dur[k]
    print("%-10s n=%d  gap before: mean %.1f us  median %.1f  p90 %.1f  max %.1f | duration mean %.1f us" % (k, len(g), sum(g) / len(g), g[len(g) // 2], g[int(0.9 * len(g))], g[-1], sum(d) / len(d)))
