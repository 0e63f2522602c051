"""dev tool: regenerate the table of DESIGN.md section 9 from profiles/r04_bench.jsonl (the lines of scripts/final_round.sh, in its order)."""
import json
import os

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = [json.loads(l) for l in open(os.path.join(R, "profiles", "r04_bench.jsonl"))]


def kern(x, names=True):
    return ", ".join(("`%s` " % k if names else "") + "%d (%.3f)" % (round(v["avg_launch_us"]), v["frac_of_8TBs"]) for k, v in x["roofline"]["kernels"].items())


def g(x):
    return x["roofline"]["step_group"]["frac_of_8TBs"]


def fr(x):
    c = x["config"]
    return c["fibre_ms_per_step"], c["rxdsp_ms_per_step"]


sf = d[0]["config"].get("single_frame") or {}
new = '''| default: C1, 1024 frames, receiver beside the next fibre | **%.4f** (0.644 in round 3) | %.1f (%.1f / %.1f beside) | %s | %.3f |
| `--no-overlap` (the fibre alone) | %.4f | %.1f (%.1f / %.1f) | %s | %.3f |
| `--frontend cohmix` (the reference's own front end) | %.4f | %.1f | %s | %.3f |
| `--power-ladder` (9 ... 169 steps per frame) | %.4f | %.1f | %s on the shrinking list | %.3f |
| `--mc` (fresh waveplates per frame and step, 100 plates) | %.4f | %.1f | %s (`k_row256r<PMD>`: FP64-bound) | %.3f |
| config[2]: `--nch 16 --spans 10 --nf 5 --frames 32` | **%.4f** (first timed this round) | %.1f (%.1f / %.1f) | %s (`k_row256r<PMD>`) | %.3f |
| 2^20-sample frames, 16 per batch, 1 span, shared plan (auto) | **%.4f** | %.1f (%.1f / receiver beside) | %s | %.3f (192 B per sample-step) |
| the same on the fused step (`--share-device no`) | %.4f | %.1f (%.1f / %.1f serial) | %s | %.3f (128 B) |
| config[4] as stated: 8 frames x 40 spans x ladder, shared plan | %.4f (0.0148 in round 3) | %.1f (%.1f / %.1f beside) | %s on the shrinking list | %.3f |
| 2^18-sample frames (`--nsymb 4096 --frames 256`: the frame `Run_my_PDM_QPSK.m:21-24` ships with; no BASELINE config) | %.4f (0.44 with `k_row` at the start of the day) | %.1f (%.1f / %.1f beside) | %s | %.3f |
| 2^20-sample 'gps-' frames (100 waveplates), 16 per batch, fused step (no BASELINE config) | %.4f (three sweeps + `k_row` 0.17 before) | %.1f (%.1f / %.1f serial) | %s | %.3f |
| Monte-Carlo leg (2 x 512, two rounds in flight) / config[3] as stated (1 x 1024) | %.0f / %.0f realisations/s (5487 in round 3) | | | |
| one frame alone (M1 read literally) | %.5f | fibre %.2f + receiver %.1f (19.8 in round 3) | | |
''' % (d[0]["value"], d[0]["ms_per_step"], *fr(d[0]), kern(d[0]), g(d[0]),
       d[1]["value"], d[1]["ms_per_step"], *fr(d[1]), kern(d[1], False), g(d[1]),
       d[2]["value"], d[2]["ms_per_step"], kern(d[2], False), g(d[2]),
       d[3]["value"], d[3]["ms_per_step"], kern(d[3], False), g(d[3]),
       d[4]["value"], d[4]["ms_per_step"], kern(d[4], False), g(d[4]),
       d[8]["value"], d[8]["ms_per_step"], *fr(d[8]), kern(d[8], False), g(d[8]),
       d[5]["value"], d[5]["ms_per_step"], fr(d[5])[0], kern(d[5]), g(d[5]),
       d[6]["value"], d[6]["ms_per_step"], *fr(d[6]), kern(d[6]), g(d[6]),
       d[7]["value"], d[7]["ms_per_step"], *fr(d[7]), kern(d[7], False), g(d[7]),
       d[9]["value"], d[9]["ms_per_step"], *fr(d[9]), kern(d[9]), g(d[9]),
       d[10]["value"], d[10]["ms_per_step"], *fr(d[10]), kern(d[10]), g(d[10]),
       d[0]["mc"]["realisations_per_s"], d[0]["mc"]["strong_scaling"]["realisations_per_s"],
       sf.get("gsample_per_s", 0), sf.get("fibre_ms", 0), sf.get("rx_ms", 0))
p = os.path.join(R, "DESIGN.md")
s = open(p).read()
a = s.index('| default: C1, 1024 frames, receiver beside the next fibre |')
b = s.index('| one frame alone (M1 read literally) |')
b = s.index('\n', b) + 1
open(p, "w").write(s[:a] + new + s[b:])
print(new)
