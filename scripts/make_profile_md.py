"""dev tool: turn gpurun_out/final/{kernel_trace_summary.md,kernel_stats.csv,bench_*.json} (scripts/final_round.sh) into the
files of record under profiles/."""
import csv, json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(R, "gpurun_out", "final")
out = []
out.append("# Round 1 final: rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame` (F = 1024 C1 frames, MI355X)\n")
out.append("Command (scripts/final_round.sh): `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame`\n")
out.append("Default path: fused column sweep `k_colx16` (inverse column pass of step s + step controller + forward column pass of step s+1) and `k_row`: two launches, two HBM sweeps per SSFM step.\n")
out.append("## Per-kernel summary from the kernel trace (scripts/prof_summary.py; 'active' = launches longer than 20 us --\nthe chunked step loop also issues launches that return at once when every frame has reached the fibre end)\n")
out.append(open(os.path.join(F, "kernel_trace_summary.md")).read())
out.append("\n## rocprofv3 --stats (kernel_stats.csv, top rows)\n\n```")
rows = list(csv.reader(open(os.path.join(F, "kernel_stats.csv"))))
for r in rows[:12]:
    out.append(",".join(x[:60] for x in r))
out.append("```\n")
d = json.loads(open(os.path.join(F, "bench_default.json")).read().strip().splitlines()[-1])
tot = sum(float(r[2]) for r in rows[1:] if "k_colx16" in r[0] or "k_row(" in r[0]) / 1e6
out.append("## Cross-check with bench.py's live HIP-event measurement (same configuration, separate run)\n")
out.append("* bench.py: fibre %.1f ms per step (HIP events on the launch stream) = %.3f ms per step-launch over %d launches, roofline.achieved %.0f GB/s (272 B x sample-steps / fibre time), frac %.3f, traffic %.3g B per step-launch (PMC, r01_traffic.json)" % (
    d["config"]["fibre_ms_per_step"], d["roofline"]["ms_per_step_launch"], d["roofline"]["launches"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["traffic"]))
out.append("* kernel trace: k_colx16 + k_row total %.1f ms over 3 batches (1 warm-up + 2 timed) = %.1f ms per batch = %.3f ms per step-launch (76 launches per batch); the HIP-event figure adds k_umax, the launch gaps of the chunked loop and, in the default run, the receiver of the previous batch sharing the GPU on its own stream." % (tot, tot / 3, tot / 3 / 76))
out.append("* one active launch processes F x nfft = 1024 x 65536 dual-pol samples; per SSFM step the two kernels move ~129 B per sample (PMC, profiles/r01_traffic.json) against the SURVEY's 272 B accounting.\n")
open(os.path.join(R, "profiles", "r01_final_kernel_trace.md"), "w").write("\n".join(out))
lines = [open(os.path.join(F, "bench_%s.json" % n)).read().strip().splitlines()[-1] for n in ("default", "cohmix", "mc")]
open(os.path.join(R, "profiles", "r01_final_bench.jsonl"), "w").write("\n".join(lines) + "\n")
print("profiles written")
