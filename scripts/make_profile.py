"""dev tool: turn gpurun_out/<tag>/* (scripts/final_round.sh lines|trace|traffic <tag>) into the files of record under profiles/.
usage: make_profile.py <tag> <round, e.g. r05>"""
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(R, "gpurun_out", sys.argv[1])
RN = sys.argv[2]
P = os.path.join(R, "profiles")


def rd(name):
    return open(os.path.join(F, name)).read()


out = ["# " + RN + ": rocprofv3 --kernel-trace --stats of the bench (F = 1024 C1 frames, 16 Tx sequences, MI355X)\n",
       "Commands (scripts/final_round.sh trace): `rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python3 bench.py --steps 2 --warmup 1 "
       "--no-cpu-baseline --no-single-frame --mc-rounds 0 --no-gateway --no-cohmix-line --configs no` (default: receiver of batch i on a second stream beside the fibre of "
       "batch i+1), the same with `--no-overlap`, and `--nsymb 16384 --frames 16 --variants 1 --no-overlap` (2^20-sample frames) and `--nch 16 --spans 2 --nf 5 --frames 32 --no-overlap` (16-channel WDM frames of BASELINE config[2]).  The summaries count as ACTIVE the launches at least half as long as the kernel's 90th-percentile launch (the chunked step loop also issues launches that return at once; a few launches that start beside the receiver's CMA waves run much longer than the rest).  The bench line of record is the un-profiled run (`" + RN + "_bench.jsonl` line 1: `roofline.kernels` from HIP events between the launches).\n",
       "Default path: `k_compact` (active list) + fused column sweep `k_colx16` (inverse column pass of step s + step controller + forward column pass of "
       "step s+1; teams of 32 workgroups claim frames one at a time; next tile staged by LDS-DMA) + `k_row256r<false>` (the register form of the 256-point "
       "row pass: one wave per 2 rows x 2 polarisations): two HBM sweeps per SSFM step.\n",
       "## Per-kernel summary, default run (scripts/prof_summary.py; 'active' = launches at least half as long as the kernel's longest)\n",
       rd("kernel_trace_summary.md"),
       "\n## Per-kernel summary, `--no-overlap` run\n", rd("kernel_trace_no_overlap_summary.md"),
       "\n## Per-kernel summary, 2^20-sample frames (16 per launch; `k_row4k` is the row pass; `--no-overlap` keeps the fused step)\n", rd("kernel_trace_2pow20_summary.md"),
       "\n## Per-kernel summary, 16-channel WDM frames (BASELINE config[2]'s frame, 'gps-' with 100 waveplates; 32 frames per launch)\n", rd("kernel_trace_wdm16_summary.md"),
       "\n## Per-kernel summary, 2^18-sample frames (4096 symbols x 64 samples, the size Run_my_PDM_QPSK.m ships with; 256 per launch; `k_rowreg` is the row pass)\n",
       rd("kernel_trace_2pow18_summary.md") if os.path.exists(os.path.join(F, "kernel_trace_2pow18_summary.md")) else "(not traced)",
       "\n## rocprofv3 --stats (kernel_stats.csv, top rows, default run)\n\n```", rd("kernel_stats_head.csv").rstrip(), "```\n"]
d = json.loads(rd("bench_default.json").strip().splitlines()[-1])
r = d["roofline"]
ROW = [k for k in r["kernels"] if k.startswith("k_row")][0]
out.append("## Cross-check with bench.py's live HIP-event figures (separate, un-profiled run: `" + RN + "_bench.jsonl` line 1)\n")
out.append("* `roofline`: kernel %s, avg active launch %.1f us, achieved %.0f GB/s, frac %.3f; `%s` %.1f us (%.3f); step group %.3f.  One active launch of a sweep "
           "= 64 B x %.1f frames x 65536 samples = %.3f GB." % (r["kernel"], r["avg_launch_us"], r["achieved"], r["frac"], ROW, r["kernels"][ROW]["avg_launch_us"],
                                                                 r["kernels"][ROW]["frac_of_8TBs"], r["step_group"]["frac_of_8TBs"],
                                                                 r["algorithmic_bytes_per_launch"] / 64 / 65536, r["algorithmic_bytes_per_launch"] / 1e9))
out.append("* the `--no-overlap` trace is the fibre ALONE; the default trace and the bench line's figures are beside the receiver.\n")
open(os.path.join(P, RN + "_kernel_trace.md"), "w").write("\n".join(out))
shutil.copy(os.path.join(F, "bench.jsonl"), os.path.join(P, RN + "_bench.jsonl"))
shutil.copy(os.path.join(F, "traffic.json"), os.path.join(P, RN + "_traffic.json"))
print("profiles/%s_kernel_trace.md, %s_bench.jsonl, %s_traffic.json written" % (RN, RN, RN))
