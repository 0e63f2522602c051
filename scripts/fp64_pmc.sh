#!/bin/bash
# FP64 VALU instructions per launch of the step kernels from the SQ counters (one rocprofv3 --pmc pass per workload, with
# --kernel-trace only): SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 count wave instructions; flop = (2 FMA + MUL + ADD) x 64 lanes
# (every lane of these kernels is active).  Writes gpurun_out/fp64/<tag>.txt and gpurun_out/fp64/fp64.json (copy the latter to
# profiles/rNN_fp64.json: bench.py's roofline.fp64 reads it, labelled offline like the HBM traffic).
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/fp64
C="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"
COMMON="--steps 1 --warmup 0 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame --no-gateway --no-cohmix-line --configs no"
run() {   # tag, bench arguments
  tag=$1; shift
  rm -rf gpurun_out/fp64/pmc
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/fp64/pmc -- python3 bench.py $COMMON "$@" > /dev/null 2>&1 || { echo "$tag: rocprofv3 failed"; return 1; }
  f=$(find gpurun_out/fp64/pmc -name "*counter_collection.csv" | head -1)
  python3 scripts/pmc_summary.py $f > gpurun_out/fp64/$tag.txt
  grep -E "k_colx16|k_row|k_col_fwd|k_col_inv" gpurun_out/fp64/$tag.txt
  rm -rf gpurun_out/fp64/pmc
}
run c1 --frames 256 || exit 1
run c1_mc --frames 256 --mc || exit 1
run c2_frame --nch 16 --frames 32 || exit 1
run big --nsymb 16384 --frames 16 --share-device no || exit 1
run big_pmd --nsymb 16384 --frames 16 --flag gps- --share-device no || exit 1
run mid --nsymb 4096 --frames 64 || exit 1
run mid_pmd --nsymb 4096 --frames 64 --flag gps- || exit 1
python3 - <<'PY' > gpurun_out/fp64/fp64.json
import json, os
d = "gpurun_out/fp64"
samples = {"c1": 256 * 65536, "c1_mc": 256 * 65536, "c2_frame": 32 * 16 * 65536, "big": 16 << 20, "big_pmd": 16 << 20, "mid": 64 << 18, "mid_pmd": 64 << 18}
what = {"c1": "C1 frames 'g-s-' (headline)", "c1_mc": "C1 frames 'gps-', 100 fresh waveplates per frame (--mc)", "c2_frame": "16-channel 'gps-' frames (config[2])",
        "big": "2^20-sample frames 'g-s-' (config[4]'s frame), fused step", "big_pmd": "2^20-sample 'gps-' frames, fused step",
        "mid": "2^18-sample frames 'g-s-'", "mid_pmd": "2^18-sample 'gps-' frames"}
out = {"_how": "scripts/fp64_pmc.sh: one rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 pass per workload of "
               "`python3 bench.py --steps 1 --warmup 0 ...`; mean over ACTIVE launches; flop = (2 FMA + MUL + ADD) x 64 lanes per wave instruction",
       "peak_TFLOPs": 78.6, "peak_source": "256 CUs x 4 SIMDs x 16 FP64 lanes x 2 flop x 2.4 GHz (AMD's vector-FP64 figure for MI355X); "
                                           "scripts/experiments/micro/fp64_peak.hip measures what independent v_fma_f64 chains sustain",
       "workloads": {}}
for tag, n in samples.items():
    p = os.path.join(d, tag + ".txt")
    if not os.path.exists(p):
        continue
    ks = {}
    for line in open(p):
        k, _, rest = line.partition(" ")
        v = {t.partition("=")[0]: float(t.partition("=")[2]) for t in rest.split()}
        if "SQ_INSTS_VALU_FMA_F64" not in v or not k.startswith(("k_col", "k_row", "k_pmd_tab")):
            continue
        flop = (2 * v["SQ_INSTS_VALU_FMA_F64"] + v.get("SQ_INSTS_VALU_MUL_F64", 0) + v.get("SQ_INSTS_VALU_ADD_F64", 0)) * 64
        ks[k] = {"fma": v["SQ_INSTS_VALU_FMA_F64"], "mul": v.get("SQ_INSTS_VALU_MUL_F64"), "add": v.get("SQ_INSTS_VALU_ADD_F64"),
                 "trans": v.get("SQ_INSTS_VALU_TRANS_F64"), "flop_per_launch": flop, "flop_per_sample_per_launch": flop / n}
    out["workloads"][tag] = {"what": what[tag], "samples_per_launch": n, "kernels": ks}
print(json.dumps(out, indent=1))
PY
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/fp64/fp64.json"))
for t, w in d["workloads"].items():
    print(t, {k: round(v["flop_per_sample_per_launch"], 1) for k, v in w["kernels"].items()})
PY
