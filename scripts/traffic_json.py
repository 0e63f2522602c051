"""Assemble profiles/rNN_traffic.json from the per-counter summaries of scripts/traffic_pmc.sh.
usage: traffic_json.py <dir> <frames>"""
import json
import os
import sys

d, F = sys.argv[1], int(sys.argv[2])
samples = F * 65536


def read(mode, counter):
    out = {}
    for line in open(os.path.join(d, "%s_%s.txt" % (mode, counter))):
        k, _, rest = line.partition(" ")
        for tok in rest.split():
            name, _, val = tok.partition("=")
            if name == counter:
                out[k] = float(val)
    return out


res = {"_how": "scripts/traffic_pmc.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes of "
               "`python3 bench.py --frames %d --steps 1 --warmup 0 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap "
               "--no-single-frame`; mean over ACTIVE launches (counter >= half of the largest).  gfx950 correction "
               "(MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 1/2 of wide coalesced reads -> x2; WRITE_SIZE exact; unit KB "
               "-> x1024." % F,
       "samples_per_launch": samples}
for mode, key, kernels in (("fused", None, ["k_colx16", "k_row"]), ("plain", "plain_three_sweep", ["k_col_fwd", "k_row", "k_col_inv"])):
    fe, wr = read(mode, "FETCH_SIZE"), read(mode, "WRITE_SIZE")
    kernels = [("k_row256r" if (k == "k_row" and "k_row256r" in fe) else k) for k in kernels]   # (the register form serves 256-point rows)
    per = {k: (2 * fe[k] + wr[k]) * 1024 / samples for k in kernels if k in fe and k in wr}
    blk = {"kernels": kernels, "fetch_kb": [fe.get(k) for k in kernels], "write_kb": [wr.get(k) for k in kernels],
           "bytes_per_sample_by_kernel": per, "bytes_per_sample_step": sum(per.values())}
    if key is None:
        res.update(blk)
        res["default_path"] = "fused column sweep k_colx16 + %s (2 sweeps per SSFM step)" % kernels[1]
    else:
        blk["selected_by"] = "PLX_SSFM_NO_FUSE=1"
        res[key] = blk
if os.path.exists(os.path.join(d, "big_FETCH_SIZE.txt")):      # 2^20-sample frames, 16 per launch (k_row4k reports as k_row4k)
    fe, wr = read("big", "FETCH_SIZE"), read("big", "WRITE_SIZE")
    sb = 16 * (1 << 20)
    names = {"k_colx16": "k_colx16", "k_row4k": "k_row4k"}
    per = {names[k]: (2 * fe[k] + wr[k]) * 1024 / sb for k in names if k in fe and k in wr}
    res["frames_2pow20"] = {"samples_per_launch": sb, "kernels": ["k_colx16", "k_row4k"], "fetch_kb": [fe.get("k_colx16"), fe.get("k_row4k")],
                            "write_kb": [wr.get("k_colx16"), wr.get("k_row4k")], "bytes_per_sample_by_kernel": per,
                            "bytes_per_sample_step": sum(per.values()),
                            "note": "k_row4k: with the users of a row's tables (frames x polarisations) dealt to one XCD the tables come out of its L2 "
                                    "(64.7 B per sample; 75.5 B under the (row, frame, polarisation) grid of round 2: betat and the inter-pass twiddles re-read from HBM)"}
if os.path.exists(os.path.join(d, "wdm_FETCH_SIZE.txt")):      # BASELINE config[2]'s frame: 16 'sepfields' channels of 2^16 samples, 'gps-', 32 frames per launch
    fe, wr = read("wdm", "FETCH_SIZE"), read("wdm", "WRITE_SIZE")
    sw = 32 * 16 * 65536
    per = {k: (2 * fe[k] + wr[k]) * 1024 / sw for k in ("k_colx16", "k_row256r") if k in fe and k in wr}
    res["wdm_16ch"] = {"samples_per_launch": sw, "kernels": list(per), "fetch_kb": [fe.get(k) for k in per], "write_kb": [wr.get(k) for k in per],
                       "bytes_per_sample_by_kernel": per, "bytes_per_sample_step": sum(per.values()),
                       "note": "a frame = 16 channels x 2^16 samples = one team of the fused sweep (512 tiles); row pass = k_row256r<PMD> (100 waveplates)"}
if os.path.exists(os.path.join(d, "mid_FETCH_SIZE.txt")):      # 2^18-sample frames, 64 per launch: k_colx16 + k_rowreg
    fe, wr = read("mid", "FETCH_SIZE"), read("mid", "WRITE_SIZE")
    sm = 64 * (1 << 18)
    per = {k: (2 * fe[k] + wr[k]) * 1024 / sm for k in ("k_colx16", "k_rowreg") if k in fe and k in wr}
    res["frames_2pow18"] = {"samples_per_launch": sm, "kernels": list(per), "fetch_kb": [fe.get(k) for k in per], "write_kb": [wr.get(k) for k in per],
                            "bytes_per_sample_by_kernel": per, "bytes_per_sample_step": sum(per.values()),
                            "note": "4096 symbols x 64 samples (Run_my_PDM_QPSK.m:21-24): 256 x 1024 split, row pass = k_rowreg<10> (one wave per row and polarisation)"}
print(json.dumps(res, indent=1))
