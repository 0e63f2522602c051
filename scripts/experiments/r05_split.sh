#!/bin/bash
# round 5: the split library against the CRC set of the round-4 build, then the GPU suite + smoke + default bench line
mkdir -p gpurun_out/r05_split
timeout -k 10 500 python scripts/crc_set.py > gpurun_out/r05_split/crc.txt 2> gpurun_out/r05_split/crc.err || { tail -5 gpurun_out/r05_split/crc.err; exit 1; }
if diff profiles/r05_crc_before_split.txt gpurun_out/r05_split/crc.txt > gpurun_out/r05_split/crc.diff; then echo "crc set identical to the round-4 build: $(wc -l < gpurun_out/r05_split/crc.txt) lines"; else echo "CRC DIFFERENCES"; head -20 gpurun_out/r05_split/crc.diff; fi
scripts/gpu_round.sh r05_split
