#!/bin/bash
# round 5, first call: CRC set of the round-4 build (twice: determinism), then the GPU suite + smoke + default bench line
mkdir -p gpurun_out/r05_base
timeout -k 10 500 python scripts/crc_set.py > gpurun_out/r05_base/crc_a.txt 2> gpurun_out/r05_base/crc_a.err || { tail -5 gpurun_out/r05_base/crc_a.err; exit 1; }
timeout -k 10 500 python scripts/crc_set.py > gpurun_out/r05_base/crc_b.txt 2>/dev/null || exit 1
diff gpurun_out/r05_base/crc_a.txt gpurun_out/r05_base/crc_b.txt && echo "crc set deterministic: $(wc -l < gpurun_out/r05_base/crc_a.txt) lines"
scripts/gpu_round.sh r05_base
