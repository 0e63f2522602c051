#!/bin/bash
# round 5, files of record, part B: kernel traces, PMC traffic, the randomised parity sweeps
scripts/final_round.sh trace r05_final || exit 1
scripts/final_round.sh traffic r05_final || exit 1
scripts/fuzz_all.sh 20261105 150 > gpurun_out/r05_final/fuzz.txt 2>&1; tail -12 gpurun_out/r05_final/fuzz.txt
