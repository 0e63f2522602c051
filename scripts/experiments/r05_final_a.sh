#!/bin/bash
# round 5, files of record, part A: GPU suite + smoke, the bench lines
scripts/gpu_round.sh r05_final > gpurun_out/r05_final_round.log 2>&1 || { tail -20 gpurun_out/r05_final_round.log; exit 1; }
tail -12 gpurun_out/r05_final_round.log | cut -c1-300
scripts/final_round.sh lines r05_final
