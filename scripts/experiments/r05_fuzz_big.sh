#!/bin/bash
mkdir -p gpurun_out/r05_fuzz_big
sed -i 's/timeout -k 10 420/timeout -k 10 1000/; s/timeout -k 10 600/timeout -k 10 1100/' scripts/fuzz_all.sh
scripts/fuzz_all.sh 20261206 600 > gpurun_out/r05_fuzz_big/fuzz.txt 2>&1; tail -14 gpurun_out/r05_fuzz_big/fuzz.txt
