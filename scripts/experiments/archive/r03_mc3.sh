#!/bin/bash
# round 3: Monte-Carlo leg against the ROUND SIZE (1024 realisations per GPU in all cases)
mkdir -p gpurun_out/r03ev
for cfgs in "128 8 7" "256 4 3" "256 4 1" "512 2 1" "512 2 0" "1024 1 0" "512 4 1" "1024 2 1"; do
  set -- $cfgs
  timeout -k 10 200 python3 bench.py --frames 64 --steps 1 --warmup 0 --variants 1 --mc-frames $1 --mc-rounds $2 --mc-depth $3 --no-cpu-baseline --no-single-frame --no-gateway --no-cohmix-line 2> gpurun_out/r03ev/err_mc.txt | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); m=d['mc']; print('round of $1 x $2 rounds, depth $3: %.0f realisations/s (%d in %.3f s) avgber %.3e' % (m['realisations_per_s'], m['realisations'], m['seconds'], m['avgber']))" || tail -3 gpurun_out/r03ev/err_mc.txt
done
