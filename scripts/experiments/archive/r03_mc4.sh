#!/bin/bash
# round 3, final build: Monte-Carlo leg by round size and rounds in flight (1024 and 2048 realisations)
O=gpurun_out/r03mc4; mkdir -p $O
C="--no-cpu-baseline --no-single-frame --no-gateway --no-cohmix-line --frames 64 --steps 1 --warmup 0"
for cfg in "512 1 2" "512 1 4" "512 2 4" "1024 1 2" "341 2 3" "256 3 4" "256 3 8" "128 7 8" "768 1 2"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py $C --mc-frames $1 --mc-depth $2 --mc-rounds $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['mc']
print('frames/round $1 depth $2 rounds $3: %.0f realisations/s (%d in %.3f s) avgber %.6g' % (m['realisations_per_s'], m['realisations'], m['seconds'], m['avgber']))"
done
