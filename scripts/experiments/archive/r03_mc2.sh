#!/bin/bash
# round 3: Monte-Carlo leg against the number of rounds in flight (and packed CMA waves)
mkdir -p gpurun_out/r03ev
for P in 256 64; do
  for D in 3 5 7 11; do
  PLX_CMA_PACK_MIN=$P timeout -k 10 200 python3 bench.py --frames 64 --steps 1 --warmup 0 --variants 1 --mc-rounds 10 --mc-depth $D --no-cpu-baseline --no-single-frame --no-gateway 2> gpurun_out/r03ev/err_mc.txt | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); m=d['mc']; print('pack_min $P depth $D: %.0f realisations/s (%d in %.3f s)' % (m['realisations_per_s'], m['realisations'], m['seconds']))" || tail -3 gpurun_out/r03ev/err_mc.txt
  done
done
