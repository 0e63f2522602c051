#!/bin/bash
# round 3: the Monte-Carlo leg -- where a round's time goes, and CMA waves packed on fewer CUs
mkdir -p gpurun_out/r03ev
timeout -k 10 120 python3 scripts/experiments/mc_round_timing.py 128 2>&1 | grep -v amdgpu.ids
for P in 256 64 32; do
  for D in 3 5; do
  PLX_CMA_PACK_MIN=$P timeout -k 10 200 python3 bench.py --frames 64 --steps 1 --warmup 0 --variants 1 --mc-rounds 16 --mc-depth $D --no-cpu-baseline --no-single-frame --no-gateway 2> gpurun_out/r03ev/err_mc.txt | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); m=d['mc']; print('pack_min $P depth $D: %.0f realisations/s (%d in %.3f s)' % (m['realisations_per_s'], m['realisations'], m['seconds']))" || tail -3 gpurun_out/r03ev/err_mc.txt
  done
done
