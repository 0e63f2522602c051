#!/bin/bash
mkdir -p gpurun_out/r03ev
run() { local label=$1; shift
  timeout -k 10 200 python3 bench.py "$@" --steps 8 --warmup 2 --mc-rounds 0 --no-cpu-baseline --no-single-frame --no-gateway --no-cohmix-line 2> gpurun_out/r03ev/err_$label.txt | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$label value %.4f  %.2f ms/step  fibre %.2f  rx %.2f  errors %s' % (d['value'], d['ms_per_step'], c['fibre_ms_per_step'], c['rxdsp_ms_per_step'], c['bit_errors_xy']))" || tail -5 gpurun_out/r03ev/err_$label.txt
}
for rep in 1 2; do
  run inline --no-rx-thread
  run thread
done
run thread_cohmix --frontend cohmix
run inline_cohmix --frontend cohmix --no-rx-thread
