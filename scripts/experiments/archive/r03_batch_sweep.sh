#!/bin/bash
# dev: fibre time per frame and per-launch kernel times against the batch size (Infinity Cache residency of the batch), current build
for F in 32 48 64 96 128 192 256 1024; do
  timeout -k 10 200 python bench.py --frames $F --steps 4 --warmup 1 --variants 1 --mc-rounds 0 --no-cpu-baseline --no-overlap --no-single-frame 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['config']['fibre_ms_per_step']; k=d['roofline']['kernels']; print('F=$F fibre ms %.2f  per frame %.4f  MB %d  col %.1f us (%.3f us/frame)  row %.1f us (%.3f us/frame)'%(f, f/$F, $F*65536*32//2**20, k['k_colx16']['avg_launch_us'], k['k_colx16']['avg_launch_us']/$F, k['k_row']['avg_launch_us'], k['k_row']['avg_launch_us']/$F))"
done
