#!/bin/bash
# dev tool: fibre time per frame against the batch size (does a batch that fits the 256 MB Infinity Cache run faster?)
for nf in 0 1; do
for F in 16 32 64 96 128 256 512; do
  if [ $nf = 1 ]; then export PLX_SSFM_NO_FUSE=1; else unset PLX_SSFM_NO_FUSE; fi
  timeout -k 10 200 python bench.py --frames $F --steps 4 --warmup 1 --no-cpu-baseline --no-overlap --no-single-frame 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['config']['fibre_ms_per_step']; print('nofuse=$nf F=$F fibre ms %.2f  per frame %.4f  MB %d'%(f, f/$F, $F*65536*32//2**20))"
done; done
