#!/bin/bash
# dev tool: the four randomised parity sweeps against the oracle, one after the other (seed, cases per tool)
S=${1:-101}; N=${2:-40}
mkdir -p gpurun_out
for t in fuzz_fibre fuzz_rx fuzz_front fuzz_batch; do
  echo "== $t seed $S cases $N"
  timeout -k 10 420 python scripts/$t.py $S $N > gpurun_out/$t.log 2>&1 || { tail -5 gpurun_out/$t.log; echo "$t FAILED"; exit 1; }
  tail -2 gpurun_out/$t.log
done
echo "== fuzz_fibre, frames of 2^16 ... 2^19 samples, seed $S cases $N"
timeout -k 10 600 python scripts/fuzz_fibre.py $S $N 16 19 > gpurun_out/fuzz_fibre_large.log 2>&1 || { tail -5 gpurun_out/fuzz_fibre_large.log; echo "fuzz_fibre (large) FAILED"; exit 1; }
tail -1 gpurun_out/fuzz_fibre_large.log
