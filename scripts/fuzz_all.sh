#!/bin/bash
# dev tool: the four randomised parity sweeps against the oracle, one after the other (seed, cases per tool)
S=${1:-101}; N=${2:-40}
mkdir -p gpurun_out
for t in fuzz_fibre fuzz_rx fuzz_front fuzz_batch; do
  echo "== $t seed $S cases $N"
  timeout -k 10 420 python scripts/$t.py $S $N > gpurun_out/$t.log 2>&1 || { tail -5 gpurun_out/$t.log; echo "$t FAILED"; exit 1; }
  tail -2 gpurun_out/$t.log
done
