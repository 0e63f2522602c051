#!/bin/bash
mkdir -p gpurun_out/r05_fp64
timeout -k 10 300 python -m pytest tests/test_gpu_configs.py -q -x -m gpu -k "another_kernel_holds" > gpurun_out/r05_fp64/pytest.log 2>&1; tail -3 gpurun_out/r05_fp64/pytest.log
hipcc --offload-arch=gfx950 -O3 scripts/experiments/micro/fp64_peak.hip -o /tmp/fp64_peak && timeout -k 5 120 /tmp/fp64_peak > gpurun_out/r05_fp64/fp64_peak.txt; cat gpurun_out/r05_fp64/fp64_peak.txt
scripts/fp64_pmc.sh > gpurun_out/r05_fp64/pmc.log 2>&1; tail -12 gpurun_out/r05_fp64/pmc.log
