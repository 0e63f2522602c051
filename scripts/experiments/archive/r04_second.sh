#!/bin/bash
O=gpurun_out/r04_second
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_mex_shims.py tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q -s -k "shim or fall or share_device or wdm_16ch or gateway_state or matrix_ssfm_gateway or adaptive or poldemux" > $O/tests.log 2>&1; rc=$?
grep -E "c2 chain|passed|failed|Error|error|assert" $O/tests.log | tail -30
[ $rc -eq 0 ] || { tail -40 $O/tests.log; exit $rc; }
(bash scripts/experiments/ab_envs.sh "16 g-s- 16384" - PLX_SSFM_NO_FUSE=1 "PLX_SSFM_NO_FUSE=1 PLX_SSFM_SHORT_ROWS=1") > $O/three_sweep_rows.txt 2>&1; grep -v amdgpu.ids $O/three_sweep_rows.txt
