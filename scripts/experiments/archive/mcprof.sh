export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/mcprof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mcprof -- python3 scripts/experiments/mc_round_timing.py 128 > /dev/null 2>&1
g=$(find gpurun_out/mcprof -name "*kernel_stats.csv" | head -1)
head -25 $g | cut -c1-160
rm -rf gpurun_out/mcprof
