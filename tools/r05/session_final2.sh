#!/bin/bash
# round 5, final code: the round script (tests, bench lines, rocprofv3 stats + PMC), a kernel-stats run WITH the geometry kernels in the
# timed region, phase stamps of the fused kernel
cd "$(dirname "$0")/../.."
R=$PWD
bash tools/gpu_round.sh r05f tests pmc
export TMPDIR=/tmp
cd /tmp
WB_BENCH_GEOMETRY=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05f/prof_geo -- python3 $R/bench.py --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 > $R/gpurun_out/r05f/prof_geo.log 2>&1
cd $R
python tools/profile_summary.py stats gpurun_out/r05f/prof_geo > gpurun_out/r05f/kernel_stats_one_batch_with_geometry.csv 2>&1
rm -rf gpurun_out/r05f/prof_geo
