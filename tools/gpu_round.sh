#!/bin/bash
# One GPU session: parity tests, the bench line, rocprofv3 kernel stats and the two PMC passes (scratch under gpurun_out/).
set -u
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3) > gpurun_out/gpu_tests.txt
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
timeout 300 python benchmarks/bench_spatial.py > gpurun_out/bench_spatial.json 2> gpurun_out/bench_spatial.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_spatial -- python3 $R/benchmarks/bench_spatial.py --reps 3 > $R/gpurun_out/prof_spatial.log 2>&1
cd $R
python tools/profile_summary.py stats gpurun_out/prof_stats > gpurun_out/kernel_stats.csv 2>&1
python tools/profile_summary.py pmc gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/hbm_traffic_pmc.txt 2>&1
python tools/profile_summary.py stats gpurun_out/prof_spatial > gpurun_out/kernel_stats_spatial.csv 2>&1
# keep only the summaries (the raw traces are large)
rm -rf gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof_spatial
cat gpurun_out/gpu_tests.txt; tail -c 3000 gpurun_out/bench.json
