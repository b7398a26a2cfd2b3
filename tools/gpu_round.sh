#!/bin/bash
# One GPU session: parity tests, the bench line (default and one batch at a time), rocprofv3 kernel stats and the two PMC
# passes; scratch under gpurun_out/$1 (summaries are copied to profiles/ by hand).
set -u
cd "$(dirname "$0")/.."
R=$PWD
O=gpurun_out/${1:-round}
export TMPDIR=/tmp
mkdir -p $O
if [ "${2:-tests}" = "tests" ]; then (timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > $O/gpu_tests.txt; fi
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --streams 1 --cpu-sample 0 > $O/bench_s1.json 2> $O/bench_s1.err
timeout 600 python bench.py --config 1 > $O/bench_config1.json 2> $O/bench_config1.err
WB_BENCH_FORCE_DIST=1 timeout 600 python bench.py --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 > $O/bench_dist1.json 2> $O/bench_dist1.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_stats -- python3 $R/bench.py --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 > $R/$O/prof_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_stats2 -- python3 $R/bench.py --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 > $R/$O/prof_stats2.log 2>&1
if [ "${3:-pmc}" = "pmc" ]; then
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 > $R/$O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 > $R/$O/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $R/$O/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 > $R/$O/pmc_sq.log 2>&1
fi
cd $R
python tools/profile_summary.py stats $O/prof_stats > $O/kernel_stats_one_batch.csv 2>&1
python tools/profile_summary.py stats $O/prof_stats2 > $O/kernel_stats.csv 2>&1
if [ "${3:-pmc}" = "pmc" ]; then
python tools/profile_summary.py pmc $O/pmc_fetch $O/pmc_write > $O/hbm_traffic_pmc.txt 2>&1
python tools/profile_summary.py sq $O/pmc_sq > $O/sq_counters.txt 2>&1
fi
# keep only the summaries (the raw traces are large)
rm -rf $O/prof_stats $O/prof_stats2 $O/pmc_fetch $O/pmc_write $O/pmc_sq
cat $O/gpu_tests.txt 2>/dev/null; tail -c 600 $O/bench.json; echo; head -30 $O/kernel_stats_one_batch.csv
