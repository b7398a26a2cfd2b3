#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the scan kernels only (two separate --pmc passes), one step of the bench batch
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/${1:-scanpmc}; export TMPDIR=/tmp; mkdir -p $O; cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 > $R/$O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 > $R/$O/pmc_write.log 2>&1
cd $R
python tools/profile_summary.py pmc $O/pmc_fetch $O/pmc_write > $O/hbm_traffic_pmc.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write
grep -E "kernel|scan" $O/hbm_traffic_pmc.txt
