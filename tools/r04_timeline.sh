#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out/tl2
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl2/raw -- python3 $R/bench.py --steps 12 --warmup 4 --cpu-sample 0 --e2e-steps 0 --streams 2 --stagger-steps 0 > $R/gpurun_out/tl2/log.txt 2>&1
cd $R
python tools/timeline2.py gpurun_out/tl2/raw > gpurun_out/tl2/timeline_s2.txt
rm -rf gpurun_out/tl2/raw
head -40 gpurun_out/tl2/timeline_s2.txt
