#!/bin/bash
# development: kernel timeline of one pass, one batch at a time
cd "$(dirname "$0")/.."
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out/tl
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl/raw -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 --streams ${1:-1} --stagger-steps 0 > $R/gpurun_out/tl/log.txt 2>&1
cd $R
python tools/timeline.py gpurun_out/tl/raw 1 > gpurun_out/tl/timeline_s${1:-1}.txt
rm -rf gpurun_out/tl/raw
cat gpurun_out/tl/timeline_s${1:-1}.txt
