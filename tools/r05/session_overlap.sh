#!/bin/bash
# round 5: how the kernels of two batches in flight share the device (kernel trace of the driver's command -> tools/r05/overlap.py)
cd "$(dirname "$0")/../.."
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r05/ovraw -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 --e2e-steps 0 > $R/gpurun_out/r05/ov_log.txt 2>&1
cd $R
python tools/r05/overlap.py gpurun_out/r05/ovraw > gpurun_out/r05/overlap.txt 2>&1
rm -rf gpurun_out/r05/ovraw
cat gpurun_out/r05/overlap.txt
