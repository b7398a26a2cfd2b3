#!/bin/bash
cd "$(dirname "$0")/.."
bash tools/r04_session_d.sh r04j
for n in 6 4; do
echo "stamps $n"
WB_HIP_LIBRARY=$PWD/build/ab/libwb_stamps$n.so WB_DEBUG_STAMPS=1 python bench.py --steps 2 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 2>&1 | grep -E "stamps" | tail -3
done 2>&1 | tee gpurun_out/r04j/stamps.txt
