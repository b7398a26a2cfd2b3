#!/bin/bash
# development: phase stamps of the kernel instrumented in wavebreaking_amd/csrc/libwb_hip_stamps*.so (built with -DWB_STAMPS=n)
cd "$(dirname "$0")/.."
for lib in ${@:-wavebreaking_amd/csrc/libwb_hip_stamps.so}; do
echo $lib
WB_HIP_LIBRARY=$PWD/$lib WB_DEBUG_STAMPS=1 WB_DEBUG_TIMING=1 python bench.py --steps 2 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 2>&1 | grep -E "stamps" | tail -2
done
