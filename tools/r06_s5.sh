#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s5; mkdir -p $O
WB_HIP_LIBRARY=$PWD/wavebreaking_amd/csrc/libwb_hip_bound2.so timeout 900 python tools/boxbound_check.py 2>&1 | grep -v amdgpu.ids | tee $O/boxbound_check.txt
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config1.py -x -q 2>&1 | tail -5) | tee $O/parity_tests.txt
for lib in libwb_hip.so libwb_hip_f64r5.so libwb_hip_f64r7.so; do WB_HIP_LIBRARY=$PWD/wavebreaking_amd/csrc/$lib timeout 300 python tools/f64_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/f64_probe.txt; done
bash tools/timeline.sh 1 > $O/timeline_one_batch.txt 2>&1; tail -42 $O/timeline_one_batch.txt
bash tools/ab_opts.sh 3 "default:X=1" "bound0:WB_HIP_LIBRARY=$PWD/wavebreaking_amd/csrc/libwb_hip_bound0.so" 2>&1 | tee $O/ab.txt
AB_ARGS="--streams 1" bash tools/ab_opts.sh 1 "one_batch:X=1" "one_batch_bound0:WB_HIP_LIBRARY=$PWD/wavebreaking_amd/csrc/libwb_hip_bound0.so" 2>&1 | tee -a $O/ab.txt
