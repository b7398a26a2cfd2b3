#!/bin/bash
# end-of-round session: the round script (tests, bench lines, rocprofv3 stats + PMC), smoke(), the one-batch kernel timeline, the
# cross-check of the boxes' exact totals (a library built with -DWB_BOXCOM_BOUND=2, if present) and the random parity hunts
cd "$(dirname "$0")/.."
T=${1:-r06f}
bash tools/gpu_round.sh $T tests pmc
O=gpurun_out/$T
(timeout 600 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -5) | tee $O/smoke.txt
bash tools/timeline.sh 1 > $O/timeline_one_batch.txt 2>&1
if [ -f wavebreaking_amd/csrc/libwb_hip_bound2.so ]; then WB_HIP_LIBRARY=$PWD/wavebreaking_amd/csrc/libwb_hip_bound2.so timeout 900 python tools/boxbound_check.py 2>&1 | grep -v amdgpu.ids | tee $O/boxbound_check.txt; fi
REPS=4 timeout 600 python tools/e2e_fp_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/e2e_fp_probe.txt
timeout 300 python tools/f64_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/f64_probe.txt
bash tools/hunts.sh ${2:-996000} $T 2>&1 | tail -30
