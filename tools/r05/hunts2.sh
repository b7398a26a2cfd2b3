#!/bin/bash
# round 5, after the rewrite of wb_k_streamer_f234 (rounds over the records): the random parity hunts and the wider sweep on the GPU
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05h
{
bash tools/hunt_round.sh 850000
WB_HUNT_GPU=1 WB_HUNT_WORKERS=48 WB_HUNT_NT=4 WB_HUNT_PARAMS=1 timeout 900 python tools/parity_hunt.py 856000 160 0.25
timeout 900 python tests/gpu_parity_sweep.py
} 2>&1 | grep -v "amdgpu.ids\|cross themselves" | tee gpurun_out/r05h/hunts2.txt
