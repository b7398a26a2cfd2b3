#!/bin/bash
# round 5, the code at the end of the round (after the second half): the random parity hunts, the wider sweep and the tracking / flags / spatial hunts on the GPU
# wider sweep and the tracking / flags / spatial hunts on the GPU
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05h
{
bash tools/hunt_round.sh 880000
WB_HUNT_GPU=1 WB_HUNT_WORKERS=32 timeout 900 python tools/track_hunt.py 885000 400
WB_HUNT_GPU=1 WB_HUNT_WORKERS=32 timeout 600 python tools/flags_hunt.py 886000 600
WB_HUNT_GPU=1 WB_HUNT_WORKERS=32 timeout 600 python tools/spatial_hunt.py 887000 600
timeout 900 python tests/gpu_parity_sweep.py
} 2>&1 | grep -v "amdgpu.ids\|cross themselves" | tee gpurun_out/r05h/hunts3.txt
