#!/bin/bash
# round 4, final code: the random parity hunts and the tracking hunt on the GPU
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04h
{
bash tools/hunt_round.sh 730000
WB_HUNT_GPU=1 WB_HUNT_WORKERS=32 timeout 900 python tools/track_hunt.py 735000 400
WB_HUNT_GPU=1 WB_HUNT_WORKERS=32 timeout 600 python tools/flags_hunt.py 736000 600
WB_HUNT_GPU=1 WB_HUNT_WORKERS=32 timeout 600 python tools/spatial_hunt.py 737000 600
} 2>&1 | grep -v "amdgpu.ids\|cross themselves" | tee gpurun_out/r04h/hunts.txt
