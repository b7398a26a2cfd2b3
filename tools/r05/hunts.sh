#!/bin/bash
# round 5, final code: the random parity hunts (fused trace kernel, keys-only scan, device geometry in every full_parity case), the
# wider sweep and the tracking / flags / spatial hunts on the GPU
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05h
{
bash tools/hunt_round.sh 830000
WB_HUNT_GPU=1 WB_HUNT_WORKERS=32 timeout 900 python tools/track_hunt.py 835000 400
WB_HUNT_GPU=1 WB_HUNT_WORKERS=32 timeout 600 python tools/flags_hunt.py 836000 600
WB_HUNT_GPU=1 WB_HUNT_WORKERS=32 timeout 600 python tools/spatial_hunt.py 837000 600
timeout 900 python tests/gpu_parity_sweep.py
} 2>&1 | grep -v "amdgpu.ids\|cross themselves" | tee gpurun_out/r05h/hunts.txt
