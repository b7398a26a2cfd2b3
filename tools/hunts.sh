#!/bin/bash
# The random parity hunts (contours, level hits, streamer filters), the tracking / flags / spatial hunts and the wider GPU-vs-oracle
# sweep, on the GPU box:  bash tools/hunts.sh [seed base] [output tag]   ->  gpurun_out/<tag>/hunts.txt
cd "$(dirname "$0")/.."
S=${1:-990000}
O=gpurun_out/${2:-hunts}
mkdir -p $O
{
bash tools/hunt_round.sh $S
WB_HUNT_GPU=1 WB_HUNT_WORKERS=32 timeout 900 python tools/track_hunt.py $((S + 5000)) 400
WB_HUNT_GPU=1 WB_HUNT_WORKERS=32 timeout 600 python tools/flags_hunt.py $((S + 6000)) 600
WB_HUNT_GPU=1 WB_HUNT_WORKERS=32 timeout 600 python tools/spatial_hunt.py $((S + 7000)) 600
timeout 900 python tests/gpu_parity_sweep.py
} 2>&1 | grep -v "amdgpu.ids\|cross themselves" | tee $O/hunts.txt
