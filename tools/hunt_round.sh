#!/bin/bash
# development: the random parity hunts on the GPU box (HIP library against the oracle), many worker processes
cd "$(dirname "$0")/.."
export WB_HUNT_GPU=1 WB_HUNT_WORKERS=${WB_HUNT_WORKERS:-48}
S=${1:-700000}
WB_HUNT_NT=6 WB_HUNT_PARAMS=1 timeout 900 python tools/parity_hunt.py $S 500 1.0
WB_HUNT_NT=6 WB_HUNT_PARAMS=1 WB_HUNT_F64=1 timeout 900 python tools/parity_hunt.py $((S+1000)) 300 2.5
WB_HUNT_NT=2 timeout 900 python tools/parity_hunt.py $((S+2000)) 120 0.25
timeout 600 python tools/levelhit_hunt.py $((S+3000)) 400
timeout 600 python tools/levelwalk_hunt.py $((S+4000)) 400
