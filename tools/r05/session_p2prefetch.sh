#!/bin/bash
# round 5: wb_k_streamer_pairs2 with the next item's descriptor fetched one item ahead; stamps (-DWB_STAMPS=2)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/p2prefetch.txt
: > $OUT
echo "== pairs2 stamps (ms of workgroup time per launch: 6 load, 0 barrier, 1 tile set-up, 2 block filter, 3 sweep, 4 append, 5 tail)" | tee -a $OUT
WB_HIP_LIBRARY=$PWD/build/ab/libwb_stamps2.so WB_DEBUG_STAMPS=1 timeout 300 python bench.py --steps 2 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 2>&1 | grep -E "stamps" | tail -1 | tee -a $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a $OUT
WB_HUNT_GPU=1 WB_HUNT_WORKERS=48 WB_HUNT_NT=6 WB_HUNT_PARAMS=1 timeout 900 python tools/parity_hunt.py 890000 300 1.0 2>&1 | grep -v amdgpu | tail -1 | tee -a $OUT
WB_HUNT_GPU=1 WB_HUNT_WORKERS=48 WB_HUNT_NT=2 timeout 900 python tools/parity_hunt.py 892000 60 0.25 2>&1 | grep -v amdgpu | tail -1 | tee -a $OUT
run() {
  label="$1"; lib="$2"
  WB_HIP_LIBRARY="$lib" timeout 300 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps ${4:-0} --streams ${3:-2} > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_fused", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props"]
print("%-24s value %6d  ms/step %.3f frac %.3f alone %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
NEW=$PWD/wavebreaking_amd/csrc/libwb_hip.so
OLD=$PWD/build/ab/libwb_prev.so
for rep in 1 2 3; do
run "old one batch" $OLD 1
run "new one batch" $NEW 1
run "old two" $OLD 2 16
run "new two" $NEW 2 16
done
