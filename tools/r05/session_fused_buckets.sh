#!/bin/bash
# round 5: fused trace, buckets by a hash of (x mod nlon, y) and the distinct x of a contour from its extent, against the library before it
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/fused_buckets.txt
: > $OUT
echo "== stamps (ms of workgroup time per launch)" | tee -a $OUT
WB_HIP_LIBRARY=$PWD/build/ab/libwb_stamps8.so WB_DEBUG_STAMPS=1 timeout 300 python bench.py --steps 2 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 2>&1 | grep -E "stamps" | tail -1 | tee -a $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a $OUT
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
