#!/bin/bash
# round 5: wb_k_streamer_f1 after the list sweeps: stamps, and 4 x 512 against 2 x 1024 threads per CU
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/f1b.txt
: > $OUT
echo "== f1 stamps (ms of workgroup time per launch: 0 cursor, 1 clear, 2 first sweep, 3 second sweep, 4 global table, 5 parts; 12..: the longest workgroup's 4 5 6)" | tee -a $OUT
WB_HIP_LIBRARY=$PWD/build/ab/libwb_stamps9.so WB_DEBUG_STAMPS=1 timeout 300 python bench.py --steps 2 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 2>&1 | grep -E "stamps" | tail -1 | tee -a $OUT
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
for rep in 1 2 3; do
run "1024 one batch" $NEW 1
run "512 one batch" $PWD/build/ab/libwb_f1_512.so 1
run "1024 two" $NEW 2 16
run "512 two" $PWD/build/ab/libwb_f1_512.so 2 16
done
