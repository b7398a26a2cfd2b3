#!/bin/bash
# build kernel with one 16 KB LDS pool (keys-only chunks of 256) against the 36 KB version
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/pool.txt
: > $OUT
WB_HIP_LIBRARY=$PWD/build/ab/libwb_next.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_parity or quarter or options or fused" 2>&1 | tail -3 | tee -a $OUT
run() {
  label="$1"; shift
  lib="$3"
  WB_HIP_LIBRARY=${lib:+$PWD/build/ab/$lib} timeout 300 python bench.py --steps 60 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams ${2:-2} > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_fused", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props"]
print("%-32s value %6d  ms/step %.3f frac %.3f alone %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
for rep in 1 2 3 4; do
run "two, in-tree" "" 2
run "two, pool" "" 2 libwb_next.so
run "one, in-tree" "" 1
run "one, pool" "" 1 libwb_next.so
done
