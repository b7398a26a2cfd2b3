#!/bin/bash
# round 5, final kernels: which phases of the ABI are entered by one batch at a time (WB_PIPE_STAGGER 0 none, 2 phase 1 (default), 3 phase 2, 1 both)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/stagger.txt
: > $OUT
run() {
  label="$1"
  WB_PIPE_STAGGER=$2 timeout 300 python bench.py --steps 60 --warmup 8 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams 2 > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_fused", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props"]
print("%-24s value %6d  ms/step %.3f frac %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
for rep in 1 2 3 4; do
run "free" 0
run "phase 1 exclusive" 2
run "phase 2 exclusive" 3
done
