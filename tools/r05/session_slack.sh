#!/bin/bash
# places the persistent launches (all-pairs, routine 1) leave free for the other batch's small kernels (option persist_slack), with the round-5 kernels
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/slack.txt
: > $OUT
run() {
  label="$1"; shift
  WB_BENCH_OPTS="$1" timeout 300 python bench.py --steps 60 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams 2 > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_fused", "select", "streamer_lists", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "fill_events", "props", "results"]
print("%-24s value %6d  ms/step %.3f frac %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
for rep in 1 2 3; do
run "slack 0" "persist_slack=0"
run "slack 8" "persist_slack=8"
run "slack 32" "persist_slack=32"
run "slack 64" "persist_slack=64"
run "slack 128" "persist_slack=128"
done
