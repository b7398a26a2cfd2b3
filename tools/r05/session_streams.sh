#!/bin/bash
# round 5 (after the record sweep of wb_k_streamer_f234): two, three and four batches in flight
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/streams.txt
: > $OUT
run() {
  label="$1"
  WB_BENCH_OPTS="$3" timeout 300 python bench.py --steps 60 --warmup 8 --cpu-sample 0 --e2e-steps 0 --stagger-steps 16 --streams $2 > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_fused", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props"]
print("%-24s value %6d  ms/step %.3f frac %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
for rep in 1 2 3; do
run "two" 2
run "three" 3
run "four" 4
done
