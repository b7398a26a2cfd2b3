#!/bin/bash
# round 5, item 2 of the review: CU-masked streams -- two batches in flight on disjoint CU sets, against time slicing
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/masks.txt
: > $OUT
run() {
  label="$1"; shift
  WB_BENCH_OPTS="$1" WB_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 60 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams ${2:-2} > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_link", "trace_rank", "trace_finish", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props"]
print("%-34s value %6d  ms/step %.3f frac %.3f alone %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
for rep in 1 2; do
run "plain two in flight" ""
run "halves 0-127 / 128-255" "0:cu_mask=128,1:cu_mask=128128"
run "halves swapped" "0:cu_mask=128128,1:cu_mask=128"
run "both 0-191, scan 192-255" "cu_mask=192,scan_cu_mask=192064"
run "both 0-127, scan 128-255" "cu_mask=128,scan_cu_mask=128128"
run "both all, scan 128-255" "scan_cu_mask=128128"
run "both all, scan 0-255 masked stream" "scan_cu_mask=256"
run "3/4 + 3/4 overlapping" "0:cu_mask=192,1:cu_mask=64192"
run "one batch plain" "" 1
run "one batch, 128 CUs" "cu_mask=128" 1
run "one batch, 192 CUs" "cu_mask=192" 1
done
cat /tmp/ab.json | python -c "import json,sys; d=json.load(sys.stdin); print(sorted(d['stage_ms'].keys()))" | tee -a $OUT
