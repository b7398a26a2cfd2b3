#!/bin/bash
# round 5: wb_k_box_com's share of a CU's LDS beside wb_k_streamer_prep / pairs2 (review item 2, second half), with and without CU halves
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/boxcom.txt
: > $OUT
run() {
  label="$1"; shift
  WB_BENCH_OPTS="$1" WB_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 60 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams ${2:-2} > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_link", "trace_rank", "trace_finish", "overturn", "box_com", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props", "polygons"]
print("%-34s value %6d  ms/step %.3f frac %.3f alone %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
H="0:cu_mask=128,1:cu_mask=128128"
for rep in 1 2; do
run "plain (boxcom 12)" ""
run "boxcom 8" "boxcom_per_cu=8"
run "boxcom 6" "boxcom_per_cu=6"
run "boxcom 4" "boxcom_per_cu=4"
run "boxcom 2" "boxcom_per_cu=2"
run "halves" "$H"
run "halves boxcom 6" "$H,boxcom_per_cu=6"
run "halves boxcom 4" "$H,boxcom_per_cu=4"
run "halves boxcom 2" "$H,boxcom_per_cu=2"
run "one batch boxcom 12" "" 1
run "one batch boxcom 4" "boxcom_per_cu=4" 1
run "one batch boxcom 2" "boxcom_per_cu=2" 1
done
