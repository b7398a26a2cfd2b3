#!/bin/bash
# round 5: wb_k_box_com on a CU-masked stream of its own (option boxcom_cu_mask = first * 1000 + count), so that its single waves do
# not take wave slots from the streamer chain's 1024-thread workgroups on every CU
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/boxmask.txt
: > $OUT
export WB_HIP_LIBRARY=$PWD/build/ab/libwb_boxmask.so
run() {
  label="$1"
  WB_BENCH_OPTS="$2" timeout 300 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps ${4:-0} --streams ${3:-2} > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_fused", "box_com", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props", "results"]
print("%-24s value %6d  ms/step %.3f frac %.3f alone %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
for rep in 1 2 3; do
for v in 0 128128 160096 192064; do
run "mask $v one batch" "boxcom_cu_mask=$v" 1
run "mask $v two" "boxcom_cu_mask=$v" 2 16
done
done
