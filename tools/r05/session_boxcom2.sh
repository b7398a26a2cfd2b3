#!/bin/bash
# round 5: wb_k_box_com later on the side stream (option boxcom_after), the bounded halo spin's cost, the GEOS-hardening GPU test
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/boxcom2.txt
: > $OUT
timeout 900 python -m pytest tests/test_geos_hardening.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 | tee -a $OUT
run() {
  label="$1"; shift
  WB_BENCH_OPTS="$1" timeout 300 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps ${3:-0} --streams ${2:-2} > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_fused", "overturn", "box_com", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props", "polygons", "geometry"]
wg = d.get("without_geometry") or {}
print("%-30s value %6d  ms/step %.3f (without geometry %.3f) frac %.3f alone %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], wg.get("ms_per_step", float("nan")), d["roofline"]["frac"], d["roofline"]["frac_alone"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
for rep in 1 2 3; do
run "one batch, boxcom_after 0" "boxcom_after=0" 1
run "one batch, boxcom_after 1" "boxcom_after=1" 1
run "one batch, boxcom_after 2" "boxcom_after=2" 1
run "two, boxcom_after 0" "boxcom_after=0" 2 16
run "two, boxcom_after 1" "boxcom_after=1" 2 16
run "two, boxcom_after 2" "boxcom_after=2" 2 16
done
