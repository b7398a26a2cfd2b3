#!/bin/bash
# round 5: fused trace with the keys kept in LDS through the ranking (no second read of the lists), splitter flags by words
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/fused5.txt
: > $OUT
echo "== stamps (ms of workgroup time per launch)" | tee -a $OUT
WB_HIP_LIBRARY=$PWD/build/ab/libwb_stamps8.so WB_DEBUG_STAMPS=1 timeout 300 python bench.py --steps 2 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 2>&1 | grep -E "stamps" | tail -1 | tee -a $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or full_parity or quarter or rough or python_surface or golden" 2>&1 | tail -3 | tee -a $OUT
run() {
  label="$1"; shift
  WB_DEBUG_TIMING=1 WB_BENCH_OPTS="$1" timeout 300 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps ${3:-0} --streams ${2:-2} > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  grep "fused trace" /tmp/ab.err | sort | uniq -c | sort -rn | head -2 | tee -a $OUT
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_fused", "overturn", "box_com", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props", "polygons", "geometry"]
wg = d.get("without_geometry") or {}
print("%-30s value %6d  ms/step %.3f (without geometry %.3f) frac %.3f alone %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], wg.get("ms_per_step", float("nan")), d["roofline"]["frac"], d["roofline"]["frac_alone"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
for rep in 1 2 3; do
run "one batch" "" 1
run "two" "" 2 16
done
