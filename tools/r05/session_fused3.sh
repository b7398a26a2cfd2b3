#!/bin/bash
# round 5: optimised wb_k_trace_fused (running list index, four-entry probes, walk cap, buckets of one meridian): stamps,
# workgroup sizes, geometry with zero-copy views
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/fused4.txt
: > $OUT
echo "== stamps (ms of workgroup time per launch)" | tee -a $OUT
WB_HIP_LIBRARY=$PWD/build/ab/libwb_stamps8.so WB_DEBUG_STAMPS=1 timeout 300 python bench.py --steps 2 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 2>&1 | grep -E "stamps" | tail -1 | tee -a $OUT
echo "== gpu tests" | tee -a $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or geometry or full_parity or quarter or python_surface or rough" 2>&1 | tail -5 | tee -a $OUT
run() {
  label="$1"; shift
  lib="$4"
  WB_HIP_LIBRARY=${lib:+$PWD/build/ab/$lib} WB_BENCH_OPTS="$1" WB_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps ${3:-0} --streams ${2:-2} > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_fused", "summary", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props", "polygons", "geometry", "results", "geometry_host"]
wg = d.get("without_geometry") or {}
print("%-30s value %6d  ms/step %.3f (without geometry %.3f) frac %.3f alone %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], wg.get("ms_per_step", float("nan")), d["roofline"]["frac"], d["roofline"]["frac_alone"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
H="0:cu_mask=128,1:cu_mask=128128"
for rep in 1 2; do
run "one batch 256 thr" "" 1 12
run "one batch 512 thr" "" 1 0 libwb_f512.so

run "two" "" 2 20
run "halves" "$H" 2 20
run "halves 512 thr" "$H" 2 0 libwb_f512.so

done
