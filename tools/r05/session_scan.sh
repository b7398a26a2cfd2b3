#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/scan_keys.txt
: > $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a $OUT
for rep in 1 2 3; do timeout 600 python tools/r05/scan_keys_probe.py 2>&1 | tail -8 | tee -a $OUT; done
run() {
  label="$1"; shift
  WB_BENCH_OPTS="$1" timeout 300 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps ${3:-0} --streams ${2:-2} > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_fused", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props", "polygons", "geometry"]
wg = d.get("without_geometry") or {}
print("%-30s value %6d  ms/step %.3f (without geometry %.3f) frac %.3f alone %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], wg.get("ms_per_step", float("nan")), d["roofline"]["frac"], d["roofline"]["frac_alone"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
for rep in 1 2 3; do
run "two, keys only" "scan_keys_only=1" 2 16
run "two, key + values" "scan_keys_only=0" 2 16
run "one, keys only" "scan_keys_only=1" 1
run "one, key + values" "scan_keys_only=0" 1
done
