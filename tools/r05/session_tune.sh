#!/bin/bash
# round 5: the pipeline's knobs with the new kernels (batches in flight, which phase is exclusive, halves), 32-bit loop indices in the fused kernel
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/tune.txt
: > $OUT
echo "== stamps (ms of workgroup time per launch), fused kernel with 32-bit loop indices" | tee -a $OUT
WB_HIP_LIBRARY=$PWD/build/ab/libwb_stamps8.so WB_DEBUG_STAMPS=1 timeout 300 python bench.py --steps 2 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 2>&1 | grep -E "stamps" | tail -1 | tee -a $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or full_parity or quarter or options" 2>&1 | tail -3 | tee -a $OUT
run() {
  label="$1"; shift
  env $4 WB_BENCH_OPTS="$1" timeout 300 python bench.py --steps 60 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps ${3:-0} --streams ${2:-2} > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_fused", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props"]
wg = d.get("with_geometry") or {}
print("%-32s value %6d  ms/step %.3f (with geometry %.3f) frac %.3f alone %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], wg.get("ms_per_step", float("nan")), d["roofline"]["frac"], d["roofline"]["frac_alone"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
H="0:cu_mask=128,1:cu_mask=128128"
for rep in 1 2 3; do
run "two (default: phase 1 exclusive)" "" 2 16
run "two, free" "" 2 0 WB_PIPE_STAGGER=0
run "three in flight" "" 3 0
run "three, free" "" 3 0 WB_PIPE_STAGGER=0
run "halves" "$H" 2 0
run "one" "" 1 0
done
