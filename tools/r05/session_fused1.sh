#!/bin/bash
# round 5: first GPU run of wb_k_trace_fused -- parity tests, then A/B against the three-kernel chain
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/fused1.txt
: > $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -15 | tee -a $OUT
run() {
  label="$1"; shift
  WB_DEBUG_TIMING=1 WB_BENCH_OPTS="$1" WB_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams ${2:-2} > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  grep "fused trace" /tmp/ab.err | sort | uniq -c | sort -rn | head -3 | tee -a $OUT
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_fused", "trace_link", "trace_rank", "trace_finish", "summary", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props"]
print("%-30s value %6d  ms/step %.3f frac %.3f alone %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
H="0:cu_mask=128,1:cu_mask=128128"
for rep in 1 2; do
run "one batch, chain" "trace_fused=0" 1
run "one batch, fused" "trace_fused=1" 1
run "two, chain" "trace_fused=0"
run "two, fused" "trace_fused=1"
run "halves, chain" "$H,trace_fused=0"
run "halves, fused" "$H,trace_fused=1"
done
