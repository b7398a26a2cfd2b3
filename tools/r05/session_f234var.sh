#!/bin/bash
# round 5: wb_k_streamer_f234 (rounds over the records) with 256 threads, 4 or 16 candidates of a failing row per round
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/f234var.txt
: > $OUT
run() {
  label="$1"; lib="$2"
  WB_BENCH_OPTS="$5" WB_HIP_LIBRARY="$lib" timeout 300 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps ${4:-0} --streams ${3:-2} > /tmp/ab.json 2>/tmp/ab.err || { echo "$label FAILED" | tee -a $OUT; tail -5 /tmp/ab.err | tee -a $OUT; return; }
  python - "$label" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
keys = ["scan", "scan_build", "trace_fused", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props"]
print("%-24s value %6d  ms/step %.3f frac %.3f alone %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"]) + " ".join("%s %.2f" % (k.replace("streamer_", "").replace("trace_", ""), s.get(k, float("nan"))) for k in keys), flush=True)
PY
}
NEW=$PWD/wavebreaking_amd/csrc/libwb_hip.so
for rep in 1 2 3; do
for v in new T256 K4 K16; do
lib=$NEW; [ $v != new ] && lib=$PWD/build/ab/libwb_f234_$v.so
run "$v one batch" $lib 1
run "$v two" $lib 2 16
done
done
