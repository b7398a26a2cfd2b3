#!/bin/bash
# the scan's patch stores as whole 128-byte lines: parity, the placement probe (is the slow mode gone?), the bench
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config1.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do WB_PROBE_SHORT=1 WB_PROBE_BUFFERS=6 python tools/stream_pattern_probe.py; echo; done
for i in 1 2 3; do
WB_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 > /tmp/ab.json 2>/tmp/ab.err || tail -5 /tmp/ab.err
python - <<'PY'
import json
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
print("two in flight: value %d  ms/step %.3f frac %.3f alone %.3f scan %.2f build %.2f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"], s["scan"], s["scan_build"]), flush=True)
PY
done
