#!/bin/bash
# two against three batches in flight, phase 1 exclusive (the default since the end of round 4)
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for st in 2 3 4; do
WB_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 60 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams $st > /tmp/ab.json 2>/tmp/ab.err || tail -5 /tmp/ab.err
python - $st <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json")); s = d["stage_ms"]
print("streams", sys.argv[1], "value %d  ms/step %.3f frac %.3f alone %.3f scan %.2f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"], s["scan"]), flush=True)
PY
done
done
