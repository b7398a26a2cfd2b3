#!/bin/bash
# end_to_end with the fingerprint taken beside the call (default) and before it (WB_INPUT_SPECULATE=0); the Python API tests on the GPU
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_python_api.py tests/test_gpu_record.py -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
for sp in 1 0; do
WB_INPUT_SPECULATE=$sp timeout 600 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --stagger-steps 0 > /tmp/e.json 2> /tmp/e.err || tail -20 /tmp/e.err
python - $sp <<'PY'
import json, sys
d = json.load(open("/tmp/e.json"))
e = d["end_to_end"]
print("speculate", sys.argv[1], "e2e %d ts/s" % e["value"], "both runs", [round(x, 4) for x in e["both_runs_s"]], {k: round(v, 4) for k, v in e["seconds"].items()}, "keep_inputs %d" % e["keep_inputs"]["value"], "contours_reuse %d" % e["contours_reuse"]["value"], "no_cache %d" % e["no_cache"]["value"], e["fingerprint"].get("beside_the_call"), "value", int(d["value"]), flush=True)
PY
done
done
