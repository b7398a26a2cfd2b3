#!/bin/bash
# development: one stage's time and the counters for library builds: tools/ab_stage.sh <stage> <lib.so|default> ...
cd "$(dirname "$0")/.."
stage=$1; shift
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset WB_HIP_LIBRARY; else export WB_HIP_LIBRARY=$PWD/$lib; fi
  timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
  python - "$lib" $stage <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], sys.argv[2], "%.3f ms" % d["stage_ms"].get(sys.argv[2], -1), "ms/step %.3f" % d["ms_per_step"], "com_rechecks", d["counts"]["com_rechecks"], flush=True)
PY
done
