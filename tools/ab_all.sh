#!/bin/bash
# development: every stage of the pass for library builds, two batches in flight: tools/ab_all.sh <lib.so|default> ...
cd "$(dirname "$0")/.."
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset WB_HIP_LIBRARY; else export WB_HIP_LIBRARY=$PWD/$lib; fi
  timeout 300 python bench.py --steps ${AB_STEPS:-30} --warmup 6 --cpu-sample 0 --e2e-steps 0 --streams ${AB_STREAMS:-2} --stagger-steps 0 > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], " ".join("%s %.2f" % (k.replace("streamer_", "s_").replace("trace_", "t_"), v) for k, v in d["stage_ms"].items()), flush=True)
PY
done
