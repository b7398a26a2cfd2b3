#!/bin/bash
# A/B of library builds on the GPU box: tools/ab_libs.sh <lib.so|default> ...   (bench at 1 and 2 batches in flight)
cd "$(dirname "$0")/.."
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset WB_HIP_LIBRARY; else export WB_HIP_LIBRARY=$PWD/$lib; fi
  for st in 1 2; do
    timeout 300 python bench.py --steps ${AB_STEPS:-30} --warmup 6 --cpu-sample 0 --e2e-steps 0 --streams $st > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
    python - "$lib" $st <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
s = d["stage_ms"]
keys = ("scan", "scan_build", "trace_link", "trace_rank", "trace_finish", "streamer_pairs", "streamer_f1", "streamer_f234", "props")
print(sys.argv[1], "streams", sys.argv[2], "value %d  ms/step %.3f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), " ".join("%s %.2f" % (k, s.get(k, -1)) for k in keys), flush=True)
PY
  done
done
