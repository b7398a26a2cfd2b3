#!/bin/bash
# round 4: GPU tests + bench at one and two batches in flight (label = $1)
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r04d}; mkdir -p $O
one() {  # label, streams, extra env ...
  label=$1; st=$2; shift 2
  env "$@" timeout 300 python bench.py --steps 30 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams $st > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
  python - "$label" $st <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
s = d["stage_ms"]
print(sys.argv[1], "streams", sys.argv[2], "value %d  ms/step %.3f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), " ".join("%s %.2f" % (k, v) for k, v in s.items()), "wall", {k: round(v, 2) for k, v in d["host_wall_ms"].items()}, flush=True)
PY
}
{
if [ "${2:-tests}" = tests ]; then timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4; fi
for rep in 1 2; do for st in 1 2; do one default $st WB_X=1; done; done
} 2>&1 | tee $O/ab.txt
