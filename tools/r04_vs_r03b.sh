#!/bin/bash
cd "$(dirname "$0")/.."
run() {
  (cd $2 && timeout 300 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams $3 > /tmp/ab.json 2> /tmp/ab.err || tail -5 /tmp/ab.err)
  python - "$1" $3 <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
s = d["stage_ms"]
ks = sum(v for k, v in s.items() if k not in ("geometry_host", "box_com", "overturn", "polygons"))
print(sys.argv[1], "streams", sys.argv[2], "ms/step %.3f scan %.2f kernels(main stream) %.2f" % (d["ms_per_step"], s["scan"], ks), "wall", {k: round(v, 2) for k, v in d["host_wall_ms"].items()}, flush=True)
PY
}
for rep in 1 2 3; do run r03 build/r03 1; run r04 . 1; done
