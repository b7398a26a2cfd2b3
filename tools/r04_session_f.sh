#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04f; mkdir -p $O
one() {  # label, streams, extra env ...
  label=$1; st=$2; shift 2
  env "$@" timeout 300 python bench.py --steps 30 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams $st > /tmp/ab.json 2> /tmp/ab.err || tail -5 /tmp/ab.err
  python - "$label" $st <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
s = d["stage_ms"]
print(sys.argv[1], "streams", sys.argv[2], "value %d  ms/step %.3f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), "wall", {k: round(v, 2) for k, v in d["host_wall_ms"].items()}, {k: d.get(k) for k in ("gather_path", "gather_verified")}, flush=True)
PY
}
{
timeout 900 python -m pytest tests/test_gpu_gather.py -x -q 2>&1 | tail -5
for rep in 1 2 3; do
one plain 2 WB_X=1
one dist1 2 WB_BENCH_FORCE_DIST=1
one dist1torch 2 WB_BENCH_FORCE_DIST=1 WB_BENCH_GATHER=torch
done
one plain 1 WB_X=1
one dist1 1 WB_BENCH_FORCE_DIST=1
} 2>&1 | tee $O/ab.txt
