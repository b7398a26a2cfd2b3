#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04e; mkdir -p $O
one() {  # label, streams, extra env ...
  label=$1; st=$2; shift 2
  env "$@" timeout 300 python bench.py --steps 30 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams $st > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
  python - "$label" $st <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
s = d["stage_ms"]
keys = ("scan", "streamer_prep", "streamer_pairs", "streamer_f234", "fill_events", "polygons", "props", "results")
print(sys.argv[1], "streams", sys.argv[2], "value %d  ms/step %.3f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), " ".join("%s %.2f" % (k, s.get(k, -1)) for k in keys), flush=True)
PY
}
{
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "group or full_parity" 2>&1 | tail -3
for st in 1 2; do
one default $st WB_X=1
one polys256 $st WB_BENCH_OPTS=polys_grid=256
one polys512 $st WB_BENCH_OPTS=polys_grid=512
one polys1024 $st WB_BENCH_OPTS=polys_grid=1024
one polys128 $st WB_BENCH_OPTS=polys_grid=128
one default $st WB_X=1
done
} 2>&1 | tee $O/ab.txt
