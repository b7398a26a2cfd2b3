#!/bin/bash
# round 4, session b: box_com split A/B, timers option, three batches in flight
cd "$(dirname "$0")/.."
O=gpurun_out/r04b; mkdir -p $O
one() {  # label, streams, extra env ...
  label=$1; st=$2; shift 2
  env "$@" timeout 300 python bench.py --steps 30 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams $st > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
  python - "$label" $st <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
s = d["stage_ms"]
keys = ("scan", "trace_link", "trace_rank", "trace_finish", "box_com", "streamer_pairs", "streamer_f1", "streamer_f234", "props", "polygons_d2h")
print(sys.argv[1], "streams", sys.argv[2], "value %d  ms/step %.3f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), " ".join("%s %.2f" % (k, s.get(k, -1)) for k in keys), "rechecks", d["counts"]["com_rechecks"], flush=True)
PY
}
{
for lib in default split4 split8 split16; do
  for st in 1 2; do
    if [ $lib = default ]; then one $lib $st WB_X=1; else one $lib $st WB_HIP_LIBRARY=$PWD/build/ab/libwb_$lib.so; fi
  done
done
one timers1 2 WB_BENCH_OPTS=timers=1
one timers1 1 WB_BENCH_OPTS=timers=1
one default 3 WB_X=1
one default 2 WB_X=1
} 2>&1 | tee $O/ab.txt
