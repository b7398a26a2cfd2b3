#!/bin/bash
# link / finish with batched requests (WB_PFOR_B): parity, bench A/B against the head of the branch, stamps
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
one() {  # label, streams, extra env ...
  label=$1; st=$2; shift 2
  env "$@" timeout 300 python -X faulthandler bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams $st > /tmp/ab.json 2> /tmp/ab.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "rc $rc"; tail -40 /tmp/ab.err; return; fi
  python - "$label" $st <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
s = d["stage_ms"]
keys = ("scan", "scan_build", "trace_link", "trace_rank", "trace_finish", "box_com", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props", "polygons")
print(sys.argv[1], "streams", sys.argv[2], "value %d  ms/step %.3f frac %.3f alone %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"]), " ".join("%s %.2f" % (k, s.get(k, -1)) for k in keys), flush=True)
PY
}
for rep in 1 2; do
one new 1 WB_X=1
one new 2 WB_X=1
one old 1 WB_HIP_LIBRARY=$PWD/build/ab/libwb_prev.so
one old 2 WB_HIP_LIBRARY=$PWD/build/ab/libwb_prev.so
done
