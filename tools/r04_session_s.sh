#!/bin/bash
# the exclusive-phase modes of the two-batch pipeline inside run_record (bench.py's timed region): value and the scan's share of the HBM peak
cd "$(dirname "$0")/.."
one() {  # label, streams, extra env ...
  label=$1; st=$2; shift 2
  env "$@" timeout 300 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams $st > /tmp/ab.json 2> /tmp/ab.err || tail -5 /tmp/ab.err
  python - "$label" $st <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
s = d["stage_ms"]
keys = ("scan", "scan_build", "trace_link", "trace_rank", "trace_finish", "streamer_pairs", "streamer_f1", "streamer_f234", "props")
print(sys.argv[1], "streams", sys.argv[2], "value %d  ms/step %.3f frac %.3f alone %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"]), " ".join("%s %.2f" % (k, s.get(k, -1)) for k in keys), flush=True)
PY
}
for rep in 1 2 3; do
one free 2 WB_PIPE_STAGGER=0
one both 2 WB_PIPE_STAGGER=1
one phase1 2 WB_PIPE_STAGGER=2
one phase2 2 WB_PIPE_STAGGER=3
done
