#!/bin/bash
# 100-step runs drift into lock step (both batches in the same kernel at the same time) more often than 20- or 40-step runs: does an exclusive phase 1 / phase 2 keep them apart?
cd "$(dirname "$0")/.."
one() {  # label, steps, extra env ...
  label=$1; steps=$2; shift 2
  env "$@" WB_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps $steps --warmup 5 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 > /tmp/ab.json 2> /tmp/ab.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "$label rc $rc"; tail -40 /tmp/ab.err; return; fi
  python - "$label" $steps <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
s = d["stage_ms"]
keys = ("scan", "trace_rank", "trace_finish", "streamer_pairs", "streamer_f1", "streamer_f234", "props")
print(sys.argv[1], "steps", sys.argv[2], "value %d  ms/step %.3f frac %.3f alone %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"]), " ".join("%s %.2f" % (k, s.get(k, -1)) for k in keys), flush=True)
PY
}
for rep in 1 2 3 4; do
one free 100 WB_PIPE_STAGGER=0
one phase1 100 WB_PIPE_STAGGER=2
one phase2 100 WB_PIPE_STAGGER=3
one both 100 WB_PIPE_STAGGER=1
one free 20 WB_PIPE_STAGGER=0
one phase1 20 WB_PIPE_STAGGER=2
done
