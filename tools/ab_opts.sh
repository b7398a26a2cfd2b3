#!/bin/bash
# A/B of context options / environment on the GPU box, round-robin so that box-to-box drift cancels:
#   tools/ab_opts.sh REPS "label:VAR=value VAR=value ..." ...      (WB_BENCH_OPTS="name=value,..." sets context options)
# prints value, ms per pass and the stage times of every run; AB_ARGS adds bench.py arguments
cd "$(dirname "$0")/.."
REPS=$1; shift
for rep in $(seq 1 $REPS); do
  for cfg in "$@"; do
    label=${cfg%%:*}; envs=${cfg#*:}
    env $envs timeout 300 python bench.py --steps ${AB_STEPS:-40} --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --stage-steps ${AB_STAGE:-6} ${AB_ARGS:-} > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
    python - "$label" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/ab.json"))
    s = d["stage_ms"]
    print("%-28s value %7d  ms/pass %.3f  frac %.3f |" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"]),
          " ".join("%s %.2f" % (k.replace("streamer_", "s_").replace("trace_", "t_"), v) for k, v in s.items() if v >= 0.045), "| host", " ".join("%s %.2f" % kv for kv in d["host_wall_ms"].items()), flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, flush=True)
PY
  done
done
