#!/bin/bash
# the whole default bench line against the same timed region without the legs that follow it (which of them, if any, makes the difference)
cd "$(dirname "$0")/.."
one() {  # label args...
  label=$1; shift
  WB_BENCH_WATCHDOG=400 timeout 600 python bench.py "$@" > /tmp/ab.json 2> /tmp/ab.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "$label rc $rc"; tail -40 /tmp/ab.err; return; fi
  python - "$label" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
w = d["host_wall_ms"]
s = d["stage_ms"]
print(sys.argv[1], "value %d  ms/step %.3f frac %.3f alone %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"]), "f234 %.2f" % s["streamer_f234"], "wall", " ".join("%s %.2f" % (k, v) for k, v in w.items()), "e2e", int(d["end_to_end"]["value"]) if "end_to_end" in d else None, flush=True)
PY
}
for rep in 1 2; do
one full
one bare --cpu-sample 0 --e2e-steps 0 --stagger-steps 0
one no_cpu --cpu-sample 0
one no_e2e --e2e-steps 0
one no_stagger --stagger-steps 0
one full_driver --gpus 1 --steps 20 --warmup 5
one bare_driver --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0
done
