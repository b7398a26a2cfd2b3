#!/bin/bash
# scan variants (ring depth x waves per SIMD) with two batches in flight: the variant was chosen for the kernel alone
cd "$(dirname "$0")/.."
one() {  # label args...
  label=$1; shift
  WB_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 "$@" > /tmp/ab.json 2> /tmp/ab.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "$label rc $rc"; tail -20 /tmp/ab.err; return; fi
  python - "$label" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
s = d["stage_ms"]
print(sys.argv[1], "value %d  ms/step %.3f frac %.3f alone %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"]), "scan %.2f pairs %.2f f234 %.2f" % (s["scan"], s["streamer_pairs"], s["streamer_f234"]), flush=True)
PY
}
for rep in 1 2 3; do
for v in 2 0 1 3 4 5; do
one "variant$v" --scan-variant $v
done
done
