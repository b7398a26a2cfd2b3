#!/bin/bash
# how many single-wave workgroups of wb_k_box_com per CU (13 KB of LDS each, on the side stream beside prep / pairs2 / f1)
cd "$(dirname "$0")/.."
one() {  # label, streams, extra env ...
  label=$1; st=$2; shift 2
  env "$@" WB_BENCH_WATCHDOG=120 timeout 300 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams $st > /tmp/ab.json 2> /tmp/ab.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "rc $rc"; tail -40 /tmp/ab.err; return; fi
  python - "$label" $st <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
s = d["stage_ms"]
keys = ("scan", "trace_link", "trace_rank", "trace_finish", "overturn", "box_com", "streamer_prep", "streamer_pairs", "streamer_f1", "streamer_f234", "props", "polygons")
print(sys.argv[1], "streams", sys.argv[2], "value %d  ms/step %.3f frac %.3f alone %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_alone"]), " ".join("%s %.2f" % (k, s.get(k, -1)) for k in keys), flush=True)
PY
}
for rep in 1 2; do
for n in 12 8 6 4 3 2; do
one percu$n 1 WB_BENCH_OPTS=boxcom_per_cu=$n
one percu$n 2 WB_BENCH_OPTS=boxcom_per_cu=$n
done
done
