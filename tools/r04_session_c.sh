#!/bin/bash
# round 4, session c: queue-based box_com A/B
cd "$(dirname "$0")/.."
O=gpurun_out/r04c; mkdir -p $O
one() {  # label, streams, extra env ...
  label=$1; st=$2; shift 2
  env "$@" timeout 300 python bench.py --steps 30 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams $st > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
  python - "$label" $st <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
s = d["stage_ms"]
keys = ("scan", "trace_link", "trace_rank", "trace_finish", "overturn", "box_com", "streamer_pairs", "streamer_f1", "streamer_f234", "props", "polygons_d2h")
print(sys.argv[1], "streams", sys.argv[2], "value %d  ms/step %.3f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), " ".join("%s %.2f" % (k, s.get(k, -1)) for k in keys), "rechecks", d["counts"]["com_rechecks"], flush=True)
PY
}
{
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "com or reference_own or full_parity" 2>&1 | tail -3
for st in 1 2; do
one default $st WB_X=1
one percu4 $st WB_BENCH_OPTS=boxcom_per_cu=4
one percu2 $st WB_BENCH_OPTS=boxcom_per_cu=2
one percu16 $st WB_BENCH_OPTS=boxcom_per_cu=16
one segs6 $st WB_HIP_LIBRARY=$PWD/build/ab/libwb_segs6.so
one segs6_16 $st WB_HIP_LIBRARY=$PWD/build/ab/libwb_segs6.so WB_BENCH_OPTS=boxcom_per_cu=16
one segs4r32 $st WB_HIP_LIBRARY=$PWD/build/ab/libwb_segs4r32.so WB_BENCH_OPTS=boxcom_per_cu=16
done
} 2>&1 | tee $O/ab.txt
