#!/bin/bash
# Does the profiler change the scan's duration?  The same one-batch-at-a-time bench, plain / under rocprofv3 / plain, and under
# rocprofv3 with the static item distribution; prints the live scan launch time of each and rocprofv3's own average.
set -u
cd "$(dirname "$0")/.."
R=$PWD
O=gpurun_out/${1:-rocprof_ab}
export TMPDIR=/tmp
mkdir -p $O
A="--cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0"
timeout 300 python3 bench.py $A > $O/plain1.json 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/p1 -- python3 $R/bench.py $A > $R/$O/prof_dynamic.json 2>/dev/null
WB_BENCH_OPTS="scan_dynamic=0" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/p2 -- python3 $R/bench.py $A > $R/$O/prof_static.json 2>/dev/null
cd $R
WB_BENCH_OPTS="scan_dynamic=0" timeout 300 python3 bench.py $A > $O/plain_static.json 2>/dev/null
timeout 300 python3 bench.py $A > $O/plain2.json 2>/dev/null
python tools/profile_summary.py stats $O/p1 | grep -E "scan_stream|wb_k_props" > $O/rocprof_dynamic.txt
python tools/profile_summary.py stats $O/p2 | grep -E "scan_stream|wb_k_props" > $O/rocprof_static.txt
rm -rf $O/p1 $O/p2
for f in plain1 prof_dynamic prof_static plain_static plain2; do
python3 - $O/$f.json $f <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-14s scan %.3f ms  props %.3f ms  value %.0f" % (sys.argv[2], d["stage_ms"]["scan"], d["stage_ms"]["props"], d["value"]))
PY
done
cat $O/rocprof_dynamic.txt $O/rocprof_static.txt
