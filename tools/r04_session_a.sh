#!/bin/bash
# round 4, session a: GPU tests, bench (default / timers=1 / one batch), kernel stats
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r04a; export TMPDIR=/tmp; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $O/gpu_tests.txt
timeout 600 python bench.py --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 > $O/bench.json 2> $O/bench.err
WB_BENCH_OPTS=timers=1 timeout 600 python bench.py --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 > $O/bench_timers1.json 2> $O/bench_timers1.err
timeout 600 python bench.py --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 > $O/bench2.json 2> $O/bench2.err
timeout 600 python bench.py --streams 1 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 > $O/bench_s1.json 2> $O/bench_s1.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_stats2 -- python3 $R/bench.py --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 > $R/$O/prof_stats2.log 2>&1
cd $R
python tools/profile_summary.py stats $O/prof_stats2 > $O/kernel_stats.csv 2>&1
rm -rf $O/prof_stats2
cat $O/gpu_tests.txt
for f in bench bench_timers1 bench2 bench_s1; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1], "value %.0f ms/step %.3f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), {k: round(v,3) for k,v in d["stage_ms"].items()})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
head -30 $O/kernel_stats.csv
