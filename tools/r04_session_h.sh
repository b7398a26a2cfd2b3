#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_config1.py -x -q -s 2>&1 | tail -4 | tee $O/config1_test.txt
for nb in 0 1 2 3 4 6; do for var in 2 0 3; do
WB_BENCH_OPTS=scan_bands=$nb timeout 600 python bench.py --config 1 --scan-variant $var --streams 1 --stagger-steps 0 --cpu-sample 0 --steps 30 > /tmp/b.json 2> /tmp/b.err || tail -5 /tmp/b.err
python - /tmp/b.json $nb $var <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("bands", sys.argv[2], "variant", sys.argv[3], "value %d ms/step %.3f scan %.3f ms frac %.3f build %.3f" % (d["value"], d["ms_per_step"], d["stage_ms"]["scan"], d["roofline"]["frac"], d["stage_ms"]["scan_build"]))
PY
done; done 2>&1 | tee $O/bands.txt
