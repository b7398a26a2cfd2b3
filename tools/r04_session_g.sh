#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_config1.py -x -q -s 2>&1 | tail -8 | tee $O/config1_test.txt
for st in 2 1; do
timeout 600 python bench.py --config 1 --streams $st --stagger-steps 0 > $O/bench_config1_s$st.json 2> $O/bench_config1_s$st.err || tail -5 $O/bench_config1_s$st.err
python - $O/bench_config1_s$st.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value %d ms/step %.3f" % (d["value"], d["ms_per_step"]), d["roofline"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, d.get("cpu_baseline"))
PY
done
