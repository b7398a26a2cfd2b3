#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s4; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "float64 or f64 or rows_without" 2>&1 | tail -5) | tee $O/f64_tests.txt
WB_HIP_LIBRARY=$PWD/wavebreaking_amd/csrc/libwb_hip_bound2.so timeout 900 python tools/boxbound_check.py 2>&1 | grep -v amdgpu.ids | tee $O/boxbound_check.txt
bash tools/timeline.sh 1 > $O/timeline_one_batch.txt 2>&1; tail -45 $O/timeline_one_batch.txt
bash tools/ab_opts.sh 3 "default:X=1" "bound0:WB_HIP_LIBRARY=$PWD/wavebreaking_amd/csrc/libwb_hip_bound0.so" 2>&1 | tee $O/ab.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_s4/bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "value_without_geometry", "geometry_adds_ms_per_step")}, d["roofline"]["frac"], d["host_wall_ms"], d["end_to_end"]["value"], d["end_to_end"]["seconds"], d["cpu_baseline"]["tables_match_gpu"])
print(json.dumps(d["float64_field"]))
PY
