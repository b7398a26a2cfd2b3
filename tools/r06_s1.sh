#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s1; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "rows_without or device_geometry or quarter" 2>&1 | tail -5) > $O/new_tests.txt; cat $O/new_tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_s1/bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "value_without_geometry", "geometry_adds_ms_per_step")}, d["roofline"]["frac"], d["stage_ms"], d["host_wall_ms"], d["end_to_end"]["value"], d["cpu_baseline"]["tables_match_gpu"])
PY
bash tools/ab_opts.sh 3 "default:X=1" "geom_sync:WB_BENCH_OPTS=geometry_async=0" "polys_eager:WB_BENCH_OPTS=index_polygons=1" "r05_like:WB_BENCH_OPTS=geometry_async=0,index_polygons=1" "ahead1:WB_PIPE_AHEAD=1" "ahead3:WB_PIPE_AHEAD=3" "streams3:WB_BENCH_STREAMS=3" "stagger0:WB_PIPE_STAGGER=0" "nogeom:WB_BENCH_GEOMETRY=0" 2>&1 | tee $O/ab.txt
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > $O/gpu_tests.txt; cat $O/gpu_tests.txt
