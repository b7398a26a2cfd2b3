#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s2; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "rows_without or device_geometry or quarter or public" 2>&1 | tail -5) > $O/new_tests.txt; cat $O/new_tests.txt
bash tools/ab_opts.sh 3 "default:X=1" "dma0:WB_BENCH_OPTS=geometry_dma=0" "stagger0:WB_PIPE_STAGGER=0" "stagger0_s3:WB_PIPE_STAGGER=0 WB_BENCH_STREAMS=3" "nogeom:WB_BENCH_GEOMETRY=0" "nogeom_st0:WB_BENCH_GEOMETRY=0 WB_PIPE_STAGGER=0" 2>&1 | tee $O/ab.txt
AB_ARGS="--streams 1" bash tools/ab_opts.sh 1 "one_batch:X=1" "one_batch_dma0:WB_BENCH_OPTS=geometry_dma=0" "one_batch_nogeom:WB_BENCH_GEOMETRY=0" 2>&1 | tee -a $O/ab.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_s2/bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "value_without_geometry", "geometry_adds_ms_per_step")}, d["roofline"]["frac"], d["stage_ms"], d["host_wall_ms"], d["end_to_end"]["value"], d["cpu_baseline"]["tables_match_gpu"])
PY
