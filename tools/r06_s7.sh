#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_s7; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "wider or float64 or rows_without or quarter or device_geometry" 2>&1 | tail -15) | tee $O/new_tests.txt
WB_BENCH_FORCE_DIST=1 timeout 600 python bench.py --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --steps 20 > $O/bench_dist1.json 2> $O/bench_dist1.err; tail -c 400 $O/bench_dist1.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_s7/bench_dist1.json"))
print("dist1", d["value"], {k: d.get(k) for k in ("gather_verified", "events_sum_over_ranks", "rings_sum_over_ranks", "ring_vertices_sum_over_ranks", "gather_path")}, d["counts"].get("gathered_rings"))
PY
bash tools/ab_opts.sh 3 "default:X=1" "nospans:WB_HIP_LIBRARY=$PWD/wavebreaking_amd/csrc/libwb_hip_nospans.so" 2>&1 | tee $O/ab.txt
AB_ARGS="--streams 1" bash tools/ab_opts.sh 1 "one_batch:X=1" "one_batch_nospans:WB_HIP_LIBRARY=$PWD/wavebreaking_amd/csrc/libwb_hip_nospans.so" 2>&1 | tee -a $O/ab.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 > $R/$O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 > $R/$O/pmc_write.log 2>&1
cd $R; python tools/profile_summary.py pmc $O/pmc_fetch $O/pmc_write > $O/hbm_traffic_pmc.txt 2>&1; rm -rf $O/pmc_fetch $O/pmc_write; grep -E "props|scan_build|kernel " $O/hbm_traffic_pmc.txt
