#!/bin/bash
# end of round 4: the wider GPU-vs-oracle sweep, the 40-year record on one GPU, the spatial kernels, the new option test
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04f
timeout 900 python tests/gpu_parity_sweep.py > gpurun_out/r04f/sweep.txt 2>&1; echo "sweep rc $?" >> gpurun_out/r04f/sweep.txt
timeout 900 python benchmarks/bench_record.py --years 40 > gpurun_out/r04f/record.json 2> gpurun_out/r04f/record.err; echo "record rc $?"
timeout 600 python benchmarks/bench_spatial.py > gpurun_out/r04f/spatial.txt 2>&1; echo "spatial rc $?"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "launch_options or reference_own" 2>&1 | tail -2
tail -3 gpurun_out/r04f/sweep.txt; tail -c 600 gpurun_out/r04f/record.json; echo; tail -5 gpurun_out/r04f/spatial.txt
