#!/bin/bash
# a silent death of one two-stream bench run (rc != 0, nothing on stderr): repeat the run with the fault handler on and print exit codes; then stamps
cd "$(dirname "$0")/.."
for i in $(seq 1 14); do
  timeout 300 python -X faulthandler bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams 2 > /tmp/ab.json 2> /tmp/ab.err
  rc=$?
  echo "run $i rc $rc $(head -c 60 /tmp/ab.json | tr -d '\n')"
  if [ $rc -ne 0 ]; then echo "---- stderr of run $i"; tail -60 /tmp/ab.err; fi
done
for n in 1 4 6 7; do
echo "stamps $n"
WB_HIP_LIBRARY=$PWD/build/ab/libwb_stamps$n.so WB_DEBUG_STAMPS=1 python bench.py --steps 2 --warmup 0 --cpu-sample 0 --e2e-steps 0 --streams 1 --stagger-steps 0 2>&1 | grep -E "stamps" | tail -2
done
