#!/bin/bash
# one two-stream run in eight of the library at the head of the morning hung (no output, killed by timeout): repeat both libraries with a watchdog that dumps the threads' stacks
cd "$(dirname "$0")/.."
run() {  # label env...
  label=$1; shift
  env "$@" WB_BENCH_WATCHDOG=90 timeout 200 python bench.py --steps 40 --warmup 6 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 --streams 2 > /tmp/ab.json 2> /tmp/ab.err
  rc=$?
  echo "$label rc $rc $(python -c "import json;d=json.load(open('/tmp/ab.json'));print(int(d['value']), round(d['ms_per_step'],3))" 2>/dev/null)"
  if [ $rc -ne 0 ]; then echo "---- stderr"; tail -80 /tmp/ab.err; rocm-smi --showuse 2>/dev/null | tail -5; fi
}
for i in $(seq 1 12); do
  run "old $i" WB_HIP_LIBRARY=$PWD/build/ab/libwb_prev.so
  run "new $i" WB_X=1
done
