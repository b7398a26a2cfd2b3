#!/bin/bash
# thirty runs of the driver's command shape on the final library: exit codes and values (a run that hangs is killed by the watchdog with every thread's stack on stderr)
cd "$(dirname "$0")/.."
for i in $(seq 1 30); do
  WB_BENCH_WATCHDOG=120 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --e2e-steps 0 --stagger-steps 0 > /tmp/ab.json 2> /tmp/ab.err
  rc=$?
  echo "run $i rc $rc $(python -c "import json;d=json.load(open('/tmp/ab.json'));print(int(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],3))" 2>/dev/null)"
  if [ $rc -ne 0 ]; then tail -60 /tmp/ab.err; fi
done
