#!/bin/bash
# the wave's patch buffer at 192 (62 KB of LDS per workgroup, inside a launch's default share), flushed at 128 or 96 patches, against 160 / 64
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for v in default lbuf192 lbuf192f96; do
  echo "== $v"
  if [ $v = default ]; then WB_PROBE_SHORT=1 WB_PROBE_BUFFERS=5 python tools/stream_pattern_probe.py
  else WB_HIP_LIBRARY=$PWD/build/ab/libwb_$v.so WB_PROBE_SHORT=1 WB_PROBE_BUFFERS=5 python tools/stream_pattern_probe.py; fi
done
done
