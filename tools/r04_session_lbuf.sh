#!/bin/bash
# patch stores in fewer, larger bursts: the wave's LDS buffer doubled (320 patches, flush at 224) / flush at 96 instead of 64
cd "$(dirname "$0")/.."
for rep in 1 2; do
for v in default lbuf320 flush96; do
  echo "== $v"
  if [ $v = default ]; then WB_PROBE_SHORT=1 WB_PROBE_BUFFERS=5 python tools/stream_pattern_probe.py
  else WB_HIP_LIBRARY=$PWD/build/ab/libwb_$v.so WB_PROBE_SHORT=1 WB_PROBE_BUFFERS=5 python tools/stream_pattern_probe.py; fi
done
done
