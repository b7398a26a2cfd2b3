#!/bin/bash
# FETCH_SIZE of the stream scan alone, with and without its stores (scan_debug bit 0)
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/scanpmc2; export TMPDIR=/tmp; mkdir -p $O; cd /tmp
for dbg in 0 1; do
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/f$dbg -- python3 $R/tools/scan_probe.py 1 1 $dbg > $R/$O/f$dbg.log 2>&1
cd $R; python tools/profile_summary.py pmc $O/f$dbg $O/f$dbg 2>&1 | grep -E "kernel|scan_stream"; cd /tmp
done
rm -rf $R/$O/f0 $R/$O/f1
