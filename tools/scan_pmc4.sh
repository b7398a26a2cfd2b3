#!/bin/bash
# L2 read requests of the stream scan (default variant) with and without its patch stores
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/scanpmc4; export TMPDIR=/tmp; mkdir -p $O; cd /tmp
for dbg in 0 1; do
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum --output-format csv -d $R/$O/p$dbg -- python3 $R/tools/scan_probe.py 1 2 $dbg 1 > $R/$O/p$dbg.log 2>&1
cd $R; echo "stores off = $dbg"; python tools/profile_summary.py raw $O/p$dbg scan_stream 2>&1 | tail -2; cd /tmp
rm -rf $R/$O/p$dbg
done
