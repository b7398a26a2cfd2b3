#!/bin/bash
# L2 request counters of the stream scan alone (calibration of the FETCH_SIZE figure)
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/scanpmc3; export TMPDIR=/tmp; mkdir -p $O; cd /tmp
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum"; do
i=$((i+1))
timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/$O/p$i -- python3 $R/tools/scan_probe.py 1 1 0 > $R/$O/p$i.log 2>&1
cd $R; python tools/profile_summary.py raw $O/p$i scan_stream 2>&1 | tail -4; cd /tmp
rm -rf $R/$O/p$i
done
