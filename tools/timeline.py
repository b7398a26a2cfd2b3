#!/usr/bin/env python
"""development: the kernels of the LAST pass of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py as a timeline
(start and end in microseconds after the pass's first kernel, queue, name):  python tools/timeline.py <dir> [passes back]"""
import csv, glob, os, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][:48]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "wb_k_scan_stream" in r[3]]
i0 = starts[-1 - back]
i1 = starts[-back] if back else len(rows)
t0 = rows[i0][0]
for s, e, q, n in rows[i0:i1]:
    print("%9.1f %9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
