#!/usr/bin/env python
"""development: every kernel of a window of a two-batches-in-flight run (rocprofv3 --kernel-trace csv) with the number of
OTHER kernels running at its start, and the share of the window during which 0 / 1 / 2 / 3+ kernels were running:
    python tools/timeline2.py <dir> [t_from_ms t_to_ms]"""
import csv, glob, os, sys
d = sys.argv[1]
rows = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][:40]))
rows.sort()
scans = [r for r in rows if "wb_k_scan_stream" in r[3]]
t_end = rows[-1][1]
w0 = scans[-9][0] if len(sys.argv) < 4 else rows[0][0] + int(float(sys.argv[2]) * 1e6)
w1 = scans[-3][0] if len(sys.argv) < 4 else rows[0][0] + int(float(sys.argv[3]) * 1e6)
ev = []
for s, e, q, n in rows:
    if e <= w0 or s >= w1:
        continue
    ev.append((max(s, w0), 1)); ev.append((min(e, w1), -1))
ev.sort()
hist = {}
cur, last = 0, w0
for t, dlt in ev:
    hist[cur] = hist.get(cur, 0) + (t - last)
    cur += dlt; last = t
hist[cur] = hist.get(cur, 0) + (w1 - last)
tot = float(w1 - w0)
print("window %.3f ms: kernels running 0: %.1f%%  1: %.1f%%  2: %.1f%%  3+: %.1f%%" % (tot / 1e6, 100 * hist.get(0, 0) / tot, 100 * hist.get(1, 0) / tot, 100 * hist.get(2, 0) / tot, 100 * sum(v for k, v in hist.items() if k >= 3) / tot))
# per kernel name: time alone vs overlapped
by = {}
for s, e, q, n in rows:
    if s < w0 or e > w1:
        continue
    by.setdefault(n, []).append((e - s) / 1e3)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("%-42s n %3d  avg %8.1f us  total %8.1f us" % (n, len(v), sum(v) / len(v), sum(v)))
print()
for s, e, q, n in rows:
    if s < w0 or s > w0 + 12e6:
        continue
    print("%9.1f %9.1f %8.1f  q%-3s %s" % ((s - w0) / 1e3, (e - w0) / 1e3, (e - s) / 1e3, q, n))
