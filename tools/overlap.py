#!/usr/bin/env python
"""development: how the kernels of the two batches in flight share the device.  From a `rocprofv3 --kernel-trace --output-format csv`
run of bench.py: over the timed passes, the time with 0 / 1 / 2 / 3+ kernels running, and per kernel its total duration, the
part of it during which it ran alone, and its mean duration.   python tools/r05/overlap.py <dir> [first pass] [last pass]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
rows = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]))
rows.sort()
scans = [r for r in rows if "wb_k_scan_stream" in r[2]]
p0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(scans) // 3
p1 = int(sys.argv[3]) if len(sys.argv) > 3 else len(scans) - 3
t0, t1 = scans[p0][0], scans[p1][0]
npass = p1 - p0
ev = []
for s, e, n in rows:
    s, e = max(s, t0), min(e, t1)
    if e > s:
        ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort()
running = defaultdict(int)
hist = defaultdict(int)
alone = defaultdict(int)
total = defaultdict(int)
calls = defaultdict(int)
with_other = defaultdict(lambda: defaultdict(int))
prev = t0
for t, k, n in ev:
    dt = t - prev
    if dt > 0:
        live = [x for x, c in running.items() if c > 0]
        m = sum(running[x] for x in live)
        hist[min(m, 3)] += dt
        for x in live:
            total[x] += dt * running[x]
            if m == 1:
                alone[x] += dt
            else:
                for y in live:
                    if y != x or running[x] > 1:
                        with_other[x][y] += dt
    running[n] += k
    if k > 0:
        calls[n] += 1
    prev = t
hist[0] += 0
span = t1 - t0
print("passes %d, %.3f ms per pass; device time with 0 / 1 / 2 / 3+ kernels running: %.3f / %.3f / %.3f / %.3f ms per pass" % (
    npass, span / npass / 1e6, (span - sum(hist[k] for k in (1, 2, 3))) / npass / 1e6, hist[1] / npass / 1e6, hist[2] / npass / 1e6, hist[3] / npass / 1e6))
print("%-42s %8s %10s %10s   %s" % ("kernel", "mean us", "ms / pass", "alone ms", "mostly beside"))
for n in sorted(total, key=lambda x: -total[x]):
    if total[n] / npass < 5e3:
        continue
    top = sorted(with_other[n].items(), key=lambda kv: -kv[1])[:3]
    print("%-42s %8.1f %10.3f %10.3f   %s" % (n, total[n] / max(calls[n], 1) / 1e3, total[n] / npass / 1e6, alone[n] / npass / 1e6,
                                            ", ".join("%s %.2f" % (y.replace("wb_k_", ""), v / npass / 1e6) for y, v in top)))
