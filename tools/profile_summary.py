#!/usr/bin/env python
"""Condense rocprofv3 output directories into the small text tables kept under profiles/.

    python tools/profile_summary.py stats  <dir of `rocprofv3 --kernel-trace --stats --output-format csv`>  > profiles/rNN_kernel_stats.csv
    python tools/profile_summary.py pmc    <dir of `--pmc FETCH_SIZE`> <dir of `--pmc WRITE_SIZE`>          > profiles/rNN_hbm_traffic_pmc.txt

The PMC table sums the counter over the dispatches of each kernel; FETCH_SIZE is doubled in the `corrected` column
(gfx950 reports half the bytes of wide coalesced reads, /opt/skills/guides/MI355X_MICROARCH.md, HBM section).
"""
import csv
import glob
import os
import sys
from collections import OrderedDict


def find(d, pattern):
    hits = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))
    if not hits:
        raise SystemExit("no %s under %s" % (pattern, d))
    return hits


def stats(d):
    for path in find(d, "*kernel_stats.csv"):
        sys.stdout.write(open(path).read())


def counters(d, name):
    acc = OrderedDict()
    for path in find(d, "*counter_collection.csv"):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != name:
                continue
            k = row["Kernel_Name"]
            v = acc.setdefault(k, [0, 0.0, set()])
            v[2].add(row.get("Dispatch_Id"))
            v[1] += float(row["Counter_Value"])
    return OrderedDict((k, (len(v[2]), v[1])) for k, v in acc.items())


def pmc(dfetch, dwrite):
    f, w = counters(dfetch, "FETCH_SIZE"), counters(dwrite, "WRITE_SIZE")
    print("%-64s%8s %16s %22s %16s" % ("kernel", "calls", "FETCH_SIZE_KB", "fetch_corrected_GB", "WRITE_SIZE_KB"))
    for k, (calls, fv) in f.items():
        wv = w.get(k, (0, 0.0))[1]
        print("%-64s%8d %16.0f %22.3f %16.0f" % (k[:60], calls, fv, 2.0 * fv * 1024 / 1e9, wv))


def sq(d):
    """per kernel: SQ counters summed over its dispatches; parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves waiting on memory
    or a barrier), issue = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES"""
    names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
             "SQ_INSTS_VMEM", "SQ_INSTS_LDS"]
    tabs = {n: counters(d, n) for n in names}
    print("%-56s%6s %14s %8s %8s %12s %12s %12s %12s" % ("kernel", "calls", "wave_cycles", "parked", "issue", "valu", "salu", "vmem", "lds"))
    for k, (calls, wc) in tabs["SQ_WAVE_CYCLES"].items():
        g = lambda n: tabs[n].get(k, (0, 0.0))[1]
        print("%-56s%6d %14.0f %8.3f %8.3f %12.0f %12.0f %12.0f %12.0f" % (
            k[:54], calls, wc, g("SQ_WAIT_ANY") / max(wc, 1), g("SQ_ACTIVE_INST_ANY") / max(wc, 1), g("SQ_INSTS_VALU"),
            g("SQ_INSTS_SALU"), g("SQ_INSTS_VMEM"), g("SQ_INSTS_LDS")))


def raw(d, pattern=""):
    """every counter found, summed over the dispatches of each kernel whose name contains `pattern`"""
    acc = OrderedDict()
    for path in find(d, "*counter_collection.csv"):
        for row in csv.DictReader(open(path)):
            if pattern in row["Kernel_Name"]:
                k = (row["Kernel_Name"][:60], row["Counter_Name"])
                v = acc.setdefault(k, [set(), 0.0])
                v[0].add(row.get("Dispatch_Id"))
                v[1] += float(row["Counter_Value"])
    for (k, c), (ids, v) in acc.items():
        print("%-62s %-28s calls %3d  sum %.0f  per call %.0f" % (k, c, len(ids), v, v / max(len(ids), 1)))


if __name__ == "__main__":
    if sys.argv[1] == "raw":
        raw(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
    elif sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "sq":
        sq(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
