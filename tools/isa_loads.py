#!/usr/bin/env python
"""development: how the global loads of every kernel are waited for.  Compiles wb_api.hip to gfx950 assembly and lists, per
kernel, its global / flat loads and how many of them are followed at once by `s_waitcnt vmcnt(0)` -- a load behind a branch
(`cond ? a[i] : 0`) or a dependent gather is waited for on the spot; eight of them in an unrolled loop are one load in flight,
not eight (wb_k_props' field sums, round 3).     python tools/isa_loads.py [kernel name fragment ...]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "wavebreaking_amd", "csrc", "wb_api.hip")
out = os.path.join(tempfile.gettempdir(), "wb_api_gfx950.s")
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                       "-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
want = sys.argv[1:]
cur, body = None, {}
for line in open(out):
    m = re.match(r"^(_Z\w+):\s+; @", line)
    if m:
        cur = m.group(1)
        body[cur] = []
    elif cur and line.startswith("\t") and not line.strip().startswith((".", ";")):
        body[cur].append(line.strip())
print("%-64s %6s %12s %10s" % ("kernel", "loads", "waited at once", "longest run"))
for name, ins in body.items():
    if "wb_k_" not in name or (want and not any(w in name for w in want)):
        continue
    loads = [k for k, l in enumerate(ins) if l.startswith(("global_load", "flat_load", "buffer_load"))]
    solo = sum(1 for k in loads if any(x.startswith("s_waitcnt vmcnt(0)") for x in ins[k + 1:k + 3]))
    run = best = 0
    for l in ins:
        if l.startswith(("global_load", "flat_load", "buffer_load")):
            run += 1
            best = max(best, run)
        elif l.startswith("s_waitcnt vmcnt"):
            run = 0
    if loads:
        print("%-64s %6d %12d %10d" % (name[:64], len(loads), solo, best))
