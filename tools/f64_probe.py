#!/usr/bin/env python
"""development: phase 1 on a float64 year of the bench record, fast path (scan_f64_fast) against the scalar-halo kernel"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavebreaking_amd import _hip, synthetic as synth
ctx = _hip.Context(0)
ctx.set_option("timers", 2)              # (every stage timed: the library times the field scan only by default)
nt = 1460
lat, lon = synth.grid(0.25)
d = ctx.alloc(nt * lat.size * lon.size * 8)
ctx.synth_fill(d, np.float64, 0, nt, lat, lon, synth.wave_params(), 0.0, synth.SEED)
for fast in (1, 0, 1, 0):
    ctx.set_option("scan_f64_fast", fast)
    ms = []
    for rep in range(4):
        b = ctx.contours(d, lat, lon, [-2.0, 2.0], 0.25, 0.25, device_shape=(nt, lat.size, lon.size), device_dtype=np.float64, want_cells=False)
        t = dict(b.timings()); b.close()
        if rep: ms.append((t["scan"], t["scan_build"], t["trace_fused"]))
    m = np.median(np.array(ms), axis=0)
    print("%s fast %d: scan %.3f ms (%.2f TB/s)  build %.3f  fused %.3f" % (os.environ.get("WB_HIP_LIBRARY", "default").split("/")[-1], fast, m[0], nt * lat.size * lon.size * 8 / m[0] / 1e9, m[1], m[2]), flush=True)
