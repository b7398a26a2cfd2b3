#!/usr/bin/env python
"""development / evidence: a library built with -DWB_BOXCOM_BOUND=2 (WB_HIP_LIBRARY) evaluates every overturning box that the
bound settles ALSO by the chained passes and counts the boxes on which the two differ (must be 0): whole years of the bench's
0.25 degree record, smooth and rough, and coarser grids."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavebreaking_amd import _hip, synthetic as synth

ctx = _hip.Context(0)
tot = [0, 0, 0]
for dlon, nt, eps, t0s in ((0.25, 1460, 0.0, (0, 1460, 2920)), (0.25, 365, 0.3, (0, 5000)), (1.0, 1460, 0.1, (0, 1460, 20000)), (0.5, 730, 0.05, (0, 9000))):
    lat, lon = synth.grid(dlon)
    d = ctx.alloc(nt * lat.size * lon.size * 4)
    for t0 in t0s:
        ctx.synth_fill(d, np.float32, t0, nt, lat, lon, synth.wave_params(), eps, synth.SEED)
        b = ctx.contours(d, lat, lon, [-2.0, 2.0], dlon, dlon, device_shape=(nt, lat.size, lon.size), device_dtype=np.float32, want_cells=False)
        b.indices(streamers=False, cutoffs=False)
        q, s, m = b.com_rechecks()[1], b.counter("box_bound"), b.counter("box_bound_mismatch")
        print("%.2f deg, %d steps from %d, eps %.2f: %d boxes queued, %d settled by the bound (%.1f %%), %d mismatches" % (dlon, nt, t0, eps, q, s, 100.0 * s / max(q, 1), m), flush=True)
        tot[0] += q; tot[1] += s; tot[2] += m
        b.close()
    ctx.free(d)
print("total: %d queued, %d settled, %d mismatches" % tuple(tot))
sys.exit(1 if tot[2] else 0)
