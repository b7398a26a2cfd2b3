"""development: lengths of the selected contours of the bench batch (what sizes the LDS of the per-contour kernels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import synth            # (development tool: the generator's parameters only)
from wavebreaking_amd import _hip
ctx = _hip.Context(0)
lat, lon = synth.grid(0.25)
nt = int(os.environ.get("NT", 1460))
d = ctx.alloc(nt * lat.size * lon.size * 4)
ctx.synth_fill(d, np.float32, 0, nt, lat, lon, synth.wave_params(), 0.0, synth.SEED)
b = ctx.contours(d, lat, lon, [-2.0, 2.0], 0.25, 0.25, device_shape=(nt, lat.size, lon.size), device_dtype=np.float32)
t = b.contour_table()
mx = t["nux"].max()
sel = t["npts"][t["nux"] == mx]
print("contours", t["npts"].size, "selected", sel.size, "points sel", sel.sum())
print("selected lengths percentiles 50/90/95/98/99/99.5/100:", np.percentile(sel, [50, 90, 95, 98, 99, 99.5, 100]).astype(int))
for cap in (2048, 2560, 3072, 3584, 4096, 5120, 6144, 8192):
    print("  <= %d: %.4f" % (cap, (sel <= cap).mean()))
allc = t["npts"]
print("all contours: n per job mean %.1f; closed %d" % (allc.size / (2.0 * nt), t["closed"].sum()))
