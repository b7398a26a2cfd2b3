"""round 5, review item 4: the streaming scan with its patches stored as keys only (the build kernel reads the values from the
field) against key + ten values, on six field buffers allocated one after the other in one process (the kernel's two modes
followed the buffer: tools/scan_placement_probe.py) -- scan and build kernel times, both formats on every buffer."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wavebreaking_amd import _hip
from wavebreaking_amd import synthetic as synth
ctx = _hip.Context(0)
ctx.set_option("timers", 2)              # (every stage timed: the library times the field scan only by default)
lat, lon = synth.grid(0.25)
nt = 1460
nlat, nlon = lat.size, lon.size
fb = nt * nlat * nlon * 4
par = synth.wave_params()
bufs = []
for k in range(int(os.environ.get("WB_PROBE_BUFFERS", 6))):
    pad = ctx.alloc((k * 37 + 1) << 20)
    d = ctx.alloc(fb)
    ctx.synth_fill(d, np.float32, 0, nt, lat, lon, par, 0.0, synth.SEED)
    bufs.append((pad, d))
    line = []
    for ko in (1, 0):
        ctx.set_option("scan_keys_only", ko)
        ts, tb, tf = [], [], []
        for rep in range(5):
            b = ctx.contours(d, lat, lon, [-2.0, 2.0], 0.25, 0.25, device_shape=(nt, nlat, nlon), device_dtype=np.float32)
            t = dict(b.timings())
            ts.append(t["scan"]); tb.append(t["scan_build"]); tf.append(t.get("trace_fused", float("nan")))
            b.close()
        line.append("keys_only %d: scan %.3f (%.3f .. %.3f) build %.3f fused %.3f" % (ko, min(ts), sorted(ts)[1], max(ts), min(tb), min(tf)))
    ctx.set_option("scan_keys_only", 1)
    print("buffer %d: " % k + " | ".join(line) + " | frac of 8 TB/s, keys only: %.3f" % (fb / (min(ts) * 1e-3) / 8e12 if False else fb / (float(line[0].split("scan ")[1].split(" ")[0]) * 1e-3) / 8e12), flush=True)
ctx.close()
