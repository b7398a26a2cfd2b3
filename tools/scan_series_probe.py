"""development probe: the scan's duration on ONE field buffer, launch after launch (is its 1.12 / 1.30 ms a property of the
buffer or a state of the device that changes in time?), with a plain streaming read of the same buffer in between"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavebreaking_amd import _hip
from wavebreaking_amd import synthetic as synth
ctx = _hip.Context(0)
lat, lon = synth.grid(0.25)
nt = 1460
nlat, nlon = lat.size, lon.size
fb = nt * nlat * nlon * 4
d = ctx.alloc(fb)
ctx.synth_fill(d, np.float32, 0, nt, lat, lon, synth.wave_params(), 0.0, synth.SEED)
out = []
t0 = time.perf_counter()
for rep in range(int(os.environ.get("WB_SERIES", 240))):
    b = ctx.contours(d, lat, lon, [-2.0, 2.0], 0.25, 0.25, device_shape=(nt, nlat, nlon), device_dtype=np.float32)
    s = dict(b.timings())["scan"]
    b.close()
    r = ctx.probe_read_ms(d, fb, 1) if rep % 4 == 3 else None
    out.append((time.perf_counter() - t0, s, r))
    if os.environ.get("WB_SERIES_SLEEP") and rep % 40 == 39:
        time.sleep(float(os.environ["WB_SERIES_SLEEP"]))
print("scan ms, launch after launch (x = a streaming read of the buffer in between, its ms in brackets):")
line = []
for t, s, r in out:
    line.append("%.2f" % s + ("(%.2f)" % r if r else ""))
    if len(line) == 20:
        print("t=%6.0f ms  " % (1e3 * t) + " ".join(line)); line = []
if line:
    print(" ".join(line))
