"""development probe: is the scan's bimodal duration (1.12 / 1.32 ms from process to process on one box) tied to where the field
buffer lies?  Several buffers allocated one after the other (all kept), the scan timed on each."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavebreaking_amd import _hip
from wavebreaking_amd import synthetic as synth
ctx = _hip.Context(0)
lat, lon = synth.grid(0.25)
nt = 1460
nlat, nlon = lat.size, lon.size
fb = nt * nlat * nlon * 4
par = synth.wave_params()
bufs = []
for k in range(6):
    pad = ctx.alloc((k * 37 + 1) << 20)          # odd-sized spacers: the next buffer starts somewhere else
    d = ctx.alloc(fb)
    ctx.synth_fill(d, np.float32, 0, nt, lat, lon, par, 0.0, synth.SEED)
    bufs.append((pad, d))
    ts = []
    for rep in range(4):
        b = ctx.contours(d, lat, lon, [-2.0, 2.0], 0.25, 0.25, device_shape=(nt, nlat, nlon), device_dtype=np.float32)
        ts.append(dict(b.timings())["scan"])
        b.close()
    if os.environ.get("WB_PROBE_BAND_ROWS"):
        per = []
        for br in [int(v) for v in os.environ["WB_PROBE_BAND_ROWS"].split(",")]:
            ctx.set_option("scan_band_rows", br)
            tt = []
            for rep in range(3):
                b = ctx.contours(d, lat, lon, [-2.0, 2.0], 0.25, 0.25, device_shape=(nt, nlat, nlon), device_dtype=np.float32)
                tt.append(dict(b.timings())["scan"])
                b.close()
            per.append("%d rows %.3f" % (br, min(tt)))
        ctx.set_option("scan_band_rows", 0)
        print("   " + ", ".join(per), flush=True)
    if os.environ.get("WB_PROBE_BANDS"):
        per = []
        for nb in [int(v) for v in os.environ["WB_PROBE_BANDS"].split(",")]:
            ctx.set_option("scan_bands", nb)
            tt = []
            for rep in range(3):
                b = ctx.contours(d, lat, lon, [-2.0, 2.0], 0.25, 0.25, device_shape=(nt, nlat, nlon), device_dtype=np.float32)
                tt.append(dict(b.timings())["scan"])
                b.close()
            per.append("%d bands %.3f" % (nb, min(tt)))
        ctx.set_option("scan_bands", 0)
        print("   " + ", ".join(per), flush=True)
    rd = ctx.probe_read_ms(d, fb, 3)
    print("   streaming read of it: %.3f ms = %.2f TB/s" % (rd, fb / rd / 1e9), flush=True)
    print("buffer %d at 0x%x (mod 1 GiB: %d MiB, mod 2 MiB: %d KiB): scan %s ms" % (k, d, (d % (1 << 30)) >> 20, (d % (2 << 20)) >> 10, ["%.3f" % t for t in ts]), flush=True)
ctx.close()
