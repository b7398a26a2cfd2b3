"""development probe: one 6 GB buffer read (a) as one dense sweep, (b) as the scan reads it -- G workgroups each streaming
chunks of C bytes -- for several G and C.  Which stream length / count loses bandwidth, and is it the loss the scan shows?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavebreaking_amd import _hip
from wavebreaking_amd import synthetic as synth
ctx = _hip.Context(0)
lat, lon = synth.grid(0.25)
nt = 1460
fb = nt * lat.size * lon.size * 4
for k in range(int(os.environ.get("WB_PROBE_BUFFERS", 3))):
    pad = ctx.alloc((k * 37 + 1) << 20)
    d = ctx.alloc(fb)
    ctx.synth_fill(d, np.float32, 0, nt, lat, lon, synth.wave_params(), 0.0, synth.SEED)
    ts = []
    for rep in range(3):
        b = ctx.contours(d, lat, lon, [-2.0, 2.0], 0.25, 0.25, device_shape=(nt, lat.size, lon.size), device_dtype=np.float32)
        ts.append(dict(b.timings())["scan"]); b.close()
    if os.environ.get("WB_PROBE_PADS"):
        per = []
        for pad in [int(v) for v in os.environ["WB_PROBE_PADS"].split(",")]:
            ctx.set_option("scan_pad", pad)
            tt = []
            for rep in range(3):
                b = ctx.contours(d, lat, lon, [-2.0, 2.0], 0.25, 0.25, device_shape=(nt, lat.size, lon.size), device_dtype=np.float32)
                tt.append(dict(b.timings())["scan"]); b.close()
            per.append("pad %d: %.3f" % (pad, min(tt)))
        ctx.set_option("scan_pad", 0)
        print("   " + ", ".join(per), flush=True)
    ctx.set_option("scan_debug", 1)                 # the scan without its patch stores (results meaningless)
    tn = []
    for rep in range(3):
        try:
            b = ctx.contours(d, lat, lon, [-2.0, 2.0], 0.25, 0.25, device_shape=(nt, lat.size, lon.size), device_dtype=np.float32)
            tn.append(dict(b.timings())["scan"]); b.close()
        except Exception as e:
            tn.append(float("nan"))
    ctx.set_option("scan_debug", 0)
    print("buffer %d: scan %.3f ms, scan without its patch stores %.3f ms, dense read %.3f ms" % (k, min(ts), min(tn), ctx.probe_read_ms(d, fb, 3)), flush=True)
    if os.environ.get("WB_PROBE_SHORT"):
        continue
    row = 5760
    for grid in (256, 512, 1024):
        per = []
        for chunk in (240 * row, 60 * row, 16 * row, 4 * row, 8192):
            per.append("%7d B: %.3f" % (chunk, ctx.probe_streams_ms(d, fb, chunk, grid, 3)))
        print("   %4d workgroups streaming chunks of  " % grid + "  ".join(per), flush=True)
