"""props / streamer-chain stage times for subsets of the three indices (development probe; run on the GPU box)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from wavebreaking_amd import _hip, synthetic as synth
ctx = _hip.Context(0)
ctx.set_option("timers", 2)              # (every stage timed: the library times the field scan only by default)
lat, lon = synth.grid(0.25)
nt = 1460
d = ctx.alloc(nt * lat.size * lon.size * 4)
di = ctx.alloc(nt * lat.size * lon.size * 4)
ctx.synth_fill(d, np.float32, 0, nt, lat, lon, synth.wave_params(), 0.0, synth.SEED)
ctx.synth_fill(di, np.float32, 0, nt, lat, lon, synth.wave_params(20241), 0.0, 20241)
for which in (dict(streamers=True, overturnings=True, cutoffs=True), dict(streamers=True, overturnings=False, cutoffs=False),
              dict(streamers=False, overturnings=True, cutoffs=True)):
    best = None
    for rep in range(3):
        b = ctx.contours(d, lat, lon, [-2.0, 2.0], 0.25, 0.25, intensity=di, device_shape=(nt, lat.size, lon.size), device_dtype=np.float32)
        b.indices(b.max_nux(), **which)
        t = dict(b.timings()); n = [b.events(k)["step"].size for k in (0, 1, 2)]
        b.close()
        if best is None or t["props"] < best["props"]:
            best = t
    print(which, n, {k: round(v, 2) for k, v in best.items() if k in ("props", "box_com", "overturn", "streamer_pairs", "streamer_f234", "polygons", "fill_events")}, flush=True)
