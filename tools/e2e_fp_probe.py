#!/usr/bin/env python
"""development: the three drop-in calls from a cold input cache, per call: seconds, seconds inside engine.verify_pending (the wait for
the fingerprints taken beside the call), seconds of wb_fingerprint"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavebreaking_amd as wb
from wavebreaking_amd import engine, synthetic as synth
from wavebreaking_amd.indices import _events

nt = int(os.environ.get("NT", 365))
ctx = engine.get_context()
lat, lon = synth.grid(0.25)
nb = nt * lat.size * lon.size * 4
d = ctx.alloc(nb)
ctx.synth_fill(d, np.float32, 0, nt, lat, lon, synth.wave_params(), 0.0, synth.SEED)
pv = ctx.to_host(d, (nt, lat.size, lon.size), np.float32)
ctx.synth_fill(d, np.float32, 0, nt, lat, lon, synth.wave_params(20241), 0.0, 20241)
it = ctx.to_host(d, (nt, lat.size, lon.size), np.float32)
ctx.free(d)
coords = {"time": synth.times(nt), "lat": wb.gridfield.Coord(lat, {"units": "degrees_north"}), "lon": wb.gridfield.Coord(lon, {"units": "degrees_east"})}
data = wb.GridField(pv, ("time", "lat", "lon"), coords, name="PV")
idata = wb.GridField(it, ("time", "lat", "lon"), coords, name="intensity")
levels = [-2.0, 2.0]
waits = [0.0]
real_verify = engine.verify_pending


def timed_verify():
    tic = time.perf_counter()
    try:
        return real_verify()
    finally:
        waits[0] += time.perf_counter() - tic


engine.verify_pending = timed_verify
_events.engine.verify_pending = timed_verify
calls = (("streamers", wb.calculate_streamers), ("overturnings", wb.calculate_overturnings), ("cutoffs", wb.calculate_cutoffs))
for f in calls:
    f[1](data=data, contour_levels=levels, intensity=idata)
for t in (8, 16, 32, 64):
    tic = time.perf_counter()
    ctx.L.fingerprint(pv, t)
    dt = time.perf_counter() - tic
    print("wb_fingerprint alone, %2d threads: %.1f ms  %.0f GB/s" % (t, 1e3 * dt, pv.nbytes / dt / 1e9))
for rep in range(int(os.environ.get("REPS", 4))):
    wb.release_inputs()
    line = []
    for name, f in calls:
        waits[0] = 0.0
        fp0 = engine.FINGERPRINT_STATS["seconds"]
        tic = time.perf_counter()
        f(data=data, contour_levels=levels, intensity=idata)
        line.append("%s %.1f ms (verify wait %.1f, fingerprints %.1f)" % (name, 1e3 * (time.perf_counter() - tic), 1e3 * waits[0], 1e3 * (engine.FINGERPRINT_STATS["seconds"] - fp0)))
    print("cold run %d:" % rep, "; ".join(line), flush=True)
