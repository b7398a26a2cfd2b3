#!/usr/bin/env python
"""development: where the time of the drop-in calls goes (host cube of 365 steps at 0.25 degree; cProfile of one call of each kind)"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavebreaking_amd as wb
from wavebreaking_amd import engine, synthetic as synth

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
wb.calculate_streamers(data=data, contour_levels=levels, intensity=idata)          # warm
cont = wb.calculate_contours(data=data, contour_levels=levels, original_coordinates=False)


def prof(label, f):
    pr = cProfile.Profile()
    tic = time.perf_counter()
    pr.enable()
    f()
    pr.disable()
    print("=====", label, "%.1f ms" % (1e3 * (time.perf_counter() - tic)))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14)
    print("\n".join(s.getvalue().splitlines()[4:24]))


prof("streamers, cached inputs + kept batch", lambda: wb.calculate_streamers(data=data, contour_levels=levels, intensity=idata))
prof("streamers, contours=frame", lambda: wb.calculate_streamers(data=data, contour_levels=levels, intensity=idata, contours=cont))
prof("calculate_contours(original_coordinates=False)", lambda: wb.calculate_contours(data=data, contour_levels=levels, original_coordinates=False))
os.environ["WB_DEBUG_TIMING"] = "1"
wb.calculate_streamers(data=data, contour_levels=levels, intensity=idata, contours=cont)


def cold():
    wb.release_inputs()
    return wb.calculate_streamers(data=data, contour_levels=levels, intensity=idata)


cold()
prof("streamers, cold cache (two uploads + everything)", cold)
prof("overturnings, cached", lambda: wb.calculate_overturnings(data=data, contour_levels=levels, intensity=idata))
