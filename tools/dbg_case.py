import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import synth
from tests import helpers
from wavebreaking_amd import _hip
ctx = _hip.Context(0)
def run(name, pv, lat, lon, tm, levels, **kw):
    try:
        helpers.full_parity(ctx, pv, lat, lon, tm, levels, **kw)
        print(name, "OK", flush=True)
    except AssertionError as e:
        print(name, "FAIL", str(e)[:200], flush=True)
        got = helpers.run_abi(ctx, pv, lat, lon, levels, **kw)
        from oracle import api as oapi
        ref = oapi.run_all(pv, levels, lat, lon, tm, **kw)
        s = got["streamers"]; c = got["contours"]
        for k, r in enumerate(ref["streamers"]):
            i1, i2 = int(s["ind1"][k]), int(s["ind2"][k])
            o = int(c["offset"][int(s["contour"][k])])
            poly = np.c_[c["x"][o + i1:o + i2 + 1], c["y"][o + i1:o + i2 + 1]]
            if not np.array_equal(poly, r["_polygon"]):
                print(" event", k, "gpu ind", i1, i2, "len", len(poly), "ref len", len(r["_polygon"]), "ref first/last", r["_polygon"][0], r["_polygon"][-1], "gpu first/last", poly[0], poly[-1], "level", r["level"], "t", r["_t"])
lat, lon = synth.grid(0.25)
pv = synth.pv_field(40, 1, lat, lon, eps=0.3)
run("f32 0.25", pv, lat, lon, synth.times(1), [-2, 2, 1.5])
run("f64 0.25", pv.astype(np.float64), lat, lon, synth.times(1), [-2, 2, 1.5])
lat, lon = synth.grid(4.0)
pv = synth.pv_field(3, 4, lat, lon, eps=0.3)
run("f32 4deg", pv, lat, lon, synth.times(4), [-2, 2], intensity=pv)
