"""Details of one parity-hunt case on the GPU (development tool): python tools/dbg_case.py seed eps t0 dlon nt level..."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import synth, api as oapi
from tests import helpers
from wavebreaking_amd import _hip
seed, eps, t0, dlon, nt = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
levels = [float(v) for v in sys.argv[6:]]
ctx = _hip.Context(0)
lat, lon = synth.grid(dlon)
pv = synth.pv_field(t0, nt, lat, lon, eps=eps, seed=1000 + seed)
for rep in range(3):
    got = helpers.run_abi(ctx, pv, lat, lon, levels, intensity=pv)
    ref = oapi.run_all(pv, levels, lat, lon, synth.times(nt), intensity=pv)
    s, c = got["streamers"], got["contours"]
    print("rep", rep, "counts", s["step"].size, len(ref["streamers"]), "ties", got.get("group_ties"))
    for k, r in enumerate(ref["streamers"]):
        if k >= s["step"].size:
            break
        i1, i2 = int(s["ind1"][k]), int(s["ind2"][k]); o = int(c["offset"][int(s["contour"][k])])
        poly = np.c_[c["x"][o + i1:o + i2 + 1], c["y"][o + i1:o + i2 + 1]]
        if not np.array_equal(poly, r["_polygon"]):
            print("  event", k, "t", r["_t"], "level", r["level"], "gpu", i1, i2, "len", len(poly), "ref len", len(r["_polygon"]),
                  "ref ends", r["_polygon"][0], r["_polygon"][-1], "gpu ends", poly[0], poly[-1])
            break
