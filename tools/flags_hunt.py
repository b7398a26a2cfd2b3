"""Hunt for to_xarray mismatches on the index functions' own polygons against the brute-force oracle (development tool).
python tools/flags_hunt.py <first seed> <count>     WB_HUNT_GPU=1: the HIP library"""
import os, sys, time
import multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def case(seed):
    import numpy as np
    import pandas as pd
    import helpers
    import wavebreaking_amd as wb
    from wavebreaking_amd import _hip, engine
    from wavebreaking_amd.processing.events import _rings
    from oracle import synth, flags as oflags
    engine._contexts[0] = _hip.Context(0) if os.environ.get("WB_HUNT_GPU") else _hip.Context(0, helpers.hostsim_library())
    rng = np.random.RandomState(seed)
    dl = float(rng.choice([1.0, 2.5]))
    nt = int(rng.randint(2, 5))
    lat, lon = synth.grid(dl)
    tm = synth.times(nt)
    pv = synth.pv_field(int(rng.randint(0, 3000)), nt, lat, lon, eps=float(rng.choice([0.1, 0.3])), seed=seed)
    data = wb.GridField(pv, ("time", "lat", "lon"), {"time": tm, "lat": wb.gridfield.Coord(lat, {"units": "degrees_north"}),
                                                       "lon": wb.gridfield.Coord(lon, {"units": "degrees_east"})}, name="PV")
    try:
        ev = pd.concat([wb.calculate_streamers(data, [-2, 2]), wb.calculate_cutoffs(data, [-2, 2]),
                        wb.calculate_overturnings(data, [-2, 2])]).reset_index(drop=True)
        if len(ev) == 0:
            return (seed, "ok")
        ev["event_id"] = np.arange(1, len(ev) + 1, dtype=float)
        steps = [int(np.nonzero(tm == np.datetime64(d))[0][0]) for d in ev["date"]]
        rings = [_rings(g) for g in ev["geometry"]]
        got = wb.to_xarray(data, ev)
        want = oflags.flag_events(nt, lat, lon, dl / 2, steps, rings)
        if not np.array_equal(got.values, want):
            return (seed, "FAIL ones: %d cells differ" % int((got.values != want).sum()))
        got = wb.to_xarray(data, ev, flag="event_id")
        want = oflags.flag_events(nt, lat, lon, dl / 2, steps, rings, ev["event_id"].values, dtype=data.values.dtype)
        if not np.array_equal(got.values, want):
            return (seed, "FAIL values: %d cells differ" % int((got.values != want).sum()))
        return (seed, "ok")
    except Exception as e:
        return (seed, "FAIL " + type(e).__name__ + " " + str(e)[:100])


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    import helpers
    if not os.environ.get("WB_HUNT_GPU"):
        helpers.build_hostsim()
    t = time.time()
    bad = 0
    with mp.get_context("spawn").Pool(int(os.environ.get("WB_HUNT_WORKERS", 6))) as pool:
        for seed, res in pool.imap_unordered(case, range(first, first + count)):
            if res != "ok":
                bad += 1
                print(seed, res, flush=True)
    print("flags hunt", first, count, "failures", bad, "%.0fs" % (time.time() - t), flush=True)
