"""Hunt for track_events mismatches on the index functions' own polygons: random fields -> calculate_* -> wb.track_events against
the exact oracle (development tool).   python tools/track_hunt.py <first seed> <count>     WB_HUNT_GPU=1: the HIP library"""
import os, sys, time
import multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def case(seed):
    import numpy as np
    import helpers
    import wavebreaking_amd as wb
    from wavebreaking_amd import _hip, engine
    from wavebreaking_amd.processing import events as pev
    from oracle import synth, tracking as otrack
    engine._contexts[0] = _hip.Context(0) if os.environ.get("WB_HUNT_GPU") else _hip.Context(0, helpers.hostsim_library())
    rng = np.random.RandomState(seed)
    dl = float(rng.choice([1.0, 2.5]))
    nt = int(rng.randint(3, 7))
    lat, lon = synth.grid(dl)
    tm = synth.times(nt)
    pv = synth.pv_field(int(rng.randint(0, 3000)), nt, lat, lon, eps=float(rng.choice([0.1, 0.3])), seed=seed)
    data = wb.GridField(pv, ("time", "lat", "lon"), {"time": tm, "lat": wb.gridfield.Coord(lat, {"units": "degrees_north"}),
                                                       "lon": wb.gridfield.Coord(lon, {"units": "degrees_east"})}, name="PV")
    try:
        for fn in (wb.calculate_streamers, wb.calculate_cutoffs, wb.calculate_overturnings):
            ev = fn(data, [-2, 2])
            if len(ev) < 2 or len(set(ev["date"])) < 2:
                continue
            rings = pev.lattice_rings(ev)
            dates = np.asarray(ev["date"].values)
            # events on several dates but no two within the time range: the reference raises (events.py:179-181) -- so must both sides
            # (seed 8001169: two overturnings twelve hours apart)
            try:
                otrack.track_events(dates, rings=rings, time_range=6.0, method="by_overlap", overlap=0.0)
            except ValueError as e_o:
                try:
                    wb.track_events(ev, time_range=6, method="by_overlap", overlap=0.0)
                    return (seed, "FAIL " + fn.__name__ + ": the oracle raises (%s), the library does not" % e_o)
                except ValueError:
                    continue
            for ov in (0.0, float(rng.choice([0.05, 0.2, 0.5]))):
                want, order, pairs, check = otrack.track_events(dates, rings=rings, time_range=6.0, method="by_overlap", overlap=ov)
                got = wb.track_events(ev, time_range=6, method="by_overlap", overlap=ov)
                if got.index.tolist() != order.tolist() or got.label.tolist() != want[order].tolist():
                    return (seed, "FAIL " + fn.__name__ + " overlap %g" % ov)
            # buffer > 0 with the default overlap: closer than 2 * buffer degrees (the path's own, seam-split polygons)
            buf = float(rng.choice([0.5, 1.0, 2.5])) * dl
            want, order, pairs, check = otrack.track_events(dates, rings=rings, time_range=6.0, method="by_overlap", buffer=buf, steps=(dl, dl))
            got = wb.track_events(ev, time_range=6, method="by_overlap", buffer=buf)
            if got.index.tolist() != order.tolist() or got.label.tolist() != want[order].tolist():
                return (seed, "FAIL " + fn.__name__ + " buffer %g" % buf)
            com = np.asarray(list(ev["com"]), dtype=np.float64)
            want, order, pairs, check = otrack.track_events(dates, com=com, time_range=6.0, method="by_distance", distance=1500.0)
            got = wb.track_events(ev, time_range=6, method="by_distance", distance=1500.0)
            if got.index.tolist() != order.tolist() or got.label.tolist() != want[order].tolist():
                return (seed, "FAIL " + fn.__name__ + " by_distance")
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
    print("track hunt", first, count, "failures", bad, "%.0fs" % (time.time() - t), flush=True)
