"""Hunt for calculate_smoothed_field / calculate_momentum_flux mismatches against the oracle (bit for bit) on random shapes,
dtypes, pass counts, kernels and boundary modes (development tool).   python tools/spatial_hunt.py <first> <count>   WB_HUNT_GPU=1"""
import os, sys, time
import multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def case(seed):
    import numpy as np
    import helpers
    from wavebreaking_amd import _hip
    from oracle import spatial as osp
    ctx = _hip.Context(0) if os.environ.get("WB_HUNT_GPU") else _hip.Context(0, helpers.hostsim_library())
    rng = np.random.default_rng(seed)
    try:
        nt, nlat, nlon = int(rng.integers(1, 4)), int(rng.integers(5, 70)), int(rng.integers(5, 300))
        dt = np.float32 if rng.random() < 0.6 else np.float64
        a = (rng.standard_normal((nt, nlat, nlon)) * 3).astype(dt)
        if rng.random() < 0.3:
            a[:, rng.integers(0, nlat), :] = np.nan
        passes = int(rng.integers(1, 11))
        mode = str(rng.choice(["wrap", "reflect", "mirror", "nearest", "constant"]))
        if rng.random() < 0.6:
            w = np.array([[0, 1, 0], [1, 2, 1], [0, 1, 0]])
            if mode != "wrap" and rng.random() < 0.5:
                mode = "wrap"
        else:
            kh, kw = int(rng.choice([1, 3, 5])), int(rng.choice([1, 3, 5, 2, 4]))
            w = rng.integers(0, 4, (kh, kw))
            if w.sum() == 0:
                w[0, 0] = 1
        got = ctx.smooth_field(a, passes, w, mode=mode)
        want = osp.smooth_field(a, passes, w, mode)
        if not np.array_equal(got, want, equal_nan=True):
            return (seed, "FAIL smooth %s passes %d mode %s w %s shape %s" % (dt.__name__, passes, mode, w.shape, a.shape))
        u = (rng.standard_normal((nt, nlat, nlon)) * 12 + 5).astype(dt)
        v = (rng.standard_normal((nt, nlat, nlon)) * 7).astype(dt)
        if rng.random() < 0.3:
            u[0, 0, : max(1, nlon // 3)] = np.nan
        if not np.array_equal(ctx.momentum_flux(u, v), osp.momentum_flux(u, v), equal_nan=True):
            return (seed, "FAIL mflux %s shape %s" % (dt.__name__, a.shape))
        return (seed, "ok")
    except Exception as e:
        return (seed, "FAIL " + type(e).__name__ + " " + str(e)[:100])
    finally:
        ctx.close()


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
    print("spatial hunt", first, count, "failures", bad, "%.0fs" % (time.time() - t), flush=True)
