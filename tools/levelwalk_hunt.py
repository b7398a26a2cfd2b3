#!/usr/bin/env python
"""Hunt over small rough fields with several cells EQUAL to the level (the generator of tests/golden/make_golden.py:
levelhit_fields, more hits): the contour table of a one-step batch must equal contour_index.py:103-156 applied to the
oracle's restatement of skimage's sequential walk (oracle/marching.py:join_by_coordinates, pinned by
tests/golden/contours_levelhit.npz) -- also where two contour ends compete for a vertex (STATUS_LEVEL_ORDER: joined by
wb_k_trace_walk) -- and carry STATUS_LEVEL_KEYERROR exactly where that walk raises KeyError.
    python tools/levelwalk_hunt.py <first seed> <count>        (WB_HUNT_GPU=1: the HIP build)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def field(seed):
    rng = np.random.RandomState(seed)
    shape = (int(rng.randint(6, 26)), int(rng.randint(6, 34)))
    a = rng.standard_normal(shape)
    for _ in range(int(rng.randint(0, 3))):
        a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1)) / 5.0
    a = (a / a.std()).astype(np.float32).astype(np.float64)
    hi = a > 0.0
    m = np.zeros_like(hi)
    m[:-1] |= hi[:-1] != hi[1:]
    m[1:] |= hi[:-1] != hi[1:]
    m[:, :-1] |= hi[:, :-1] != hi[:, 1:]
    m[:, 1:] |= hi[:, :-1] != hi[:, 1:]
    c = np.argwhere(m)
    for _ in range(int(rng.randint(2, 14))):
        if len(c) == 0:
            break
        j, i = c[rng.randint(len(c))]
        a[j, i] = 0.0
        if rng.rand() < 0.5:
            a[min(max(j + rng.randint(-1, 2), 0), shape[0] - 1), min(max(i + rng.randint(-1, 2), 0), shape[1] - 1)] = 0.0
    if rng.rand() < 0.15:
        a[rng.randint(shape[0]), rng.randint(shape[1])] = np.nan
    return a


def check(ctx, a, level=0.0):
    """-> "ok" / "walked" / "keyerror" (all agreeing with the oracle); raises AssertionError on a difference"""
    from oracle import contours as ocont
    from oracle import marching
    from wavebreaking_amd import _hip
    nlat, nlon = a.shape
    lat, lon = np.arange(nlat, dtype=np.float64), np.arange(nlon, dtype=np.float64)
    b = ctx.contours(a[None], lat, lon, [level], 1.0, 1.0, periodic_add=0.0)
    got, st = b.contour_table(), int(b.status()[0])
    b.close()
    try:
        raw = marching.find_contours(a, level)
    except KeyError:
        assert st & _hip.STATUS_LEVEL_KEYERROR, ("the walk raises KeyError, status", st)
        return "keyerror"
    assert not (st & _hip.STATUS_LEVEL_KEYERROR), ("status KEYERROR, the walk returns contours", st)
    items, closed = [], []
    for item in raw:                                                      # contour_index.py:108-117
        xy, cl = ocont.round_dedupe(item)
        if len(xy) >= 4:
            items.append(xy)
            closed.append(cl)
    drop = ocont.check_duplicates(items, nlon)                            # :119-147
    keep = [i for i in range(len(items)) if i not in drop]
    assert got["step"].size == len(keep), (got["step"].size, len(keep))
    for k, i in enumerate(keep):
        o, n = int(got["offset"][k]), int(got["npts"][k])
        assert np.array_equal(np.c_[got["x"][o:o + n], got["y"][o:o + n]], items[i]), k
        assert bool(got["closed"][k]) == bool(closed[i]), k
    if st & _hip.STATUS_LEVEL_ORDER:
        assert st & _hip.STATUS_LEVEL_WALKED
        return "walked"
    return "ok"


if __name__ == "__main__":
    from tests import helpers
    from wavebreaking_amd import _hip
    first, count = int(sys.argv[1]), int(sys.argv[2])
    ctx = _hip.Context(0) if os.environ.get("WB_HUNT_GPU") else _hip.Context(0, helpers.hostsim_library())
    tally = {}
    for seed in range(first, first + count):
        try:
            r = check(ctx, field(seed))
        except AssertionError as e:
            r = "FAIL"
            print(seed, "FAIL", str(e)[:200], flush=True)
        tally[r] = tally.get(r, 0) + 1
    print("levelwalk hunt", first, count, tally, flush=True)
