#!/usr/bin/env python
"""Generate the golden vectors that pin the oracle to the third-party libraries
the reference calls.  Fixtures are DATA (inputs + expected outputs); this script
is how they were made.  Nothing from /root/reference is imported or copied.

Run in the build container (the two interpreters hold different libraries):

    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden.py skimage   # scikit-image 0.18.3
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden.py levelhit  # scikit-image 0.18.3, cells equal to the level
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden.py levelwalk # scikit-image 0.18.3, order-dependent joins / KeyError
    python3 tests/golden/make_golden.py sklearn                             # scikit-learn 1.7.2
    python3 tests/golden/make_golden.py pandas                              # pandas 2.3.3
    python3 tests/golden/make_golden.py props                               # pandas 2.3.3 (groupby sums of calculate_properties)
    python3 tests/golden/make_golden.py track                               # pandas 2.3.3 + scikit-learn 1.7.2 (track_events)

Outputs: tests/golden/contours_cases.npz, contours_random.npz, haversine.npz,
pandas_rules.npz, props_pandas.npz, contours_levelhit.npz, track_pinned.npz (all < 200 KB).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _pack(list_of_arrays):
    """ragged list of (n,2) arrays -> (concatenated, offsets)."""
    off = np.zeros(len(list_of_arrays) + 1, dtype=np.int64)
    for i, a in enumerate(list_of_arrays):
        off[i + 1] = off[i] + len(a)
    cat = np.concatenate(list_of_arrays, axis=0) if list_of_arrays else np.zeros((0, 2))
    return cat, off


def _smooth(rng, shape, passes):
    a = rng.standard_normal(shape)
    for _ in range(passes):
        a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1)) / 5.0
    return a


def fields():
    """Named small fields covering the nasty cases of SURVEY.md section 7 step 1."""
    rng = np.random.RandomState(12345)
    out = {}
    # 1. random smoothed fields of several shapes (closed loops, edge-cut loops, saddles)
    for n, (shape, passes) in enumerate([((12, 17), 1), ((14, 23), 2), ((33, 40), 3), ((24, 64), 2),
                                         ((40, 96), 4), ((9, 9), 0), ((20, 31), 1), ((64, 48), 5)]):
        out["smooth%d" % n] = (_smooth(rng, shape, passes).astype(np.float32).astype(np.float64), 0.0)
    # 2. NaN border rows as calculate_smoothed_field leaves them (spatial.py:106-107) + NaN cells inside
    a = _smooth(rng, (30, 44), 2)
    a[[0, 1, -2, -1], :] = np.nan
    out["nanrows"] = (a, 0.05)
    a = _smooth(rng, (26, 37), 1)
    a[rng.randint(0, 26, 12), rng.randint(0, 37, 12)] = np.nan
    out["nancells"] = (a, -0.02)
    # 3. checkerboard: every interior square is a saddle (cases 6 / 9)
    yy, xx = np.mgrid[0:10, 0:13]
    out["checker"] = (((yy + xx) % 2).astype(np.float64) + 0.25 * rng.rand(10, 13), 0.6)
    # 4. periodic extension exactly as contour_index.py:94-100 does it (jet + cutoffs; pad = 1/3 of nlon)
    lat = np.linspace(-60, 60, 41)[:, None]
    lon = np.linspace(0, 360, 48, endpoint=False)[None, :]
    jet = lat + 14 * np.sin(np.radians(3 * lon)) + 9 * np.cos(np.radians(5 * lon + 40)) - 10
    bump = 25 * np.exp(-(((lat - 35) / 6.0) ** 2 + ((lon - 20) / 9.0) ** 2))
    hole = -25 * np.exp(-(((lat + 30) / 5.0) ** 2 + ((lon - 340) / 8.0) ** 2))
    base = (jet - bump - hole).astype(np.float32).astype(np.float64)
    out["periodic"] = (np.concatenate([base, base[:, :16]], axis=1), 2.0)
    # 5. float32-valued folded field with steep gradients (f32 -> f64 cast path)
    a = (8 * np.tanh(_smooth(rng, (36, 50), 2) * 6)).astype(np.float32).astype(np.float64)
    out["steep"] = (a, 2.0)
    # 6. 2x2 and thin shapes
    out["tiny"] = (np.array([[1.0, 4.0], [1.0, 1.0]]), 2.0)
    out["thin_row"] = (_smooth(rng, (2, 29), 0), 0.1)
    out["thin_col"] = (_smooth(rng, (31, 2), 0), -0.1)
    # 7. all below / all above (no contours)
    out["empty"] = (np.zeros((7, 8)), 1.0)
    return out


def make_skimage():
    from skimage.measure import find_contours
    from skimage.measure._find_contours_cy import _get_contour_segments
    import skimage

    # (a) the 16 binary squares: case table
    seg_from, seg_to, seg_case = [], [], []
    for case in range(16):
        a = np.array([[case & 1, (case >> 1) & 1], [(case >> 2) & 1, (case >> 3) & 1]], dtype=float)
        for f, t in _get_contour_segments(a, 0.5, False, mask=None):
            seg_case.append(case)
            seg_from.append(f)
            seg_to.append(t)
    np.savez_compressed(os.path.join(HERE, "contours_cases.npz"), case=np.array(seg_case),
                        p_from=np.array(seg_from), p_to=np.array(seg_to),
                        skimage_version=np.array(skimage.__version__))

    # (b) named fields: raw segment list + assembled contours
    store = {"skimage_version": np.array(skimage.__version__)}
    names = []
    for name, (a, level) in fields().items():
        names.append(name)
        segs = _get_contour_segments(a.astype(np.double), float(level), False, mask=None)
        sf = np.array([s[0] for s in segs]).reshape(-1, 2)
        st = np.array([s[1] for s in segs]).reshape(-1, 2)
        cont = find_contours(a, level)
        cat, off = _pack([np.asarray(c) for c in cont])
        store[name + "__field"] = a
        store[name + "__level"] = np.array(level)
        store[name + "__seg_from"] = sf
        store[name + "__seg_to"] = st
        store[name + "__pts"] = cat
        store[name + "__off"] = off
    store["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "contours_random.npz"), **store)
    print("skimage goldens written:", names)


def levelhit_fields():
    """Fields with cells EQUAL to the level next to the contour (contour_index.py:103 on packed / rounded PV): the
    crossing then sits on a grid vertex and skimage joins through a coordinate shared by up to eight segment ends."""
    rng = np.random.RandomState(2024)
    out = {}
    lat = np.linspace(-60, 60, 25)[:, None]
    lon = np.linspace(0, 360, 40, endpoint=False)[None, :]
    base = (lat / 10.0 + 1.4 * np.sin(np.radians(3 * lon)) + 0.9 * np.cos(np.radians(5 * lon + 40))).astype(np.float32).astype(np.float64)

    def near_contour(a, level):
        hi = a > level
        m = np.zeros_like(hi)
        m[:-1] |= hi[:-1] != hi[1:]
        m[1:] |= hi[:-1] != hi[1:]
        m[:, :-1] |= hi[:, :-1] != hi[:, 1:]
        m[:, 1:] |= hi[:, :-1] != hi[:, 1:]
        return np.argwhere(m)

    # hand-made: interior hit, hit on every image edge, a saddle through a hit, two adjacent hits, a hit next to a NaN
    a = base.copy(); j, i = near_contour(a, 2.0)[7]; a[j, i] = 2.0; out["interior"] = (a, 2.0)
    a = base.copy()
    for j, i in near_contour(a, 2.0):
        if i == 0 or i == a.shape[1] - 1:
            a[j, i] = 2.0
    out["on_left_right_edge"] = (a, 2.0)
    a = base.T.copy()
    for j, i in near_contour(a, 2.0):
        if j == 0 or j == a.shape[0] - 1:
            a[j, i] = 2.0
    out["on_top_bottom_edge"] = (a, 2.0)
    out["saddle"] = (np.array([[3.0, 1.0, 3.0, 1.0], [1.0, 2.0, 1.0, 3.0], [3.0, 1.0, 3.0, 1.0], [1.0, 3.0, 2.0, 3.0]]), 2.0)
    out["plus"] = (np.array([[1.0, 3.0, 1.0], [3.0, 2.0, 3.0], [1.0, 3.0, 1.0]]), 2.0)
    out["cross_low"] = (np.array([[3.0, 1.0, 3.0], [1.0, 2.0, 1.0], [3.0, 1.0, 3.0]]), 2.0)
    out["ring_of_hits"] = (np.array([[1, 1, 1, 1, 1], [1, 2, 2, 2, 1], [1, 2, 3, 2, 1], [1, 2, 2, 2, 1], [1, 1, 1, 1, 1]], dtype=float), 2.0)
    a = base.copy(); c = near_contour(a, 2.0); j, i = c[11]; a[j, i] = 2.0; a[min(j + 1, a.shape[0] - 1), i] = 2.0; out["adjacent_pair"] = (a, 2.0)
    a = base.copy(); j, i = near_contour(a, 2.0)[19]; a[j, i] = 2.0; a[max(j - 1, 0), max(i - 1, 0)] = np.nan; out["next_to_nan"] = (a, 2.0)
    # periodic pad exactly as contour_index.py:94-100 builds it: the hit column appears twice (x and x + nlon)
    a = base.copy(); c = near_contour(a, 2.0); c = c[c[:, 1] < 10]; j, i = c[3]; a[j, i] = 2.0
    out["periodic_seam"] = (np.concatenate([a, a[:, :13]], axis=1), 2.0)
    a = base.copy(); a[near_contour(a, 2.0)[5][0], 0] = 2.0; a[near_contour(a, 2.0)[9][0], a.shape[1] - 1] = 2.0
    out["periodic_seam_edge"] = (np.concatenate([a, a[:, :13]], axis=1), 2.0)
    # random: smooth fields with 1..6 hits (some adjacent), a few with NaN cells; keeps every case where skimage raises
    k = 0
    nerr = 0
    from skimage.measure import find_contours
    while k < 40:
        shape = (int(rng.randint(6, 22)), int(rng.randint(6, 28)))
        a = _smooth(rng, shape, int(rng.randint(0, 4)))
        a = (a / a.std()).astype(np.float32).astype(np.float64)
        c = near_contour(a, 0.0)
        for _ in range(int(rng.randint(1, 7))):
            if len(c) == 0:
                break
            j, i = c[rng.randint(len(c))]
            a[j, i] = 0.0
            if rng.rand() < 0.3:
                a[min(max(j + rng.randint(-1, 2), 0), shape[0] - 1), min(max(i + rng.randint(-1, 2), 0), shape[1] - 1)] = 0.0
        if rng.rand() < 0.2:
            a[rng.randint(shape[0]), rng.randint(shape[1])] = np.nan
        try:
            find_contours(a, 0.0)
            raises = False
        except KeyError:
            raises = True
        if raises:
            nerr += 1
        elif k >= 34 and nerr < 4:
            continue                                   # the last slots are reserved for KeyError cases
        out["random%02d" % k] = (a, 0.0)
        k += 1
    return out


def levelwalk_field(seed):
    """the generator of tools/levelwalk_hunt.py:field (numpy's legacy RandomState: the same stream in every numpy)"""
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


def make_levelwalk():
    """scikit-image's own find_contours on the fields of tests/golden/levelwalk_seeds.txt: the seeds (of 0 .. 1499) on which two
    contour ends compete for a grid vertex -- the order-dependent joins (48) and the KeyError cases (6) that
    tools/levelwalk_hunt.py found with the library's sequential walk and the oracle's restatement agreeing."""
    from skimage.measure import find_contours
    import skimage

    store = {"skimage_version": np.array(skimage.__version__)}
    seeds, kinds = [], []
    for line in open(os.path.join(HERE, "levelwalk_seeds.txt")):
        seed, kind = line.split()
        a = levelwalk_field(int(seed))
        name = "s%04d" % int(seed)
        store[name + "__field"] = a.astype(np.float32)
        store[name + "__level"] = np.array(0.0)
        try:
            cont = find_contours(a, 0.0)
            cat, off = _pack([np.asarray(c) for c in cont])
            store[name + "__raises"] = np.array(0)
        except KeyError:
            cat, off = np.zeros((0, 2)), np.zeros(1, dtype=np.int64)
            store[name + "__raises"] = np.array(1)
        assert (kind == "keyerror") == bool(store[name + "__raises"]), (seed, kind)
        store[name + "__pts"] = cat
        store[name + "__off"] = off
        seeds.append(int(seed))
        kinds.append(kind)
    store["seeds"] = np.array(seeds)
    store["kinds"] = np.array(kinds)
    store["names"] = np.array(["s%04d" % q for q in seeds])
    np.savez_compressed(os.path.join(HERE, "contours_levelwalk.npz"), **store)
    print("levelwalk goldens written:", len(seeds), "fields,", kinds.count("keyerror"), "KeyError")


def make_levelhit():
    from skimage.measure import find_contours
    import skimage

    store = {"skimage_version": np.array(skimage.__version__)}
    names, nraise = [], 0
    for name, (a, level) in levelhit_fields().items():
        names.append(name)
        assert np.array_equal(a.astype(np.float32).astype(np.float64), a, equal_nan=True)
        store[name + "__field"] = a.astype(np.float32)          # every value is float32-representable
        store[name + "__level"] = np.array(level)
        try:
            cont = find_contours(a, level)
            cat, off = _pack([np.asarray(c) for c in cont])
            store[name + "__raises"] = np.array(0)
        except KeyError:
            cat, off = np.zeros((0, 2)), np.zeros(1, dtype=np.int64)
            store[name + "__raises"] = np.array(1)
            nraise += 1
        store[name + "__pts"] = cat
        store[name + "__off"] = off
    store["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "contours_levelhit.npz"), **store)
    print("level-hit goldens written: %d fields, %d of them make skimage raise KeyError" % (len(names), nraise))


def make_sklearn():
    import sklearn
    from sklearn.metrics import DistanceMetric

    dist = DistanceMetric.get_metric("haversine")
    rng = np.random.RandomState(777)
    # lattice-like coordinates as the streamer index sees them (streamer_index.py:124-130): [lat, lon] deg
    lat = np.round(rng.uniform(-89, 89, 60) * 4) / 4
    lon = np.round(rng.uniform(-180, 179.75, 60) * 4) / 4
    coords = np.c_[lat, lon]
    geo = dist.pairwise(np.radians(coords)) * 6371
    # a contour-like path: consecutive neighbours one lattice step apart
    steps = rng.randint(-1, 2, size=(300, 2))
    steps[(steps == 0).all(1)] = (1, 0)
    path = np.cumsum(steps, axis=0) * 0.25 + np.array([40.0, -30.0])
    gm = dist.pairwise(np.radians(path)) * 6371
    on = np.insert(np.diagonal(gm, 1), 0, 0)                      # streamer_index.py:133
    on_mat = np.triu(np.tile(on, (len(on), 1)), k=1)             # :134
    cont = np.cumsum(on_mat, axis=1)                             # :135
    np.savez_compressed(os.path.join(HERE, "haversine.npz"), coords=coords, geo=geo, path=path,
                        path_on=on, path_geo_row0=gm[0], path_cont_rows=cont[[0, 7, 150]],
                        sklearn_version=np.array(sklearn.__version__))
    print("sklearn goldens written")


def make_pandas():
    import pandas as pd

    rng = np.random.RandomState(99)
    store = {"pandas_version": np.array(pd.__version__)}
    # (a) streamer check_duplicates rule (streamer_index.py:160-183) on synthetic base-point frames
    nlon = 36
    for n in range(6):
        m = 14
        x1 = rng.randint(0, 10, m)
        y1 = rng.randint(0, 4, m)
        x2 = x1 + rng.randint(1, 6, m)
        y2 = rng.randint(0, 4, m)
        # append periodic twins of a random subset (exactly one twin each)
        tw = rng.choice(m, 5, replace=False)
        df = pd.DataFrame({"x1": np.r_[x1, x1[tw] + nlon], "y1": np.r_[y1, y1[tw]],
                           "x2": np.r_[x2, x2[tw] + nlon], "y2": np.r_[y2, y2[tw]]})
        df = df.drop_duplicates().sort_values(["x1", "y1", "x2", "y2"]).reset_index(drop=True)
        # unique rows only (a frame of candidate pairs never repeats a pair)
        temp = pd.concat([df.x1 % nlon, df.y1, df.x2 % nlon, df.y2], axis=1)
        temp = temp[temp.duplicated(keep=False)]
        if len(temp) == 0:
            check = []
        else:
            groups = temp.groupby(list(temp)).apply(lambda x: tuple(x.index)).tolist()
            if len({len(g) for g in groups}) != 1:
                continue
            check = list(np.asarray(groups)[:, 1])
        keep = df.drop(check).index.values
        store["dup%d__frame" % n] = df.values
        store["dup%d__keep" % n] = keep
    store["dup_nlon"] = np.array(nlon)
    # (b) overturning labels (overturning_index.py:123-134)
    for n in range(6):
        x = np.sort(rng.randint(0, 60, 90))
        lons, counts = np.unique(x, return_counts=True)
        ot = pd.DataFrame({"lon": lons[counts >= 3]})
        for rg in (5.0, 2.0):
            ot2 = ot.copy()
            ot2["label"] = (ot2.diff() > rg / 1.0).cumsum()
            if len(ot2) == 0:
                mm = np.zeros((0, 2), dtype=int)
            else:
                g = ot2.groupby("label").agg(["min", "max"]).astype("int").reset_index(drop=True)
                mm = g.values
            store["ot%d_rg%d__x" % (n, int(rg))] = x
            store["ot%d_rg%d__minmax" % (n, int(rg))] = mm
    # (c) numpy rounding rules the contour index relies on (contour_index.py:111,155)
    v = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 3.4999999999999996, 2.5000000000000004, 1e-17, 7.5, 8.5])
    store["round_in"] = v
    store["round_out"] = np.round(v).astype("int")
    m = np.array([12.345, 0.125, 0.375, 100.005, 45.675, 33.33333333])
    store["round2_in"] = m
    store["round2_out"] = np.round(m, 2)
    np.savez_compressed(os.path.join(HERE, "pandas_rules.npz"), **store)
    print("pandas goldens written")


def make_props():
    """The arithmetic of calculate_properties (index_utils.py:58-111) on per-cell frames: one row per (event, cell) in
    the join's row order (grid points numbered y-major, x-minor, :37-41), the per-row products and
    ``groupby("id").agg(sum)`` evaluated by pandas itself.  Boxes with min_lon + max_lon even make x_com / areas an
    integer mathematically, which is where the summation order decides the truncated centre-of-mass index (:105-110)."""
    import pandas as pd

    rng = np.random.RandomState(4242)
    store = {"pandas_version": np.array(pd.__version__), "numpy_version": np.array(np.__version__)}
    ev_id, ev_x, ev_y, ev_val, ev_int = [], [], [], [], []
    boxes, grid_of = [], []
    grids = {}
    for gi, d in enumerate((1.0, 0.25)):
        lat = -90.0 + d * np.arange(int(round(180 / d)) + 1)
        area_cell = np.round(6371 * 2 * np.pi / (360 / ((d + d) / 2))) ** 2        # index_utils.py:58-61
        grids[gi] = np.cos(np.radians(lat)) * area_cell                              # :62
        store["weight_lat_%d" % gi] = grids[gi]
    eid = 0

    def add(gi, xs, ys, with_data):
        nonlocal eid
        n = len(xs)
        ev_id.append(np.full(n, eid)); ev_x.append(xs); ev_y.append(ys)
        ev_val.append((rng.standard_normal(n) * 3).astype(np.float32) if with_data else np.zeros(n, np.float32))
        ev_int.append((rng.standard_normal(n) * 40).astype(np.float32) if with_data else np.zeros(n, np.float32))
        grid_of.append(gi)
        eid += 1

    # (a) boxes, cells in (y, x) order as np.meshgrid(...).flatten() numbers them
    for k in range(40):
        gi = k % 2
        nlat = len(grids[gi])
        cols = int(rng.randint(5, 28 if gi == 0 else 110))
        x0 = int(rng.randint(0, 300 if gi == 0 else 1700))
        x1 = x0 + cols - 1
        if k % 4 != 3 and (x0 + x1) % 2 == 1:                                       # three of four: min_lon + max_lon even
            x1 += 1
        rows = int(rng.randint(3, 22 if gi == 0 else 60))
        y0 = int(rng.randint(nlat // 2 + 5, nlat - rows - 8)) if k % 3 else int(rng.randint(8, nlat // 2 - rows - 5))
        y1 = y0 + rows - 1
        gx, gy = np.meshgrid(np.arange(x0, x1 + 1), np.arange(y0, y1 + 1))
        boxes.append((eid, x0, y0, x1, y1))
        add(gi, gx.flatten(), gy.flatten(), with_data=(k < 12))
    # (b) ragged cell sets (streamer / cutoff like): random row spans
    for k in range(10):
        gi = k % 2
        nlat = len(grids[gi])
        rows = int(rng.randint(4, 30))
        y0 = int(rng.randint(10, nlat - rows - 10))
        xs, ys = [], []
        c = int(rng.randint(20, 200))
        for r in range(rows):
            a = c + int(rng.randint(-4, 5)); b = a + int(rng.randint(1, 40))
            xs.append(np.arange(a, b + 1)); ys.append(np.full(b - a + 1, y0 + r))
            if rng.rand() < 0.3:                                                     # a second span in the row
                a2 = b + int(rng.randint(2, 9)); b2 = a2 + int(rng.randint(0, 12))
                xs.append(np.arange(a2, b2 + 1)); ys.append(np.full(b2 - a2 + 1, y0 + r))
        add(gi, np.concatenate(xs), np.concatenate(ys), with_data=True)
    ids = np.concatenate(ev_id); X = np.concatenate(ev_x); Y = np.concatenate(ev_y)
    V = np.concatenate(ev_val); I = np.concatenate(ev_int)
    gof = np.asarray(grid_of)
    sums = np.zeros((eid, 5))
    for gi in grids:
        m = gof[ids] == gi
        merged = pd.DataFrame({"id": ids[m], "x": X[m], "y": Y[m]})
        merged["areas"] = grids[gi][merged.y]                                        # :63
        merged["mean_var"] = merged.areas * V[m].astype(np.float64)                  # :66-75 (float32 field * float64)
        merged["intensity"] = merged.areas * I[m].astype(np.float64)                 # :77-87
        merged["x_com"] = merged.x * merged.areas                                    # :91
        merged["y_com"] = merged.y * merged.areas                                    # :92
        agg = merged.groupby("id").agg({"areas": "sum", "mean_var": "sum", "intensity": "sum", "x_com": "sum",
                                        "y_com": "sum"})                             # :94-102
        sums[agg.index.values] = agg[["areas", "mean_var", "intensity", "x_com", "y_com"]].values
    com_ix = (sums[:, 3] / sums[:, 0]).astype("int")                                 # :105-110
    com_iy = (sums[:, 4] / sums[:, 0]).astype("int")
    store.update(ids=ids.astype(np.int32), x=X.astype(np.int16), y=Y.astype(np.int16), val=V, inten=I, grid_of=gof,
                 boxes=np.asarray(boxes), sums=sums, com_ix=com_ix, com_iy=com_iy)
    np.savez_compressed(os.path.join(HERE, "props_pandas.npz"), **store)
    r = sums[:, 3] / sums[:, 0]
    print("props goldens written: %d events, %d cells, %d with x_com/areas within 1e-9 of an integer"
          % (eid, len(ids), int(np.sum(np.abs(r - np.round(r)) < 1e-9))))


def make_track():
    """What track_events (events.py:113-241) gets from pandas and scikit-learn: the hours between rows and the time-range
    test (:151-167), the haversine distance between two centres of mass with the threshold test (:189-200), and the final
    sort_values(by=["label", "date"]) (:241).  Several cases: shuffled rows, repeated dates, an irregular time axis, a
    range that spans more than one step, distances within a few ulps of the threshold."""
    import pandas as pd
    import sklearn
    from sklearn.metrics import DistanceMetric

    dist = DistanceMetric.get_metric("haversine")
    rng = np.random.RandomState(4242)
    out = {}
    cases = [dict(n=40, steps=6, step_h=6, time_range=None, distance=1000.0, shuffle=False),
             dict(n=60, steps=8, step_h=6, time_range=12, distance=2500.0, shuffle=True),
             dict(n=50, steps=7, step_h=1.5, time_range=2.0, distance=800.0, shuffle=True),
             dict(n=30, steps=5, step_h=24, time_range=24, distance=4000.0, shuffle=False)]
    for c, case in enumerate(cases):
        n = case["n"]
        step = rng.randint(0, case["steps"], n)
        if not case["shuffle"]:
            step = np.sort(step)
        ns = (np.datetime64("1979-01-01T00", "ns").astype(np.int64)
              + (step * case["step_h"] * 3600e9).astype(np.int64)
              + (rng.randint(0, 3, n) == 0) * (c == 2) * np.int64(1800e9))           # case 2: an irregular axis
        dates = ns.astype("datetime64[ns]")
        lon = np.round(rng.uniform(-180, 179.75, n) * 4) / 4 + rng.uniform(-0.1, 0.1, n)
        lat = np.round(rng.uniform(-80, 80, n) * 4) / 4 + rng.uniform(-0.1, 0.1, n)
        events = pd.DataFrame({"date": dates, "com": list(zip(lon, lat))})
        if case["time_range"] is None:
            dif = events.date.diff()
            time_range = dif[dif > pd.Timedelta(0)].min().total_seconds() / 3600
        else:
            time_range = float(case["time_range"])
        pairs = set()
        for i in events.index:
            diffs = (events.date - events.date.iloc[i]).dt.total_seconds() / 3600
            pairs.update((i, int(j)) for j in events[(diffs > 0) & (diffs <= time_range)].index)
        pairs = np.asarray(sorted(pairs), dtype=np.int64).reshape(-1, 2)
        com1 = np.asarray(list(events.iloc[pairs[:, 0]].com))
        com2 = np.asarray(list(events.iloc[pairs[:, 1]].com))
        d = np.asarray([dist.pairwise(np.radians([p1, p2]))[0, 1] for p1, p2 in zip(com1, com2)])
        # labels of a fixed pseudo-random partition: only the SORT is pinned here
        label = rng.randint(0, max(2, n // 4), n)
        order = events.assign(label=label).sort_values(by=["label", "date"]).index.values
        out.update({"c%d_dates" % c: ns, "c%d_lon" % c: lon, "c%d_lat" % c: lat, "c%d_time_range" % c: np.float64(time_range),
                    "c%d_time_range_arg" % c: np.float64(np.nan if case["time_range"] is None else case["time_range"]),
                    "c%d_pairs" % c: pairs, "c%d_dist" % c: d, "c%d_distance" % c: np.float64(case["distance"]),
                    "c%d_check" % c: d * 6371 < case["distance"], "c%d_label_in" % c: label, "c%d_order" % c: order})
    # thresholds a few ulps either side of one pair's distance
    p1, p2 = np.array([12.25, 47.5]), np.array([19.75, 52.0])
    d0 = dist.pairwise(np.radians([p1, p2]))[0, 1] * 6371
    thr = np.array([np.nextafter(d0, 0), d0, np.nextafter(d0, np.inf), d0 * (1 - 1e-9), d0 * (1 + 1e-9)])
    out.update(edge_p1=p1, edge_p2=p2, edge_d=np.float64(d0), edge_thr=thr, edge_check=d0 < thr)
    out.update(ncases=np.int64(len(cases)), sklearn_version=np.array(sklearn.__version__), pandas_version=np.array(pd.__version__))
    np.savez_compressed(os.path.join(HERE, "track_pinned.npz"), **out)
    print("track goldens written")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else ""
    {"skimage": make_skimage, "sklearn": make_sklearn, "pandas": make_pandas, "props": make_props,
     "levelhit": make_levelhit, "levelwalk": make_levelwalk, "track": make_track}[which]()
