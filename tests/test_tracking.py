"""track_events (processing/events.py:113-241): the oracle against the pandas / scikit-learn goldens and against an
independent exact area formula and hand-derived truth tables; the kernels (CPU: host-simulation build, -m gpu: the HIP
library) against the oracle; the public function the way the reference's own test calls it."""
import os
from fractions import Fraction

import numpy as np
import pandas as pd
import pytest

import wavebreaking_amd as wb
from oracle import tracking as otrack
from tests import helpers
from wavebreaking_amd import _hip, engine
from wavebreaking_amd import gridfield as gf
from wavebreaking_amd.processing import events as pev

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "track_pinned.npz")


# ------------------------------------------------------------------------------------------------ helpers
def _signed_area2(r):
    r = np.asarray(r, dtype=np.int64)
    x, y = r[:, 0], r[:, 1]
    return int(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))


def _edge_pair_area(rings_a, rings_b):
    """area(A n B), exact, by a formula that shares nothing with the row decomposition: a simple positively oriented
    ring is the signed sum of the regions between its edges and the line x = 0, so the area of the intersection is the
    double sum over edge pairs of sign_e * sign_f * integral of min(x_e(y), x_f(y)) over their common y-range (all x >= 0)."""
    def edges(rings):
        out = []
        for r in rings:
            r = [(int(x), int(y)) for x, y in r]
            if r[0] == r[-1]:
                r = r[:-1]
            if _signed_area2(r) < 0:
                r = r[::-1]                                   # counter-clockwise: the inside is to the left of every edge
            for (x0, y0), (x1, y1) in zip(r, r[1:] + r[:1]):
                if y0 != y1:
                    out.append((x0, y0, x1, y1))
        return out

    def x_at(e, y):
        x0, y0, x1, y1 = e
        return x0 + Fraction((x1 - x0) * 1, y1 - y0) * (y - y0)

    total = Fraction(0)
    for e in edges(rings_a):
        for f in edges(rings_b):
            lo, hi = max(min(e[1], e[3]), min(f[1], f[3])), min(max(e[1], e[3]), max(f[1], f[3]))
            if lo >= hi:
                continue
            # area = closed integral of x dy: an edge that goes up is a right-hand boundary (+), one that goes down a left-hand one (-)
            se = 1 if e[3] > e[1] else -1
            sf = 1 if f[3] > f[1] else -1
            cuts = [Fraction(lo), Fraction(hi)]
            da, db = x_at(e, lo) - x_at(f, lo), x_at(e, hi) - x_at(f, hi)
            if (da > 0 and db < 0) or (da < 0 and db > 0):
                cuts.insert(1, lo + (hi - lo) * da / (da - db))
            part = Fraction(0)
            for y0, y1 in zip(cuts[:-1], cuts[1:]):
                m0 = min(x_at(e, y0), x_at(f, y0))
                m1 = min(x_at(e, y1), x_at(f, y1))
                part += (y1 - y0) * (m0 + m1) / 2
            total += se * sf * part
    return total


def _star(rng, cx, cy, n, rmin, rmax):
    """a star-shaped (hence simple) lattice polygon around (cx, cy)"""
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    rad = rng.uniform(rmin, rmax, n)
    pts = np.c_[np.rint(cx + rad * np.cos(ang)), np.rint(cy + rad * np.sin(ang))].astype(int)
    keep = [0]
    for i in range(1, len(pts)):
        if (pts[i] != pts[keep[-1]]).any():
            keep.append(i)
    if len(keep) > 1 and (pts[keep[-1]] == pts[keep[0]]).all():
        keep.pop()
    return pts[keep]


def _is_simple(ring):
    from oracle import geom
    r = [tuple(int(v) for v in p) for p in ring]
    n = len(r)
    if n < 3 or len(set(r)) < n:
        return False
    for i in range(n):
        for j in range(i + 1, n):
            if j == i + 1 or (i == 0 and j == n - 1):
                continue
            if geom.segments_intersect(r[i], r[(i + 1) % n], r[j], r[(j + 1) % n]):
                return False
    # neighbouring edges must not fold back onto each other
    for i in range(n):
        a, b, c = r[i - 1], r[i], r[(i + 1) % n]
        if geom.orient(*a, *b, *c) == 0 and (b[0] - a[0]) * (c[0] - b[0]) + (b[1] - a[1]) * (c[1] - b[1]) < 0:
            return False
    return True


def _random_polygons(seed, count, spread=40, rmax=18):
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < count:
        ring = _star(rng, rng.randint(25, 25 + spread), rng.randint(25, 25 + spread), rng.randint(4, 14), 3, rmax)
        if _is_simple(ring) and _signed_area2(ring) != 0:
            out.append([ring])
    return out


SQ = lambda x0, y0, x1, y1: np.array([(x0, y0), (x1, y0), (x1, y1), (x0, y1)])      # noqa: E731

# hand-derived truth table: (rings A, rings B, area of the intersection)
TRUTH = [
    ([SQ(0, 0, 10, 10)], [SQ(0, 0, 10, 10)], Fraction(100)),                 # identical (the reference's own test)
    ([SQ(0, 0, 10, 10)], [SQ(5, 5, 15, 15)], Fraction(25)),                  # corner overlap
    ([SQ(0, 0, 10, 10)], [SQ(10, 0, 20, 10)], Fraction(0)),                  # share an edge
    ([SQ(0, 0, 10, 10)], [SQ(10, 10, 20, 20)], Fraction(0)),                 # share a corner
    ([SQ(0, 0, 10, 10)], [SQ(2, 3, 5, 9)], Fraction(18)),                    # nested
    ([SQ(0, 0, 10, 10)], [SQ(11, 0, 20, 10)], Fraction(0)),                  # apart
    ([SQ(0, 0, 10, 10)], [SQ(0, 10, 10, 12)], Fraction(0)),                  # on top, share a horizontal edge
    ([np.array([(0, 0), (10, 0), (0, 10)])], [np.array([(10, 10), (0, 10), (10, 0)])], Fraction(0)),   # share the diagonal
    ([np.array([(0, 0), (10, 0), (0, 10)])], [np.array([(10, 10), (0, 10), (2, 6), (10, 0)])], Fraction(10)),          # B dips below the diagonal: triangle (0,10) (2,6) (10,0)
    ([np.array([(0, 0), (4, 2), (0, 4)])], [np.array([(2, 1), (6, 3), (2, 5)])], None),  # collinear edge pieces overlap partly
    ([np.array([(0, 0), (7, 3), (0, 3)])], [np.array([(0, 3), (7, 0), (7, 3)])], None),  # edges cross inside a row
    ([SQ(0, 0, 4, 4), SQ(10, 0, 14, 4)], [SQ(2, 2, 12, 6)], Fraction(8)),    # a two-part polygon
    ([SQ(0, 0, 10, 10), SQ(3, 3, 7, 7)], [SQ(4, 4, 6, 6)], Fraction(0)),     # B sits in the hole of A
    ([SQ(0, 0, 10, 10), SQ(3, 3, 7, 7)], [SQ(2, 2, 8, 8)], Fraction(20)),    # B covers the hole and a rim of A
]


def _truth_cases():
    cases = []
    for a, b, want in TRUTH:
        if want is None:
            want = _edge_pair_area(a, b)
        cases.append((a, b, want))
    return cases


# ------------------------------------------------------------------------------------------------ oracle: pinned parts
def test_oracle_pairs_distance_and_order_against_pandas_sklearn():
    g = np.load(GOLD)
    assert int(g["ncases"]) == 4
    for c in range(int(g["ncases"])):
        dates = g["c%d_dates" % c].astype("datetime64[ns]")
        arg = float(g["c%d_time_range_arg" % c])
        tr = otrack.default_time_range(dates) if np.isnan(arg) else arg
        assert tr == float(g["c%d_time_range" % c])
        pairs = otrack.time_pairs(dates, tr)
        assert np.array_equal(pairs, g["c%d_pairs" % c])
        com = np.c_[g["c%d_lon" % c], g["c%d_lat" % c]]
        d = np.array([otrack.haversine(com[i], com[j]) for i, j in pairs])
        assert np.array_equal(d, g["c%d_dist" % c])                                      # bit for bit
        assert np.array_equal(otrack.distance_check(com, pairs, float(g["c%d_distance" % c])), g["c%d_check" % c])
        assert g["c%d_check" % c].any() and not g["c%d_check" % c].all()
        assert np.array_equal(otrack.final_order(g["c%d_label_in" % c], dates), g["c%d_order" % c])
    d0 = otrack.haversine(g["edge_p1"], g["edge_p2"]) * 6371
    assert d0 == float(g["edge_d"])
    assert np.array_equal(d0 < g["edge_thr"], g["edge_check"])


def test_oracle_labels_follow_the_reference_rule():
    # events.py:224-238 by hand: rows 0..6, links 5-2, 2-6, 4-1 -> components {2,5,6} named 2, {1,4} named 1; names
    # 0,1,2,3 renumbered 0,1,2,3
    lab = otrack.labels_from_pairs(7, np.array([(5, 2), (2, 6), (4, 1)]))
    assert lab.tolist() == [0, 1, 2, 3, 1, 2, 2]
    assert otrack.labels_from_pairs(3, np.zeros((0, 2), dtype=np.int64)).tolist() == [0, 1, 2]


# ------------------------------------------------------------------------------------------------ oracle: overlap areas
def test_oracle_overlap_truth_table():
    for a, b, want in _truth_cases():
        assert otrack.overlap_area(a, b) == want
        assert otrack.overlap_area(b, a) == want


def test_oracle_overlap_against_the_edge_pair_formula():
    polys = _random_polygons(11, 16)
    nonzero = 0
    for i in range(len(polys)):
        assert otrack.rings_area(polys[i]) == Fraction(abs(_signed_area2(polys[i][0])), 2)
        for j in range(i + 1, len(polys)):
            got = otrack.overlap_area(polys[i], polys[j])
            assert got == _edge_pair_area(polys[i], polys[j])
            nonzero += got > 0
    assert nonzero >= 20


# ------------------------------------------------------------------------------------------------ kernels vs oracle
def _kernel_overlaps(ctx):
    cases = _truth_cases()
    rings, pairs = [], []
    for a, b, _ in cases:
        pairs.append((len(rings), len(rings) + 1))
        pairs.append((len(rings) + 1, len(rings)))
        rings += [a, b]
    inter, area = ctx.track_overlaps(rings, pairs)
    for q, (a, b, want) in enumerate(cases):
        for v in inter[2 * q:2 * q + 2]:
            assert (v == 0.0) if want == 0 else abs(v - float(want)) <= 1e-13 * float(want), (q, v, want)
        assert abs(area[2 * q] - float(otrack.rings_area(a))) <= 1e-13 * max(1.0, area[2 * q])
    polys = _random_polygons(23, 28)
    pairs = [(i, j) for i in range(len(polys)) for j in range(len(polys)) if i != j]
    inter, area = ctx.track_overlaps(polys, pairs)
    positive = 0
    for (i, j), v in zip(pairs, inter):
        want = otrack.overlap_area(polys[i], polys[j])
        if want == 0:
            assert v == 0.0
        else:
            positive += 1
            assert abs(v - float(want)) <= 1e-12 * float(want)
    assert positive >= 100
    for e, p in enumerate(polys):
        assert abs(area[e] - float(otrack.rings_area(p))) <= 1e-13 * area[e]


def _tracking_case(seed, n, steps, shuffle=True, numeric=False):
    rng = np.random.RandomState(seed)
    step = rng.randint(0, steps, n)
    if not shuffle:
        step = np.sort(step)
    if numeric:
        dates = step.astype(np.float64) * 6.0
    else:
        dates = (np.datetime64("1979-01-01T00", "ns") + step * np.timedelta64(6, "h")).astype("datetime64[ns]")
    polys = _random_polygons(seed + 1, n, spread=60, rmax=14)
    com = np.c_[rng.uniform(-180, 180, n), rng.uniform(-80, 80, n)]
    return dates, polys, com


def _kernel_tracking(ctx):
    for seed, n, steps, shuffle, numeric, tr in ((3, 40, 6, False, False, 6.0), (5, 48, 7, True, False, 12.0),
                                                 (7, 36, 5, True, True, 6.0)):
        dates, polys, com = _tracking_case(seed, n, steps, shuffle, numeric)
        for method, kw in (("by_overlap", dict(overlap=0.0)), ("by_overlap", dict(overlap=0.1)),
                           ("by_distance", dict(distance=4000.0))):
            want, order, pairs, check = otrack.track_events(dates, com=com, rings=polys, time_range=tr, method=method, **kw)
            if numeric:
                pairs_fwd = pairs[pairs[:, 0] < pairs[:, 1]]            # |difference|: both directions are in the reference's list
                assert len(pairs_fwd) * 2 == len(pairs)
            label, area, stats = ctx.track_events(dates, method, tr, com=com, rings_per_event=polys, **kw)
            assert np.array_equal(label, want), (seed, method, kw)
            assert stats["pairs_in_range"] == (len(pairs) // 2 if numeric else len(pairs))
            assert stats["pairs_linked"] == (int(check.sum()) // 2 if numeric else int(check.sum()))
            assert 0 < check.sum() < len(check)
            assert len(set(want.tolist())) < n                          # something was linked


def _kernel_thresholds(ctx):
    """the link tests at their thresholds: IoU exactly equal to `overlap` is NOT linked (events.py:213 `>`), touching
    polygons are not linked at overlap 0, every pair is linked at a negative overlap; haversine at the golden's edge"""
    dates = np.array(["2000-01-01T00", "2000-01-01T06"], dtype="datetime64[ns]")
    a, b = [SQ(0, 0, 10, 10)], [SQ(5, 0, 15, 10)]                        # IoU = 50 / 150
    third = 50.0 / 150.0
    # equality with the threshold is decided with a guard of 1e-13 relative (the trapezoid sum is rounded; lattice polygons
    # give ratios like 1/2 or 1/5 on the dot, tools/track_hunt.py): at the threshold, and within the guard of it, not linked
    for ov, linked in ((third, False), (np.nextafter(third, 0), False), (third * (1 - 1e-9), True), (third * (1 + 1e-9), False), (0.0, True)):
        label, _, _ = ctx.track_events(dates, "by_overlap", 6.0, overlap=ov, rings_per_event=[a, b])
        assert label.tolist() == ([0, 0] if linked else [0, 1]), ov
    touch = [SQ(10, 0, 20, 10)]
    assert ctx.track_events(dates, "by_overlap", 6.0, overlap=0.0, rings_per_event=[a, touch])[0].tolist() == [0, 1]
    assert ctx.track_events(dates, "by_overlap", 6.0, overlap=-1.0, rings_per_event=[a, touch])[0].tolist() == [0, 0]
    g = np.load(GOLD)
    com = np.array([g["edge_p1"], g["edge_p2"]])
    d0 = float(g["edge_d"])
    for thr, linked in ((d0 * (1 + 1e-9), True), (d0 * (1 - 1e-9), False)):
        label, _, _ = ctx.track_events(dates, "by_distance", 6.0, distance=thr, com=com)
        assert label.tolist() == ([0, 0] if linked else [0, 1])


def _kernel_ratio_on_the_threshold(ctx):
    """intersection over union EXACTLY 1/2 and 1/5 (found by tools/track_hunt.py on the path's own polygons; here squares and
    an L-shape): not above a threshold of 0.5 / 0.2, above anything smaller"""
    dates = np.array(["2000-01-01T00", "2000-01-01T06"], dtype="datetime64[ns]")
    a, b = [SQ(0, 0, 3, 3)], [np.array([(0, 0), (3, 0), (3, 3), (2, 3), (2, 1), (0, 1)])]     # area 9 and 5, b inside a: 5 / 9
    c, d = [SQ(0, 0, 4, 2)], [SQ(0, 0, 2, 2)]                                                   # 4 / 8 = 1 / 2
    e, f = [SQ(0, 0, 5, 1)], [SQ(0, 0, 1, 1)]                                                   # 1 / 5
    for x, y, num, den in ((a, b, 5, 9), (c, d, 1, 2), (e, f, 1, 5)):
        exact = num / den
        assert ctx.track_events(dates, "by_overlap", 6.0, overlap=exact, rings_per_event=[x, y])[0].tolist() == [0, 1]
        assert ctx.track_events(dates, "by_overlap", 6.0, overlap=exact * (1 - 1e-9), rings_per_event=[x, y])[0].tolist() == [0, 0]


def _kernel_golden_distance(ctx):
    g = np.load(GOLD)
    for c in range(int(g["ncases"])):
        dates = g["c%d_dates" % c].astype("datetime64[ns]")
        com = np.c_[g["c%d_lon" % c], g["c%d_lat" % c]]
        tr, dist = float(g["c%d_time_range" % c]), float(g["c%d_distance" % c])
        d = g["c%d_dist" % c] * 6371
        assert np.abs(d - dist).min() > 1e-6 * dist                      # no pair within an ulp of the threshold
        want = otrack.labels_from_pairs(len(dates), g["c%d_pairs" % c][g["c%d_check" % c]])
        label, _, stats = ctx.track_events(dates, "by_distance", tr, distance=dist, com=com)
        assert np.array_equal(label, want)
        assert stats["pairs_in_range"] == len(d) and stats["pairs_linked"] == int(g["c%d_check" % c].sum())


def _kernel_buffered(ctx):
    """track_events(buffer > 0, overlap = 0) (events.py:203-214): linked iff the polygons are closer than 2 * buffer degrees --
    labels against the exact-rational oracle on random catalogues (isotropic and anisotropic grid steps), and the decisions
    at the threshold: a gap of exactly 2 * buffer is NOT linked (the discs touch), anything closer is; one polygon inside the
    other's hole-free interior is linked; corner-to-corner distances"""
    for seed, n, steps, tr, buf, st in ((3, 40, 6, 6.0, 0.5, (0.25, 0.25)), (5, 48, 7, 12.0, 1.0, (0.25, 0.25)), (7, 36, 5, 6.0, 0.9, (1.0, 0.5)),
                                        (11, 40, 4, 6.0, 2.0, (0.3, 0.3))):
        dates, polys, com = _tracking_case(seed, n, steps, shuffle=True)
        want, order, pairs, check = otrack.track_events(dates, rings=polys, time_range=tr, method="by_overlap", buffer=buf, steps=st)
        plain = otrack.track_events(dates, rings=polys, time_range=tr, method="by_overlap", overlap=0)[3]
        label, stats = ctx.track_events_buffered(dates, tr, buf, st, rings_per_event=polys)
        assert np.array_equal(label, want), (seed, buf)
        assert stats["pairs_in_range"] == len(pairs) and stats["pairs_linked"] == int(check.sum())
        assert int(check.sum()) > int(plain.sum()) > 0                    # the buffer links pairs that do not overlap
        assert stats["pairs_linked_by_distance"] == int((check & ~plain).sum())
    dates = np.array(["2000-01-01T00", "2000-01-01T06"], dtype="datetime64[ns]")
    a = [SQ(0, 0, 10, 10)]
    for b, buf, st, linked in (([SQ(12, 0, 20, 10)], 0.25, (0.25, 0.25), False),      # gap 2 steps = 0.5 degrees = 2 * buffer: touching discs
                               ([SQ(12, 0, 20, 10)], 0.2500001, (0.25, 0.25), True),
                               ([SQ(11, 0, 20, 10)], 0.25, (0.25, 0.25), True),
                               ([SQ(13, 13, 20, 20)], 0.5, (0.25, 0.25), False),       # corner to corner: 3 * sqrt(2) * 0.25 = 1.06 > 1
                               ([SQ(13, 13, 20, 20)], 0.54, (0.25, 0.25), True),
                               ([SQ(2, 2, 4, 4)], 0.001, (0.25, 0.25), True),           # inside
                               ([SQ(10, 0, 20, 10)], 0.001, (0.25, 0.25), True),        # shared edge: distance 0
                               ([SQ(12, 0, 20, 10)], 0.6, (0.5, 2.0), False),           # anisotropic: 2 steps in x = 1.0 degree < 1.2 ...
                               ([SQ(0, 12, 10, 20)], 0.6, (0.5, 2.0), False)):          # ... 2 steps in y = 4.0 degrees
        if b[0][0][0] == 12 and st == (0.5, 2.0):
            linked = True                                                              # (1.0 < 1.2)
        label, stats = ctx.track_events_buffered(dates, 6.0, buf, st, rings_per_event=[a, b])
        want, _, _, check = otrack.track_events(dates, rings=[a, b], time_range=6.0, method="by_overlap", buffer=buf, steps=st)
        assert label.tolist() == want.tolist() == ([0, 0] if linked else [0, 1]), (b, buf, st)
    # the arc band: a corner-to-corner pair just inside the threshold is counted
    b = [SQ(13, 13, 20, 20)]
    d = 3 * 2 ** 0.5 * 0.25
    label, stats = ctx.track_events_buffered(dates, 6.0, d / 2 * 1.001, (0.25, 0.25), rings_per_event=[a, b])
    assert label.tolist() == [0, 0] and stats["pairs_in_arc_band"] == 1
    label, stats = ctx.track_events_buffered(dates, 6.0, d / 2 * 1.01, (0.25, 0.25), rings_per_event=[a, b])
    assert label.tolist() == [0, 0] and stats["pairs_in_arc_band"] == 0


@pytest.fixture
def hostsim_ctx():
    ctx = _hip.Context(0, helpers.hostsim_library())
    yield ctx
    ctx.close()


@pytest.mark.parametrize("reverse", [0, 1])
def test_track_kernels_hostsim(hostsim_ctx, reverse):
    helpers.hostsim_library().lib.wb_hostsim_set_reverse(reverse)
    try:
        _kernel_overlaps(hostsim_ctx)
        _kernel_tracking(hostsim_ctx)
        _kernel_thresholds(hostsim_ctx)
        _kernel_ratio_on_the_threshold(hostsim_ctx)
        _kernel_golden_distance(hostsim_ctx)
        _kernel_buffered(hostsim_ctx)
    finally:
        helpers.hostsim_library().lib.wb_hostsim_set_reverse(0)


@pytest.mark.gpu
def test_track_kernels_gpu():
    ctx = _hip.Context(0)
    try:
        _kernel_overlaps(ctx)
        _kernel_tracking(ctx)
        _kernel_thresholds(ctx)
        _kernel_ratio_on_the_threshold(ctx)
        _kernel_golden_distance(ctx)
        _kernel_buffered(ctx)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_track_many_events_gpu():
    """a catalogue of a few thousand events: labels against a union-find over the oracle's pairs, whose link tests are
    evaluated with the library's own overlap list (validated above) -- a size the exact oracle cannot finish"""
    ctx = _hip.Context(0)
    try:
        n, steps = 6000, 400
        dates, polys, com = _tracking_case(101, n, steps, shuffle=True)
        label, area, stats = ctx.track_events(dates, "by_overlap", 6.0, overlap=0.05, rings_per_event=polys)
        ns = dates.astype(np.int64)
        order = np.argsort(ns, kind="stable")
        pairs = []
        by_step = {}
        for i in order:
            by_step.setdefault(int(ns[i]), []).append(int(i))
        for t, rows in by_step.items():
            for j in by_step.get(t + 6 * 3600 * 10 ** 9, []):
                pairs += [(i, j) for i in rows]
        pairs = np.asarray(pairs, dtype=np.int64)
        assert stats["pairs_in_range"] == len(pairs)
        inter, area2 = ctx.track_overlaps(polys, pairs)
        assert np.array_equal(area, area2)
        linked = inter / (area[pairs[:, 0]] + area[pairs[:, 1]] - inter) > 0.05
        assert stats["pairs_linked"] == int(linked.sum())
        parent = np.arange(n)

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x

        for i, j in pairs[linked]:
            a, b = find(i), find(j)
            if a != b:
                parent[max(a, b)] = min(a, b)
        roots = np.array([find(i) for i in range(n)])
        want = np.searchsorted(np.unique(roots), roots)
        assert np.array_equal(label, want)
    finally:
        ctx.close()


# ------------------------------------------------------------------------------------------------ the public function
def _frame(dates, polys_lonlat, com=None):
    cols = {"date": dates}
    if com is not None:
        cols["com"] = [tuple(c) for c in com]
    return gf.make_frame(cols, [gf.make_polygon(p) for p in polys_lonlat])


def _public_function():
    # the reference's own test (tests/test_wavebreaking.py:130-144)
    date1, date2 = np.datetime64("1959-06-03T12"), np.datetime64("1959-06-03T18")
    square = np.array([(0, 0), (10, 0), (10, 10), (0, 10)], dtype=float)
    events = _frame([date1, date2], [[square], [square]])
    tracked = wb.track_events(events=events, method="by_overlap")
    assert isinstance(tracked, pd.DataFrame)
    assert tracked.iloc[0].label == 0 and tracked.label.tolist() == [0, 0]
    # a catalogue on a 0.25 degree grid, shuffled rows, both methods, against the oracle
    dates, polys, com = _tracking_case(9, 30, 5, shuffle=True)
    lonlat = [[np.c_[-180 + 0.25 * r[:, 0], -90 + 0.25 * r[:, 1]] for r in rs] for rs in polys]
    events = _frame(dates, lonlat, com)
    for method, kw in (("by_overlap", dict(overlap=0.02)), ("by_distance", dict(distance=5000))):
        want, order, _, _ = otrack.track_events(dates, com=com, rings=polys, time_range=6.0, method=method, **kw)
        tracked = wb.track_events(events, method=method, **kw)               # time_range=None -> 6 h
        assert tracked.index.tolist() == order.tolist()
        assert tracked.label.tolist() == want[order].tolist()
        assert tracked.label.dtype == np.int64
        assert "label" not in events.columns                                 # the caller's frame is not touched
    # the grid the index functions attach is used instead of the inferred one
    events.attrs["wavebreaking_grid"] = dict(lon=(-180.0, 0.25), lat=(-90.0, 0.25))
    assert wb.track_events(events, overlap=0.02).label.tolist() == wb.track_events(_frame(dates, lonlat, com), overlap=0.02).label.tolist()
    # errors, worded as the reference words them (events.py:179-181, 219-222)
    with pytest.raises(ValueError, match="No events detected in the time range: 3"):
        wb.track_events(events, time_range=3)
    with pytest.raises(ValueError, match="'by_radius' not supported as method!"):
        wb.track_events(events, method="by_radius")
    with pytest.raises(TypeError):
        wb.track_events([1, 2, 3])
    with pytest.raises(ValueError, match="is empty"):
        wb.track_events(events.iloc[:0])
    # buffer > 0 with the default overlap: closer than 2 * buffer degrees (events.py:203-214); with overlap > 0: not built
    want, order, _, check = otrack.track_events(dates, rings=polys, time_range=6.0, method="by_overlap", buffer=0.75, steps=(0.25, 0.25))
    tracked = wb.track_events(events, buffer=0.75)
    assert tracked.index.tolist() == order.tolist() and tracked.label.tolist() == want[order].tolist()
    assert tracked.attrs["track_stats"]["pairs_linked"] == int(check.sum())
    with pytest.raises(NotImplementedError):
        wb.track_events(events, buffer=1.0, overlap=0.1)
    with pytest.raises(NotImplementedError):
        wb.track_events(events, buffer=-1.0)
    off_grid = _frame(dates[:2], [[np.array([(0, 0), (1, 0), (2 ** 0.5 / 3, 0.77)])], [np.array([(0, 0), (1, 0), (0.5, 3 ** 0.5 / 2)])]])
    with pytest.raises(ValueError, match="regular grid"):
        wb.track_events(off_grid, time_range=1000)


def test_track_events_public_hostsim(monkeypatch):
    ctx = _hip.Context(0, helpers.hostsim_library())
    monkeypatch.setitem(engine._contexts, 0, ctx)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    try:
        _public_function()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_track_events_public_gpu():
    engine._contexts.clear()
    _public_function()


def test_tracking_of_the_index_functions_own_polygons_hostsim(monkeypatch):
    # the polygons the path itself produces (long, thin, concave, seam-split), not synthetic stars: labels and order of
    # wb.track_events against the exact oracle, for each index, at overlap 0 and 0.1
    from oracle import synth
    ctx = _hip.Context(0, helpers.hostsim_library())
    monkeypatch.setitem(engine._contexts, 0, ctx)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    try:
        lat, lon = synth.grid(1.0)
        tm = synth.times(5)
        pv = synth.pv_field(7, 5, lat, lon, eps=0.3, seed=3)
        data = wb.GridField(pv, ("time", "lat", "lon"), {"time": tm, "lat": gf.Coord(lat, {"units": "degrees_north"}),
                                                           "lon": gf.Coord(lon, {"units": "degrees_east"})}, name="PV")
        for fn in (wb.calculate_streamers, wb.calculate_cutoffs, wb.calculate_overturnings):
            ev = fn(data, [-2, 2])
            assert "wavebreaking_grid" in ev.attrs and len(ev) > 20
            rings = pev.lattice_rings(ev)
            dates = np.asarray(ev["date"].values)
            for ov in (0.0, 0.1):
                want, order, pairs, check = otrack.track_events(dates, rings=rings, time_range=6.0, method="by_overlap", overlap=ov)
                got = wb.track_events(ev, time_range=6, method="by_overlap", overlap=ov)
                assert got.index.tolist() == order.tolist() and got.label.tolist() == want[order].tolist()
                assert 0 < check.sum() < len(check)
    finally:
        ctx.close()


def test_lattice_recovery():
    assert pev._float_gcd([10, 25]) == 5.0
    assert abs(pev._float_gcd([0.1 * 3, 0.1 * 7, 0.1 * 2]) - 0.1) < 1e-12
    assert pev._float_gcd([0.25, 0.5, 1.75]) == 0.25
