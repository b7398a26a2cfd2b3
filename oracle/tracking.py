"""track_events restated on plain arrays (TEST INFRASTRUCTURE -- only tests/, smoke() and bench.py's cpu_baseline leg may
import this package).

Follows wavebreaking/processing/events.py:113-241.  What is pinned and what is not:
  * the time-range pairs (:155-181), the haversine distance of method "by_distance" (:187-200, sklearn's
    DistanceMetric "haversine", a third-party dependency: sklearn/metrics/_dist_metrics.pyx) and the final ordering
    (:241, pandas sort_values) are pinned against pandas + scikit-learn by tests/golden/track_pinned.npz;
  * method "by_overlap" (:202-217) calls GEOS (buffer / make_valid / intersection / area), which cannot be installed in
    the build container: **parity unpinned**.  The areas are restated EXACTLY (Python Fractions) for polygons whose
    vertices are lattice points -- every event polygon the reference produces is one, transform_polygons truncates the
    vertices to grid indices (index_utils.py:135) -- under the even-odd rule, which is the region of a valid polygon;
    tests/test_tracking.py checks the restatement against a second, structurally different exact formula and against
    hand-derived truth tables.
"""
import math
from fractions import Fraction

import numpy as np


# ------------------------------------------------------------------------------------------------------ time pairs
def hours_between(dates, i):
    """events.py:162-165: datetime64 -> hours as float64 (pandas: nanoseconds / 1e9 / 3600), anything else -> |difference|"""
    d = np.asarray(dates)
    if np.issubdtype(d.dtype, np.datetime64):
        ns = d.astype("datetime64[ns]").astype(np.int64)
        return (ns - ns[i]) / 1e9 / 3600
    return np.abs(d - d[i])


def default_time_range(dates):
    """events.py:151-153: the smallest positive difference of consecutive rows, in hours"""
    ns = np.asarray(dates).astype("datetime64[ns]").astype(np.int64)
    dif = np.diff(ns)
    return float(dif[dif > 0].min() / 1e9 / 3600)


def time_pairs(dates, time_range):
    """events.py:156-181: all (i, j) with 0 < date_j - date_i <= time_range (in hours), as a sorted (n, 2) array (the
    reference's order is that of a Python set -- nothing downstream depends on it)"""
    out = []
    for i in range(len(dates)):
        diffs = hours_between(dates, i)
        for j in np.nonzero((diffs > 0) & (diffs <= time_range))[0]:
            out.append((i, int(j)))
    return np.asarray(sorted(set(out)), dtype=np.int64).reshape(-1, 2)


# ------------------------------------------------------------------------------------------------------ by_distance
def haversine(p1, p2):
    """sklearn HaversineDistance.rdist/dist on radians (_dist_metrics.pyx: ``sin_0 = sin(0.5 * (x1[0] - x2[0]))``,
    ``sin_1 = sin(0.5 * (x1[1] - x2[1]))``, ``2 * asin(sqrt(sin_0**2 + cos(x1[0]) * cos(x2[0]) * sin_1**2))``).  The
    reference passes ``com = (lon, lat)`` (events.py:189-195), so the FIRST coordinate plays the latitude's role --
    restated as is."""
    a0, a1 = (float(v) for v in np.radians(np.asarray(p1, dtype=np.float64)))
    b0, b1 = (float(v) for v in np.radians(np.asarray(p2, dtype=np.float64)))
    s0 = math.sin(0.5 * (a0 - b0))                  # the C library's functions, as the Cython code calls them
    s1 = math.sin(0.5 * (a1 - b1))
    return 2 * math.asin(math.sqrt(s0 * s0 + math.cos(a0) * math.cos(b0) * s1 * s1))


def distance_check(com, pairs, distance):
    """events.py:189-200"""
    return np.asarray([haversine(com[i], com[j]) * 6371 < distance for i, j in pairs], dtype=bool)


# ------------------------------------------------------------------------------------------------------ by_overlap
def _row_pieces(rings, k):
    """the edges of the rings that cross the unit row k <= y <= k + 1, each as (x at y = k, x at y = k + 1) in exact
    Fractions, sorted left to right (edges of a valid polygon do not cross inside a row)"""
    out = []
    for r in rings:
        r = [(int(x), int(y)) for x, y in r]
        if r[0] != r[-1]:
            r = r + [r[0]]
        for (x0, y0), (x1, y1) in zip(r[:-1], r[1:]):
            lo, hi = min(y0, y1), max(y0, y1)
            if lo <= k < hi:
                fa = x0 + Fraction((x1 - x0) * (k - y0), y1 - y0)
                fb = x0 + Fraction((x1 - x0) * (k + 1 - y0), y1 - y0)
                out.append((fa, fb))
    out.sort(key=lambda p: (p[0] + p[1], p[0]))
    return out


def _spans(pieces):
    """even-odd rule: the inside of the row lies between pieces 0-1, 2-3, ..."""
    return [(pieces[i], pieces[i + 1]) for i in range(0, len(pieces) - 1, 2)]


def _span_overlap(sa, sb):
    """area of the intersection of two trapezoids of one unit row, exact: the integral over t in [0, 1] of
    max(0, min(rA, rB) - max(lA, lB)) -- the integrand is concave piecewise linear, so its positive part is ONE interval"""
    (la, ra), (lb, rb) = sa, sb

    def lin(p):
        return lambda t: p[0] + (p[1] - p[0]) * t

    fl = [lin(la), lin(lb)]
    fr = [lin(ra), lin(rb)]

    def g(t):
        return min(f(t) for f in fr) - max(f(t) for f in fl)

    cuts = {Fraction(0), Fraction(1)}
    lines = [(r, l) for r in (ra, rb) for l in (la, lb)]             # the four differences r_i - l_j
    for r, l in lines:
        p, q = r[0] - l[0], r[1] - l[1]
        if (p > 0) != (q > 0) and p != q:
            cuts.add(p / (p - q))                                    # zero of this difference
    for u, v in ((la, lb), (ra, rb)):                                # kinks of max(l) and of min(r)
        p, q = u[0] - v[0], u[1] - v[1]
        if p != q and 0 < p / (p - q) < 1:
            cuts.add(p / (p - q))
    cuts = sorted(c for c in cuts if 0 <= c <= 1)
    area = Fraction(0)
    for t0, t1 in zip(cuts[:-1], cuts[1:]):
        g0, g1 = g(t0), g(t1)
        if g0 >= 0 and g1 >= 0:                                      # linear between two cuts, sign constant
            area += (t1 - t0) * (g0 + g1) / 2
    return area


def rings_area(rings):
    """even-odd area of a set of rings, exact"""
    ys = [int(y) for r in rings for _, y in r]
    total = Fraction(0)
    if not ys:                                       # an empty geometry (transform_polygons can return one, index_utils.py:167-168)
        return total
    for k in range(min(ys), max(ys)):
        for l, r in _spans(_row_pieces(rings, k)):
            total += ((r[0] - l[0]) + (r[1] - l[1])) / 2
    return total


def overlap_area(rings_a, rings_b):
    """area of (region A) n (region B) for two sets of lattice rings, exact (a Fraction) -- what
    ``geom1.intersection(geom2).area`` (events.py:209-213) is for valid polygons, in index units"""
    ya = [int(y) for r in rings_a for _, y in r]
    yb = [int(y) for r in rings_b for _, y in r]
    total = Fraction(0)
    if not ya or not yb:
        return total
    for k in range(max(min(ya), min(yb)), min(max(ya), max(yb))):
        sa = _spans(_row_pieces(rings_a, k))
        sb = _spans(_row_pieces(rings_b, k))
        for a in sa:
            for b in sb:
                total += _span_overlap(a, b)
    return total


def overlap_check(rings, pairs, overlap):
    """events.py:209-214: intersection over union > overlap (areas converted to float64 the way the kernel's result is
    compared; the comparison with 0 is exact)"""
    out = np.zeros(len(pairs), dtype=bool)
    ratio = np.zeros(len(pairs))
    area = {}
    for n, (i, j) in enumerate(pairs):
        for e in (i, j):
            if e not in area:
                area[e] = rings_area(rings[e])
        inter = overlap_area(rings[i], rings[j])
        union = area[i] + area[j] - inter
        ratio[n] = float(inter / union) if union != 0 else np.nan
        out[n] = ratio[n] > overlap
    return out, ratio


# ------------------------------------------------------------------------------------------------------ by_overlap, buffer > 0
def _seg_seg_d2_exact(a, b, c, d):
    """squared distance of the closed segments [a, b] and [c, d] (points with Fraction coordinates), exact"""
    def orient(p, q, r):
        return (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0])

    def on(p, q, r):
        return min(p[0], q[0]) <= r[0] <= max(p[0], q[0]) and min(p[1], q[1]) <= r[1] <= max(p[1], q[1])
    o1, o2, o3, o4 = orient(a, b, c), orient(a, b, d), orient(c, d, a), orient(c, d, b)
    if o1 * o2 < 0 and o3 * o4 < 0:
        return Fraction(0)
    if (o1 == 0 and on(a, b, c)) or (o2 == 0 and on(a, b, d)) or (o3 == 0 and on(c, d, a)) or (o4 == 0 and on(c, d, b)):
        return Fraction(0)

    def pt_seg(p, q, r):
        ux, uy, wx, wy = r[0] - q[0], r[1] - q[1], p[0] - q[0], p[1] - q[1]
        L = ux * ux + uy * uy
        t = (wx * ux + wy * uy) / L if L else Fraction(0)
        t = min(max(t, Fraction(0)), Fraction(1))
        ex, ey = wx - t * ux, wy - t * uy
        return ex * ex + ey * ey
    return min(pt_seg(a, c, d), pt_seg(b, c, d), pt_seg(c, a, b), pt_seg(d, a, b))


def _edges(rings):
    out = []
    for r in rings:
        v = np.asarray(r, dtype=np.int64).reshape(-1, 2)
        if len(v):
            out.append(np.c_[v, np.roll(v, -1, axis=0)])
    return np.concatenate(out) if out else np.zeros((0, 4), dtype=np.int64)


def rings_distance2(rings_a, rings_b, sx=1, sy=1):
    """squared planar distance, in degrees^2, of two events given as lattice rings on a grid with sx / sy degrees per step:
    0 if the closed regions share a point (they overlap, touch, or one lies inside the other), else the smallest distance
    between their boundaries.  Exact: whether two edges meet is decided in integers for every pair of edges; the minimum
    over the others is taken in Fractions over the pairs whose float64 distance comes within 1e-9 of the smallest (the
    float evaluation is good to ~1e-13: the exact minimum is among them)."""
    if overlap_area(rings_a, rings_b) > 0:
        return Fraction(0)
    ea, eb = _edges(rings_a), _edges(rings_b)
    if not len(ea) or not len(eb):
        return None
    A = ea[:, None, :]
    B = eb[None, :, :]

    def orient(px, py, qx, qy, rx, ry):
        return (qx - px) * (ry - py) - (qy - py) * (rx - px)

    def on(px, py, qx, qy, rx, ry):
        return (np.minimum(px, qx) <= rx) & (rx <= np.maximum(px, qx)) & (np.minimum(py, qy) <= ry) & (ry <= np.maximum(py, qy))
    o1 = orient(A[..., 0], A[..., 1], A[..., 2], A[..., 3], B[..., 0], B[..., 1])
    o2 = orient(A[..., 0], A[..., 1], A[..., 2], A[..., 3], B[..., 2], B[..., 3])
    o3 = orient(B[..., 0], B[..., 1], B[..., 2], B[..., 3], A[..., 0], A[..., 1])
    o4 = orient(B[..., 0], B[..., 1], B[..., 2], B[..., 3], A[..., 2], A[..., 3])
    meet = ((o1 * o2 < 0) & (o3 * o4 < 0)) \
        | ((o1 == 0) & on(A[..., 0], A[..., 1], A[..., 2], A[..., 3], B[..., 0], B[..., 1])) \
        | ((o2 == 0) & on(A[..., 0], A[..., 1], A[..., 2], A[..., 3], B[..., 2], B[..., 3])) \
        | ((o3 == 0) & on(B[..., 0], B[..., 1], B[..., 2], B[..., 3], A[..., 0], A[..., 1])) \
        | ((o4 == 0) & on(B[..., 0], B[..., 1], B[..., 2], B[..., 3], A[..., 2], A[..., 3]))
    if meet.any():
        return Fraction(0)
    fa = ea.astype(np.float64) * np.array([sx, sy, sx, sy], dtype=np.float64)
    fb = eb.astype(np.float64) * np.array([sx, sy, sx, sy], dtype=np.float64)

    def pt_seg(px, py, qx, qy, rx, ry):
        ux, uy, wx, wy = rx - qx, ry - qy, px - qx, py - qy
        L = ux * ux + uy * uy
        t = np.clip(np.where(L > 0, (wx * ux + wy * uy) / np.where(L > 0, L, 1.0), 0.0), 0.0, 1.0)
        ex, ey = wx - t * ux, wy - t * uy
        return ex * ex + ey * ey
    FA, FB = fa[:, None, :], fb[None, :, :]
    d = np.minimum(np.minimum(pt_seg(FA[..., 0], FA[..., 1], FB[..., 0], FB[..., 1], FB[..., 2], FB[..., 3]),
                              pt_seg(FA[..., 2], FA[..., 3], FB[..., 0], FB[..., 1], FB[..., 2], FB[..., 3])),
                   np.minimum(pt_seg(FB[..., 0], FB[..., 1], FA[..., 0], FA[..., 1], FA[..., 2], FA[..., 3]),
                              pt_seg(FB[..., 2], FB[..., 3], FA[..., 0], FA[..., 1], FA[..., 2], FA[..., 3])))
    near = np.argwhere(d <= d.min() * (1 + 1e-9))
    fx, fy = Fraction(sx), Fraction(sy)
    best = None
    for i, j in near[:64]:
        a = (Fraction(int(ea[i, 0])) * fx, Fraction(int(ea[i, 1])) * fy)
        b = (Fraction(int(ea[i, 2])) * fx, Fraction(int(ea[i, 3])) * fy)
        c = (Fraction(int(eb[j, 0])) * fx, Fraction(int(eb[j, 1])) * fy)
        e = (Fraction(int(eb[j, 2])) * fx, Fraction(int(eb[j, 3])) * fy)
        d2 = _seg_seg_d2_exact(a, b, c, e)
        if best is None or d2 < best:
            best = d2
    return best


def buffered_overlap_check(rings, pairs, buffer, steps=(1, 1)):
    """events.py:203-214 with buffer > 0 and overlap == 0: the buffered geometries overlap at all.  GEOS' round buffer
    (geopandas: 16 segments per quadrant) taken as the disc of radius ``buffer``: linked iff the polygons' planar distance
    in degrees is BELOW 2 * buffer -- **parity unpinned** (no GEOS here); ``band`` marks the pairs within
    2 * (1 - cos(pi / 64)) of that distance, where GEOS' inscribed arcs could fall short of the disc."""
    reach = Fraction(2) * Fraction(buffer)
    lo = reach * (1 - Fraction(0.0024091))
    out = np.zeros(len(pairs), dtype=bool)
    band = np.zeros(len(pairs), dtype=bool)
    for n, (i, j) in enumerate(pairs):
        d2 = rings_distance2(rings[i], rings[j], steps[0], steps[1])
        out[n] = d2 is not None and d2 < reach * reach
        band[n] = out[n] and d2 >= lo * lo
    return out, band


# ------------------------------------------------------------------------------------------------------ labels
def combine_shared(pairs):
    """index_utils.py:187-214 on pairs: connected components, as sets"""
    groups = []
    for a, b in pairs:
        hit = [g for g in groups if a in g or b in g]
        merged = {int(a), int(b)}.union(*hit) if hit else {int(a), int(b)}
        groups = [g for g in groups if g not in hit] + [merged]
    return groups


def labels_from_pairs(n, linked_pairs):
    """events.py:224-238: every event starts with its own row number, a component takes the smallest row number of its
    members, then the labels are renumbered 0, 1, 2, ... in ascending order"""
    label = np.arange(n, dtype=np.int64)
    for g in combine_shared(linked_pairs):
        label[sorted(g)] = min(g)
    rank = {v: i for i, v in enumerate(sorted(set(label.tolist())))}
    return np.asarray([rank[v] for v in label.tolist()], dtype=np.int64)


def final_order(label, dates):
    """events.py:241: sort_values(by=["label", "date"]) -- pandas sorts several keys with a stable lexsort"""
    d = np.asarray(dates)
    if np.issubdtype(d.dtype, np.datetime64):
        d = d.astype("datetime64[ns]").astype(np.int64)
    return np.lexsort((d, np.asarray(label)))


def track_events(dates, com=None, rings=None, time_range=None, method="by_overlap", overlap=0, distance=1000, buffer=0, steps=(1, 1)):
    """events.py:113-241 -> (label per input row, row order of the returned frame, the pairs in range, which are linked)"""
    if time_range is None:
        time_range = default_time_range(dates)
    pairs = time_pairs(dates, time_range)
    if len(pairs) == 0:
        raise ValueError("No events detected in the time range: {}".format(time_range))
    if method == "by_distance":
        check = distance_check(com, pairs, distance)
    elif method == "by_overlap" and buffer > 0:
        assert overlap == 0, "buffer > 0 is restated for overlap == 0 only"
        check, _ = buffered_overlap_check(rings, pairs, buffer, steps)
    elif method == "by_overlap":
        check, _ = overlap_check(rings, pairs, overlap)
    else:
        raise ValueError("'{}' not supported as method! Supported methods are 'by_overlap' and 'by_distance'".format(method))
    label = labels_from_pairs(len(dates), pairs[check])
    return label, final_order(label, dates), pairs, check
