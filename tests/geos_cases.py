"""Deterministic inputs for the GEOS-dependent rows (8f row 1): the cases the hardening tests
(tests/test_geos_hardening.py) and the GEOS pin kit (tests/golden/make_golden_geos.py) share.  numpy + the oracle's exact
integer predicates only -- importable where the HIP library is not (the kit runs on any machine with shapely)."""
import numpy as np

from oracle import geom


def dedupe(points):
    """np.round(...).astype(int) + dict.fromkeys: the reference's contour, contour_index.py:110-112"""
    return np.array(list(dict.fromkeys(map(tuple, points))), dtype=np.int64)


def walk(rng, n, revisit):
    """a lattice walk of unit and diagonal steps that doubles back on itself every now and then (so that the deduped contour
    has long steps OVER earlier vertices)"""
    steps = np.array([(1, 0), (1, 0), (1, 1), (1, -1), (0, 1), (0, -1), (-1, 0), (-1, 1), (-1, -1)])
    p = np.array([int(rng.randint(20, 40)), int(rng.randint(20, 40))])
    out = [p.copy()]
    k = 0
    while len(out) < n:
        if rng.rand() < revisit and len(out) > 6:
            # go back over the last few points (these revisits are dropped by the dedupe and leave a long step behind)
            m = int(rng.randint(2, 6))
            back = out[-2:-2 - m:-1]
            out.extend(q.copy() for q in back)
            out.extend(q.copy() for q in back[::-1][1:])
            p = out[-1].copy()
            # ... and beyond where we came from
            p = p + steps[int(rng.randint(0, 4))]
            out.append(p.copy())
            continue
        p = p + steps[int(rng.randint(0, len(steps)))]
        out.append(p.copy())
        k += 1
    return np.array(out)


def contours(rng, count):
    out = []
    while len(out) < count:
        c = dedupe(walk(rng, int(rng.randint(12, 60)), float(rng.choice([0.0, 0.1, 0.25]))))
        if len(c) >= 4 and c.min() >= 0:
            out.append(c)
    return out


def corner_cases():
    """hand-made: the contour's FIRST vertex lies on a later long segment, and chords run through it"""
    cases = []
    # first vertex (5, 5); the dedupe leaves the long step (4, 5) -> (7, 5)?? no: the line comes back OVER (5, 5), (6, 5)
    raw = [(5, 5), (6, 5), (7, 6), (7, 7), (6, 8), (4, 8), (3, 7), (3, 6), (4, 5), (5, 5), (6, 5), (7, 5), (8, 4), (6, 2), (5, 3), (5, 4), (5, 6), (5, 7)]
    c = dedupe(np.array(raw))
    cases.append(c)
    # last vertex on an earlier segment is impossible after the dedupe (it would have been dropped); first vertex on a LATER
    # vertical long step:
    raw = [(10, 10), (11, 10), (12, 11), (12, 12), (10, 13), (9, 12), (10, 11), (10, 10), (10, 9), (10, 8), (9, 7), (8, 9), (9, 10), (11, 10), (12, 10)]
    cases.append(dedupe(np.array(raw)))
    return cases


def touches_cases(seed, ncontours, per):
    rng = np.random.RandomState(seed)
    cs = corner_cases() + contours(rng, ncontours)
    for c in cs:
        n = len(c)
        # every chord through the first / last vertex's location is interesting: add all pairs collinear with them, + random pairs
        pairs = set()
        for _ in range(per):
            i, j = sorted(int(v) for v in rng.randint(0, n, 2))
            if i != j:
                pairs.add((i, j))
        for e in (0, n - 1):
            for i in range(n):
                for j in range(i + 1, n):
                    if i != e and j != e and geom.orient(c[i, 0], c[i, 1], c[j, 0], c[j, 1], c[e, 0], c[e, 1]) == 0 and len(pairs) < 4 * per:
                        pairs.add((i, j))
        ij = np.array(sorted(pairs), dtype=np.int64)
        yield c, ij


def intersect_cases(seed, n):
    rng = np.random.RandomState(seed)
    s = rng.randint(0, 9, size=(n, 8))
    # many degenerate configurations: shared end points, collinear, touching
    k = n // 4
    s[:k, 4:6] = s[:k, 0:2]                                                        # shared end point
    s[k:2 * k, 4:8] = s[k:2 * k, 0:4] + np.repeat(rng.randint(-2, 3, size=(k, 2)), 2, axis=1) * 0 + np.tile(rng.randint(-1, 2, size=(k, 1)), (1, 4)) * np.array([1, 0, 1, 0])   # shifted along x
    d = s[2 * k:3 * k, 2:4] - s[2 * k:3 * k, 0:2]
    s[2 * k:3 * k, 4:6] = s[2 * k:3 * k, 0:2] + d * rng.randint(-1, 3, size=(k, 1))     # third point collinear with the first segment
    keep = (s[:, 0:2] != s[:, 2:4]).any(axis=1) & (s[:, 4:6] != s[:, 6:8]).any(axis=1)  # no zero-length segments (a LineString needs two distinct points)
    return s[keep]




def star(rng, cx, cy, n, rmin, rmax):
    """a star-shaped lattice polygon around (cx, cy): vertices sorted by angle, consecutive repeats dropped"""
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


def is_simple(ring):
    """no two non-neighbouring edges meet, no repeated vertex, no edge folding back onto its predecessor (exact)"""
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
    for i in range(n):
        a, b, c = r[i - 1], r[i], r[(i + 1) % n]
        if geom.orient(*a, *b, *c) == 0 and (b[0] - a[0]) * (c[0] - b[0]) + (b[1] - a[1]) * (c[1] - b[1]) < 0:
            return False
    return True


def _area2(ring):
    r = np.asarray(ring, dtype=np.int64)
    return int(np.sum(r[:, 0] * np.roll(r[:, 1], -1) - np.roll(r[:, 0], -1) * r[:, 1]))


def simple_polygons(seed, count, lo=6, hi=40, rmin=1.5, rmax=12):
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < count:
        ring = star(rng, rng.randint(lo, hi), rng.randint(lo, hi), rng.randint(4, 14), rmin, rmax)
        if ring.min() >= 0 and is_simple(ring) and _area2(ring) != 0:
            out.append(ring)
    return out


def cell_cases(quick=False):
    """(polygon vertices (x, y) int, r, grid width, grid height): calculate_properties' membership rule, index_utils.py:37-51.
    r = 1/2 (1 degree) and 1/8 (0.25 degree): the hand table, the streamer / cutoff polygons and overturning boxes the oracle
    finds on small synthetic fields (contour stretches as the reference builds them, streamer_index.py:267-272,
    overturning_index.py:211, cutoff_index.py:96 -- closed by Polygon(), self-touching where the contour doubles back),
    random star polygons, thin slivers and zero-height boxes"""
    from tests.golden import geos_truth
    out = [(list(p), float(r), 12, 8) for p, r, _ in geos_truth.CELLS]
    from oracle import api as oapi
    from oracle import synth
    for dl, eps, nt in ((2.5, 0.2, 1 if quick else 3), (5.0, 0.3, 1 if quick else 2)):
        lat, lon = synth.grid(dl)
        pv = synth.pv_field(0, nt, lat, lon, eps=eps)
        rows = oapi.run_all(pv, [-2.0, 2.0], lat, lon, synth.times(nt))
        width = lon.size + int(120.0 / dl)
        for name in ("streamers", "overturnings", "cutoffs"):
            for r in rows[name][:: 3 if quick else 1]:
                out.append((np.asarray(r["_polygon"]).tolist(), 0.5, width, lat.size))
                out.append((np.asarray(r["_polygon"]).tolist(), 0.125, width, lat.size))
    for k, ring in enumerate(simple_polygons(5, 10 if quick else 60)):
        out.append((ring.tolist(), 0.5 if k % 2 else 0.125, 56, 56))
    rng = np.random.RandomState(6)
    for _ in range(4 if quick else 30):                               # slivers: a long edge passing lattice points closely
        x0, y0 = int(rng.randint(2, 10)), int(rng.randint(2, 10))
        dx, dy = int(rng.randint(5, 30)), int(rng.randint(1, 4))
        out.append(([(x0, y0), (x0 + dx, y0 + dy), (x0 + dx, y0 + dy + int(rng.randint(0, 3))), (x0, y0 + int(rng.randint(0, 2)))],
                    0.125 if rng.rand() < 0.5 else 0.5, 48, 24))
    return out


def split_cases(quick=False):
    """(index polygon, nlon, nlat): transform_polygons' seam split, index_utils.py:148-173"""
    hand = {
        "box_across": [(34, 2), (38, 2), (38, 4), (34, 4)],
        "box_right_of_seam": [(37, 2), (39, 2), (39, 4), (37, 4)],
        "inside_the_strip_only": [(35, 3), (36, 3), (36, 5), (35, 5)],
        "vertex_on_both_lines": [(33, 2), (35, 6), (36, 2), (38, 7), (34, 9)],
        "u_shape": [(30, 2), (40, 2), (40, 4), (33, 4), (33, 6), (40, 6), (40, 8), (30, 8)],
        "u_shape_mirrored": [(44, 2), (32, 2), (32, 4), (41, 4), (41, 6), (32, 6), (32, 8), (44, 8)],
        "comb": [(31, 1), (39, 1), (39, 3), (33, 3), (33, 5), (39, 5), (39, 7), (33, 7), (33, 9), (39, 9), (39, 11), (31, 11)],
        "edge_on_the_line": [(32, 2), (35, 2), (35, 6), (36, 8), (35, 10), (32, 10)],
        "slanted": [(30, 1), (41, 4), (38, 12), (33, 9)],
    }
    out = [(v, 36, 19) for v in hand.values()]
    rng = np.random.RandomState(32)
    want = 15 if quick else 150
    while len(out) < want:
        ring = star(rng, 35.5 + rng.uniform(-2, 2), 9.0 + rng.uniform(-2, 2), int(rng.randint(4, 14)), 1.5, 8.0)
        if ring[:, 1].min() < 0 or ring[:, 1].max() > 18 or ring[:, 0].min() < 0 or not is_simple(ring) or _area2(ring) == 0:
            continue
        if ring[:, 0].max() < 36:
            continue                                                     # (only polygons that reach x >= nlon are split, :136)
        out.append(([tuple(int(v) for v in p) for p in ring], 36, 19))
    return out


def iou_cases(quick=False):
    """(ring set A, ring set B) in index coordinates (a regular grid scales every area alike, so the ratio of events.py:209-213
    is the same in lon / lat): squares of the hand table, random simple polygons that overlap partly / not at all / nest,
    and two-part ring sets"""
    sq = lambda x0, y0, x1, y1: np.array([(x0, y0), (x1, y0), (x1, y1), (x0, y1)], dtype=float)      # noqa: E731
    out = [([sq(0, 0, 10, 10)], [sq(0, 0, 10, 10)]), ([sq(0, 0, 10, 10)], [sq(5, 5, 15, 15)]), ([sq(0, 0, 10, 10)], [sq(10, 0, 20, 10)]),
           ([sq(0, 0, 10, 10)], [sq(2, 3, 5, 9)]), ([sq(0, 0, 10, 10)], [sq(11, 0, 20, 10)]),
           ([sq(0, 0, 4, 4), sq(10, 0, 14, 4)], [sq(2, 2, 12, 6)]),
           ([np.array([(0, 0), (10, 0), (0, 10)], dtype=float)], [np.array([(10, 10), (0, 10), (2, 6), (10, 0)], dtype=float)])]
    polys = simple_polygons(9, 16 if quick else 80, lo=15, hi=40, rmin=3, rmax=14)
    for k in range(0, len(polys) - 1, 2):
        out.append(([polys[k].astype(float)], [polys[k + 1].astype(float)]))
    return out
