"""Hardening of the GEOS-unpinned rows of SURVEY.md 8(a) S6 with the evidence that IS available in this image (the round-4
review's item 5): randomized differential tests of ``LineString.touches`` / ``intersects`` as the streamer index calls them
(streamer_index.py:185-251) -- oracle (oracle/geom.py), device functions (csrc/wb_core.h, run by a test kernel through
include/wavebreaking_hip_dev.h) and an INDEPENDENT exact-rational statement written from the OGC text (tests/ogc_reference.py)
-- on the inputs the hand-made truth tables do not enumerate: contours that run over themselves after the first-occurrence
dedupe of contour_index.py:110-112 (steps up to (3, 0), an end vertex lying on another segment of the same line, collinear
runs).  matplotlib's ``Path.intersects_path`` is a second opinion for ``intersects``.  None of this pins the rows to GEOS
(nothing in this image can); it is where a silent divergence would hide."""
import ctypes as C

import numpy as np
import pytest

from oracle import geom
from tests import helpers, ogc_reference as ogc
from wavebreaking_amd import _hip


def _dedupe(points):
    """np.round(...).astype(int) + dict.fromkeys: the reference's contour, contour_index.py:110-112"""
    return np.array(list(dict.fromkeys(map(tuple, points))), dtype=np.int64)


def _walk(rng, n, revisit):
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


def _contours(rng, count):
    out = []
    while len(out) < count:
        c = _dedupe(_walk(rng, int(rng.randint(12, 60)), float(rng.choice([0.0, 0.1, 0.25]))))
        if len(c) >= 4 and c.min() >= 0:
            out.append(c)
    return out


def _corner_cases():
    """hand-made: the contour's FIRST vertex lies on a later long segment, and chords run through it"""
    cases = []
    # first vertex (5, 5); the dedupe leaves the long step (4, 5) -> (7, 5)?? no: the line comes back OVER (5, 5), (6, 5)
    raw = [(5, 5), (6, 5), (7, 6), (7, 7), (6, 8), (4, 8), (3, 7), (3, 6), (4, 5), (5, 5), (6, 5), (7, 5), (8, 4), (6, 2), (5, 3), (5, 4), (5, 6), (5, 7)]
    c = _dedupe(np.array(raw))
    cases.append(c)
    # last vertex on an earlier segment is impossible after the dedupe (it would have been dropped); first vertex on a LATER
    # vertical long step:
    raw = [(10, 10), (11, 10), (12, 11), (12, 12), (10, 13), (9, 12), (10, 11), (10, 10), (10, 9), (10, 8), (9, 7), (8, 9), (9, 10), (11, 10), (12, 10)]
    cases.append(_dedupe(np.array(raw)))
    return cases


def _device_touches(ctx, contour, i1, i2):
    x = np.ascontiguousarray(contour[:, 0], dtype=np.int32)
    y = np.ascontiguousarray(contour[:, 1], dtype=np.int32)
    a = np.ascontiguousarray(i1, dtype=np.int32)
    b = np.ascontiguousarray(i2, dtype=np.int32)
    out = np.zeros(a.size, dtype=np.int32)
    p = C.POINTER(C.c_int32)
    lib = ctx.L.lib
    lib.wb_test_chord_touches.argtypes = [C.c_void_p, C.c_int, p, p, C.c_int, p, p, p]
    rc = lib.wb_test_chord_touches(ctx.h, int(x.size), x.ctypes.data_as(p), y.ctypes.data_as(p), int(a.size), a.ctypes.data_as(p), b.ctypes.data_as(p), out.ctypes.data_as(p))
    assert rc == 0, ctx.error()
    return out.astype(bool)


def _device_intersects(ctx, seg8):
    s = np.ascontiguousarray(seg8, dtype=np.int32)
    out = np.zeros(len(s), dtype=np.int32)
    p = C.POINTER(C.c_int32)
    lib = ctx.L.lib
    lib.wb_test_segments_intersect.argtypes = [C.c_void_p, C.c_int, p, p]
    assert lib.wb_test_segments_intersect(ctx.h, len(s), s.ctypes.data_as(p), out.ctypes.data_as(p)) == 0, ctx.error()
    return out.astype(bool)


def _touches_cases(seed, ncontours, per):
    rng = np.random.RandomState(seed)
    cs = _corner_cases() + _contours(rng, ncontours)
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


def _run_touches(ctx, seed, ncontours, per):
    total = boundary_cases = rejected = 0
    for c, ij in _touches_cases(seed, ncontours, per):
        want = np.array([ogc.chord_touches_linestring(c[i], c[j], c) for i, j in ij])
        orc = np.array([geom.chord_touches_contour(c[i], c[j], c) for i, j in ij])
        assert np.array_equal(orc, want), ("oracle", c.tolist(), ij[orc != want].tolist())
        if ctx is not None:
            dev = _device_touches(ctx, c, ij[:, 0], ij[:, 1])
            assert np.array_equal(dev, want), ("device", c.tolist(), ij[dev != want].tolist())
        total += len(ij)
        rejected += int((~want).sum())
        # chords whose open segment passes through an end vertex's location (the cases that turn on the boundary as a point set)
        for e in (0, len(c) - 1):
            for i, j in ij:
                if i != e and j != e and geom.orient(c[i, 0], c[i, 1], c[j, 0], c[j, 1], c[e, 0], c[e, 1]) == 0 \
                        and min(c[i, 0], c[j, 0]) <= c[e, 0] <= max(c[i, 0], c[j, 0]) and min(c[i, 1], c[j, 1]) <= c[e, 1] <= max(c[i, 1], c[j, 1]):
                    boundary_cases += 1
    return total, boundary_cases, rejected


def test_touches_oracle_and_host_simulation_against_the_independent_statement():
    ctx = _hip.Context(0, helpers.hostsim_library())
    try:
        total, bd, rej = _run_touches(ctx, 2024, 220, 60)
    finally:
        ctx.close()
    assert total >= 10000 and bd >= 50 and rej > total // 10 and total - rej > total // 10, (total, bd, rej)


def test_end_vertex_under_a_later_segment_is_boundary_as_a_point():
    """the case the index-based rule got wrong until round 5: the open chord properly crosses a long segment exactly at the
    contour's first vertex.  Point-set boundary (OGC): not an interior-interior meeting."""
    c = _corner_cases()[1]                          # starts at (10, 10); the step (10, 11) -> (10, 9) runs over it
    assert tuple(c[0]) == (10, 10)
    k = [tuple(v) for v in c.tolist()]
    i, j = k.index((9, 10)), k.index((11, 10))      # the chord (9, 10) - (11, 10) passes through (10, 10) and crosses that step there
    seg = [(a, b) for a, b in zip(k[:-1], k[1:]) if a[0] == b[0] == 10 and min(a[1], b[1]) < 10 < max(a[1], b[1])]
    assert seg, k
    want = ogc.chord_touches_linestring(c[i], c[j], c)
    assert geom.chord_touches_contour(c[i], c[j], c) == want


def _intersect_cases(seed, n):
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


def test_intersects_oracle_host_simulation_and_matplotlib_against_the_independent_statement():
    from matplotlib.path import Path
    s = _intersect_cases(7, 14000)
    want = np.array([ogc.segments_intersect(r[0:2], r[2:4], r[4:6], r[6:8]) for r in s])
    orc = np.array([geom.segments_intersect(r[0:2], r[2:4], r[4:6], r[6:8]) for r in s])
    assert np.array_equal(orc, want)
    ctx = _hip.Context(0, helpers.hostsim_library())
    try:
        assert np.array_equal(_device_intersects(ctx, s), want)
    finally:
        ctx.close()
    # matplotlib: Path.intersects_path(filled=False) is a floating-point segment test -- a second opinion on lattice inputs,
    # where its arithmetic is exact too; it does not count every collinear end-point contact, so it is compared on the proper
    # crossings and the disjoint pairs, where no convention enters
    agree = checked = 0
    for r, w in zip(s[-3000:], want[-3000:]):
        d1 = geom.orient(*r[0:2], *r[2:4], *r[4:6]); d2 = geom.orient(*r[0:2], *r[2:4], *r[6:8])
        d3 = geom.orient(*r[4:6], *r[6:8], *r[0:2]); d4 = geom.orient(*r[4:6], *r[6:8], *r[2:4])
        if 0 in (d1, d2, d3, d4):
            continue
        checked += 1
        agree += Path([r[0:2], r[2:4]]).intersects_path(Path([r[4:6], r[6:8]]), filled=False) == w
    assert checked > 500 and agree == checked
    assert len(s) >= 10000 and want.sum() > len(s) // 10 and (~want).sum() > len(s) // 10


@pytest.mark.gpu
def test_touches_and_intersects_on_the_device_against_the_independent_statement():
    """the same cases through the device functions wb_k_streamer_f234 calls (>= 10^4 chords, >= 10^4 segment pairs)"""
    ctx = _hip.Context(0)
    try:
        total, bd, rej = _run_touches(ctx, 2025, 200, 60)
        assert total >= 10000 and bd >= 50
        s = _intersect_cases(11, 14000)
        want = np.array([ogc.segments_intersect(r[0:2], r[2:4], r[4:6], r[6:8]) for r in s])
        assert np.array_equal(_device_intersects(ctx, s), want) and len(s) >= 10000
    finally:
        ctx.close()
