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
from tests import geos_cases, helpers, ogc_reference as ogc
from tests.geos_cases import corner_cases as _corner_cases, intersect_cases as _intersect_cases, touches_cases as _touches_cases
from wavebreaking_amd import _hip


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
