"""The GEOS pin (tests/golden/geos_rules.npz, written by tests/golden/make_golden_geos.py wherever shapely >= 2.0.1 is
importable -- not in the build image): the reference's own GEOS calls on stored inputs, against the oracle's statements and,
where a device function exists, the kernels' (host simulation here, the device under -m gpu).

WITHOUT the file every comparison below SKIPS, loudly: parity of the GEOS-dependent rows (SURVEY.md 8a S6 routines 2 and 4,
P1 cell membership, P2 seam split, 8f by_overlap) then rests on builder-authored statements.
``test_kit_plumbing_with_stand_in_predicates`` always runs: it drives the kit with the oracle's functions in place of the five
GEOS calls and feeds the file it writes to the same comparisons -- a check of the file format, NOT a pin."""
import os
from fractions import Fraction

import numpy as np
import pytest

from oracle import geom
from oracle import properties as oprop
from oracle import tracking as otrack

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "geos_rules.npz")
HOWTO = ("GEOS PIN ABSENT -- tests/golden/geos_rules.npz is not there: the GEOS-dependent rows are checked against builder-authored "
         "statements only.  One command on any machine with shapely >= 2.0.1:  python tests/golden/make_golden_geos.py")


def _load(path=GOLDEN):
    if not os.path.exists(path):
        pytest.skip(HOWTO)
    return dict(np.load(path))


def _slices(off, rows):
    return [rows[off[k]:off[k + 1]] for k in range(len(off) - 1)]


# ------------------------------------------------------------------------------------------------ the comparisons (file -> asserts)
def check_touches(d, device_touches=None):
    cont = _slices(d["touches_contour_offset"], d["touches_contour_xy"])
    po, pairs, want = d["touches_pair_offset"], d["touches_pairs"], d["touches_result"]
    n = 0
    for k, c in enumerate(cont):
        ij, w = pairs[po[k]:po[k + 1]], want[po[k]:po[k + 1]]
        got = np.array([geom.chord_touches_contour(c[i], c[j], c) for i, j in ij], dtype=bool)
        assert np.array_equal(got, w), ("oracle touches", c.tolist(), ij[got != w].tolist())
        if device_touches is not None and len(ij):
            dev = device_touches(c, ij[:, 0], ij[:, 1])
            assert np.array_equal(dev, w), ("device touches", c.tolist(), ij[dev != w].tolist())
        n += len(ij)
    return n


def check_intersects(d, device_intersects=None):
    s, want = d["inter_segments"], d["inter_result"]
    got = np.array([geom.segments_intersect(r[0:2], r[2:4], r[4:6], r[6:8]) for r in s], dtype=bool)
    assert np.array_equal(got, want), s[got != want][:5].tolist()
    if device_intersects is not None:
        assert np.array_equal(device_intersects(s), want)
    return len(s)


def check_cells(d):
    polys = _slices(d["cells_poly_offset"], d["cells_poly_xy"])
    cells = _slices(d["cells_offset"], d["cells_xy"])
    for poly, (r, width, nlat), want in zip(polys, d["cells_meta"], cells):
        xs, ys = geom.polygon_cells(poly, float(r), int(width), int(nlat))
        order = np.lexsort((xs, ys))
        got = np.c_[np.asarray(xs)[order], np.asarray(ys)[order]]
        assert np.array_equal(got, want), ("cells", poly.tolist(), float(r))
    return len(polys)


def _truncated(ring):
    return np.asarray(ring).astype("int").astype(float)            # index_utils.py:135


def check_split(d):
    polys = _slices(d["split_poly_offset"], d["split_poly_xy"])
    rings = _slices(d["split_ring_offset"], d["split_ring_xy"])
    rc = d["split_ring_count"]
    multi = 0
    for k, (poly, (nlon, nlat)) in enumerate(zip(polys, d["split_meta"])):
        want = sorted(oprop.canonical_ring(_truncated(r)) for r in rings[rc[k]:rc[k + 1]])
        got = sorted(oprop.canonical_ring(_truncated(r)) for r in oprop.split_index_polygon(poly, int(nlon)))
        assert got == want, ("split", poly.tolist(), got, want)
        multi += len(want) > 2
    return len(polys), multi


def check_iou(d, rtol=1e-9):
    ra = _slices(d["iou_a_offset"], d["iou_a_xy"])
    rb = _slices(d["iou_b_offset"], d["iou_b_xy"])
    ca, cb = d["iou_a_count"], d["iou_b_count"]
    for k, (area_a, area_b, inter) in enumerate(d["iou_areas"]):
        a = [np.asarray(r).astype(int) for r in ra[ca[k]:ca[k + 1]]]
        b = [np.asarray(r).astype(int) for r in rb[cb[k]:cb[k + 1]]]
        ga, gb, gi = float(otrack.rings_area(a)), float(otrack.rings_area(b)), float(otrack.overlap_area(a, b))
        for got, want in ((ga, area_a), (gb, area_b), (gi, inter)):
            assert abs(got - want) <= rtol * max(1.0, abs(want)), ("iou", k, got, want)
        # the decision of events.py:209-213 with overlap = 0 is exact in the oracle: an intersection of measure zero is none
        assert (Fraction(otrack.overlap_area(a, b)) > 0) == (inter > 1e-12), ("iou > 0", k)
    return len(d["iou_areas"])


def _hostsim_predicates():
    from tests import helpers
    from tests.test_geos_hardening import _device_intersects, _device_touches
    from wavebreaking_amd import _hip
    ctx = _hip.Context(0, helpers.hostsim_library())
    return ctx, (lambda c, i, j: _device_touches(ctx, c, i, j)), (lambda s: _device_intersects(ctx, s))


# ------------------------------------------------------------------------------------------------ with the file
def test_touches_and_intersects_against_geos():
    d = _load()
    ctx, dt, di = _hostsim_predicates()
    try:
        assert check_touches(d, dt) >= 1000 and check_intersects(d, di) >= 1000
    finally:
        ctx.close()


def test_cell_membership_against_geos_buffer_and_sjoin():
    assert check_cells(_load()) >= 20


def test_seam_split_against_geos_polygonize():
    n, multi = check_split(_load())
    assert n >= 10


def test_overlap_areas_against_geos_intersection():
    assert check_iou(_load()) >= 10


@pytest.mark.gpu
def test_touches_and_intersects_on_the_device_against_geos():
    d = _load()
    from tests.test_geos_hardening import _device_intersects, _device_touches
    from wavebreaking_amd import _hip
    ctx = _hip.Context(0)
    try:
        check_touches(d, lambda c, i, j: _device_touches(ctx, c, i, j))
        check_intersects(d, lambda s: _device_intersects(ctx, s))
    finally:
        ctx.close()


# ------------------------------------------------------------------------------------------------ the kit's plumbing (pins nothing)
def test_kit_plumbing_with_stand_in_predicates(tmp_path, monkeypatch):
    """make_golden_geos.build() with the oracle's functions in place of the five GEOS calls: the file it writes goes through
    the same comparisons (oracle against oracle: they must pass) -- the generators run, the ragged arrays round-trip, the
    consumers read what the kit writes.  What GEOS would answer is NOT checked here."""
    from tests.golden import make_golden_geos as kit
    monkeypatch.setattr(kit, "geos_touches", lambda c, i, j: geom.chord_touches_contour(c[i], c[j], c))
    monkeypatch.setattr(kit, "geos_intersects", lambda s: geom.segments_intersect(s[0:2], s[2:4], s[4:6], s[6:8]))

    def cells(poly, r, width, nlat):
        xs, ys = geom.polygon_cells(np.asarray(poly), r, width, nlat)
        order = np.lexsort((xs, ys))
        return np.c_[np.asarray(xs)[order], np.asarray(ys)[order]]
    monkeypatch.setattr(kit, "geos_cells", cells)
    monkeypatch.setattr(kit, "geos_split", lambda poly, nlon, nlat: oprop.split_index_polygon(np.asarray(poly), nlon))
    monkeypatch.setattr(kit, "geos_iou", lambda a, b: (float(otrack.rings_area([np.asarray(r).astype(int) for r in a])),
                                                       float(otrack.rings_area([np.asarray(r).astype(int) for r in b])),
                                                       float(otrack.overlap_area([np.asarray(r).astype(int) for r in a], [np.asarray(r).astype(int) for r in b]))))
    monkeypatch.setattr(kit, "versions", lambda: {"shapely": "stand-in", "geos": "stand-in", "geopandas": ""})
    out = str(tmp_path / "rules.npz")
    kit.build(out, quick=True)
    d = _load(out)
    assert str(d["version_shapely"]) == "stand-in"
    assert check_touches(d) >= 500 and check_intersects(d) >= 1000
    assert check_cells(d) >= 20
    n, multi = check_split(d)
    assert n >= 10 and multi >= 1
    assert check_iou(d) >= 10


def test_the_pin_is_reported_when_absent():
    """the skip above must be visible in a default run's summary: the file is either there or the reason names the command"""
    if os.path.exists(GOLDEN):
        d = dict(np.load(GOLDEN))
        assert str(d["version_shapely"]) not in ("", "stand-in")
    else:
        assert "make_golden_geos.py" in HOWTO
