"""transform_polygons' seam split (index_utils.py:148-173): the product's vertical-line clip (csrc/wb_geometry.h, the code
behind wb_get_event_geometry, reached here through wb_transform_polygons of the host-simulation build of the same sources)
against an oracle built on a DIFFERENT algorithm -- exact rational arithmetic,
line stretches assigned by point-in-polygon tests of their midpoints -- ring set against ring set, and both against a
ring-free statement of the region's area (Green's theorem with the abscissa clamped at the line).  GEOS itself is not
installable here: the restatement is unpinned, the two independent derivations agree."""
from fractions import Fraction

import numpy as np
import pytest

from oracle import properties as oprop
from tests import helpers
from wavebreaking_amd import _hip

NLON = 36
LAT = np.linspace(-45.0, 45.0, 19)
LON = np.arange(0.0, 360.0, 10.0)


_ctx = []


def _context():
    if not _ctx:
        _ctx.append(_hip.Context(0, helpers.hostsim_library()))
    return _ctx[0]


def _ring_sets(poly):
    got = _context().transform_polygons([np.asarray(poly)], LON, LAT, NLON)[0]
    want = oprop.transform_polygon(np.asarray(poly), LAT, LON)
    return sorted(oprop.canonical_ring(r) for r in got), sorted(oprop.canonical_ring(r) for r in want)


def _same_rings(a, b):
    return len(a) == len(b) and all(len(p) == len(q) and all(np.array_equal(u, v) for u, v in zip(p, q)) for p, q in zip(a, b))


def _check(poly):
    got, want = _ring_sets(poly)
    assert got == want, (poly, got, want)
    # the kernels wb_indices runs for a pass's events (csrc/wb_geom.h, here in the host simulation): ring for ring and bit for
    # bit the host function's, in lon / lat and in index space
    for ispace in (False, True):
        host = _context().transform_polygons([np.asarray(poly)], LON, LAT, NLON, index_space=ispace)
        dev = _context().transform_polygons([np.asarray(poly)], LON, LAT, NLON, index_space=ispace, device=True)
        assert _same_rings(host, dev), (poly, ispace, host, dev)
    # the index-space pieces (before truncation and the lon / lat look-up) against the clamped-abscissa area
    pieces = _context().transform_polygons([np.asarray(poly)], LON, LAT, NLON, index_space=True)[0]
    area = sum(abs(oprop.ring_area(p)) for p in pieces)
    exact = oprop.clipped_area_exact(poly, NLON - 1, True) + oprop.clipped_area_exact(poly, NLON, False)
    assert abs(area - float(exact)) <= 1e-9 * max(1.0, float(exact)), (poly, area, float(exact))
    return got


HAND = {
    "box_across": [(34, 2), (38, 2), (38, 4), (34, 4)],
    "box_right_of_seam": [(37, 2), (39, 2), (39, 4), (37, 4)],
    "triangle_left": [(3, 2), (5, 2), (5, 4)],
    "inside_the_strip_only": [(35, 3), (36, 3), (36, 5), (35, 5)],
    "vertex_on_both_lines": [(33, 2), (35, 6), (36, 2), (38, 7), (34, 9)],
    # a U lying on its side: two prongs reach across the seam -> two pieces on the right, one on the left
    "u_shape": [(30, 2), (40, 2), (40, 4), (33, 4), (33, 6), (40, 6), (40, 8), (30, 8)],
    # ... and mirrored: two pieces on the left
    "u_shape_mirrored": [(44, 2), (32, 2), (32, 4), (41, 4), (41, 6), (32, 6), (32, 8), (44, 8)],
    # a comb with three teeth across the seam
    "comb": [(31, 1), (39, 1), (39, 3), (33, 3), (33, 5), (39, 5), (39, 7), (33, 7), (33, 9), (39, 9), (39, 11), (31, 11)],
    # boundary running along the line x = nlon - 1 and touching x = nlon in a single vertex
    "edge_on_the_line": [(32, 2), (35, 2), (35, 6), (36, 8), (35, 10), (32, 10)],
    # slanted edges: fractional crossing ordinates (truncated by index_utils.py:135)
    "slanted": [(30, 1), (41, 4), (38, 12), (33, 9)],
}


@pytest.mark.parametrize("name", sorted(HAND))
def test_hand_cases(name):
    _check(HAND[name])


def test_expected_piece_counts():
    assert len(_check(HAND["u_shape"])) == 3
    assert len(_check(HAND["u_shape_mirrored"])) == 3
    assert len(_check(HAND["comb"])) == 4
    assert len(_check(HAND["inside_the_strip_only"])) == 0
    assert len(_check(HAND["box_across"])) == 2


def test_random_simple_lattice_polygons_across_the_seam():
    rng = np.random.RandomState(31)
    multi = 0
    for trial in range(300):
        # star-shaped around a centre near the seam: vertices sorted by angle -> a simple polygon
        cx, cy = NLON - 0.5 + rng.uniform(-2, 2), 9.0 + rng.uniform(-2, 2)
        n = int(rng.randint(4, 14))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        rad = rng.uniform(1.5, 8.0, n)
        pts = np.c_[np.rint(cx + rad * np.cos(ang)), np.rint(cy + rad * np.sin(ang))].astype(int)
        pts = pts[np.r_[True, np.any(np.diff(pts, axis=0) != 0, axis=1)]]
        if len(pts) > 1 and (pts[0] == pts[-1]).all():
            pts = pts[:-1]
        if len(set(map(tuple, pts))) != len(pts) or len(pts) < 3 or pts[:, 1].min() < 0 or pts[:, 1].max() > 18 or pts[:, 0].min() < 0:
            continue
        if not _simple(pts):
            continue
        multi += len(_check([tuple(p) for p in pts])) > 2
    assert multi >= 3


def test_random_combs_give_many_pieces():
    # a body left of the seam with 2..5 teeth of random length and slant reaching across it (or the mirror image): every
    # tooth is a piece of its own on the far side
    rng = np.random.RandomState(77)
    done = 0
    for trial in range(120):
        nteeth = int(rng.randint(2, 6))
        x_body = int(rng.randint(26, 33))
        ring = [(x_body, 0)]
        y = 0
        for k in range(nteeth):
            tip = int(rng.randint(37, 45))
            h = int(rng.randint(1, 3))
            slant = int(rng.randint(-1, 2))
            ring += [(tip, y + max(0, slant)), (tip + int(rng.randint(0, 2)), y + h + max(0, slant))]     # the tooth's far end
            y += h + 1 + abs(slant)
            gap_x = int(rng.randint(x_body + 1, 35))
            if k < nteeth - 1:
                ring += [(gap_x, y - 1), (gap_x, y)]                                                     # the notch between two teeth
        ring += [(x_body, y)]
        pts = np.asarray(ring)
        if rng.rand() < 0.5:                                   # mirror about the seam
            pts = np.c_[2 * NLON - 1 - pts[:, 0], pts[:, 1]]
        pts = pts[np.r_[True, np.any(np.diff(pts, axis=0) != 0, axis=1)]]
        if pts[:, 1].max() > 18 or len(set(map(tuple, pts))) != len(pts) or not _simple(pts):
            continue
        assert len(_check([tuple(p) for p in pts])) >= nteeth
        done += 1
    assert done >= 40


def _simple(pts):
    from oracle import geom
    n = len(pts)
    for i in range(n):
        for j in range(i + 1, n):
            if j == i + 1 or (i == 0 and j == n - 1):
                continue
            if geom.segments_intersect(pts[i], pts[(i + 1) % n], pts[j], pts[(j + 1) % n]):
                return False
    # adjacent edges must not fold back onto each other
    for i in range(n):
        a, b, c = pts[i - 1], pts[i], pts[(i + 1) % n]
        if geom.orient(a[0], a[1], b[0], b[1], c[0], c[1]) == 0 and np.dot(b - a, c - b) < 0:
            return False
    return True


def test_device_kernels_on_many_polygons_in_one_call():
    """count -> offsets -> fill over a table of polygons (several workgroups, empty and whole and cut events mixed), and an
    event with more cut edges than the kernel holds (sent to the host function)"""
    rng = np.random.RandomState(5)
    polys = [HAND[k] for k in sorted(HAND)] * 7
    rng.shuffle(polys)
    teeth = [(30, 0)]
    for k in range(40):                                         # 40 teeth: 80 cut edges > WB_GEOM_XMAX
        teeth += [(40, 2 * k), (40, 2 * k + 1), (33, 2 * k + 1), (33, 2 * k + 2)]
    teeth += [(30, 80)]
    lat = np.linspace(-45.0, 45.0, 91)
    for extra in ([], [teeth]):
        allp = [np.asarray(q) for q in polys + extra]
        host = _context().transform_polygons(allp, LON, lat, NLON)
        dev = _context().transform_polygons(allp, LON, lat, NLON, device=True)
        assert _same_rings(host, dev)
    assert len(host[-1]) == 41


def test_clamped_area_formula_on_a_known_case():
    # a 4 x 2 box from x = 34 to 38: one column left of x = 35, two columns right of x = 36
    box = HAND["box_across"]
    assert oprop.clipped_area_exact(box, 35, True) == Fraction(2)
    assert oprop.clipped_area_exact(box, 36, False) == Fraction(4)
