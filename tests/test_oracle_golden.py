"""The oracle against the golden vectors of the third-party libraries the reference calls
(CPU only).  These tests are what makes the oracle trustworthy as the parity checker."""
import numpy as np
import pytest

from oracle import contours as oc
from oracle import geom, indices, marching, properties
from tests.golden import geos_truth


def test_case_table_matches_skimage(golden_dir):
    g = np.load(golden_dir + "/contours_cases.npz")
    k = 0
    for case in range(16):
        a = np.array([[case & 1, (case >> 1) & 1], [(case >> 2) & 1, (case >> 3) & 1]], dtype=float)
        seg = marching.contour_segments(a, 0.5)
        for s in range(len(seg["r"])):
            assert g["case"][k] == case
            assert np.array_equal(seg["p_from"][s], g["p_from"][k])
            assert np.array_equal(seg["p_to"][s], g["p_to"][k])
            k += 1
    assert k == len(g["case"])


def test_find_contours_bit_exact_vs_skimage(golden_dir):
    g = np.load(golden_dir + "/contours_random.npz")
    total = 0
    for name in g["names"]:
        a, lv = g[name + "__field"], float(g[name + "__level"])
        seg = marching.contour_segments(a, lv)
        assert np.array_equal(seg["p_from"].reshape(-1, 2), g[name + "__seg_from"]), name
        assert np.array_equal(seg["p_to"].reshape(-1, 2), g[name + "__seg_to"]), name
        cont = marching.find_contours(a, lv)
        off, pts = g[name + "__off"], g[name + "__pts"]
        assert len(cont) == len(off) - 1, name
        for i, c in enumerate(cont):
            assert np.array_equal(c, pts[off[i]:off[i + 1]]), (name, i)
        total += len(cont)
    assert total > 400


def test_tiny_example_from_survey():
    # [[1,4],[1,1]], level 2 -> ((0.6667,1) -> (0,0.3333))
    c = marching.find_contours(np.array([[1.0, 4.0], [1.0, 1.0]]), 2.0)
    assert len(c) == 1
    assert np.allclose(c[0], [[2 / 3, 1.0], [0.0, 1 / 3]])


def test_haversine_formula_vs_sklearn(golden_dir):
    g = np.load(golden_dir + "/haversine.npz")
    geo = indices.haversine_formula(np.radians(g["coords"])) * 6371
    assert np.allclose(geo, g["geo"], rtol=0, atol=1e-9)
    assert np.array_equal(np.diag(geo), np.zeros(len(geo)))
    # along-contour distance: sequential row cumsum of `on` (streamer_index.py:133-135)
    on = g["path_on"]
    for row, i in zip(g["path_cont_rows"], [0, 7, 150]):
        acc, ref = 0.0, np.zeros(len(on))
        for j in range(i + 1, len(on)):
            acc = acc + on[j]
            ref[j] = acc
        assert np.array_equal(ref, row)


def test_streamer_duplicate_rule_vs_pandas(golden_dir):
    g = np.load(golden_dir + "/pandas_rules.npz")
    nlon = int(g["dup_nlon"])
    seen = 0
    for n in range(6):
        if "dup%d__frame" % n not in g:
            continue
        fr = g["dup%d__frame" % n]
        # build a fake contour whose vertex k is frame point k (two vertices per pair)
        xy = np.concatenate([fr[:, 0:2], fr[:, 2:4]], axis=0)
        ij = np.c_[np.arange(len(fr)), np.arange(len(fr)) + len(fr)]
        kept = indices._f_duplicates(ij, xy, nlon)
        assert np.array_equal(kept[:, 0], g["dup%d__keep" % n])
        seen += 1
    assert seen >= 3


def test_overturning_labels_vs_pandas(golden_dir):
    g = np.load(golden_dir + "/pandas_rules.npz")
    for n in range(6):
        for rg in (5, 2):
            x = g["ot%d_rg%d__x" % (n, rg)]
            assert np.array_equal(indices.overturning_groups(x, float(rg), 1.0), g["ot%d_rg%d__minmax" % (n, rg)])


def test_rounding_rules(golden_dir):
    g = np.load(golden_dir + "/pandas_rules.npz")
    assert np.array_equal(np.rint(g["round_in"]).astype(int), g["round_out"])
    xy, closed = oc.round_dedupe(np.array([[0.5, 1.5], [0.4, 2.5], [1.5, 2.5], [0.5, 1.5]]))
    # rows round to (0,2),(0,2),(2,2),(0,2): first occurrences kept, (row,col)->(x,y)
    assert xy.tolist() == [[2, 0], [2, 2]] and closed


def test_combine_shared_reference_known_answer():
    # tests/test_wavebreaking.py:101-104
    assert indices.combine_shared([[1, 2, 3], [2, 3, 4], [5, 6]]) == [[1, 2, 3, 4], [5, 6]]


@pytest.mark.parametrize("row", geos_truth.TOUCHES)
def test_touches_truth_table(row):
    contour, i, j, expect, why = row
    c = np.array(contour)
    assert geom.chord_touches_contour(c[i], c[j], c) is expect, why


@pytest.mark.parametrize("row", geos_truth.INTERSECTS)
def test_intersects_truth_table(row):
    p1, q1, p2, q2, expect = row
    assert geom.segments_intersect(p1, q1, p2, q2) is expect
    assert geom.segments_intersect(p2, q2, p1, q1) is expect


@pytest.mark.parametrize("row", geos_truth.CELLS)
def test_buffer_contains_truth_table(row):
    poly, r, expect = row
    xs, ys = geom.polygon_cells(np.array(poly), r, 64, 64)
    assert sorted(zip(xs.tolist(), ys.tolist())) == sorted(expect)


def test_contour_periodic_duplicates_and_columns():
    # a cutoff inside the periodic overlap appears twice; the later copy goes (contour_index.py:119-147)
    a = np.array([[10, 5], [11, 5], [11, 6], [10, 6]])
    items = [a, a + np.array([36, 0]), np.array([[3, 1], [4, 1], [4, 2], [3, 2], [2, 2]])]
    assert oc.check_duplicates(items, 36) == {1}
    # a partial copy (subset, shorter) goes whichever comes first
    assert oc.check_duplicates([a[:3] + np.array([36, 0]), a], 36) == {0}


def test_seam_split_areas():
    lat = np.arange(-5.0, 6.0)
    lon = np.arange(-18.0, 18.0) * 10.0       # nlon = 36
    # rectangle x 34..38, y 2..4 straddles the seam: left piece 34..35, right piece 36..38 -> lon[0..2]
    rings = properties.transform_polygon(np.array([[34, 2], [38, 2], [38, 4], [34, 4]]), lat, lon)
    areas = sorted(abs(properties.ring_area(r)) for r in rings)
    assert len(rings) == 2
    assert np.allclose(areas, [10.0 * 2, 20.0 * 2])
    # entirely right of the seam: one ring, shifted
    rings = properties.transform_polygon(np.array([[37, 2], [39, 2], [39, 4], [37, 4]]), lat, lon)
    assert len(rings) == 1 and rings[0][:, 0].min() == lon[1]
    # not touching the seam: untouched
    rings = properties.transform_polygon(np.array([[3, 2], [5, 2], [5, 4]]), lat, lon)
    assert len(rings) == 1 and np.array_equal(rings[0][:, 0], lon[[3, 5, 5]])


class _CellValues:
    """stands in for data[t, ys, xs % nlon]: hands back the fixture's per-cell float32 values"""

    def __init__(self, v):
        self.v = v

    def __getitem__(self, key):
        return self.v


def test_property_sums_vs_pandas(golden_dir):
    """calculate_properties' per-cell products and groupby sums (index_utils.py:58-111): the oracle's restatement of
    pandas' compensated group sum against pandas 2.3.3 itself, bit for bit, on frames whose x_com / areas is an integer
    mathematically (symmetric boxes) -- the case where the summation order decides the truncated index."""
    g = np.load(golden_dir + "/props_pandas.npz")
    ids, X, Y = g["ids"], g["x"].astype(np.int64), g["y"].astype(np.int64)
    nev = len(g["sums"])
    near = 0
    for e in range(nev):
        m = ids == e
        w = g["weight_lat_%d" % int(g["grid_of"][e])]
        s = properties.event_sums(X[m], Y[m], 0, _CellValues(g["val"][m]), _CellValues(g["inten"][m]), w, 1 << 20)
        assert tuple(s) == tuple(g["sums"][e]), (e, s, g["sums"][e])
        assert int(s[3] / s[0]) == g["com_ix"][e] and int(s[4] / s[0]) == g["com_iy"][e]
        r = s[3] / s[0]
        near += abs(r - round(r)) < 1e-9
    assert near >= 20
    # the boxes of the fixture are what they claim to be: cells in (y, x) order, as np.meshgrid(...).flatten() numbers them
    for e, x0, y0, x1, y1 in g["boxes"]:
        gx, gy = np.meshgrid(np.arange(x0, x1 + 1), np.arange(y0, y1 + 1))
        assert np.array_equal(X[ids == e], gx.flatten()) and np.array_equal(Y[ids == e], gy.flatten())


def test_group_sum_is_not_a_plain_sum(golden_dir):
    """guards the fixture's power: a naive left-to-right sum of the same rows misses pandas' value on some events"""
    g = np.load(golden_dir + "/props_pandas.npz")
    ids, X, Y = g["ids"], g["x"].astype(np.int64), g["y"].astype(np.int64)
    differ = 0
    for e in range(len(g["sums"])):
        m = ids == e
        w = g["weight_lat_%d" % int(g["grid_of"][e])]
        acc = 0.0
        for v in (X[m].astype(np.float64) * w[Y[m]]).tolist():
            acc += v
        differ += acc != g["sums"][e][3]
    assert differ >= 5


def test_cells_equal_to_the_level_vs_skimage(golden_dir):
    """find_contours with cells EQUAL to the level (contour_index.py:103 on packed or rounded PV): crossings on grid
    vertices, zero-length segments, dictionary keys shared by several contour ends -- bit-exact against scikit-image
    0.18.3, including the fields on which the third-party routine raises KeyError."""
    g = np.load(golden_dir + "/contours_levelhit.npz")
    nraise = nmachine = 0
    for name in g["names"]:
        a, lv = g[name + "__field"].astype(np.float64), float(g[name + "__level"])
        assert np.count_nonzero(a == lv) > 0, name
        if int(g[name + "__raises"]):
            with pytest.raises(KeyError):
                marching.find_contours(a, lv)
            nraise += 1
            assert marching.find_contours_vertex_machine(a, lv)[1] > 0, name     # ... and the kernel's formulation knows
            continue
        cont = marching.find_contours(a, lv)
        off, pts = g[name + "__off"], g[name + "__pts"]
        assert len(cont) == len(off) - 1, name
        for i, c in enumerate(cont):
            assert np.array_equal(c, pts[off[i]:off[i + 1]]), (name, i)
        # the order-independent formulation the HIP link kernel uses: equal whenever no pending end was overwritten
        mach, overwritten = marching.find_contours_vertex_machine(a, lv)
        if overwritten == 0:
            nmachine += 1
            assert len(mach) == len(cont) and all(np.array_equal(x, y) for x, y in zip(mach, cont)), name
    assert nraise >= 3 and nmachine >= 40


def test_edge_linking_and_coordinate_join_agree_without_level_hits(golden_dir):
    g = np.load(golden_dir + "/contours_random.npz")
    for name in g["names"]:
        a, lv = g[name + "__field"], float(g[name + "__level"])
        seg = marching.contour_segments(a, lv)
        assert seg["level_hits"] == 0
        joined = marching.join_by_coordinates(seg["p_from"], seg["p_to"])
        linked = marching.find_contours(a, lv)
        assert len(joined) == len(linked) and all(np.array_equal(x, y) for x, y in zip(joined, linked)), name


def test_lattice_rasterisation_interior_vs_matplotlib():
    """geom.polygon_cells (the oracle's restatement of buffer(r) + contains, index_utils.py:37-51) on random simple
    lattice polygons: its interior term against matplotlib.path.Path.contains_points (a third-party point-in-polygon,
    importable here), its boundary / rim term against exact point-segment distances."""
    mpath = pytest.importorskip("matplotlib.path")
    rng = np.random.RandomState(8)
    checked = 0
    for trial in range(150):
        n = int(rng.randint(3, 12))
        cx, cy = rng.uniform(12, 28), rng.uniform(12, 28)
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        rad = rng.uniform(2.0, 11.0, n)
        pts = np.c_[np.rint(cx + rad * np.cos(ang)), np.rint(cy + rad * np.sin(ang))].astype(int)
        if len(set(map(tuple, pts))) != len(pts):
            continue
        ok = True
        for i in range(n):                                   # simple?
            for j in range(i + 1, n):
                if j == i + 1 or (i == 0 and j == n - 1):
                    continue
                ok &= not geom.segments_intersect(pts[i], pts[(i + 1) % n], pts[j], pts[(j + 1) % n])
        for i in range(n):
            a, b, c = pts[i - 1], pts[i], pts[(i + 1) % n]
            if geom.orient(a[0], a[1], b[0], b[1], c[0], c[1]) == 0 and np.dot(b - a, c - b) < 0:
                ok = False
        if not ok:
            continue
        gx, gy = np.meshgrid(np.arange(0, 41), np.arange(0, 41))
        P = np.c_[gx.ravel(), gy.ravel()]
        # exact distance of every lattice point to the boundary
        on_boundary = np.zeros(len(P), dtype=bool)
        d2min = np.full(len(P), np.inf)
        for k in range(n):
            a, b = pts[k].astype(float), pts[(k + 1) % n].astype(float)
            ab = b - a
            t = np.clip(((P - a) @ ab) / (ab @ ab), 0.0, 1.0)
            d2 = np.sum((P - (a + t[:, None] * ab)) ** 2, axis=1)
            d2min = np.minimum(d2min, d2)
        on_boundary = d2min < 1e-18
        inside_mpl = mpath.Path(np.r_[pts, pts[:1]], closed=True).contains_points(P) & ~on_boundary
        for r in (0.125, 0.5):
            want = inside_mpl | (d2min < r * r - 1e-12)
            xs, ys = geom.polygon_cells(pts, r, 41, 41)
            got = np.zeros((41, 41), dtype=bool)
            got[ys, xs] = True
            # a lattice point off a lattice segment is never at distance exactly r for r = 1/8, 1/2
            assert not np.any(np.abs(np.sqrt(d2min[~on_boundary]) - r) < 1e-9)
            assert np.array_equal(got.ravel(), want), trial
        checked += 1
    assert checked >= 60


def test_order_dependent_joins_vs_skimage(golden_dir):
    """the constellations in which two contour ends compete for one grid vertex (found by tools/levelwalk_hunt.py among 1 500
    random rough fields, scikit-image's own outputs in contours_levelwalk.npz): the oracle's restatement of the sequential
    walk returns skimage's contours on the 48 it returns on and raises KeyError on the 6 it raises on; the vertex state
    machine sees an overwritten end on every one of them"""
    g = np.load(golden_dir + "/contours_levelwalk.npz")
    nraise = nwalk = 0
    for name in g["names"]:
        a, lv = g[name + "__field"].astype(np.float64), float(g[name + "__level"])
        assert marching.find_contours_vertex_machine(a, lv)[1] > 0, name
        if int(g[name + "__raises"]):
            with pytest.raises(KeyError):
                marching.find_contours(a, lv)
            nraise += 1
            continue
        cont = marching.find_contours(a, lv)
        off, pts = g[name + "__off"], g[name + "__pts"]
        assert len(cont) == len(off) - 1, name
        for i, c in enumerate(cont):
            assert np.array_equal(c, pts[off[i]:off[i + 1]]), (name, i)
        nwalk += 1
    assert nraise == 6 and nwalk == 48


def test_a_contours_distinct_x_are_its_extent():
    """wb_k_trace_fused takes exp_lon (contour_index.py:154: the number of distinct x of the contour's points) as xmax - xmin + 1:
    consecutive crossings of the marching squares lie on the edges of one cell and round to corners of that cell, so the points
    form a lattice path.  Here: the oracle's contours in index coordinates (len(set(x)) as the reference counts, contour_index.py:154
    with original_coordinates=False, the index functions' case) on smooth to rough fields of three grids all have exp_lon equal to
    their extent."""
    import numpy as np
    from oracle import contours as oc, synth
    n = 0
    for dlon, eps, t0, nt in ((2.5, 0.2, 0, 4), (2.5, 0.8, 100, 4), (1.0, 0.0, 0, 2), (1.0, 0.6, 40, 2), (0.25, 0.3, 7, 1)):
        lat, lon = synth.grid(dlon)
        pv = synth.pv_field(t0, nt, lat, lon, eps=eps)
        frame = oc.calculate_contours(pv, [-2, 2], lat, lon, synth.times(nt), original_coordinates=False)   # (index coordinates: what the index functions select on)
        for c in frame:
            x = np.asarray(c["geometry"])[:, 0]
            assert round(c["exp_lon"] / dlon) == len(set(x.tolist())) == int(x.max() - x.min()) + 1, (dlon, eps, c["exp_lon"])
            n += 1
    assert n > 100
