#!/usr/bin/env python
"""GEOS pin kit: ONE command that pins the GEOS-dependent rows of SURVEY.md 8(a) wherever shapely >= 2.0.1 is importable.

    pip install "shapely>=2.0.1" geopandas numpy        # (geopandas optional: without it the script calls the same GEOS
    python tests/golden/make_golden_geos.py             #  predicates through shapely directly and says so in the file)
    python -m pytest tests/test_geos_pin.py -q          # now compares oracle (and host-simulated / device kernels) with GEOS

The build image has neither shapely nor a GEOS library, so the repository's statements of ``touches`` /
``intersects`` (oracle/geom.py), of ``buffer(r)`` + ``sjoin(contains)`` on the lattice (oracle/geom.py:polygon_cells), of the
seam split (oracle/properties.py:split_index_polygon) and of ``intersection().area`` (oracle/tracking.py) are checked against
hand-derived tables and an independent exact-rational statement only -- builder against builder.  This script runs the
reference's OWN calls, on inputs that are stored beside the answers (``tests/golden/geos_rules.npz``: numeric arrays and
version strings, no code), so the tests that consume the file need neither shapely nor this script's generators:

    section   reference call (file:line)                                          stored
    touches   LineString([p, q]).touches(contour)      streamer_index.py:192-198   contours, chord end indices -> bool
    inter     dline.intersects(dline)                  streamer_index.py:233-236   segment pairs -> bool
    cells     geometry.buffer((dlon + dlat) / 4) + sjoin(predicate="contains")     polygons, r, grid -> lattice cells
                                                       index_utils.py:37-51
    split     linemerge / unary_union / polygonize / make_valid                    index polygons, nlon, nlat -> exterior
                                                       index_utils.py:148-173      rings of the kept pieces
    iou       buffer(0).make_valid(), intersection(), area   events.py:205-213    ring sets A, B -> areas and the ratio

The five ``geos_*`` functions below are the ONLY places GEOS is called; everything else is plumbing (which
tests/test_geos_pin.py::test_kit_plumbing_with_stand_in_predicates exercises in the build image by swapping the five for the
oracle's functions -- that test checks the file format, it pins nothing)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

OUT = os.path.join(ROOT, "tests", "golden", "geos_rules.npz")


# ------------------------------------------------------------------------------------------------ the five GEOS calls
def geos_touches(contour, i, j):
    """streamer_index.py:189-198: the chord of a base-point pair against the contour it was taken from"""
    from shapely.geometry import LineString
    c = [(float(x), float(y)) for x, y in contour]
    return bool(LineString([c[i], c[j]]).touches(LineString(c)))


def geos_intersects(seg8):
    """streamer_index.py:233-236: two chords (closed segments)"""
    from shapely.geometry import LineString
    s = [float(v) for v in seg8]
    return bool(LineString([(s[0], s[1]), (s[2], s[3])]).intersects(LineString([(s[4], s[5]), (s[6], s[7])])))


def geos_cells(polygon, r, width, nlat):
    """index_utils.py:37-51: lattice points (x in [0, width), y in [0, nlat)) contained in polygon.buffer(r); (x, y) rows
    sorted by (y, x).  With geopandas: the reference's own frame operations; without: shapely.contains_xy on the same
    buffered geometry (the predicate sjoin evaluates)"""
    from shapely.geometry import Polygon
    poly = Polygon([(float(x), float(y)) for x, y in polygon])
    x, y = np.meshgrid(np.arange(0, width), np.arange(0, nlat))
    x, y = x.flatten(), y.flatten()
    try:
        import geopandas as gpd
        import pandas as pd
        points = gpd.GeoDataFrame(pd.DataFrame({"x": x, "y": y}), geometry=gpd.points_from_xy(x, y))
        events = gpd.GeoDataFrame(pd.DataFrame({"k": [0]}), geometry=[poly])
        buffer = events.copy()
        buffer.geometry = buffer.geometry.buffer(r)
        merged = gpd.sjoin(buffer, points, how="inner", predicate="contains").sort_index()
        cx, cy = np.asarray(merged.x, dtype=np.int64), np.asarray(merged.y, dtype=np.int64)
    except ImportError:
        import shapely
        inside = shapely.contains_xy(poly.buffer(r), x.astype(float), y.astype(float))
        cx, cy = x[inside].astype(np.int64), y[inside].astype(np.int64)
    order = np.lexsort((cx, cy))
    return np.c_[cx[order], cy[order]]


def geos_split(polygon, nlon, nlat):
    """index_utils.py:148-166 (split_polys up to the coordinate look-up): exterior rings, in index coordinates and NOT yet
    truncated (:135 does that, the consumer applies it), of the pieces the reference keeps"""
    from shapely.geometry import Polygon
    from shapely.ops import linemerge, polygonize, unary_union
    from shapely.validation import make_valid
    polygon = Polygon([(float(x), float(y)) for x, y in polygon])
    p00, p01 = [nlon - 1, nlat], [nlon - 1, 0]
    p10, p11 = [nlon, 0], [nlon, nlat]
    meridian = Polygon([p00, p01, p10, p11])
    merged = linemerge([polygon.boundary, meridian.boundary])
    borders = unary_union(merged)
    polygons = [p for p in polygonize(borders) if not meridian.contains(p) and make_valid(polygon).contains(p)]
    return [np.asarray(p.exterior.coords.xy).T for p in polygons]


def geos_iou(rings_a, rings_b):
    """events.py:205-213 with buffer = 0: (area A, area B, area of the intersection); a ring set of several rings is a
    MultiPolygon of its parts (what transform_polygons leaves for an event split at the last meridian)"""
    from shapely.geometry import MultiPolygon, Polygon
    from shapely.validation import make_valid

    def geom(rings):
        polys = [Polygon([(float(x), float(y)) for x, y in r]) for r in rings]
        g = polys[0] if len(polys) == 1 else MultiPolygon(polys)
        return make_valid(g.buffer(0))                           # events.py:205-206: .buffer(buffer).make_valid()
    a, b = geom(rings_a), geom(rings_b)
    inter = a.intersection(b)
    return float(a.area), float(b.area), float(inter.area)


def versions():
    import shapely
    v = {"shapely": shapely.__version__, "geos": getattr(shapely, "geos_version_string", "?")}
    try:
        import geopandas
        v["geopandas"] = geopandas.__version__
    except ImportError:
        v["geopandas"] = ""
    return v


# ------------------------------------------------------------------------------------------------ plumbing
def _ragged(arrays, dtype):
    """list of (n_k, 2) arrays -> (offsets[n + 1], rows[sum n_k, 2])"""
    off = np.zeros(len(arrays) + 1, dtype=np.int64)
    for k, a in enumerate(arrays):
        off[k + 1] = off[k] + len(a)
    rows = np.concatenate([np.asarray(a, dtype=dtype).reshape(-1, 2) for a in arrays]) if arrays else np.zeros((0, 2), dtype=dtype)
    return off, rows


def build(out=OUT, quick=False):
    from tests import geos_cases as gc
    d = {}
    # touches
    contours, pair_off, pairs, res = [], [0], [], []
    for c, ij in gc.touches_cases(2026, 40 if quick else 220, 20 if quick else 60):
        contours.append(c)
        for i, j in ij:
            pairs.append((i, j))
            res.append(geos_touches(c, int(i), int(j)))
        pair_off.append(len(pairs))
    d["touches_contour_offset"], d["touches_contour_xy"] = _ragged(contours, np.int64)
    d["touches_pair_offset"] = np.asarray(pair_off, dtype=np.int64)
    d["touches_pairs"] = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    d["touches_result"] = np.asarray(res, dtype=bool)
    # intersects
    s = gc.intersect_cases(13, 2000 if quick else 14000)
    d["inter_segments"] = np.asarray(s, dtype=np.int64)
    d["inter_result"] = np.asarray([geos_intersects(r) for r in s], dtype=bool)
    # cells
    polys, meta, cells = [], [], []
    for poly, r, width, nlat in gc.cell_cases(quick):
        polys.append(np.asarray(poly, dtype=np.int64))
        meta.append((r, width, nlat))
        cells.append(geos_cells(poly, r, width, nlat))
    d["cells_poly_offset"], d["cells_poly_xy"] = _ragged(polys, np.int64)
    d["cells_meta"] = np.asarray(meta, dtype=np.float64).reshape(-1, 3)
    d["cells_offset"], d["cells_xy"] = _ragged(cells, np.int64)
    # split
    polys, meta, ring_count, rings = [], [], [0], []
    for poly, nlon, nlat in gc.split_cases(quick):
        polys.append(np.asarray(poly, dtype=np.int64))
        meta.append((nlon, nlat))
        pieces = geos_split(poly, nlon, nlat)
        rings.extend(pieces)
        ring_count.append(len(rings))
    d["split_poly_offset"], d["split_poly_xy"] = _ragged(polys, np.int64)
    d["split_meta"] = np.asarray(meta, dtype=np.int64).reshape(-1, 2)
    d["split_ring_count"] = np.asarray(ring_count, dtype=np.int64)
    d["split_ring_offset"], d["split_ring_xy"] = _ragged(rings, np.float64)
    # intersection over union
    ra, rb, ca, cb, areas = [], [], [0], [0], []
    for a, b in gc.iou_cases(quick):
        ra.extend(a); ca.append(len(ra))
        rb.extend(b); cb.append(len(rb))
        areas.append(geos_iou(a, b))
    d["iou_a_count"], d["iou_b_count"] = np.asarray(ca, dtype=np.int64), np.asarray(cb, dtype=np.int64)
    d["iou_a_offset"], d["iou_a_xy"] = _ragged(ra, np.float64)
    d["iou_b_offset"], d["iou_b_xy"] = _ragged(rb, np.float64)
    d["iou_areas"] = np.asarray(areas, dtype=np.float64).reshape(-1, 3)
    v = versions()
    for k in ("shapely", "geos", "geopandas"):
        d["version_" + k] = np.asarray(v[k])
    np.savez_compressed(out, **d)
    return d


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=OUT)
    ap.add_argument("--quick", action="store_true", help="a tenth of the cases (a first look)")
    a = ap.parse_args()
    try:
        import shapely
    except ImportError:
        raise SystemExit("make_golden_geos.py needs shapely >= 2.0.1 (pip install shapely geopandas); it is not importable here")
    if tuple(int(x) for x in shapely.__version__.split(".")[:2]) < (2, 0):
        raise SystemExit("make_golden_geos.py needs shapely >= 2.0.1, found " + shapely.__version__)
    d = build(a.out, a.quick)
    print("wrote %s: %d chords, %d segment pairs, %d cell sets, %d split polygons, %d overlap pairs; shapely %s, GEOS %s, geopandas %s"
          % (a.out, len(d["touches_result"]), len(d["inter_result"]), len(d["cells_meta"]), len(d["split_meta"]), len(d["iou_areas"]),
             d["version_shapely"], d["version_geos"], d["version_geopandas"] or "absent"))
