"""Index helpers -- mirror of wavebreaking/utils/index_utils.py.

``calculate_properties`` and the polygon work are split between the device (cells and area-weighted sums,
csrc/wb_props.h) and this module (the few floating-point operations per event that turn the sums into
the reference's columns, and the lon/lat polygons).
"""
import functools

import numpy as np

from wavebreaking_amd import gridfield as _gf


def properties_from_sums(sums, lon, lat, nlon):
    """index_utils.py:104-120 from the five per-event sums (areas, mean_var, intensity, x_com, y_com)."""
    sums = np.asarray(sums, dtype=np.float64).reshape(-1, 5)
    areas = sums[:, 0]
    com_x = np.asarray(lon)[(sums[:, 3] / areas).astype("int") % nlon]
    com_y = np.asarray(lat)[(sums[:, 4] / areas).astype("int")]
    com = list(map(tuple, np.c_[com_x, com_y]))
    return com, np.round(sums[:, 1] / areas, 2), np.round(sums[:, 2] / areas, 2), np.round(areas, 2)


def _clip_halfplane(ring, L, keep_le):
    """Pieces of a closed ring on one side of the vertical line x = L (see transform_polygons)."""
    v = np.asarray(ring, dtype=np.float64)
    n = len(v)
    ins = (v[:, 0] <= L) if keep_le else (v[:, 0] >= L)
    if ins.all():
        return [v]
    if not ins.any():
        return []
    s = int(np.argmin(ins))
    chains, cur = [], None
    for k in range(n):
        ia, ib = (s + k) % n, (s + k + 1) % n
        a, b = v[ia], v[ib]
        if not ins[ia] and ins[ib]:
            t = (L - a[0]) / (b[0] - a[0])
            cur = [np.array([L, a[1] + t * (b[1] - a[1])]), b.copy()]
        elif ins[ia] and ins[ib]:
            cur.append(b.copy())
        elif ins[ia] and not ins[ib]:
            t = (L - a[0]) / (b[0] - a[0])
            cur.append(np.array([L, a[1] + t * (b[1] - a[1])]))
            chains.append(cur)
            cur = None
    pts = []
    for ci, ch in enumerate(chains):
        pts.append((ch[0][1], 0, ci))
        pts.append((ch[-1][1], 1, ci))
    pts.sort(key=lambda q: (q[0], q[1]))
    partner = {}
    for k in range(0, len(pts) - 1, 2):
        partner[(pts[k][1], pts[k][2])] = (pts[k + 1][1], pts[k + 1][2])
        partner[(pts[k + 1][1], pts[k + 1][2])] = (pts[k][1], pts[k][2])
    used = [False] * len(chains)
    pieces = []
    for c0 in range(len(chains)):
        if used[c0]:
            continue
        piece, c, guard = [], c0, 0
        while not used[c] and guard <= len(chains):
            used[c] = True
            piece.extend(chains[c])
            c = partner.get((1, c), (0, c))[1]
            guard += 1
        piece = np.asarray(piece)
        x, y = piece[:, 0], piece[:, 1]
        if len(piece) >= 3 and abs(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y)) > 0:
            pieces.append(piece)
    return pieces


def split_index_polygon(polygon, nlon):
    """index_utils.py:148-166: the index polygon without the strip nlon - 1 < x < nlon, as float rings in index coordinates"""
    coords = np.asarray(polygon).astype("int")
    return _clip_halfplane(coords, nlon - 1, True) + _clip_halfplane(coords, nlon, False)


def transform_polygon_rings(polygon, lon, lat, nlon):
    """One event: index polygon -> list of (lon, lat) rings (index_utils.py:129-173).

    A polygon reaching x >= nlon is split at the last meridian: the strip nlon-1 < x < nlon (between
    lon[-1] and the repeated lon[0]) is removed and what lies right of it is shifted by -nlon.  The
    reference does this with GEOS (linemerge / unary_union / polygonize / make_valid, :148-173); here it
    is an exact vertical-line clip of the integer polygon (GEOS absent -> unpinned, areas compared)."""
    coords = np.asarray(polygon).astype("int")

    def to_lonlat(c):
        c = np.asarray(c).astype("int")                                   # :135 truncation
        return np.c_[np.asarray(lon)[c[:, 0] % nlon], np.asarray(lat)[c[:, 1]]]   # :139-146

    if not (coords[:, 0] >= nlon).any():                                   # :136
        return [to_lonlat(coords)]
    return [to_lonlat(p) for p in split_index_polygon(coords, nlon)]


def transform_polygons(polygons, lon, lat, nlon):
    """index_utils.py:129-184: geometry objects for a list of index polygons."""
    return [_gf.make_polygon(transform_polygon_rings(p, lon, lat, nlon)) for p in polygons]


def combine_shared(lst):
    """Merge the lists that share at least one element, transitively (index_utils.py:187-214): connected components
    of the "shares an element" relation by union-find over the elements; components come out in the order of their
    first list, each as a list of its distinct elements.  (The index kernels do the same by label propagation,
    csrc/wb_indices.h; this host version exists because the reference's tests call it, tests/test_wavebreaking.py:101-104.)"""
    groups = [list(g) for g in lst]
    parent = {}

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for g in groups:
        for x in g:
            parent.setdefault(x, x)
        for x in g[1:]:
            ra, rb = find(g[0]), find(x)
            if ra != rb:
                parent[rb] = ra
    merged, slot = [], {}
    for g in groups:
        if not g:
            continue
        root = find(g[0])
        if root not in slot:
            slot[root] = len(merged)
            merged.append([])
    for x in parent:                                   # elements in first-seen order
        merged[slot[find(x)]].append(x)
    return merged


def _iterate_over(keyword, values_of):
    """decorator factory behind the two iteration decorators: call the function once per value with kwargs[keyword] set
    and stack the frames it returns (index_utils.py:217-258).  The HIP path does not loop -- every (time step, level) of
    a call is one job of a batched launch -- these exist because the reference's tests use them as public helpers."""

    def decorator(func):
        @functools.wraps(func)
        def wrapper(data, contour_levels, *args, **kwargs):
            import pandas as pd
            frames = []
            for value in values_of(data, contour_levels, kwargs):
                frames.append(func(data, contour_levels, *args, **dict(kwargs, **{keyword: value})))
            return pd.concat(frames).reset_index(drop=True)

        return wrapper

    return decorator


def _levels_of(data, contour_levels, kwargs):
    try:
        return list(contour_levels)
    except TypeError:
        return [contour_levels]


iterate_time_dimension = _iterate_over("step", lambda data, contour_levels, kwargs: data[kwargs["time_name"]])
iterate_time_dimension.__doc__ = "decorator: once per time step, kwargs['step'] set (index_utils.py:217-235)"
iterate_contour_levels = _iterate_over("level", _levels_of)
iterate_contour_levels.__doc__ = "decorator: once per contour level, kwargs['level'] set (index_utils.py:238-258)"
