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


def transform_polygon_rings(polygon, lon, lat, nlon, index_space=False):
    """One event: index polygon -> list of (lon, lat) rings (index_utils.py:129-173).

    A polygon reaching x >= nlon is split at the last meridian: the strip nlon-1 < x < nlon (between lon[-1] and the repeated
    lon[0]) is removed and what lies right of it is shifted by -nlon.  The reference does this with GEOS (linemerge /
    unary_union / polygonize / make_valid, :148-173); here it is the library's exact vertical-line clip of the integer
    polygon (csrc/wb_geometry.h behind ``wb_transform_polygons`` -- the code the index functions' frames come from; GEOS
    absent -> unpinned, checked against an exact-rational splitter in tests/test_seam_split.py)."""
    from wavebreaking_amd import engine
    return engine.get_context().transform_polygons([polygon], lon, lat, nlon, index_space)[0]


def transform_polygons(polygons, lon, lat, nlon):
    """index_utils.py:129-184: geometry objects for a list of index polygons."""
    from wavebreaking_amd import engine
    rings = engine.get_context().transform_polygons(list(polygons), lon, lat, nlon)
    return [_gf.make_polygon(r) for r in rings]


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
            merged.append([])                          # the reference emits an empty group for an empty list (:196-212)
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
            values, contour_levels = values_of(data, contour_levels, kwargs)
            for value in values:
                frames.append(func(data, contour_levels, *args, **dict(kwargs, **{keyword: value})))
            return pd.concat(frames).reset_index(drop=True)

        return wrapper

    return decorator


def _levels_of(data, contour_levels, kwargs):
    # a scalar becomes [scalar] and the wrapped function is handed the LIST, as the reference rebinds it (:246-256)
    try:
        iter(contour_levels)
    except Exception:
        contour_levels = [contour_levels]
    return contour_levels, contour_levels


iterate_time_dimension = _iterate_over("step", lambda data, contour_levels, kwargs: (data[kwargs["time_name"]], contour_levels))
iterate_time_dimension.__doc__ = "decorator: once per time step, kwargs['step'] set (index_utils.py:217-235)"
iterate_contour_levels = _iterate_over("level", _levels_of)
iterate_contour_levels.__doc__ = "decorator: once per contour level, kwargs['level'] set (index_utils.py:238-258)"
