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
    """Merge the lists that share at least one element, transitively (index_utils.py:187-214).  Components come out in the
    order of their first list, and the members of a component in the order the reference returns them: ``list(set)`` of a
    CPython set that was built by exactly these operations -- the set of the first list, then, sweep after sweep over the
    lists not yet absorbed, an in-place union with the set of every list that meets it (streamer_index.py:241-246 picks a
    group's survivor by position in this list, so the order is part of the result: ``[[1, 8]] -> [[8, 1]]``).  Pinned
    against the reference's own function by tests/golden/reference_rules.npz.  (The index kernels restate the same
    insertion sequence on the device, csrc/wb_core.h:wb_pyset_*.)"""
    waiting = list(lst)
    components = []
    while waiting:
        grown = set(waiting[0])
        others = waiting[1:]
        size = -1
        while len(grown) > size:                       # until a sweep adds nothing
            size = len(grown)
            apart = []
            for member in others:
                as_set = set(member)
                if grown.intersection(as_set):
                    grown |= as_set
                else:
                    apart.append(member)
            others = apart
        components.append(list(grown))
        waiting = others
    return components


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
