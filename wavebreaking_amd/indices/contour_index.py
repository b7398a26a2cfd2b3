"""calculate_contours / decorator_contour_calculation -- mirror of
wavebreaking/indices/contour_index.py (same signature, defaults, columns, row order and errors); the
per-(step, level) work of :87-156 runs in the HIP kernels wb_k_scan + wb_k_trace."""
import functools
import logging

import numpy as np

from wavebreaking_amd import engine
from wavebreaking_amd import gridfield as _gf
from wavebreaking_amd.utils.data_utils import check_argument_types, get_dimension_attributes

logger = logging.getLogger(__name__)


def _contour_frame(tables, prep, levels, original_coordinates):
    """tables: [(contour table, level positions, first time step)] -> frame in the reference's order (time, level, contour)."""
    rows = []
    for ti, (ct, pos, t0) in enumerate(tables):
        for k in range(ct["step"].size):
            rows.append((int(ct["step"][k]) + t0, pos[int(ct["level_idx"][k])], ti, k))
    rows.sort(key=lambda r: (r[0], r[1]))          # stable: contour order inside a (step, level) is kept
    cols = dict(date=[], level=[], closed=[], exp_lon=[], mean_lat=[])
    geoms = []
    for step, lpos, ti, k in rows:
        ct = tables[ti][0]
        o, n = int(ct["offset"][k]), int(ct["npts"][k])
        x, y = ct["x"][o:o + n], ct["y"][o:o + n]
        cols["date"].append(prep.times[step])
        cols["level"].append(levels[lpos])
        cols["closed"].append(bool(ct["closed"][k]))
        if original_coordinates:                                             # contour_index.py:177-194
            co = np.c_[prep.lon[x % prep.nlon], prep.lat[y]]
            _, first = np.unique(co, axis=0, return_index=True)
            co = co[np.sort(first)]
            cols["exp_lon"].append(len(set(co[:, 0].tolist())) * prep.dlon)
            cols["mean_lat"].append(np.round(co[:, 1].mean(), 2))
            geoms.append(_gf.make_linestring(co))
        else:
            cols["exp_lon"].append(int(ct["nux"][k]) * prep.dlon)            # :154
            cols["mean_lat"].append(np.round(int(ct["sum_y"][k]) / n, 2))    # :155
            geoms.append(_gf.make_linestring(np.c_[x, y]))
    return _gf.make_frame(cols, geoms)


@check_argument_types(["data"], [_gf.data_array_types()])
@get_dimension_attributes("data")
def calculate_contours(data, contour_levels, periodic_add=120, original_coordinates=True, *args, **kwargs):
    """
    Calculate contour lines for a set of contour levels (reference: contour_index.py:45-194).

    Returns a frame with the columns "date", "level", "closed", "exp_lon", "mean_lat", "geometry";
    rows are ordered time-major, then by level, then in scikit-image's contour order.
    """
    keep = kwargs.pop("_keep_batches", None)
    prep = engine.Prepared(data, None, kwargs)
    levels = engine.as_levels(contour_levels)
    batches = engine.contour_batches(prep, levels, periodic_add, {})
    tables = [(b.contour_table(), pos, t0) for b, pos, t0 in batches]
    frame = _contour_frame(tables, prep, levels, original_coordinates)
    if keep is not None:
        keep.update(batches=batches, prep=prep, levels=levels, nrows=len(frame), frame=frame)
    else:
        for b, *_ in batches:
            b.close()
    return frame


def _validate_contour_frame(contours, contour_levels):
    """a contour frame handed in by the caller (contour_index.py:209-245): a non-empty frame in index coordinates; levels
    that are asked for but absent only give a warning"""
    if not isinstance(contours, _gf.frame_types()):
        raise TypeError("contours has to be a geopandas.GeoDataFrame!")
    if contours.empty:
        raise ValueError("contours geopandas.GeoDataFrame is empty!")
    present = set(contours.level)
    missing = [lev for lev in engine.as_levels(contour_levels) if lev not in present]
    if missing:
        logger.warning("\n The contour levels {} are not present in 'contours'".format(missing))
    first = np.asarray(contours.iloc[0].geometry.coords.xy).T
    if not ((first.astype("int") == first).all() and (first[:, 0] >= 0).all()):
        raise ValueError("Original coordinates not supported for the index calculation. "
                         "Use original_coordinates=False in the contour calculation.")


def decorator_contour_calculation(func):
    """decorator to wrap the contour calculation around the index functions (contour_index.py:197-249)"""

    @functools.wraps(func)
    def wrapper(data, contour_levels, periodic_add=120, *args, **kwargs):
        if "contours" not in kwargs:
            dim_kwargs = {k: kwargs[k] for k in ("time_name", "lon_name", "lat_name", "ntime", "nlon", "nlat", "dlon", "dlat")
                          if k in kwargs}
            state = {}
            kwargs["contours"] = calculate_contours(data, contour_levels, periodic_add, original_coordinates=False,
                                                    _keep_batches=state, **dim_kwargs)
            kwargs["_wb_state"] = state
        else:
            _validate_contour_frame(kwargs["contours"], contour_levels)
        return func(data, contour_levels, periodic_add=periodic_add, *args, **kwargs)

    return wrapper
