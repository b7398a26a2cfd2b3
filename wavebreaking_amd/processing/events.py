"""Event post-processing with the call surface of wavebreaking/processing/events.py: ``to_xarray`` (:37-110; the flagging
runs in csrc/wb_flags.h, wb_flag_events) and ``track_events`` (:113-241; pairs, link test, components and labels run in
csrc/wb_track.h, wb_track_events)."""
import logging

import numpy as np

from wavebreaking_amd import engine
from wavebreaking_amd import gridfield as _gf
from wavebreaking_amd.engine import _time_key
from wavebreaking_amd.utils.data_utils import check_argument_types, check_empty_dataframes, get_dimension_attributes

_DA = _gf.data_array_types()

logger = logging.getLogger(__name__)


def _rings(geom):
    """Exterior and interior rings of a (Multi)Polygon -- shapely's or this package's light classes -- as (n, 2) arrays."""
    if geom is None or getattr(geom, "is_empty", False):
        return []
    out = []
    for part in (geom.geoms if hasattr(geom, "geoms") else [geom]):
        if getattr(part, "is_empty", False):
            continue
        out.append(np.asarray(part.exterior.coords, dtype=np.float64).reshape(-1, 2))
        for hole in getattr(part, "interiors", []):
            out.append(np.asarray(hole.coords, dtype=np.float64).reshape(-1, 2))
    return out


@check_argument_types(["data", "events"], [_DA, _gf.frame_types()])
@check_empty_dataframes
@get_dimension_attributes("data")
def to_xarray(data, events, flag="ones", name="flag", *args, **kwargs):
    """
    Create a gridded flag field from events stored in a (Geo)DataFrame (events.py:37-110).
    Grid cells where an event is present are flagged with the value 1 (int8) or, with ``flag`` naming a column
    of ``events``, with that column's value (dtype of ``data``).

    Parameters
    ----------
        data : xarray.DataArray / GridField
            data used for the index calculation
        events : geopandas.GeoDataFrame (pandas.DataFrame with a geometry column when geopandas is absent)
            date and geometry of each event
        flag : string, optional
            column of ``events`` whose value is written where an event is present; "ones" (default) writes 1
        name : string, optional
            name of the result

    Returns
    -------
        flag field with the dimensions and coordinates of ``data``
    """
    tn, la, lo = kwargs["time_name"], kwargs["lat_name"], kwargs["lon_name"]
    times = np.asarray(data[tn].values)
    lat = np.asarray(data[la].values, dtype=np.float64)
    lon = np.asarray(data[lo].values, dtype=np.float64)
    if flag == "ones":
        values, dtype = None, np.dtype(np.int8)
    else:
        if flag not in events.columns:
            raise KeyError("{} is not a column of the events geopandas.GeoDataFrame.".format(flag))
        values = np.asarray(events[flag].values, dtype=np.float64)
        dtype = np.dtype(np.asarray(data.values).dtype)
        if dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            dtype = np.dtype(np.float64)
    tkey = {_time_key(t): i for i, t in enumerate(times)}
    steps = []
    for d in events["date"]:
        k = _time_key(d)
        if k not in tkey:
            raise KeyError("event date {} not found in the time coordinate of data".format(d))
        steps.append(tkey[k])
    rings = [_rings(g) for g in events["geometry"]]
    # the kernel wants ascending coordinates; flip descending axes (ERA5 latitudes) around the call
    flip_lat, flip_lon = lat.size > 1 and lat[0] > lat[-1], lon.size > 1 and lon[0] > lon[-1]
    radius = ((kwargs["dlon"] + kwargs["dlat"]) / 2) / 2                      # events.py:76-78
    cube = engine.get_context().flag_events(len(times), lat[::-1] if flip_lat else lat, lon[::-1] if flip_lon else lon,
                                            float(radius), steps, rings, values, dtype=dtype)
    if flip_lat:
        cube = cube[:, ::-1, :]
    if flip_lon:
        cube = cube[:, :, ::-1]
    order = [(tn, la, lo).index(d) for d in data.dims]
    cube = np.ascontiguousarray(np.transpose(cube, order))
    attrs = dict(data.attrs)
    attrs["long_name"] = "flag wave breaking"
    if _gf._xr is not None and isinstance(data, _gf._xr.DataArray):           # pragma: no cover - xarray not in the image
        return _gf._xr.DataArray(cube, coords=[data[d] for d in data.dims], dims=data.dims, name=name, attrs=attrs)
    return _gf.GridField(cube, data.dims, dict(data.coords), attrs=attrs, name=name)


def _pandas_frame_type():
    import pandas as pd
    return pd.DataFrame


def _float_gcd(values, tol=1e-6):
    """largest step of which every value is a whole multiple (within tol): Euclid's algorithm on floats"""
    step = 0.0
    for v in values:
        a, b = max(step, abs(float(v))), min(step, abs(float(v)))
        while b > tol * a:
            a, b = b, a % b
            if b > a * (1 - tol):                      # a remainder one rounding error short of the divisor
                b = 0.0
        step = a
    return step


def _axis_lattice(coords, given=None):
    """(origin, step) of the regular grid the coordinates lie on: from the grid the frame carries, else inferred as the
    greatest common step of the distinct values"""
    if given is not None:
        return float(given[0]), float(given[1])
    u = np.unique(coords)
    if u.size < 2:
        return float(u[0]) if u.size else 0.0, 1.0
    return float(u[0]), _float_gcd(np.diff(u))


def lattice_rings(events, with_steps=False):
    """The event polygons as integer rings on the data's grid (``with_steps``: and the grid's (dlon, dlat)).  transform_polygons (index_utils.py:129-184) puts every
    vertex on a grid point (it truncates to indices, :135), so this loses nothing for frames the index functions made;
    the grid comes from ``events.attrs["wavebreaking_grid"]`` (set by this package's index functions) or is inferred from
    the coordinates.  Raises ValueError for polygons that are not on a regular grid."""
    rings = [_rings(g) for g in events["geometry"]]
    flat = [r for rs in rings for r in rs]
    if not flat:
        return ([[] for _ in rings], (1.0, 1.0)) if with_steps else [[] for _ in rings]
    steps = []
    allv = np.concatenate(flat)
    grid = getattr(events, "attrs", {}).get("wavebreaking_grid")
    out_axes = []
    for axis, key in ((0, "lon"), (1, "lat")):
        origin, step = _axis_lattice(allv[:, axis], None if grid is None else grid[key])
        if not step > 0:
            raise ValueError("track_events(method='by_overlap'): cannot find the grid step of the polygon coordinates")
        idx = (allv[:, axis] - origin) / step
        if np.abs(idx - np.rint(idx)).max() > 1e-6 or np.abs(idx).max() >= 32768:
            raise ValueError("track_events(method='by_overlap') needs polygons whose vertices lie on a regular grid "
                             "(as the index functions produce them); these do not")
        out_axes.append(np.rint(idx).astype(np.int32))
        steps.append(step)
    xy = np.c_[out_axes[0], out_axes[1]]
    out, pos = [], 0
    for rs in rings:
        mine = []
        for r in rs:
            mine.append(xy[pos:pos + len(r)])
            pos += len(r)
        out.append(mine)
    return (out, (steps[0], steps[1])) if with_steps else out


@check_argument_types(["events"], [_pandas_frame_type()])
@check_empty_dataframes
def track_events(events, time_range=None, method="by_overlap", buffer=0, overlap=0, distance=1000):
    """
    Temporal tracking of events (events.py:113-241).
    Events receive the same label if they spatially overlap at step t
    and t + time_range.

    Parameters
    ----------
        events : geopandas.GeoDataFrame (pandas.DataFrame with a geometry column when geopandas is absent)
            date and coordinates of each identified event
        time_range: int or float, optional
            Time range for temporally tracking the events, in hours if the dates are np.datetime64.
            If not specified, the smallest time difference larger than zero is used.
        method : {"by_overlap", "by_distance"}, optional
            * "by_overlap": same label if the polygons overlap at step t and t + time_range
            * "by_distance": same label if the centres of mass are closer than "distance"
        buffer : float, optional
            buffer around the event polygons in degrees for "by_overlap".  0: exact intersection-over-union of the grid
            polygons.  > 0 with ``overlap=0`` (the default): a pair is linked when its buffered polygons overlap at all,
            i.e. when the polygons are closer than 2 * buffer degrees (GEOS' round buffer taken as a disc: pairs within
            0.24 % of that distance, where GEOS' inscribed arcs may fall short, are counted in
            ``attrs["track_stats"]["pairs_in_arc_band"]``).  > 0 with ``overlap > 0`` needs the AREA of buffered polygons
            (GEOS' arc approximation) and raises NotImplementedError, as does a negative buffer
        overlap : float, optional
            minimum intersection-over-union for "by_overlap"
        distance : int or float, optional
            maximum distance in km between two events for "by_distance"

    Returns
    -------
        events: the frame with a ``label`` column, sorted by label and date
    """
    events = events.reset_index(drop=True)
    dates = np.asarray(events["date"].values)
    is_time = np.issubdtype(dates.dtype, np.datetime64)
    key = dates.astype("datetime64[ns]").astype(np.int64) if is_time else dates.astype(np.float64)
    if time_range is None:                                                   # events.py:151-153
        dif = np.diff(key)
        dif = dif[dif > 0]
        if dif.size == 0:
            raise ValueError("No events detected in the time range: {}".format(np.nan))
        time_range = float(dif.min() / 1e9 / 3600) if is_time else float(dif.min())
    gaps = np.diff(np.unique(key))                                           # a pair in range exists iff the smallest gap fits
    gaps = gaps / 1e9 / 3600 if is_time else gaps
    if gaps.size == 0 or not (gaps.min() <= time_range):
        raise ValueError("No events detected in the time range: {}".format(time_range))   # events.py:179-181
    if method not in ("by_overlap", "by_distance"):                          # events.py:219-222
        raise ValueError("'{}' not supported as method!".format(method) + " Supported methods are 'by_overlap' and 'by_distance'")
    ctx = engine.get_context()
    if method == "by_distance":
        com = np.asarray(list(events["com"]), dtype=np.float64).reshape(len(events), 2)
        label, _, stats = ctx.track_events(dates, method, time_range, distance=distance, com=com)
    else:
        if buffer < 0 or (buffer > 0 and overlap != 0):
            raise NotImplementedError("track_events(method='by_overlap'): buffer > 0 is built for overlap=0 only (do the buffered "
                                      "polygons overlap at all = are the polygons closer than 2 * buffer); a ratio of AREAS of "
                                      "buffered polygons needs GEOS' arc approximation, a negative buffer its erosion")
        if buffer > 0:                                                      # events.py:203-214 with overlap == 0
            rings, steps = lattice_rings(events, with_steps=True)
            label, stats = ctx.track_events_buffered(dates, time_range, buffer, steps, rings_per_event=rings)
            if stats["pairs_in_arc_band"]:
                logger.warning("track_events(buffer=%g): %d pairs lie within 0.24 %% of the linking distance, where GEOS' polygonal "
                               "arcs can decide differently from the disc used here" % (buffer, stats["pairs_in_arc_band"]))
        else:
            label, _, stats = ctx.track_events(dates, method, time_range, overlap=overlap, rings_per_event=lattice_rings(events))
        if stats["polygons_not_simple"]:
            # the reference repairs such geometries with buffer(0).make_valid() before it intersects them (events.py:203-204):
            # GEOS, not restated here -- their regions are taken by the even-odd rule and their labels may differ
            logger.warning("track_events(method='by_overlap'): %d event polygons cross themselves; the reference repairs them with "
                           "buffer(0).make_valid() (GEOS, not reproduced) -- the labels of these events may differ"
                           % stats["polygons_not_simple"])
    events["label"] = label.astype(np.int64)                                 # events.py:224-238
    order = np.lexsort((key, label))                                         # events.py:241: stable, label then date
    out = events.iloc[order]
    out.attrs["track_stats"] = stats
    return out
