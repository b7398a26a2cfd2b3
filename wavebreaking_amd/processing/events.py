"""Event post-processing -- ``to_xarray`` with the call surface of wavebreaking/processing/events.py:37-110; the
flagging runs in csrc/wb_flags.h (wb_flag_events).  ``track_events`` is not built."""
import numpy as np

from wavebreaking_amd import engine
from wavebreaking_amd import gridfield as _gf
from wavebreaking_amd.engine import _time_key
from wavebreaking_amd.utils.data_utils import check_argument_types, check_empty_dataframes, get_dimension_attributes

_DA = _gf.data_array_types()


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
