"""Argument and dimension helpers of the index functions.

Same public names, argument meaning, defaults and error texts as wavebreaking/utils/data_utils.py (the reference's tests
call them directly, tests/test_wavebreaking.py:57-79), organised differently: ONE table describes how each of the three
dimensions is recognised and one resolver walks it.  Works on xarray.DataArray when xarray is importable and on
wavebreaking_amd.GridField otherwise.  The reference's global ``warnings.filterwarnings("ignore")`` (data_utils.py:23-25)
is deliberately not replicated.
"""
import functools

import numpy as np

from wavebreaking_amd import gridfield as _gf

# how a dimension is recognised (data_utils.py:76-126): `units` values that identify it, whether a substring of the units
# is enough, whether the datetime64[ns] dtype identifies it, its conventional names, and the error when nothing matches
_DIMENSIONS = {
    "time": dict(units=("since",), substring=True, datetime=True, names=("time",),
                 error="'time' dimension (dtype='datetime64[ns]') not found."
                       " Add time dimension with xarray.DataArray.expand_dims('time')."),
    "lon": dict(units=("degree_east", "degrees_east"), substring=False, datetime=False, names=("lon", "longitude", "x"),
                error="'longitude' dimension (units='degrees_east') not found."),
    "lat": dict(units=("degree_north", "degrees_north"), substring=False, datetime=False, names=("lat", "latitude", "y"),
                error="latitude' dimension (units='degrees_north') not found."),
}
# keyword <- (kind, what): the eight keywords get_dimension_attributes fills in (data_utils.py:160-176)
_KEYWORDS = (("time_name", "time", "name"), ("lon_name", "lon", "name"), ("lat_name", "lat", "name"),
             ("ntime", "time", "size"), ("nlon", "lon", "size"), ("nlat", "lat", "size"),
             ("dlon", "lon", "step"), ("dlat", "lat", "step"))


def _units_of(coord):
    """every `units` string a coordinate carries (attributes first, then the netCDF encoding)"""
    return [holder["units"] for holder in (coord.attrs, getattr(coord, "encoding", {})) if "units" in holder]


def _find_dimension(data, kind):
    rule = _DIMENSIONS[kind]
    for dim in data.dims:
        coord = data[dim]
        units = _units_of(coord) if rule["substring"] else _units_of(coord)[:1] if "units" in coord.attrs else []
        by_units = any((any(u in text for u in rule["units"]) if rule["substring"] else text in rule["units"]) for text in units)
        by_dtype = rule["datetime"] and coord.dtype == np.dtype("datetime64[ns]")
        if by_units or by_dtype or dim in rule["names"]:
            return dim
    raise ValueError(rule["error"])


def get_time_name(data):
    """name of the time dimension (data_utils.py:76-94)"""
    return _find_dimension(data, "time")


def get_lon_name(data):
    """name of the longitude dimension (data_utils.py:97-110)"""
    return _find_dimension(data, "lon")


def get_lat_name(data):
    """name of the latitude dimension (data_utils.py:113-126)"""
    return _find_dimension(data, "lat")


def get_spatial_resolution(data, dim):
    """the one grid step of a regular coordinate (data_utils.py:129-144)"""
    coord = np.asarray(data[dim].data)
    steps = abs(np.unique(coord[1:] - coord[:-1]))
    if len(steps) > 1:
        raise ValueError("No regular grid found for dimension {}.".format(dim))
    if steps[0] == 0:
        raise ValueError("Two equivalent coordinates found for dimension {}.".format(dim))
    return steps[0]


def check_argument_types(arguments, types):
    """decorator: the named arguments must be instances of the given types (data_utils.py:28-50); an argument is looked up
    by keyword first, else by its position in `arguments`"""
    wanted = list(zip(range(len(arguments)), arguments, types))

    def decorator(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            for position, name, kind in wanted:
                value = kwargs[name] if name in kwargs else args[position]
                if not isinstance(value, kind):
                    shown = kind[0] if isinstance(kind, tuple) else kind
                    raise TypeError(name + " has to be a " + str(shown)[8:-2] + "!")
            return func(*args, **kwargs)

        return wrapper

    return decorator


def check_empty_dataframes(func):
    """decorator: no frame among the arguments may be empty (data_utils.py:53-73)"""
    frames = _gf.frame_types()

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        labelled = [("", a) for a in args] + [(key + " ", a) for key, a in kwargs.items()]
        for label, value in labelled:
            if isinstance(value, frames) and value.empty:
                raise ValueError(label + "geopandas.GeoDataFrame is empty!")
        return func(*args, **kwargs)

    return wrapper


def get_dimension_attributes(arg_name):
    """decorator: fills time_name / lon_name / lat_name, ntime / nlon / nlat and dlon / dlat for the data argument unless
    the caller passed them (data_utils.py:147-193)"""

    def decorator(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            data = kwargs[arg_name] if arg_name in kwargs else args[0]
            for keyword, kind, what in _KEYWORDS:
                if keyword in kwargs:
                    continue
                if what == "name":
                    kwargs[keyword] = _find_dimension(data, kind)
                elif what == "size":
                    kwargs[keyword] = len(data[kwargs[kind + "_name"]])
                else:
                    kwargs[keyword] = get_spatial_resolution(data, kwargs[kind + "_name"])
            if len(data.dims) > 3:
                known = {kwargs[kind + "_name"] for kind in _DIMENSIONS}
                raise ValueError("Unexpected dimension(s): {}. Select dimensions first.".format(
                    [dim for dim in data.dims if dim not in known]))
            return func(*args, **kwargs)

        return wrapper

    return decorator


def correct_dimension_orientation(data, *args, **kwargs):
    """latitude and longitude ascending (data_utils.py:196-213).  Done ONCE per call here; the reference re-sorts the
    cube for every step x level (contour_index.py:87)."""
    for axis in ("lat_name", "lon_name"):
        if np.average(np.diff(np.asarray(data[kwargs[axis]].values))) < 0:
            data = data.sortby(kwargs[axis], ascending=True)
    return data


def as_time_lat_lon(data, **kwargs):
    """numpy view (time, lat, lon) + the three coordinate vectors, whatever the dimension order."""
    order = [data.dims.index(kwargs[k]) for k in ("time_name", "lat_name", "lon_name")]
    values = np.transpose(np.asarray(data.values), order)
    return (values, np.asarray(data[kwargs["time_name"]].values), np.asarray(data[kwargs["lat_name"]].values, dtype=np.float64),
            np.asarray(data[kwargs["lon_name"]].values, dtype=np.float64))
