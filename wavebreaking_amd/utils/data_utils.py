"""Argument / dimension helpers -- same names, argument meaning and error behaviour as
wavebreaking/utils/data_utils.py (reference lines cited per function).  Works on xarray.DataArray when
xarray is importable and on wavebreaking_amd.GridField otherwise.  The reference's global
``warnings.filterwarnings("ignore")`` (data_utils.py:23-25) is deliberately NOT replicated.
"""
import functools

import numpy as np

from wavebreaking_amd import gridfield as _gf


def _type_name(t):
    if isinstance(t, tuple):
        t = t[0]
    return str(t)[8:-2]


def check_argument_types(arguments, types):
    """decorator to check the type of function arguments (data_utils.py:28-50)"""

    def decorator(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            for (arg_index, arg_name), arg_type in zip(enumerate(arguments), types):
                value = kwargs[arg_name] if arg_name in kwargs else args[arg_index]
                if not isinstance(value, arg_type):
                    errmsg = arg_name + " has to be a " + _type_name(arg_type) + "!"
                    raise TypeError(errmsg)
            return func(*args, **kwargs)

        return wrapper

    return decorator


def check_empty_dataframes(func):
    """decorator to check if there is an empty DataFrame (data_utils.py:53-73)"""

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        frames = _gf.frame_types()
        for item in args:
            if isinstance(item, frames) and item.empty:
                raise ValueError("geopandas.GeoDataFrame is empty!")
        for key, item in kwargs.items():
            if isinstance(item, frames) and item.empty:
                raise ValueError(key + " geopandas.GeoDataFrame is empty!")
        return func(*args, **kwargs)

    return wrapper


def get_time_name(data):
    """check for 'time' dimension and return name (data_utils.py:76-94)"""
    for dim in data.dims:
        if (
            ("units" in data[dim].attrs and "since" in data[dim].attrs["units"])
            or ("units" in data[dim].encoding and "since" in data[dim].encoding["units"])
            or (data[dim].dtype == np.dtype("datetime64[ns]"))
            or (dim in ["time"])
        ):
            return dim
    errmsg = "'time' dimension (dtype='datetime64[ns]') not found."
    hint = " Add time dimension with xarray.DataArray.expand_dims('time')."
    raise ValueError(errmsg + hint)


def get_lon_name(data):
    """check for 'longitude' dimension and return name (data_utils.py:97-110)"""
    for dim in data.dims:
        if (
            "units" in data[dim].attrs and data[dim].attrs["units"] in ["degree_east", "degrees_east"]
        ) or dim in ["lon", "longitude", "x"]:
            return dim
    raise ValueError("'longitude' dimension (units='degrees_east') not found.")


def get_lat_name(data):
    """check for 'latitude' dimension and return name (data_utils.py:113-126)"""
    for dim in data.dims:
        if (
            "units" in data[dim].attrs and data[dim].attrs["units"] in ["degree_north", "degrees_north"]
        ) or dim in ["lat", "latitude", "y"]:
            return dim
    raise ValueError("latitude' dimension (units='degrees_north') not found.")


def get_spatial_resolution(data, dim):
    """check resolution of the longitude and latitude coordinate (data_utils.py:129-144)"""
    coord = np.asarray(data[dim].data)
    delta = abs(np.unique((coord[1:] - coord[:-1])))
    if len(delta) > 1:
        raise ValueError("No regular grid found for dimension {}.".format(dim))
    elif delta[0] == 0:
        raise ValueError("Two equivalent coordinates found for dimension {}.".format(dim))
    return delta[0]


def get_dimension_attributes(arg_name):
    """decorator to get the dimension, size and resolution of the input data (data_utils.py:147-193)"""

    def decorator(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            data = kwargs[arg_name] if arg_name in kwargs else args[0]
            names = ["time_name", "lon_name", "lat_name"]
            sizes = ["ntime", "nlon", "nlat"]
            resolutions = ["dlon", "dlat"]
            get_dims = [get_time_name, get_lon_name, get_lat_name]
            for name, get_dim in zip(names, get_dims):
                if name not in kwargs:
                    kwargs[name] = get_dim(data)
            for size, name in zip(sizes, names):
                if size not in kwargs:
                    kwargs[size] = len(data[kwargs[name]])
            for res, name in zip(resolutions, names[1:]):
                if res not in kwargs:
                    kwargs[res] = get_spatial_resolution(data, kwargs[name])
            if len(data.dims) > 3:
                err_dims = [dim for dim in data.dims if dim not in [kwargs[name] for name in names]]
                raise ValueError("Unexpected dimension(s): {}. Select dimensions first.".format(err_dims))
            return func(*args, **kwargs)

        return wrapper

    return decorator


def correct_dimension_orientation(data, *args, **kwargs):
    """check the orientation of the latitude and longitude dimension (data_utils.py:196-213).
    Done ONCE per call here (the reference re-sorts the cube for every step x level, contour_index.py:87)."""
    wrong_lat_orientation = np.average(np.diff(np.asarray(data[kwargs["lat_name"]].values))) < 0
    wrong_lon_orientation = np.average(np.diff(np.asarray(data[kwargs["lon_name"]].values))) < 0
    if wrong_lat_orientation:
        data = data.sortby(kwargs["lat_name"], ascending=True)
    if wrong_lon_orientation:
        data = data.sortby(kwargs["lon_name"], ascending=True)
    return data


def as_time_lat_lon(data, **kwargs):
    """numpy view (time, lat, lon) + the three coordinate vectors, whatever the dimension order."""
    order = [data.dims.index(kwargs[k]) for k in ("time_name", "lat_name", "lon_name")]
    values = np.transpose(np.asarray(data.values), order)
    return (values, np.asarray(data[kwargs["time_name"]].values), np.asarray(data[kwargs["lat_name"]].values, dtype=np.float64),
            np.asarray(data[kwargs["lon_name"]].values, dtype=np.float64))
