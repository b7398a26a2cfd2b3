"""Spatial pre-processing -- same names, arguments and error behaviour as wavebreaking/processing/spatial.py;
the arithmetic runs in the HIP kernels of csrc/wb_spatial.h (wb_smooth_field / wb_momentum_flux), bit-identical to
the scipy.ndimage.convolve / numpy nanmean evaluation the reference performs with numpy >= 2.
"""
import numpy as np

from wavebreaking_amd import engine
from wavebreaking_amd import gridfield as _gf
from wavebreaking_amd.utils.data_utils import check_argument_types, get_dimension_attributes

_DA = _gf.data_array_types()


def _cube(data, kwargs):
    """(time, lat, lon) numpy view of ``data`` + the permutation back to its own dimension order."""
    order = [data.dims.index(kwargs[k]) for k in ("time_name", "lat_name", "lon_name")]
    values = np.transpose(np.asarray(data.values), order)
    return values, list(np.argsort(order))


def _like(data, values, name, attrs):
    """A labelled array with ``data``'s dimensions and coordinates around ``values``."""
    if _gf._xr is not None and isinstance(data, _gf._xr.DataArray):  # pragma: no cover - xarray is not in the build image
        return _gf._xr.DataArray(values, coords=[data[d] for d in data.dims], dims=data.dims, name=name, attrs=attrs)
    return _gf.GridField(values, data.dims, dict(data.coords), attrs=attrs, name=name)


def _time_chunks(nt):
    step = engine.max_steps_per_batch()
    return [(t0, min(nt, t0 + step)) for t0 in range(0, nt, step)]


@check_argument_types(["u", "v"], [_DA, _DA])
@get_dimension_attributes("u")
def calculate_momentum_flux(u, v, *args, **kwargs):
    """
    Calculate the momentum flux derived from the product of the deviations
    of both wind components from the zonal mean (spatial.py:27-57).

    Parameters
    ----------
        u : xarray.DataArray / GridField
            zonal (x) component of the wind
        v : xarray.DataArray / GridField
            meridional (y) component of the wind

    Returns
    -------
        momentum flux, named "mflux", with the dimensions of ``u``

    The zonal mean is evaluated as numpy does for a contiguous longitude axis (pairwise summation in the array dtype,
    NaNs skipped); cubes are brought to (time, lat, lon) order for it.
    """
    uu, back = _cube(u, kwargs)
    order_v = [v.dims.index(kwargs[k]) for k in ("time_name", "lat_name", "lon_name")]
    vv = np.transpose(np.asarray(v.values), order_v)
    if uu.shape != vv.shape:
        raise ValueError("u and v must have the same shape")
    dt = np.result_type(uu.dtype, vv.dtype)
    if dt not in (np.float32, np.float64):
        dt = np.dtype(np.float64)
    ctx = engine.get_context()
    out = np.empty(uu.shape, dtype=dt)
    for t0, t1 in _time_chunks(uu.shape[0]):
        out[t0:t1] = ctx.momentum_flux(np.ascontiguousarray(uu[t0:t1], dtype=dt), np.ascontiguousarray(vv[t0:t1], dtype=dt))
    return _like(u, np.transpose(out, back), "mflux", {})


@check_argument_types(["data"], [_DA])
@get_dimension_attributes("data")
def calculate_smoothed_field(data, passes, weights=np.array([[0, 1, 0], [1, 2, 1], [0, 1, 0]]), mode="wrap", *args, **kwargs):
    """
    Calculate smoothed field based on a two-dimensional weight kernel
    and multiple smoothing passes (spatial.py:60-128). Default weight kernel is a 3x3
    5-point smoothing with double-weighted centre. "weights" and "mode" are those of
    scipy.ndimage.convolve. Values at the latitude border are always set to NaN.

    Parameters
    ----------
        data : xarray.DataArray / GridField
            data to smooth
        passes : int
            number of smoothing passes
        weights : array_like, optional
            two-dimensional array of weights (up to 7 x 7)
        mode : string, optional
            boundary extension: "wrap" (default), "reflect", "mirror", "nearest", "constant"

    Returns
    -------
        smoothed data, (time, lat, lon), float64, named "smooth_" + data.name, attrs of ``data`` + "smooth_passes"
    """
    weights = np.asarray(weights)
    values, _ = _cube(data, kwargs)
    if values.dtype not in (np.float32, np.float64):
        values = values.astype(np.float64)
    npass = len(range(passes))                                  # the reference's loop count; TypeError for a float
    border = int(weights.shape[0] / 2 + 0.5)                    # spatial.py:105
    rows = np.arange(-border, border)
    if npass == 0:                                              # nothing to compute: the field with its border rows blanked
        out = np.array(values, copy=True)
        out[:, rows, :] = np.nan
    else:
        ctx = engine.get_context()
        out = np.empty(values.shape, dtype=np.float64)
        for t0, t1 in _time_chunks(values.shape[0]):
            out[t0:t1] = ctx.smooth_field(values[t0:t1], npass, weights, mode)
    attrs = dict(data.attrs)
    attrs["smooth_passes"] = passes
    name = "smooth_" + data.name                                # TypeError for an unnamed array, as in the reference
    tn, la, lo = kwargs["time_name"], kwargs["lat_name"], kwargs["lon_name"]
    if _gf._xr is not None and isinstance(data, _gf._xr.DataArray):  # pragma: no cover
        return _gf._xr.DataArray(out, coords=[data[tn], data[la], data[lo]], name=name, attrs=attrs)
    return _gf.GridField(out, (tn, la, lo), {tn: data.coords[tn], la: data.coords[la], lo: data.coords[lo]}, attrs=attrs,
                         name=name)
