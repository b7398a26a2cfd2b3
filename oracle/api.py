"""End-to-end oracle entry points on plain arrays (TEST INFRASTRUCTURE).

Mirrors the four public functions of the reference
(wavebreaking/__init__.py:31-34) without xarray/geopandas: inputs are
``data[time, lat, lon]`` + coordinate vectors, outputs are lists of dict rows
with the reference's column names in the reference's column order
(tests/test_wavebreaking.py:175,208-211,225-237,250-253).
"""
import numpy as np

from . import contours as _c
from . import indices as _i
from . import properties as _p

CONTOUR_COLUMNS = ["date", "level", "closed", "exp_lon", "mean_lat", "geometry"]
EVENT_COLUMNS = ["date", "level", "com", "mean_var", "intensity", "event_area", "geometry"]
OVERTURNING_COLUMNS = EVENT_COLUMNS[:-1] + ["orientation", "geometry"]


def correct_orientation(data, lat, lon, intensity=None):
    """wavebreaking/utils/data_utils.py:196-213: sort lat / lon ascending."""
    lat = np.asarray(lat)
    lon = np.asarray(lon)
    if np.average(np.diff(lat)) < 0:
        o = np.argsort(lat, kind="stable")
        lat, data = lat[o], data[:, o, :]
        intensity = None if intensity is None else intensity[:, o, :]
    if np.average(np.diff(lon)) < 0:
        o = np.argsort(lon, kind="stable")
        lon, data = lon[o], data[:, :, o]
        intensity = None if intensity is None else intensity[:, :, o]
    return data, lat, lon, intensity


def calculate_contours(data, contour_levels, lat, lon, times, periodic_add=120, original_coordinates=True):
    data, lat, lon, _ = correct_orientation(np.asarray(data), lat, lon)
    return _c.calculate_contours(data, contour_levels, lat, lon, times, periodic_add, original_coordinates)


def _finish(events, data, intensity, lat, lon, periodic_add):
    rows = _p.calculate_properties(events, data, intensity, lat, lon, periodic_add)
    for row, ev in zip(rows, events):
        row["geometry"] = _p.transform_polygon(ev["polygon"], lat, lon)
        row["_polygon"] = ev["polygon"]
    return rows


def calculate_streamers(data, contour_levels, lat, lon, times, contours=None, geo_dis=800, cont_dis=1500,
                        intensity=None, periodic_add=120, stats=None):
    data, lat, lon, intensity = correct_orientation(np.asarray(data), lat, lon, intensity)
    if contours is None:
        contours = _c.calculate_contours(data, contour_levels, lat, lon, times, periodic_add, False)
    ev = _i.calculate_streamers(contours, lat, lon, geo_dis, cont_dis, stats)
    return _finish(ev, data, intensity, lat, lon, periodic_add)


def calculate_overturnings(data, contour_levels, lat, lon, times, contours=None, range_group=5, min_exp=5,
                           intensity=None, periodic_add=120):
    data, lat, lon, intensity = correct_orientation(np.asarray(data), lat, lon, intensity)
    if contours is None:
        contours = _c.calculate_contours(data, contour_levels, lat, lon, times, periodic_add, False)
    ev = _i.calculate_overturnings(contours, lat, lon, range_group, min_exp)
    return _finish(ev, data, intensity, lat, lon, periodic_add)


def calculate_cutoffs(data, contour_levels, lat, lon, times, contours=None, min_exp=5, intensity=None,
                      periodic_add=120):
    data, lat, lon, intensity = correct_orientation(np.asarray(data), lat, lon, intensity)
    if contours is None:
        contours = _c.calculate_contours(data, contour_levels, lat, lon, times, periodic_add, False)
    ev = _i.calculate_cutoffs(contours, min_exp)
    return _finish(ev, data, intensity, lat, lon, periodic_add)


def run_all(data, contour_levels, lat, lon, times, intensity=None, periodic_add=120, geo_dis=800,
            cont_dis=1500, range_group=5, min_exp=5, stats=None):
    """Contours once, then all three indices (what bench.py's cpu_baseline leg times)."""
    data, lat, lon, intensity = correct_orientation(np.asarray(data), lat, lon, intensity)
    contours = _c.calculate_contours(data, contour_levels, lat, lon, times, periodic_add, False)
    return dict(
        contours=contours,
        streamers=_finish(_i.calculate_streamers(contours, lat, lon, geo_dis, cont_dis, stats),
                          data, intensity, lat, lon, periodic_add),
        overturnings=_finish(_i.calculate_overturnings(contours, lat, lon, range_group, min_exp),
                             data, intensity, lat, lon, periodic_add),
        cutoffs=_finish(_i.calculate_cutoffs(contours, min_exp), data, intensity, lat, lon, periodic_add),
    )
