"""calculate_cutoffs -- mirror of wavebreaking/indices/cutoff_index.py:33-104: closed contours with
min_exp <= exp_lon < max(exp_lon) become cutoff polygons."""
from wavebreaking_amd import _hip
from wavebreaking_amd import gridfield as _gf
from wavebreaking_amd.indices._events import run_index
from wavebreaking_amd.indices.contour_index import decorator_contour_calculation
from wavebreaking_amd.utils.data_utils import check_argument_types, get_dimension_attributes


@check_argument_types(["data"], [_gf.data_array_types()])
@get_dimension_attributes("data")
@decorator_contour_calculation
def calculate_cutoffs(data, contour_level, contours=None, min_exp=5, intensity=None, periodic_add=120, *args, **kwargs):
    """
    Identify cutoff structures.  Output columns: "date", "level", "com", "mean_var", "intensity",
    "event_area", "geometry".
    """
    if contours is None:
        contours = kwargs["contours"]
    return run_index(_hip.CUTOFFS, data, contour_level, contours, intensity, periodic_add,
                     dict(co_min_exp=min_exp), kwargs)
