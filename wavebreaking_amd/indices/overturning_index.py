"""calculate_overturnings -- mirror of wavebreaking/indices/overturning_index.py:40-232 (Barnes &
Hartmann 2012 overturning index); per-longitude crossing counts, grouping and filters run in
wb_k_overturn, the cell sums in wb_k_props."""
from wavebreaking_amd import _hip
from wavebreaking_amd import gridfield as _gf
from wavebreaking_amd.indices._events import run_index
from wavebreaking_amd.indices.contour_index import decorator_contour_calculation
from wavebreaking_amd.utils.data_utils import check_argument_types, get_dimension_attributes


@check_argument_types(["data"], [_gf.data_array_types()])
@get_dimension_attributes("data")
@decorator_contour_calculation
def calculate_overturnings(data, contour_levels, contours=None, range_group=5, min_exp=5, intensity=None,
                           periodic_add=120, *args, **kwargs):
    """
    Identify overturning structures.  Output columns: "date", "level", "com", "mean_var", "intensity",
    "event_area", "orientation", "geometry".
    """
    if contours is None:
        contours = kwargs["contours"]
    return run_index(_hip.OVERTURNINGS, data, contour_levels, contours, intensity, periodic_add,
                     dict(range_group=range_group, ot_min_exp=min_exp), kwargs)
