"""calculate_streamers -- mirror of wavebreaking/indices/streamer_index.py:44-289 (Wernli & Sprenger 2007
streamer index).  The N x N distance test, the four base-point filters and the cell sums run in the HIP
kernels wb_k_streamer_* and wb_k_props."""
from wavebreaking_amd import _hip
from wavebreaking_amd import gridfield as _gf
from wavebreaking_amd.indices._events import run_index
from wavebreaking_amd.indices.contour_index import decorator_contour_calculation
from wavebreaking_amd.utils.data_utils import check_argument_types, get_dimension_attributes


@check_argument_types(["data"], [_gf.data_array_types()])
@get_dimension_attributes("data")
@decorator_contour_calculation
def calculate_streamers(data, contour_level, contours=None, geo_dis=800, cont_dis=1500, intensity=None,
                        periodic_add=120, *args, **kwargs):
    """
    Identify streamer structures.  Same parameters, defaults and output columns
    ("date", "level", "com", "mean_var", "intensity", "event_area", "geometry") as the reference.
    """
    if contours is None:
        contours = kwargs["contours"]
    return run_index(_hip.STREAMERS, data, contour_level, contours, intensity, periodic_add,
                     dict(geo_dis=geo_dis, cont_dis=cont_dis), kwargs)
