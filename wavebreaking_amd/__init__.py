"""wavebreaking_amd -- MI355X-native implementation of the WaveBreaking hot path.

Same call surface as the reference package for this path (wavebreaking/__init__.py:31-34):
calculate_contours, calculate_streamers, calculate_overturnings, calculate_cutoffs, plus the two pre-processing
functions calculate_smoothed_field and calculate_momentum_flux, the event-to-grid conversion to_xarray and the temporal
tracking track_events (wavebreaking/__init__.py:20-25) -- xarray.DataArray
(or wavebreaking_amd.GridField) in, GeoDataFrame (or pandas.DataFrame when geopandas is absent) out.
"""
__version__ = "0.1.0"

from wavebreaking_amd.engine import keep_inputs, release_inputs  # noqa: F401
from wavebreaking_amd.gridfield import GridField  # noqa: F401
from wavebreaking_amd.indices.contour_index import calculate_contours  # noqa: F401
from wavebreaking_amd.indices.cutoff_index import calculate_cutoffs  # noqa: F401
from wavebreaking_amd.indices.overturning_index import calculate_overturnings  # noqa: F401
from wavebreaking_amd.indices.streamer_index import calculate_streamers  # noqa: F401
from wavebreaking_amd.processing.spatial import calculate_momentum_flux, calculate_smoothed_field  # noqa: F401
from wavebreaking_amd.processing.events import to_xarray, track_events  # noqa: F401
