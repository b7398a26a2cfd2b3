"""wavebreaking_amd -- MI355X-native implementation of the WaveBreaking hot path.

Same call surface as the reference package for this path
(wavebreaking/__init__.py:31-34): calculate_contours, calculate_streamers,
calculate_overturnings, calculate_cutoffs.
"""
__version__ = "0.1.0"
