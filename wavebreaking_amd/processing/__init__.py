"""Pre-processing next to the hot path (wavebreaking/processing/spatial.py)."""
