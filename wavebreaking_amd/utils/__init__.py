"""Helpers the indices sit on (mirror of wavebreaking/utils/{data_utils,index_utils}.py)."""
