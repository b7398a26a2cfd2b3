"""The four index functions (mirror of wavebreaking/indices/)."""
