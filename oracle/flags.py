"""CPU restatement of to_xarray's flagging rule (wavebreaking/processing/events.py:37-110) -- TEST INFRASTRUCTURE.

events.py:63-79: every grid point (lon_i, lat_j) becomes a shapely Point, every event geometry is buffered by
``r = ((dlon + dlat) / 2) / 2`` and ``sjoin(buffer, points, predicate="contains")`` pairs events with the points they
contain; events.py:93-99 then assigns ``set_val`` at (date, lat, lon) in event-index order, so a later event overwrites
an earlier one.  GEOS / geopandas are not installable in the build container (parity unpinned at that boundary, see
DESIGN.md 4); restated from the definitions: ``buffer(r).contains(p)``  <=>  p is in the closed polygon or closer than
r to its boundary.  For the lattice polygons transform_polygons produces and r = dlon / 2 no grid point is at distance
exactly r from a lattice segment, and GEOS' polygonal arc approximation (inscribed) only matters for points within
0.1 % of r from a vertex, which cannot happen on the lattice either.

Brute force: every cell of the event's bounding box against every edge (numpy broadcasting), crossing-number parity with
the half-open rule, exact point-segment distance; deliberately not the scanline algorithm of csrc/wb_flags.h.
"""
import numpy as np


def _rings_contain(px, py, rings, r):
    """bool array: points (px, py) contained in buffer(r) of the even-odd union of ``rings`` (list of (n, 2) arrays)."""
    inside = np.zeros(px.shape, dtype=bool)
    near = np.zeros(px.shape, dtype=bool)
    for ring in rings:
        ring = np.asarray(ring, dtype=np.float64)
        a = ring
        b = np.roll(ring, -1, axis=0)
        for (x1, y1), (x2, y2) in zip(a, b):
            if y1 != y2:
                span = (min(y1, y2) <= py) & (py < max(y1, y2))
                with np.errstate(all="ignore"):
                    xc = x1 + (py - y1) * (x2 - x1) / (y2 - y1)
                inside ^= span & (px < xc)
            dx, dy = x2 - x1, y2 - y1
            len2 = dx * dx + dy * dy
            if len2 > 0.0:
                t = np.clip(((px - x1) * dx + (py - y1) * dy) / len2, 0.0, 1.0)
            else:
                t = np.zeros(px.shape)
            ex, ey = px - (x1 + t * dx), py - (y1 + t * dy)
            near |= ex * ex + ey * ey < r * r
    return inside | near


def flag_events(nt, lat, lon, radius, steps, rings_per_event, values=None, dtype=np.int8):
    """(nt, nlat, nlon) cube; ``rings_per_event[k]`` = list of (n, 2) lon/lat rings of event k, ``steps[k]`` its time index."""
    lat = np.asarray(lat, dtype=np.float64)
    lon = np.asarray(lon, dtype=np.float64)
    out = np.zeros((nt, lat.size, lon.size), dtype=dtype)
    for k, rings in enumerate(rings_per_event):
        allv = np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1, 2) for r in rings]) if len(rings) else np.zeros((0, 2))
        if len(allv) == 0:
            continue
        ii = np.nonzero((lon >= allv[:, 0].min() - radius - 1e-9) & (lon <= allv[:, 0].max() + radius + 1e-9))[0]
        jj = np.nonzero((lat >= allv[:, 1].min() - radius - 1e-9) & (lat <= allv[:, 1].max() + radius + 1e-9))[0]
        if ii.size == 0 or jj.size == 0:
            continue
        px, py = np.meshgrid(lon[ii], lat[jj])
        hit = _rings_contain(px, py, rings, radius)
        sub = out[steps[k]][np.ix_(jj, ii)]
        sub[hit] = 1 if values is None else values[k]
        out[steps[k]][np.ix_(jj, ii)] = sub
    return out
