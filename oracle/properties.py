"""calculate_properties / transform_polygons restated on numpy arrays (TEST INFRASTRUCTURE).

Follows wavebreaking/utils/index_utils.py:35-184.  GEOS-dependent pieces
(buffer+contains, seam split) are **parity unpinned** -- see oracle/geom.py.
"""
import numpy as np

from . import geom


def area_weights(lat, dlon, dlat):
    """index_utils.py:58-62"""
    area_cell = np.round(6371 * 2 * np.pi / (360 / ((dlon + dlat) / 2))) ** 2
    return np.cos(np.radians(np.asarray(lat, dtype=np.float64))) * area_cell


def event_cells(polygon, dlon, dlat, width, nlat):
    """index_utils.py:37-51: lattice points contained in polygon.buffer((dlon+dlat)/4)."""
    return geom.polygon_cells(polygon, ((dlon + dlat) / 2) / 2, width, nlat)


def group_sum(values):
    """pandas ``groupby(...).sum()`` of one group of float64 values (index_utils.py:94-102).

    Third-party arithmetic, not under /root/reference: pandas ``_libs/groupby.pyx:group_sum`` adds the rows of a group
    one after the other in frame order with Kahan compensation (``y = v - c; t = s + y; c = (t - s) - y; s = t``, the
    compensation reset to 0 when it becomes NaN).  Pinned bit-for-bit against pandas 2.3.3 by
    tests/golden/props_pandas.npz."""
    s = 0.0
    c = 0.0
    for v in values:
        y = v - c
        t = s + y
        c = (t - s) - y
        if c != c:
            c = 0.0
        s = t
    return s


def event_sums(xs, ys, t, data, intensity, weight_lat, nlon):
    """index_utils.py:63-102 for one event, in the reference's arithmetic: one frame row per cell in the join's row
    order -- the grid points are numbered y-major, x-minor (np.meshgrid(...).flatten(), :37-41), ``xs, ys`` arrive in
    that order --, per-row products ``areas = weight_lat[y]`` (:63), ``areas * data`` (:66-75), ``x * areas``,
    ``y * areas`` (:91-92) rounded once each, and the five columns summed by pandas' groupby sum (:94-102)."""
    xs = np.asarray(xs, dtype=np.int64)
    ys = np.asarray(ys, dtype=np.int64)
    areas = np.asarray(weight_lat, dtype=np.float64)[ys]
    val = np.asarray(data[t, ys, xs % nlon], dtype=np.float64)
    cols = [areas, areas * val]
    if intensity is not None:
        cols.append(areas * np.asarray(intensity[t, ys, xs % nlon], dtype=np.float64))
    else:
        cols.append(np.zeros(len(xs)))                                               # :88-89 (intensity = 0)
    cols.append(xs.astype(np.float64) * areas)
    cols.append(ys.astype(np.float64) * areas)
    return tuple(group_sum(c.tolist()) for c in cols)


def calculate_properties(events, data, intensity, lat, lon, periodic_add):
    """Rows with the reference's property columns, plus the raw sums (``_sums``) and the
    cell list (``_cells``) for parity checks."""
    lat = np.asarray(lat, dtype=np.float64)
    lon = np.asarray(lon, dtype=np.float64)
    nlon, nlat = len(lon), len(lat)
    dlon = abs(np.unique(np.diff(lon)))[0]
    dlat = abs(np.unique(np.diff(lat)))[0]
    width = nlon + int(periodic_add / dlon)
    w = area_weights(lat, dlon, dlat)
    out = []
    for ev in events:
        xs, ys = event_cells(ev["polygon"], dlon, dlat, width, nlat)
        areas, mv, it, xc, yc = event_sums(xs, ys, ev["_t"], data, intensity, w, nlon)
        com = (lon[int(xc / areas) % nlon], lat[int(yc / areas)])                      # :105-111
        row = dict(date=ev["date"], level=ev["level"], com=com,
                   mean_var=np.round(mv / areas, 2), intensity=np.round(it / areas, 2),
                   event_area=np.round(areas, 2))                                      # :113-120
        if "orientation" in ev:
            row["orientation"] = ev["orientation"]
        row["_sums"] = (areas, mv, it, xc, yc)
        row["_cells"] = (xs, ys)
        row["_t"] = ev["_t"]
        out.append(row)
    return out


# --------------------------------------------------------------------------- transform_polygons

def _clip_halfplane(ring, L, keep_le):
    """Pieces of the closed ring (n,2 float) on one side of the vertical line x = L
    (closed half-plane).  Proper multi-piece split: boundary chains on the kept side are
    re-connected along the line by pairing the line points sorted by y."""
    v = np.asarray(ring, dtype=np.float64)
    n = len(v)

    def inside(p):
        return p[0] <= L if keep_le else p[0] >= L

    ins = [inside(p) for p in v]
    if all(ins):
        return [v]
    if not any(ins):
        return []
    # rotate so that we start on an outside vertex
    s = ins.index(False)
    order = [(s + k) % n for k in range(n)]
    chains = []
    cur = None
    for k in range(n):
        a, b = v[order[k]], v[order[(k + 1) % n]]
        ia, ib = ins[order[k]], ins[order[(k + 1) % n]]
        if not ia and ib:                      # entering
            t = (L - a[0]) / (b[0] - a[0])
            cur = [np.array([L, a[1] + t * (b[1] - a[1])]), b.copy()]
        elif ia and ib:
            cur.append(b.copy())
        elif ia and not ib:                    # leaving
            t = (L - a[0]) / (b[0] - a[0])
            cur.append(np.array([L, a[1] + t * (b[1] - a[1])]))
            chains.append(cur)
            cur = None
    # pair the line points by y
    pts = []
    for ci, ch in enumerate(chains):
        pts.append((ch[0][1], 0, ci))          # entry
        pts.append((ch[-1][1], 1, ci))         # exit
    pts.sort(key=lambda p: (p[0], p[1]))
    partner = {}
    for k in range(0, len(pts) - 1, 2):
        partner[(pts[k][1], pts[k][2])] = (pts[k + 1][1], pts[k + 1][2])
        partner[(pts[k + 1][1], pts[k + 1][2])] = (pts[k][1], pts[k][2])
    used = [False] * len(chains)
    pieces = []
    for c0 in range(len(chains)):
        if used[c0]:
            continue
        piece = []
        c = c0
        guard = 0
        while not used[c] and guard <= len(chains):
            used[c] = True
            piece.extend(chains[c])
            kind, nxt = partner.get((1, c), (0, c))
            c = nxt
            guard += 1
        piece = np.asarray(piece)
        if len(piece) >= 3 and abs(ring_area(piece)) > 0:
            pieces.append(piece)
    return pieces


def ring_area(r):
    r = np.asarray(r, dtype=np.float64)
    x, y = r[:, 0], r[:, 1]
    return 0.5 * float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))


def transform_polygon(polygon, lat, lon):
    """index_utils.py:129-173 for one event: list of rings in (lon, lat) coordinates
    (one ring = Polygon, several = MultiPolygon, none = empty Polygon)."""
    lat = np.asarray(lat, dtype=np.float64)
    lon = np.asarray(lon, dtype=np.float64)
    nlon = len(lon)
    coords = np.asarray(polygon).astype("int")

    def to_lonlat(c):
        c = np.asarray(c).astype("int")                                    # :135 (truncation)
        return np.c_[lon[c[:, 0] % nlon], lat[c[:, 1]]]                     # :139-146

    if not (coords[:, 0] >= nlon).any():                                    # :136
        return [to_lonlat(coords)]
    # :148-173: drop the strip nlon-1 < x < nlon, keep what is left on either side
    pieces = _clip_halfplane(coords, nlon - 1, True) + _clip_halfplane(coords, nlon, False)
    return [to_lonlat(p) for p in pieces]
