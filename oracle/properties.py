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
        row["_not_simple"] = "orientation" not in ev and geom.polygon_not_simple(ev["polygon"], width, nlat)
        row["_cells"] = (xs, ys)
        row["_t"] = ev["_t"]
        out.append(row)
    return out


# --------------------------------------------------------------------------- transform_polygons
# The seam split (index_utils.py:148-173) is GEOS work in the reference (linemerge / unary_union / polygonize /
# make_valid: **parity unpinned**, no GEOS here).  The restatement below is deliberately NOT the product's algorithm
# (wavebreaking_amd/utils/index_utils.py pairs the line points by sorting them): it works in exact rational arithmetic,
# decides which stretch of the cutting line lies inside the polygon by an explicit point-in-polygon test of its midpoint,
# and `clipped_area_exact` gives a third, ring-free statement of the same region (Green's theorem with the abscissa
# clamped at the line).
from fractions import Fraction


def _inside_exact(px, py, v):
    """strictly inside the closed integer ring v (even-odd), exact; points on the boundary count as outside"""
    n = len(v)
    inside = False
    for k in range(n):
        ax, ay = v[k]
        bx, by = v[(k + 1) % n]
        # on the boundary?
        cr = (bx - ax) * (py - ay) - (by - ay) * (px - ax)
        if cr == 0 and min(ax, bx) <= px <= max(ax, bx) and min(ay, by) <= py <= max(ay, by):
            return False
        if (ay > py) != (by > py):
            xc = ax + Fraction((py - ay) * (bx - ax), (by - ay))
            if xc > px:
                inside = not inside
    return inside


def split_rings_exact(ring, L, keep_le):
    """Pieces of the closed integer ring on one side of the vertical line x = L (closed half-plane) as lists of exact
    (Fraction) points.  Boundary chains on the kept side are linked across the line from a chain's exit point to the
    neighbouring line point (above or below) whose connecting stretch of the line has its midpoint strictly inside the
    polygon."""
    v = [(int(x), int(y)) for x, y in np.asarray(ring)]
    if len(v) > 1 and v[0] == v[-1]:
        v = v[:-1]
    n = len(v)

    def keep(p):
        return p[0] <= L if keep_le else p[0] >= L

    nodes = []                                   # boundary nodes in order: vertices + strict crossings of the line
    for k in range(n):
        a, b = v[k], v[(k + 1) % n]
        nodes.append((Fraction(a[0]), Fraction(a[1])))
        if (a[0] < L < b[0]) or (b[0] < L < a[0]):
            t = Fraction(L - a[0], b[0] - a[0])
            nodes.append((Fraction(L), a[1] + t * (b[1] - a[1])))
    kp = [keep(p) for p in nodes]
    if all(kp):
        return [nodes]
    if not any(kp):
        return []
    m = len(nodes)
    s = kp.index(False)
    chains, cur = [], None
    for q in range(1, m + 1):
        i = (s + q) % m
        if kp[i]:
            if cur is None:
                cur = []
            cur.append(nodes[i])
        elif cur is not None:
            chains.append(cur)
            cur = None
    if cur is not None:
        chains.append(cur)
    chains = [c for c in chains if len(c) >= 2]                      # a lone point touching the line from outside
    # every chain starts and ends ON the line
    assert all(c[0][0] == L and c[-1][0] == L for c in chains)
    ys = sorted({c[0][1] for c in chains} | {c[-1][1] for c in chains})
    entry_at = {}
    for ci, c in enumerate(chains):
        entry_at.setdefault(c[0][1], []).append(ci)

    def partner(y0):
        """the line point joined to (L, y0) through the polygon's interior along the line"""
        k = ys.index(y0)
        for kk in (k + 1, k - 1):
            if 0 <= kk < len(ys):
                mid = (Fraction(L), (y0 + ys[kk]) / 2)
                if _inside_exact(mid[0], mid[1], v):
                    return ys[kk]
        return y0                                                    # the chain closes on itself (exit = entry point)

    used = [False] * len(chains)
    pieces = []
    for c0 in range(len(chains)):
        if used[c0]:
            continue
        piece, c = [], c0
        while not used[c]:
            used[c] = True
            piece.extend(chains[c])
            yn = partner(chains[c][-1][1])
            cand = [ci for ci in entry_at.get(yn, []) if not used[ci]] or entry_at.get(yn, [c0])
            c = cand[0]
        if len(piece) >= 3 and ring_area_exact(piece) != 0:
            pieces.append(piece)
    return pieces


def ring_area_exact(r):
    n = len(r)
    return sum(r[k][0] * r[(k + 1) % n][1] - r[(k + 1) % n][0] * r[k][1] for k in range(n)) / 2


def clipped_area_exact(ring, L, keep_le):
    """|area| of (closed integer polygon) intersected with the half-plane x <= L (or x >= L), by Green's theorem with the
    abscissa clamped at the line: area = | closed integral of min(x, L) dy | -- exact, no rings involved"""
    v = [(int(x), int(y)) for x, y in np.asarray(ring)]
    n = len(v)
    total = Fraction(0)
    for k in range(n):
        (ax, ay), (bx, by) = v[k], v[(k + 1) % n]
        pts = [(Fraction(ax), Fraction(ay))]
        if (ax < L < bx) or (bx < L < ax):
            t = Fraction(L - ax, bx - ax)
            pts.append((Fraction(L), ay + t * (by - ay)))
        pts.append((Fraction(bx), Fraction(by)))
        for (x0, y0), (x1, y1) in zip(pts[:-1], pts[1:]):
            c0 = min(x0, L) if keep_le else max(x0, L) - L
            c1 = min(x1, L) if keep_le else max(x1, L) - L
            total += (c0 + c1) / 2 * (y1 - y0)                      # the clamped abscissa is linear on each piece
    return abs(total)


def ring_area(r):
    r = np.asarray(r, dtype=np.float64)
    x, y = r[:, 0], r[:, 1]
    return 0.5 * float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))


def split_index_polygon(polygon, nlon):
    """index_utils.py:148-166: the parts of the index polygon left of x = nlon - 1 and right of x = nlon (the strip
    between the last meridian and its periodic copy removed), as float rings in index coordinates"""
    coords = np.asarray(polygon).astype("int")
    pieces = split_rings_exact(coords, nlon - 1, True) + split_rings_exact(coords, nlon, False)
    return [np.array([[float(x), float(y)] for x, y in p]) for p in pieces]


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
    return [to_lonlat(p) for p in split_index_polygon(coords, nlon)]


def canonical_ring(r, ndigits=9):
    """a ring as a rotation- and direction-independent tuple (for comparing ring SETS): consecutive repeats dropped, then
    the lexicographically smallest rotation of the ring or its reversal"""
    pts = [tuple(round(float(c), ndigits) for c in p) for p in np.asarray(r)]
    if len(pts) > 1 and pts[0] == pts[-1]:
        pts = pts[:-1]
    pts = [p for i, p in enumerate(pts) if p != pts[i - 1]] or pts[:1]
    # vertices in the middle of a straight stretch carry no shape (a polygon touching the cutting line from the far side
    # leaves such a node on the line; whether the reference's GEOS output keeps it is unpinned)
    changed = True
    while changed and len(pts) > 3:
        changed = False
        for i in range(len(pts)):
            a, b, c = pts[i - 1], pts[i], pts[(i + 1) % len(pts)]
            if abs((b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])) < 1e-9 and \
                    (b[0] - a[0]) * (c[0] - b[0]) + (b[1] - a[1]) * (c[1] - b[1]) >= 0:
                del pts[i]
                changed = True
                break
    best = None
    for seq in (pts, pts[::-1]):
        for k in range(len(seq)):
            rot = tuple(seq[k:] + seq[:k])
            if best is None or rot < best:
                best = rot
    return best
