"""Streamer / overturning / cutoff indices restated on plain numpy arrays
(TEST INFRASTRUCTURE).  Follows

* wavebreaking/indices/streamer_index.py:100-289
* wavebreaking/indices/overturning_index.py:100-232
* wavebreaking/indices/cutoff_index.py:84-104
* wavebreaking/utils/index_utils.py:187-214 (combine_shared)

``contours`` is the row list of ``oracle.contours.calculate_contours(...,
original_coordinates=False)``; events are dict rows carrying the index-space
polygon; ``oracle.properties`` turns them into the reference's output columns.
"""
import numpy as np

from . import geom

try:  # the reference's own distance routine when it is importable (streamer_index.py:27-29)
    from sklearn.metrics import DistanceMetric
    _dist = DistanceMetric.get_metric("haversine")
except Exception:  # pragma: no cover
    _dist = None


def haversine_pairwise(coords_rad):
    """sklearn DistanceMetric('haversine').pairwise on [lat, lon] radians, float64.

    Closed form of the third-party routine (scikit-learn, unpinned in the reference's
    setup.py:13-24; pinned here to 1.7.2 through tests/golden/haversine.npz):
    ``2 asin(sqrt(sin^2(dlat/2) + cos(lat1) cos(lat2) sin^2(dlon/2)))``."""
    if _dist is not None:
        return _dist.pairwise(coords_rad)
    la, lo = coords_rad[:, 0], coords_rad[:, 1]
    s0 = np.sin(0.5 * (la[:, None] - la[None, :]))
    s1 = np.sin(0.5 * (lo[:, None] - lo[None, :]))
    return 2 * np.arcsin(np.sqrt(s0 * s0 + np.cos(la)[:, None] * np.cos(la)[None, :] * s1 * s1))


def haversine_formula(coords_rad):
    la, lo = coords_rad[:, 0], coords_rad[:, 1]
    s0 = np.sin(0.5 * (la[:, None] - la[None, :]))
    s1 = np.sin(0.5 * (lo[:, None] - lo[None, :]))
    return 2 * np.arcsin(np.sqrt(s0 * s0 + np.cos(la)[:, None] * np.cos(la)[None, :] * s1 * s1))


def combine_shared(lst):
    """index_utils.py:187-214, literally (CPython set iteration order included)."""
    elements = list(lst)
    output = []
    while len(elements) > 0:
        first, *rest = elements
        first = set(first)
        lf = -1
        while len(first) > lf:
            lf = len(first)
            rest2 = []
            for r in rest:
                if len(first.intersection(set(r))) > 0:
                    first |= set(r)
                else:
                    rest2.append(r)
            rest = rest2
        output.append(list(first))
        elements = rest
    return output


def select_max_exp(contours):
    """streamer_index.py:106 / overturning_index.py:105: exp_lon == max over the WHOLE frame."""
    if not contours:
        return []
    m = max(c["exp_lon"] for c in contours)
    return [c for c in contours if c["exp_lon"] == m]


# --------------------------------------------------------------------------- streamers

def candidate_pairs(xy, lat, lon, nlon, geo_dis, cont_dis, stats=None):
    """streamer_index.py:121-157: (ind1, ind2) candidates in row-major order and ``on``."""
    x, y = xy[:, 0], xy[:, 1]
    coords = np.c_[lat[y], lon[x % nlon]]
    geo = haversine_pairwise(np.radians(coords)) * 6371                  # :130
    on = np.insert(np.diagonal(geo, 1), 0, 0)                            # :133
    on_mat = np.triu(np.tile(on, (len(on), 1)), k=1)                     # :134
    cont = np.cumsum(on_mat, axis=1)                                     # :135
    check = (geo < geo_dis) * (cont > cont_dis)                          # :138
    if stats is not None:
        # how close the two threshold tests come to a tie (libm-ulp sensitivity, SURVEY 7 "hard parts")
        iu = np.triu_indices(len(on), 1)
        stats["min_rel_geo"] = float(np.min(np.abs(geo[iu] - geo_dis)) / geo_dis) if len(iu[0]) else 1.0
        stats["min_rel_cont"] = float(np.min(np.abs(cont[iu] - cont_dis)) / cont_dis) if len(iu[0]) else 1.0
    ij = np.transpose(check.nonzero())                                   # :139
    keep = np.abs(x[ij[:, 0]] - x[ij[:, 1]]) <= 120                      # :157 (index units, hard-coded)
    return ij[keep], on


def _f_duplicates(ij, xy, nlon):
    """streamer_index.py:160-183: of two pairs equal after x mod nlon, the second goes."""
    x, y = xy[:, 0], xy[:, 1]
    key = np.c_[x[ij[:, 0]] % nlon, y[ij[:, 0]], x[ij[:, 1]] % nlon, y[ij[:, 1]]]
    _, inv, cnt = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    drop = np.zeros(len(ij), dtype=bool)
    for g in np.nonzero(cnt > 1)[0]:
        members = np.nonzero(inv == g)[0]
        if len(members) != 2:
            # np.asarray(list of ragged tuples)[:, 1] raises in the reference (periodic_add > 360)
            raise ValueError("duplicate group of size %d: reference behaviour undefined" % len(members))
        drop[members[1]] = True
    return ij[~drop]


def _f_intersections(ij, xy):
    """:185-200: keep pairs whose chord ``touches`` the contour."""
    keep = [geom.chord_touches_contour(xy[a], xy[b], xy) for a, b in ij]
    return ij[np.asarray(keep, dtype=bool)]


def _f_overlapping(ij):
    """:202-222: drop a pair whose [ind1, ind2] range lies inside another pair's range."""
    i1, i2 = ij[:, 0], ij[:, 1]
    drop = np.zeros(len(ij), dtype=bool)
    for k in range(len(ij)):
        inside = (i1 <= i1[k]) & (i2[k] <= i2)
        inside[k] = False
        drop[k] = inside.any()
    return ij[~drop]


def _f_groups(ij, xy, on):
    """:224-251: chords that intersect are grouped transitively; the member with the
    longest along-contour distance survives; output in group order."""
    n = len(ij)
    combos = [(a, b) for a in range(n) for b in range(a, n)]             # combinations_with_replacement
    crossing = [c for c in combos
                if geom.segments_intersect(xy[ij[c[0], 0]], xy[ij[c[0], 1]], xy[ij[c[1], 0]], xy[ij[c[1], 1]])]
    groups = combine_shared(np.asarray(crossing))
    keep = [item[int(np.argmax([on[int(ij[m, 0]): int(ij[m, 1]) + 1].sum() for m in item]))] for item in groups]
    return ij[keep]


def streamer_basepoints(xy, lat, lon, nlon, geo_dis=800, cont_dis=1500, stats=None):
    """All of streamer_index.py:119-264 for one contour -> surviving (ind1, ind2) rows, in output order."""
    ij, on = candidate_pairs(xy, lat, lon, nlon, geo_dis, cont_dis, stats)
    routines = [lambda p: _f_duplicates(p, xy, nlon), lambda p: _f_intersections(p, xy),
                _f_overlapping, lambda p: _f_groups(p, xy, on)]
    i = 0
    counts = [len(ij)]
    while len(ij) > 1 and i <= 3:                                        # :261-264
        ij = routines[i](ij)
        counts.append(len(ij))
        i += 1
    if stats is not None:
        stats["stage_counts"] = counts
    return ij


def calculate_streamers(contours, lat, lon, geo_dis=800, cont_dis=1500, stats=None):
    nlon = len(lon)
    lat = np.asarray(lat, dtype=np.float64)
    lon = np.asarray(lon, dtype=np.float64)
    events = []
    for c in select_max_exp(contours):
        xy = np.asarray(c["geometry"]).astype("int")
        st = {} if stats is not None else None
        ij = streamer_basepoints(xy, lat, lon, nlon, geo_dis, cont_dis, st)
        if stats is not None:
            stats.setdefault("per_contour", []).append(st)
        for a, b in ij:
            events.append(dict(date=c["date"], level=c["level"], _t=c["_t"],
                               polygon=xy[int(a): int(b) + 1], ind=(int(a), int(b))))   # :269-272
    return events


# --------------------------------------------------------------------------- overturnings

def overturning_groups(x, range_group, dlon):
    """overturning_index.py:123-134: (min_lon, max_lon) per label."""
    lons, counts = np.unique(x, return_counts=True)
    ot = lons[counts >= 3]
    if len(ot) == 0:
        return np.zeros((0, 2), dtype=np.int64)
    gap = np.diff(ot) > range_group / dlon
    label = np.concatenate([[0], np.cumsum(gap)])
    out = []
    for l in range(label[-1] + 1):
        m = ot[label == l]
        out.append((int(m.min()), int(m.max())))
    return np.asarray(out, dtype=np.int64)


def _ot_duplicates(df, nlon):
    """:136-162"""
    temp = [np.array(range(a, b + 1)) % nlon for a, b in df]
    drop = set()
    for i0 in range(len(df)):
        for i1 in range(len(df)):
            if i0 == i1:
                continue
            if set(temp[i0].tolist()).issubset(set(temp[i1].tolist())):
                l0, l1 = len(temp[i0]), len(temp[i1])
                if l0 == l1:
                    drop.add(max(i0, i1))
                else:
                    drop.add(i0 if l0 < l1 else i1)
    return df[[i for i in range(len(df)) if i not in drop]]


def calculate_overturnings(contours, lat, lon, range_group=5, min_exp=5):
    nlon = len(lon)
    lat = np.asarray(lat, dtype=np.float64)
    dlon = abs(np.unique(np.diff(np.asarray(lon, dtype=np.float64))))[0]
    events = []
    for c in select_max_exp(contours):
        xy = np.asarray(c["geometry"]).astype("int")
        x, y = xy[:, 0], xy[:, 1]
        df = overturning_groups(x, range_group, dlon)
        i = 0
        lat_exp = None
        while len(df) > 0 and i <= 2:                                    # :185-188
            if i == 0:
                df = _ot_duplicates(df, nlon)
            elif i == 1:
                df = df[(df[:, 1] - df[:, 0]) >= min_exp / dlon]         # :164-166
            else:
                lat_exp = []
                for a, b in df:                                          # :168-180
                    sel = y[(x >= a) & (x <= b)]
                    lat_exp.append((int(sel.min()), int(sel.max())))
            i += 1
        if len(df) == 0:
            continue
        for (a, b), (lo, hi) in zip(df, lat_exp):
            lat_west = lat[y[np.nonzero(x == a)[0][0]]]                  # :192-194
            lat_east = lat[y[np.nonzero(x == b)[0][-1]]]                 # :195-197
            orientation = "cyclonic" if abs(lat_west) <= abs(lat_east) else "anticyclonic"
            box = np.array([[b, lo], [b, hi], [a, hi], [a, lo]])         # shapely.geometry.box(a,lo,b,hi) ccw=True
            events.append(dict(date=c["date"], level=c["level"], _t=c["_t"], orientation=orientation,
                               polygon=box, box=(int(a), int(lo), int(b), int(hi))))
    return events


# --------------------------------------------------------------------------- cutoffs

def calculate_cutoffs(contours, min_exp=5):
    """cutoff_index.py:89-98"""
    if not contours:
        return []
    m = max(c["exp_lon"] for c in contours)
    events = []
    for c in contours:
        if c["exp_lon"] < m and c["exp_lon"] >= min_exp and c["closed"]:
            events.append(dict(date=c["date"], level=c["level"], _t=c["_t"],
                               polygon=np.asarray(c["geometry"]).astype("int")))
    return events
