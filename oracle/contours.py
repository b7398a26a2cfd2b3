"""calculate_contours restated on plain numpy arrays (TEST INFRASTRUCTURE).

Follows wavebreaking/indices/contour_index.py:87-194 step by step; the iteration
order (time-major, level-minor) is that of the two decorators
wavebreaking/utils/index_utils.py:217-258.
"""
import numpy as np

from . import marching


def spatial_resolution(coord, name):
    """wavebreaking/utils/data_utils.py:129-144"""
    delta = abs(np.unique(coord[1:] - coord[:-1]))
    if len(delta) > 1:
        raise ValueError("No regular grid found for dimension {}.".format(name))
    elif delta[0] == 0:
        raise ValueError("Two equivalent coordinates found for dimension {}.".format(name))
    return delta[0]


def as_levels(contour_levels):
    """index_utils.py:247-250: a scalar becomes a one-element list."""
    try:
        iter(contour_levels)
    except Exception:
        contour_levels = [contour_levels]
    return list(contour_levels)


def round_dedupe(item):
    """contour_index.py:109-112: closed flag on the float coordinates, round-half-even,
    order-preserving global first-occurrence dedupe, swap (row, col) -> (x, y)."""
    closed = bool(np.all(item[0] == item[-1]))
    ints = np.round(item).astype("int")
    keys = ints[:, 0] * (1 << 20) + ints[:, 1]
    _, first = np.unique(keys, return_index=True)
    keep = np.sort(first)
    return ints[keep][:, ::-1], closed


def check_duplicates(list_of_arrays, nlon):
    """contour_index.py:119-147: periodic-duplicate rule.  Returns the set of dropped positions."""
    temp = [np.unique((a[:, 0] % nlon) * (1 << 20) + a[:, 1]) for a in list_of_arrays]
    lens = [len(a) for a in list_of_arrays]
    drop = set()
    n = len(temp)
    for i1 in range(n):
        s1 = temp[i1]
        for i2 in range(n):
            if i1 == i2 or len(s1) > len(temp[i2]):
                continue
            s2 = temp[i2]
            pos = np.searchsorted(s2, s1)
            if pos[-1] >= len(s2) or not np.array_equal(s2[pos], s1):
                continue
            # set(e1) is a subset of set(e2)
            if lens[i1] == lens[i2]:
                drop.add(max(i1, i2))
            else:
                drop.add(i1 if lens[i1] < lens[i2] else i2)
    return drop


def contours_one_step(field2d, level, nlon, dlon, periodic_add):
    """One (time step, level): list of (xy int array, closed) in index coordinates of the
    periodically extended field (contour_index.py:93-147)."""
    padd = int(periodic_add / dlon)
    ext = np.concatenate([field2d, field2d[:, :padd]], axis=1)          # :94-100
    raw = marching.find_contours(ext, level)                            # :103
    items, closed = [], []
    for item in raw:                                                    # :108-117
        xy, cl = round_dedupe(item)
        if len(xy) >= 4:
            items.append(xy)
            closed.append(cl)
    drop = check_duplicates(items, nlon)                                # :143
    keep = [i for i in range(len(items)) if i not in drop]
    return [items[i] for i in keep], [closed[i] for i in keep]


def calculate_contours(data, contour_levels, lat, lon, times, periodic_add=120,
                       original_coordinates=True):
    """Rows (dicts) in the reference's frame order with the reference's columns
    ``date, level, closed, exp_lon, mean_lat, geometry`` (contour_index.py:158-169);
    ``geometry`` is an (n,2) array (x, y) standing for the LineString."""
    data = np.asarray(data)
    nlon, dlon = len(lon), spatial_resolution(np.asarray(lon), "lon")
    rows = []
    for t in range(data.shape[0]):
        for level in as_levels(contour_levels):
            items, closed = contours_one_step(data[t], level, nlon, dlon, periodic_add)
            if original_coordinates:                                    # :177-191
                conv = []
                for it in items:
                    co = np.c_[np.asarray(lon)[it[:, 0] % nlon], np.asarray(lat)[it[:, 1]]]
                    _, first = np.unique(co, axis=0, return_index=True)
                    conv.append(co[np.sort(first)])
                items = conv
            for it, cl in zip(items, closed):
                rows.append(dict(date=times[t], level=level, closed=cl,
                                 exp_lon=len(set(it[:, 0].tolist())) * dlon,        # :154
                                 mean_lat=np.round(it[:, 1].mean(), 2),            # :155
                                 geometry=it, _t=t))
    return rows
