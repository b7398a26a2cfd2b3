"""The oracle -- and the package's public helpers -- against outputs of the REFERENCE'S OWN CODE
(tests/golden/reference_rules.npz, made by tests/golden/make_golden_reference.py: statements and nested functions of
/root/reference/wavebreaking/**.py executed in the build container with the real numpy / pandas / scikit-learn /
scikit-image on random inputs).  CPU only; the GPU side of the same cases is tests/test_gpu_reference.py.  What needs
GEOS (check_intersections, check_groups' intersects, the cell join, transform_polygons, by_overlap) is not in the file:
those rows stay unpinned."""
import numpy as np
import pytest

from oracle import contours as oc
from oracle import indices, marching, properties, synth
from oracle import tracking as otrack


@pytest.fixture(scope="module")
def ref(golden_dir):
    return np.load(golden_dir + "/reference_rules.npz")


def _unpack(cat, off):
    return [cat[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def contour_case(ref, name):
    level, nlon, dlon, padd = ref["contours__%s__par" % name]
    return ref["contours__%s__field" % name], float(level), int(nlon), float(dlon), float(padd)


def test_round_dedupe_and_duplicate_rule_against_the_reference(ref):
    # contour_index.py:105-147 run on scikit-image's own find_contours output
    kept = dropped = 0
    for name in ref["contours__names"]:
        a, level, nlon, dlon, padd = contour_case(ref, name)
        ext = np.concatenate([a, a[:, :int(padd / dlon)]], axis=1)
        items, closed = [], []
        for item in marching.find_contours(ext, level):
            xy, cl = oc.round_dedupe(item)
            if len(xy) >= 4:
                items.append(xy)
                closed.append(cl)
        before = _unpack(ref["contours__%s__before" % name], ref["contours__%s__before_off" % name])
        assert len(items) == len(before), name
        for mine, theirs in zip(items, before):
            assert np.array_equal(mine, theirs), name
        assert np.array_equal(closed, ref["contours__%s__before_closed" % name]), name
        drop = oc.check_duplicates(items, nlon)
        assert sorted(drop) == ref["contours__%s__drop" % name].tolist(), name
        got, got_closed = oc.contours_one_step(a, level, nlon, dlon, padd)
        want = _unpack(ref["contours__%s__pts" % name], ref["contours__%s__off" % name])
        assert len(got) == len(want), name
        for mine, theirs in zip(got, want):
            assert np.array_equal(mine, theirs), name
        assert np.array_equal(got_closed, ref["contours__%s__closed" % name]), name
        kept += len(want)
        dropped += len(drop)
    assert kept > 150 and dropped > 40


def test_contour_frame_columns_against_the_reference(ref):
    # contour_index.py:154-155 (exp_lon, mean_lat) and :179-191 (original coordinates, second dedupe)
    n = 0
    for name in ref["contours__names"]:
        a, level, nlon, dlon, padd = contour_case(ref, name)
        lat = np.linspace(-90, 90, a.shape[0])
        lon = np.arange(nlon) * dlon - 180.0
        rows = oc.calculate_contours(a[None], [level], lat, lon, [0], padd, original_coordinates=False)
        assert [r["exp_lon"] for r in rows] == ref["contours__%s__exp_lon" % name].tolist(), name
        assert [r["mean_lat"] for r in rows] == ref["contours__%s__mean_lat" % name].tolist(), name
        rows = oc.calculate_contours(a[None], [level], lat, lon, [0], padd, original_coordinates=True)
        want = _unpack(ref["contours__%s__orig" % name], ref["contours__%s__orig_off" % name])
        assert len(rows) == len(want), name
        for r, w in zip(rows, want):
            assert np.array_equal(np.asarray(r["geometry"], dtype=np.float64), w), name
        n += len(rows)
    assert n > 150


def _combine_cases(ref):
    na, nl = ref["combine__count"]
    for k in range(int(na)):
        key = "combine__a%03d" % k
        yield ref[key + "__in"], [g.tolist() for g in _unpack(ref[key + "__out"], ref[key + "__out_off"])]
    for k in range(int(nl)):
        key = "combine__l%03d" % k
        yield ([g.tolist() for g in _unpack(ref[key + "__in"], ref[key + "__in_off"])],
               [g.tolist() for g in _unpack(ref[key + "__out"], ref[key + "__out_off"])])


def test_combine_shared_in_the_reference_order(ref):
    # index_utils.py:187-214: components in the order of their first list, members in CPython set order
    from wavebreaking_amd.utils.index_utils import combine_shared
    n = reordered = 0
    for lst, want in _combine_cases(ref):
        for f in (indices.combine_shared, combine_shared):
            got = [[int(v) for v in g] for g in f(lst)]
            assert got == want
        reordered += any(g != sorted(g) for g in want)
        n += 1
    assert n >= 300 and reordered > 20           # (the set order is observable: it is not ascending)
    assert combine_shared([[1, 8]]) == [[8, 1]] and combine_shared([[], [3, 1]]) == [[], [1, 3]]


def test_candidate_pairs_and_pair_filters_against_the_reference(ref):
    # streamer_index.py:124-157 (haversine matrix, on, cumsum matrix, both thresholds, the 120-column drop), :160-183, :202-222
    npairs = ndup = novl = 0
    for k in range(int(ref["streamer__count"])):
        key = "streamer__s%02d" % k
        xy = ref[key + "__xy"]
        dlon, geo_dis, cont_dis, nlon = ref[key + "__par"]
        lat, lon = ref["streamer__lat_%.1f" % dlon], ref["streamer__lon_%.1f" % dlon]
        ij, on = indices.candidate_pairs(xy, lat, lon, int(nlon), geo_dis, cont_dis)
        assert np.array_equal(on, ref[key + "__on"]), key
        assert np.array_equal(ij, ref[key + "__pairs"]), key
        after = indices._f_duplicates(ij, xy, int(nlon)) if len(ij) else ij
        assert np.array_equal(after, ref[key + "__after_dup"]), key
        assert np.array_equal(indices._f_overlapping(after), ref[key + "__after_ovl"]), key
        assert np.array_equal(indices._f_overlapping(ref[key + "__sub"]), ref[key + "__sub_ovl"]), key
        npairs += len(ij)
        ndup += len(ij) - len(after)
        novl += len(after) - len(ref[key + "__after_ovl"])
    assert npairs > 10000 and ndup > 1000 and novl > 5000


def _overturn_rows(xy):
    return [dict(date=0, level=2.0, _t=0, exp_lon=1.0, closed=False, geometry=xy)]


def test_overturning_groups_and_filters_against_the_reference(ref):
    # overturning_index.py:123-188: counts >= 3, labels, min / max, check_duplicates, check_expansion, find_lat_expansion
    boxes = 0
    for k in range(int(ref["overturn__count"])):
        key = "overturn__o%03d" % k
        xy = ref[key + "__xy"]
        dlon, nlon, range_group, min_exp = ref[key + "__par"]
        assert np.array_equal(indices.overturning_groups(xy[:, 0], range_group, dlon), ref[key + "__groups"]), key
        lat, lon = synth.grid(dlon)
        ev = indices.calculate_overturnings(_overturn_rows(xy), lat, lon, range_group, min_exp)
        got = np.asarray([[e["box"][0], e["box"][2], e["box"][1], e["box"][3]] for e in ev], dtype=np.int64).reshape(-1, 4)
        assert np.array_equal(got, ref[key + "__out"]), key
        boxes += len(got)
    assert boxes > 300


def test_contour_selection_and_cutoff_filter_against_the_reference(ref):
    # streamer_index.py:106-108 (exp_lon == exp_lon.max() over the frame), cutoff_index.py:89-93
    for k in range(int(ref["select__count"])):
        key = "select__%02d" % k
        nux, closed = ref[key + "__nux"], ref[key + "__closed"]
        dlon, min_exp = ref[key + "__par"]
        rows = [dict(exp_lon=int(n) * dlon, closed=bool(c), date=0, level=2.0, _t=0, geometry=np.full((4, 2), i, dtype=np.int64), orig=i)
                for i, (n, c) in enumerate(zip(nux, closed))]
        assert [r["orig"] for r in indices.select_max_exp(rows)] == ref[key + "__selected"].tolist(), key
        cut = indices.calculate_cutoffs(rows, min_exp)          # (a row is recognised by its polygon: every vertex is the row number)
        assert [int(e["polygon"][0, 0]) for e in cut] == ref[key + "__cutoffs"].tolist(), key


def test_spatial_resolution_against_the_reference(ref):
    # data_utils.py:129-144, errors included
    import types
    from wavebreaking_amd.utils import data_utils
    raised = 0
    for k in range(int(ref["resolution__count"])):
        coord, want, err = ref["resolution__%02d__coord" % k], float(ref["resolution__%02d__out" % k]), str(ref["resolution__%02d__err" % k])
        for f in (lambda c: oc.spatial_resolution(c, "d"), lambda c: data_utils.get_spatial_resolution({"d": types.SimpleNamespace(data=c)}, "d")):
            if err:
                with pytest.raises(ValueError) as e:
                    f(coord)
                assert str(e.value) == err
            else:
                assert float(f(coord)) == want
        raised += bool(err)
    assert raised >= 4


_sim = []


def _sim_context():
    if not _sim:
        from tests import helpers
        from wavebreaking_amd import _hip
        _sim.append(_hip.Context(0, helpers.hostsim_library()))
    return _sim[0]


def test_property_sums_and_columns_against_the_reference(ref):
    # index_utils.py:54-63, 91-120 (pandas' groupby sum, com truncation, rounding) on the oracle's cell lists
    from wavebreaking_amd.utils.index_utils import properties_from_sums
    events = 0
    for k in range(int(ref["props__count"])):
        key = "props__p%02d" % k
        dlon, t0, nt, eps, seed = ref[key + "__field"]
        lat, lon = synth.grid(dlon)
        pv = synth.pv_field(int(t0), int(nt), lat, lon, eps=float(eps), seed=int(seed))
        inten = synth.pv_field(int(t0) + 7, int(nt), lat, lon, eps=float(eps), seed=int(seed) + 50)
        w = properties.area_weights(lat, dlon, dlon)
        assert np.array_equal(w, ref[key + "__weight_lat"])
        cells = ref[key + "__cells"]
        want = ref[key + "__sums"]
        for e in range(len(want)):
            m = cells[:, 0] == e
            got = properties.event_sums(cells[m, 1], cells[m, 2], int(cells[m, 3][0]), pv, inten, w, len(lon))
            assert np.array_equal(np.asarray(got), want[e]), (key, e)
            events += 1
        com, mean_var, intensity, area = properties_from_sums(want, lon, lat, len(lon))
        # ... and the device function wb_indices runs for every event (csrc/wb_props.h:wb_props_columns), here in the host simulation
        dev = _sim_context().props_columns(want, lon, lat)
        assert np.array_equal(dev[:, :2], ref[key + "__com"]) and np.array_equal(dev[:, 2], ref[key + "__mean_var"])
        assert np.array_equal(dev[:, 3], ref[key + "__intensity"]) and np.array_equal(dev[:, 4], ref[key + "__event_area"])
        assert np.array_equal(np.asarray(com), ref[key + "__com"])
        assert np.array_equal(mean_var, ref[key + "__mean_var"]) and np.array_equal(intensity, ref[key + "__intensity"])
        assert np.array_equal(area, ref[key + "__event_area"])
        cols = ref[key + "__columns"].tolist()
        assert cols[:6] == ["date", "level", "com", "mean_var", "intensity", "event_area"] and cols[6:] in ([], ["orientation"])
    assert events > 100


def test_track_events_by_distance_against_the_reference(ref):
    # processing/events.py:113-241 executed whole (method="by_distance")
    linked = raised = 0
    for k in range(int(ref["track__count"])):
        key = "track__%02d" % k
        dates = ref[key + "__dates"].astype("datetime64[ns]")
        com = ref[key + "__com"]
        time_range, distance = ref[key + "__par"]
        time_range = None if np.isnan(time_range) else float(time_range)
        err = str(ref[key + "__err"])
        if err:
            with pytest.raises(ValueError) as e:
                otrack.track_events(dates, com=com, time_range=time_range, method="by_distance", distance=distance)
            assert str(e.value) == err
            raised += 1
            continue
        label, order, pairs, check = otrack.track_events(dates, com=com, time_range=time_range, method="by_distance", distance=distance)
        assert np.array_equal(order, ref[key + "__rows"]), key
        assert np.array_equal(label[order], ref[key + "__labels"]), key
        linked += int(check.sum())
    assert linked > 50
