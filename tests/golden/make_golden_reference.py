#!/usr/bin/env python
"""Pin the oracle to the REFERENCE'S OWN CODE (build container only).

The reference cannot be imported here (xarray / geopandas / shapely / cartopy are absent), but large parts of its hot
path need nothing beyond numpy / pandas / itertools / scikit-learn / scikit-image.  This script reads the reference's
source files where they lie (/root/reference/wavebreaking/**.py), takes the statements and (nested) functions named
below out of their syntax trees with ``ast`` -- by line range, checked against the names they assign -- compiles those
very statements and EXECUTES them on random inputs with the real libraries of the build container.  Closure variables
(``kwargs``, ``contour_index``, ``data`` ...) are supplied as real objects (dicts, pandas frames / series); no module is
replaced by a stand-in, and nothing that needs GEOS is run (``check_intersections``, ``check_groups``' ``intersects``,
``calculate_properties``' ``sjoin``, ``transform_polygons``, ``by_overlap`` stay unpinned).  Inputs and outputs are
stored as plain arrays in tests/golden/reference_rules.npz; NO reference text is committed, and this script never runs
on the GPU box (it reads /root/reference).

    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_reference.py contours   # scikit-image 0.18.3 + the reference's C3-C6
    python3 tests/golden/make_golden_reference.py rules                                  # pandas 2.3.3, scikit-learn 1.7.2

What is executed (reference file:lines, skaderli/WaveBreaking 0.3.8):
  contours   indices/contour_index.py:105-117 (closed flag, round, dedupe, >= 4), :119-147 (check_duplicates and the drop),
             :154-155 (exp_lon, mean_lat), :179-191 (original coordinates + second dedupe), on find_contours' own output
  combine    utils/index_utils.py:187-214 (combine_shared)
  streamer   indices/streamer_index.py:29 (dist), :124-157 (coordinates, haversine matrix, on, cumsum matrix, candidate
             pairs, the |x1 - x2| > 120 drop), :160-183 (check_duplicates), :202-222 (check_overlapping)
  overturn   indices/overturning_index.py:123-134 (counts, labels, min / max), :136-188 (check_duplicates, check_expansion,
             find_lat_expansion and the loop that applies them)
  select     indices/streamer_index.py:106-108 (exp_lon == max), indices/cutoff_index.py:89-93 (cutoff filter)
  resolution utils/data_utils.py:129-144 (get_spatial_resolution) incl. its two ValueErrors
  props      utils/index_utils.py:54-63, 91-120 (area weights, per-cell products, groupby sums, com, rounding) on the
             oracle's cell lists (the join that produces them is GEOS: unpinned)
  track      processing/events.py:113-241 (track_events, method="by_distance": range pairs, haversine of com, threshold,
             combine_shared, labels, final sort)
"""
import ast
import itertools
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/wavebreaking"
OUT = os.path.join(HERE, "reference_rules.npz")


# ------------------------------------------------------------------------------------------------ syntax-tree access
class Source:
    def __init__(self, rel):
        self.path = os.path.join(REF, rel)
        self.tree = ast.parse(open(self.path).read(), filename=self.path)

    def function(self, name, inside=None):
        """the FunctionDef called ``name`` (anywhere below ``inside``, default the module)"""
        for node in ast.walk(inside or self.tree):
            if isinstance(node, ast.FunctionDef) and node.name == name and node is not inside:
                return node
        raise KeyError(name)

    def statements(self, holder, lo, hi):
        """the statements below ``holder`` (a FunctionDef) that lie completely inside the lines [lo, hi], in source order;
        loops / branches that only overlap the range are entered, nested functions are taken whole or not at all"""
        out = []

        def visit(body):
            for st in body:
                if lo <= st.lineno and st.end_lineno <= hi:
                    out.append(st)
                elif isinstance(st, (ast.For, ast.While, ast.If, ast.With, ast.Try)) and st.lineno <= hi and st.end_lineno >= lo:
                    visit(st.body)
                    visit(getattr(st, "orelse", []))
        visit(holder.body)
        return out

    def run(self, stmts, env):
        mod = ast.Module(body=list(stmts), type_ignores=[])
        exec(compile(mod, self.path, "exec"), env)
        return env


def assigned(stmts):
    names = []
    for st in stmts:
        if isinstance(st, ast.FunctionDef):
            names.append(st.name)
        for node in ast.walk(st) if not isinstance(st, ast.FunctionDef) else []:
            if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Store):
                names.append(node.id)
    return names


def expect(stmts, *names):
    got = assigned(stmts)
    for n in names:
        assert n in got, "the reference's statements at these lines no longer assign %r (found %s)" % (n, sorted(set(got)))


def check_version():
    src = open(os.path.join(REF, "__init__.py")).read()
    assert '__version__ = "0.3.8"' in src, "line ranges below are those of WaveBreaking 0.3.8"


def pack(list_of_arrays, width=2, dtype=np.int64):
    off = np.zeros(len(list_of_arrays) + 1, dtype=np.int64)
    for i, a in enumerate(list_of_arrays):
        off[i + 1] = off[i] + len(a)
    cat = (np.concatenate([np.asarray(a, dtype=dtype).reshape(-1, width) for a in list_of_arrays], axis=0)
           if list_of_arrays else np.zeros((0, width), dtype=dtype))
    return cat, off


def merge_into(path, new):
    old = dict(np.load(path, allow_pickle=False)) if os.path.exists(path) else {}
    old.update(new)
    np.savez_compressed(path, **old)


# ------------------------------------------------------------------------------------------------ contours (scikit-image)
def contour_fields():
    """(name, field, level, nlon, dlon, periodic_add): small global-looking fields incl. cutoffs inside the periodic
    overlap (the duplicate rule's reason to exist), rough noise (revisited points), NaN border rows, a cell equal to the
    level"""
    rng = np.random.RandomState(8128)
    out = []
    for k in range(64):
        nlon = int(rng.choice([36, 48, 72, 90]))
        dlon = 360.0 / nlon
        nlat = int(rng.choice([19, 25, 37]))
        lat = np.linspace(-90, 90, nlat)[:, None]
        lon = (np.arange(nlon) * dlon)[None, :]
        a = lat / 12.0
        for w in range(2, 6):
            a = a + rng.uniform(0.2, 1.2) * np.sin(np.radians(w * lon + rng.uniform(0, 360))) * np.cos(np.radians(lat)) ** 2
        for _ in range(int(rng.randint(5, 16))):                     # blobs: cutoffs, some of them in the first 120 degrees
            la0, lo0 = rng.uniform(-60, 60), rng.uniform(0, 360 if rng.rand() < 0.5 else 110)
            dl = np.minimum(np.abs(lon - lo0), 360 - np.abs(lon - lo0))
            a = a + rng.choice([-1, 1]) * rng.uniform(3, 7) * np.exp(-(((lat - la0) / rng.uniform(5, 12)) ** 2 + (dl / rng.uniform(6, 18)) ** 2))
        if k % 3 == 1:
            a = a + rng.uniform(0.3, 1.5) * rng.uniform(-1, 1, a.shape)
        a = a.astype(np.float32).astype(np.float64)
        level = float(rng.choice([-2.0, 0.5, 2.0]))
        if k % 9 == 4:
            a[[0, 1, -2, -1], :] = np.nan                            # calculate_smoothed_field's border (spatial.py:106-107)
        if k % 9 == 7:
            j, i = int(rng.randint(2, nlat - 2)), int(rng.randint(0, nlon))
            a[j, i] = level                                          # a cell equal to the level
        padd = float(rng.choice([120, 120, 120, 60, 0, 360]))
        out.append(("c%02d" % k, a, level, nlon, dlon, padd))
    return out


def make_contours():
    from skimage import measure
    import skimage
    check_version()
    src = Source("indices/contour_index.py")
    fn = src.function("calculate_contours")
    round_block = src.statements(fn, 105, 117)                       # contours_index_expanded, closed, the loop
    expect(round_block, "contours_index_expanded", "closed", "check_closed", "indices", "check_len")
    dup_fn = src.function("check_duplicates", inside=fn)
    drop_block = src.statements(fn, 142, 147)
    expect(drop_block, "drop", "contours_index_expanded", "closed")
    frame_fn = src.function("contour_to_dataframe", inside=fn)
    stats = src.statements(frame_fn, 154, 155)
    expect(stats, "exp_lon", "mean_lat")
    orig = src.statements(fn, 179, 191)
    expect(orig, "contours_coordinates_original")
    import pandas as pd
    store = {"contours__skimage_version": np.array(skimage.__version__), "contours__numpy_version": np.array(np.__version__)}
    names = []
    for name, a, level, nlon, dlon, padd in contour_fields():
        ext = np.concatenate([a, a[:, :int(padd / dlon)]], axis=1)    # contour_index.py:94-100 (xarray: data movement, ours)
        try:
            cfm = measure.find_contours(ext, level)                   # :103, the real library
        except KeyError:
            continue                                                  # (order-dependent level hits: contours_levelwalk.npz has those)
        lat = np.linspace(-90, 90, a.shape[0])
        lon = np.arange(nlon) * dlon - 180.0
        kwargs = {"nlon": nlon, "dlon": dlon, "lon_name": "lon", "lat_name": "lat", "level": level}
        env = {"np": np, "itertools": itertools, "kwargs": kwargs, "contours_from_measure": cfm,
               "data": {"lon": pd.Series(lon), "lat": pd.Series(lat)}}
        src.run(round_block, env)
        before = [np.array(c) for c in env["contours_index_expanded"]]
        closed_before = list(env["closed"])
        src.run([dup_fn], env)
        src.run(drop_block, env)
        kept = [np.array(c) for c in env["contours_index_expanded"]]
        env["list_of_arrays"] = env["contours_index_expanded"]
        src.run(stats, env)
        src.run(orig, env)
        names.append(name)
        store["contours__%s__field" % name] = a
        store["contours__%s__par" % name] = np.array([level, nlon, dlon, padd])
        cat, off = pack(before)
        store["contours__%s__before" % name], store["contours__%s__before_off" % name] = cat, off
        store["contours__%s__before_closed" % name] = np.array(closed_before, dtype=bool)
        store["contours__%s__drop" % name] = np.array(sorted(env["drop"]), dtype=np.int64)
        cat, off = pack(kept)
        store["contours__%s__pts" % name], store["contours__%s__off" % name] = cat, off
        store["contours__%s__closed" % name] = np.array(env["closed"], dtype=bool)
        store["contours__%s__exp_lon" % name] = np.array(env["exp_lon"], dtype=np.float64)
        store["contours__%s__mean_lat" % name] = np.array(env["mean_lat"], dtype=np.float64)
        cat, off = pack(env["contours_coordinates_original"], dtype=np.float64)
        store["contours__%s__orig" % name], store["contours__%s__orig_off" % name] = cat, off
    store["contours__names"] = np.array(names)
    merge_into(OUT, store)
    print("contours:", len(names), "fields,", sum(len(store["contours__%s__drop" % n]) for n in names), "contours dropped by the duplicate rule,",
          sum(len(store["contours__%s__closed" % n]) for n in names), "kept")


# ------------------------------------------------------------------------------------------------ the pure numpy / pandas rules
def reference_combine_shared():
    src = Source("utils/index_utils.py")
    return src.run([src.function("combine_shared")], {})["combine_shared"]


def make_combine(store):
    f = reference_combine_shared()
    rng = np.random.RandomState(77)
    cases_a, cases_l = [], []
    for k in range(260):                                              # as check_groups / track_events call it: an (m, 2) int64 array
        n = int(rng.randint(1, 40))
        m = int(rng.randint(0, 3 * n))
        pairs = rng.randint(0, n, size=(m, 2))
        if k % 2 == 0:                                                # check_groups: every chord crosses itself
            pairs = np.concatenate([np.c_[np.arange(n), np.arange(n)], pairs], axis=0)
            pairs = pairs[np.lexsort((pairs[:, 1], pairs[:, 0]))]
        if k % 5 == 0:
            pairs = pairs * int(rng.choice([7, 8, 64, 1000]))          # other hash slots
        cases_a.append(pairs.astype(np.int64))
    for k in range(60):                                               # plain lists, ragged, empty members included
        lst = [list(map(int, rng.randint(0, 30, size=int(rng.randint(0, 5))))) for _ in range(int(rng.randint(0, 12)))]
        cases_l.append(lst)
    cases_l += [[[1, 8]], [[], [3, 1]], [[8, 1], [1, 16]], [[0, 1], [2, 3], [1, 2]], []]
    n = 0
    for kind, cases in (("a", cases_a), ("l", cases_l)):
        for k, c in enumerate(cases):
            out = f(c if kind == "a" else [list(x) for x in c])
            key = "combine__%s%03d" % (kind, k)
            if kind == "a":
                store[key + "__in"] = c
            else:
                flat, off = pack([np.asarray(x, dtype=np.int64).reshape(-1, 1) for x in c], width=1)
                store[key + "__in"], store[key + "__in_off"] = flat.reshape(-1), off
            flat, off = pack([np.asarray([int(v) for v in g], dtype=np.int64).reshape(-1, 1) for g in out], width=1)
            store[key + "__out"], store[key + "__out_off"] = flat.reshape(-1), off
            n += 1
    store["combine__count"] = np.array([len(cases_a), len(cases_l)])
    print("combine_shared:", n, "cases")


def synth_contours(dlon, t0, nt, eps, levels, seed):
    sys.path.insert(0, ROOT)
    from oracle import contours as ocont
    from oracle import synth
    lat, lon = synth.grid(dlon)
    pv = synth.pv_field(t0, nt, lat, lon, eps=eps, seed=seed)
    rows = ocont.calculate_contours(pv, levels, lat, lon, synth.times(nt), 120, original_coordinates=False)
    return lat, lon, pv, rows


def make_streamer(store):
    import pandas as pd
    src = Source("indices/streamer_index.py")
    fn = src.function("calculate_streamers")
    mod_dist = [st for st in src.tree.body if (isinstance(st, ast.ImportFrom) and st.module == "sklearn.metrics")
                or (isinstance(st, ast.Assign) and getattr(st.targets[0], "id", "") == "dist")]
    expect(mod_dist, "dist")
    pairs_block = src.statements(fn, 124, 157)
    expect(pairs_block, "contour_coords", "geo_matrix", "on", "cont_matrix", "check", "df_bp")
    dup_fn = src.function("check_duplicates", inside=fn)
    ovl_fn = src.function("check_overlapping", inside=fn)
    import sklearn
    store["streamer__sklearn_version"] = np.array(sklearn.__version__)
    store["streamer__pandas_version"] = np.array(pd.__version__)
    rng = np.random.RandomState(5)
    k = 0
    tot = [0, 0, 0]
    for dlon, t0, nt, eps, seed, params in ((2.5, 11, 6, 0.0, 1001, (800, 1500)), (2.5, 400, 6, 0.2, 1002, (800, 1500)),
                                            (2.5, 77, 4, 0.1, 1003, (1200, 1000)), (1.0, 3601, 2, 0.0, 42289, (800, 1500)),
                                            (1.0, 95, 2, 0.1, 1004, (600, 2000))):
        lat, lon, pv, rows = synth_contours(dlon, t0, nt, eps, [-2, 2], seed)
        nlon = len(lon)
        mx = max(r["exp_lon"] for r in rows)
        for r in rows:
            if r["exp_lon"] != mx:
                continue
            xy = np.asarray(r["geometry"]).astype("int")
            contour_index = pd.DataFrame(xy, columns=["x", "y"]).astype("int")     # streamer_index.py:121-123 without shapely
            env = {"np": np, "pd": pd, "itertools": itertools, "contour_index": contour_index,
                   "data": {"lat": pd.Series(lat), "lon": pd.Series(lon)},
                   "kwargs": {"lat_name": "lat", "lon_name": "lon", "nlon": nlon}, "geo_dis": params[0], "cont_dis": params[1]}
            src.run(mod_dist, env)
            src.run(pairs_block, env)
            df_bp = env["df_bp"].reset_index(drop=True)                # :263 (the loop's reset_index)
            src.run([dup_fn, ovl_fn], env)
            after_dup = env["check_duplicates"](df_bp).reset_index(drop=True)
            after_ovl = env["check_overlapping"](after_dup).reset_index(drop=True)
            # check_overlapping as the loop would reach it (after check_intersections: GEOS, not run) stands on any subset of
            # the pairs: a random half as a second input
            sub = after_dup[rng.rand(len(after_dup)) < 0.5].reset_index(drop=True)
            sub_ovl = env["check_overlapping"](sub).reset_index(drop=True)
            key = "streamer__s%02d" % k
            store[key + "__xy"] = xy.astype(np.int64)
            store[key + "__par"] = np.array([dlon, params[0], params[1], nlon], dtype=np.float64)
            store[key + "__on"] = np.asarray(env["on"], dtype=np.float64)
            store[key + "__pairs"] = df_bp[["ind1", "ind2"]].to_numpy().astype(np.int64)
            store[key + "__after_dup"] = after_dup[["ind1", "ind2"]].to_numpy().astype(np.int64)
            store[key + "__after_ovl"] = after_ovl[["ind1", "ind2"]].to_numpy().astype(np.int64)
            store[key + "__sub"] = sub[["ind1", "ind2"]].to_numpy().astype(np.int64)
            store[key + "__sub_ovl"] = sub_ovl[["ind1", "ind2"]].to_numpy().astype(np.int64)
            tot[0] += len(df_bp); tot[1] += len(df_bp) - len(after_dup); tot[2] += len(after_dup) - len(after_ovl)
            k += 1
    store["streamer__count"] = np.array(k)
    store["streamer__lat_2.5"], store["streamer__lon_2.5"] = synth_grid(2.5)
    store["streamer__lat_1.0"], store["streamer__lon_1.0"] = synth_grid(1.0)
    print("streamer:", k, "contours,", tot[0], "candidate pairs,", tot[1], "dropped as duplicates,", tot[2], "as overlapped")


def synth_grid(dlon):
    sys.path.insert(0, ROOT)
    from oracle import synth
    return synth.grid(dlon)


def make_overturn(store):
    import pandas as pd
    src = Source("indices/overturning_index.py")
    fn = src.function("calculate_overturnings")
    groups_block = src.statements(fn, 123, 134)
    expect(groups_block, "lons", "counts", "ot_lons", "groups", "df_ot")
    fns = [src.function(n, inside=fn) for n in ("check_duplicates", "check_expansion", "find_lat_expansion")]
    loop = src.statements(fn, 183, 188)
    expect(loop, "routines", "i", "df_ot")
    k = 0
    nev = 0
    rng = np.random.RandomState(9)
    for dlon, t0, nt, eps, seed in ((2.5, 11, 8, 0.0, 2001), (2.5, 500, 8, 0.25, 2002), (1.0, 3601, 3, 0.0, 2003), (1.0, 40, 3, 0.15, 2004)):
        lat, lon, pv, rows = synth_contours(dlon, t0, nt, eps, [-2, 2], seed)
        nlon = len(lon)
        mx = max(r["exp_lon"] for r in rows)
        for r in rows:
            if r["exp_lon"] != mx:
                continue
            for range_group, min_exp in ((5, 5), (float(rng.choice([2.5, 10, 20])), float(rng.choice([0, 2.5, 7.5, 15])))):
                xy = np.asarray(r["geometry"]).astype("int")
                contour_index = pd.DataFrame(xy, columns=["x", "y"]).astype("int")  # overturning_index.py:120-122 without shapely
                env = {"np": np, "pd": pd, "itertools": itertools, "contour_index": contour_index,
                       "kwargs": {"dlon": dlon, "nlon": nlon}, "range_group": range_group, "min_exp": min_exp}
                src.run(groups_block, env)
                groups0 = env["df_ot"].to_numpy().astype(np.int64).reshape(-1, 2)
                src.run(fns, env)
                src.run(loop, env)
                df = env["df_ot"]
                key = "overturn__o%03d" % k
                store[key + "__xy"] = xy.astype(np.int64)
                store[key + "__par"] = np.array([dlon, nlon, range_group, min_exp], dtype=np.float64)
                store[key + "__groups"] = groups0
                store[key + "__out"] = (df[["min_lon", "max_lon", "min_lat", "max_lat"]].to_numpy().astype(np.int64)
                                        if len(df) else np.zeros((0, 4), dtype=np.int64))
                nev += len(df)
                k += 1
    store["overturn__count"] = np.array(k)
    print("overturning:", k, "contours x parameters,", nev, "boxes")


def make_select(store):
    import pandas as pd
    s_src = Source("indices/streamer_index.py")
    sel = s_src.statements(s_src.function("calculate_streamers"), 106, 108)
    expect(sel, "contours")
    c_src = Source("indices/cutoff_index.py")
    cut = c_src.statements(c_src.function("calculate_cutoffs"), 89, 93)
    expect(cut, "contours")
    rng = np.random.RandomState(3)
    for k in range(40):
        dlon = float(rng.choice([0.25, 0.5, 1.0, 2.5, 0.3]))
        n = int(rng.randint(1, 60))
        W = int(360 / dlon) + int(120 / dlon)
        nux = rng.randint(1, W + 1, size=n)
        if k % 2 == 0:
            nux[rng.randint(0, n, size=1 + n // 8)] = W
        exp_lon = nux * dlon                                           # contour_index.py:154
        closed = rng.rand(n) < 0.5
        min_exp = float(rng.choice([5, 2.5, 0, 7.3]))
        frame = pd.DataFrame({"orig": np.arange(n), "exp_lon": exp_lon, "closed": closed})
        env = {"contours": frame.copy()}
        s_src.run(sel, env)
        picked = env["contours"]["orig"].to_numpy()
        env = {"contours": frame.copy(), "min_exp": min_exp}
        c_src.run(cut, env)
        key = "select__%02d" % k
        store[key + "__nux"] = nux.astype(np.int64)
        store[key + "__closed"] = closed
        store[key + "__par"] = np.array([dlon, min_exp])
        store[key + "__selected"] = picked.astype(np.int64)
        store[key + "__cutoffs"] = env["contours"]["orig"].to_numpy().astype(np.int64)
    store["select__count"] = np.array(40)
    print("selection / cutoff filter: 40 frames")


def make_resolution(store):
    src = Source("utils/data_utils.py")
    f = src.run([src.function("get_spatial_resolution")], {"np": np})["get_spatial_resolution"]
    cases = [np.arange(-180, 180, 0.25), np.arange(-90, 90.1, 1.0), np.linspace(-90, 90, 721), np.linspace(-90, 90, 181),
             np.linspace(90, -90, 73), np.arange(0, 360, 2.5)[::-1], np.array([0.0, 1.0, 2.0, 4.0]), np.array([5.0, 5.0, 5.0]),
             np.arange(0, 360, 0.3), np.linspace(-89, 89, 179), np.arange(-180, 180, 1.0).astype(np.float32),
             np.array([0.0, 1.0, 1.0, 2.0]), np.arange(-180, 180, 0.1), np.linspace(0, 359.75, 1440)]
    for k, c in enumerate(cases):
        try:
            res, err = float(f({"d": types.SimpleNamespace(data=c)}, "d")), ""
        except ValueError as e:
            res, err = np.nan, str(e)
        store["resolution__%02d__coord" % k] = c
        store["resolution__%02d__out" % k] = np.array(res)
        store["resolution__%02d__err" % k] = np.array(err)
    store["resolution__count"] = np.array(len(cases))
    print("get_spatial_resolution:", len(cases), "axes,", sum(1 for k in range(len(cases)) if str(store["resolution__%02d__err" % k])), "raise")


def make_props(store):
    import pandas as pd
    sys.path.insert(0, ROOT)
    from oracle import api as oapi
    from oracle import synth
    src = Source("utils/index_utils.py")
    fn = src.function("calculate_properties")
    coords = src.statements(fn, 54, 63)                               # merged["lon"/"lat"], area_cell, weight_lat, merged["areas"]
    expect(coords, "area_cell", "weight_lat")
    com_cols = src.statements(fn, 91, 92)                             # x_com, y_com
    tail = src.statements(fn, 94, 120)                                # agg_merged, com_x, com_y, com, prop_dict
    expect(tail, "agg_merged", "com_x", "com_y", "com", "prop_dict")
    k = 0
    for dlon, t0, nt, eps, seed in ((2.5, 11, 3, 0.0, 3001), (2.5, 300, 3, 0.2, 3002), (1.0, 3601, 1, 0.05, 3003)):
        lat, lon = synth.grid(dlon)
        pv = synth.pv_field(t0, nt, lat, lon, eps=eps, seed=seed)
        inten = synth.pv_field(t0 + 7, nt, lat, lon, eps=eps, seed=seed + 50)
        tm = synth.times(nt)
        ref = oapi.run_all(pv, [-2, 2], lat, lon, tm, intensity=inten)
        for name in ("streamers", "overturnings", "cutoffs"):
            rows = ref[name]
            if not rows:
                continue
            ids = np.concatenate([np.full(len(r["_cells"][0]), i) for i, r in enumerate(rows)])
            xs = np.concatenate([r["_cells"][0] for r in rows]).astype(np.int64)
            ys = np.concatenate([r["_cells"][1] for r in rows]).astype(np.int64)
            ts = np.concatenate([np.full(len(r["_cells"][0]), r["_t"]) for r in rows])
            events = pd.DataFrame({"id": np.arange(len(rows)), "date": [r["date"] for r in rows], "level": [r["level"] for r in rows]})
            if name == "overturnings":
                events["orientation"] = [r["orientation"] for r in rows]
            merged = pd.DataFrame({"id": ids, "x": xs, "y": ys})       # what the join (:44-51, GEOS) would return, (y, x) order per event
            kwargs = {"lon_name": "lon", "lat_name": "lat", "nlon": len(lon), "nlat": len(lat), "dlon": dlon, "dlat": dlon}
            env = {"np": np, "pd": pd, "merged": merged, "events": events, "kwargs": kwargs,
                   "data": {"lon": pd.Series(lon), "lat": pd.Series(lat)}}
            src.run(coords, env)
            # :66-89 multiply by the xarray point selection of the field (xarray: absent; the selection is data movement)
            merged["mean_var"] = merged.areas * pv[ts, ys, xs % len(lon)].astype(np.float64)
            merged["intensity"] = merged.areas * inten[ts, ys, xs % len(lon)].astype(np.float64)
            src.run(com_cols, env)
            src.run(tail, env)
            pdict = env["prop_dict"]
            agg = env["agg_merged"]
            key = "props__p%02d" % k
            store[key + "__cells"] = np.c_[ids, xs, ys, ts].astype(np.int64)
            store[key + "__field"] = np.array([dlon, t0, nt, eps, seed], dtype=np.float64)
            store[key + "__sums"] = agg[["areas", "mean_var", "intensity", "x_com", "y_com"]].to_numpy().astype(np.float64)
            store[key + "__com"] = np.asarray(env["com"], dtype=np.float64).reshape(-1, 2)
            store[key + "__mean_var"] = np.asarray(pdict["mean_var"], dtype=np.float64)
            store[key + "__intensity"] = np.asarray(pdict["intensity"], dtype=np.float64)
            store[key + "__event_area"] = np.asarray(pdict["event_area"], dtype=np.float64)
            store[key + "__columns"] = np.array(list(pdict.keys()))
            store[key + "__weight_lat"] = np.asarray(env["weight_lat"], dtype=np.float64)
            k += 1
    store["props__count"] = np.array(k)
    print("calculate_properties tail:", k, "event tables")


def make_track(store):
    import pandas as pd
    src = Source("processing/events.py")
    mod_dist = [st for st in src.tree.body if (isinstance(st, ast.ImportFrom) and st.module == "sklearn.metrics")
                or (isinstance(st, ast.Assign) and getattr(st.targets[0], "id", "") == "dist")]
    expect(mod_dist, "dist")
    fn = src.function("track_events")
    fn.decorator_list = []                                            # (type / emptiness checks: they need geopandas' names)
    env = {"np": np, "pd": pd, "itertools": itertools,
           "index_utils": types.SimpleNamespace(combine_shared=reference_combine_shared())}
    src.run(mod_dist, env)
    track = src.run([fn], env)["track_events"]
    rng = np.random.RandomState(21)
    for k in range(24):
        n = int(rng.randint(2, 60))
        steps = np.sort(rng.randint(0, int(rng.randint(2, 12)), size=n))
        if k % 4 == 3:
            rng.shuffle(steps)
        dates = (np.datetime64("1980-01-01T00") + steps * np.timedelta64(6, "h")).astype("datetime64[ns]")
        lon = np.round(rng.uniform(-180, 180, n) * 4) / 4
        lat = np.round(rng.uniform(-85, 85, n) * 4) / 4
        if k % 3 == 0:                                                # clusters, so that chains of links form
            c0 = rng.randint(0, 4, n)
            lon = np.round((np.array([-120, -30, 40, 130])[c0] + rng.uniform(-6, 6, n) + steps * 1.5) * 4) / 4
            lat = np.round((np.array([50, -45, 60, -55])[c0] + rng.uniform(-4, 4, n)) * 4) / 4
        events = pd.DataFrame({"date": dates, "com": list(zip(lon.tolist(), lat.tolist())), "row": np.arange(n)})
        time_range = None if k % 5 == 0 else float(rng.choice([6, 12, 18]))
        distance = float(rng.choice([500, 1000, 2500]))
        try:
            out = track(events, time_range=time_range, method="by_distance", distance=distance)
            rows, labels, err = out["row"].to_numpy(), out["label"].to_numpy(), ""
        except ValueError as e:
            rows, labels, err = np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), str(e)
        key = "track__%02d" % k
        store[key + "__dates"] = dates.astype(np.int64)
        store[key + "__com"] = np.c_[lon, lat]
        store[key + "__par"] = np.array([np.nan if time_range is None else time_range, distance])
        store[key + "__rows"] = rows.astype(np.int64)
        store[key + "__labels"] = labels.astype(np.int64)
        store[key + "__err"] = np.array(err)
    store["track__count"] = np.array(24)
    print("track_events(by_distance): 24 catalogues")


def make_rules():
    import pandas as pd
    check_version()
    store = {"rules__pandas_version": np.array(pd.__version__), "rules__numpy_version": np.array(np.__version__),
             "rules__python_version": np.array(sys.version.split()[0])}
    make_combine(store)
    make_resolution(store)
    make_select(store)
    make_overturn(store)
    make_streamer(store)
    make_props(store)
    make_track(store)
    merge_into(OUT, store)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else ""
    if not os.path.isdir(REF):
        raise SystemExit("this generator runs in the build container only (it executes statements of /root/reference)")
    if which == "contours":
        make_contours()
    elif which == "rules":
        make_rules()
    else:
        raise SystemExit(__doc__)
