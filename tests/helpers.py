"""Shared parity helpers: run the C ABI (HIP build on the GPU box, host-simulation build on CPU)
and the oracle on the same input and compare every table."""
import os
import subprocess

import numpy as np

from oracle import api as oapi
from oracle import contours as ocont
from oracle import synth
from wavebreaking_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTSIM_DIR = os.path.join(ROOT, "tests", "_hostsim")
HOSTSIM_LIB = os.path.join(HOSTSIM_DIR, "libwb_hostsim.so")
CSRC = os.path.join(ROOT, "wavebreaking_amd", "csrc")


def build_hostsim(force=False, variant="", defines=(), sanitize=False):
    """g++ build of the kernel sources with -DWB_HOSTSIM (test infrastructure only).  ``variant`` / ``defines``
    build a second copy with other compile-time limits (e.g. a tiny WB_TRACE_LDS_N to force the big-job paths);
    ``sanitize``: with AddressSanitizer + UndefinedBehaviorSanitizer (loaded in a child process with libasan preloaded)."""
    if sanitize:
        variant = (variant + "_" if variant else "") + "asan"
    lib = HOSTSIM_LIB if not variant else HOSTSIM_LIB.replace(".so", "_%s.so" % variant)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    srcs.append(os.path.join(ROOT, "include", "wavebreaking_hip.h"))
    srcs.append(os.path.join(ROOT, "include", "wavebreaking_hip_dev.h"))
    if (not force and os.path.exists(lib)
            and os.path.getmtime(lib) >= max(os.path.getmtime(s) for s in srcs)):
        return lib
    os.makedirs(HOSTSIM_DIR, exist_ok=True)
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"] if sanitize else []
    cmd = ["g++", "-O1" if sanitize else "-O2", "-g", *san, "-DWB_HOSTSIM", *["-D" + d for d in defines], "-x", "c++", "-std=c++17",
           "-ffp-contract=off", "-fno-strict-aliasing", "-shared", "-fPIC", os.path.join(CSRC, "wb_api.hip"), "-o", lib]
    subprocess.check_call(cmd)
    return lib


_hostsim = None


def hostsim_library():
    global _hostsim
    if _hostsim is None:
        _hostsim = _hip.Library(build_hostsim(), allow_hostsim=True)
    return _hostsim


def run_abi(ctx, data, lat, lon, levels, intensity=None, periodic_add=120.0, want_cells=True, **kw):
    dlon = ocont.spatial_resolution(np.asarray(lon, dtype=np.float64), "lon")
    dlat = ocont.spatial_resolution(np.asarray(lat, dtype=np.float64), "lat")
    b = ctx.contours(data, lat, lon, levels, dlon, dlat, periodic_add=periodic_add, intensity=intensity,
                     want_cells=want_cells, **kw)
    b.indices()
    out = dict(contours=b.contour_table(), streamers=b.events(_hip.STREAMERS), overturnings=b.events(_hip.OVERTURNINGS),
               cutoffs=b.events(_hip.CUTOFFS), status=b.status(), band=b.band_hits(), max_nux=b.max_nux(),
               com_rechecks=b.com_rechecks(),
               timings=b.timings(),
               geometry=[b.event_geometry(k) for k in (_hip.STREAMERS, _hip.OVERTURNINGS, _hip.CUTOFFS)])
    b.close()
    return out


GEOMETRY_CHECKED = {"events": 0, "split": 0}


def check_event_geometry(ctx, got, lat, lon):
    """transform_polygons (index_utils.py:129-184): the rings wb_indices left in host memory -- cut, truncated and mapped by the
    device kernels of csrc/wb_geom.h (the host simulation runs the same sources) -- against the HOST function on the very
    polygons of the event tables, ring for ring and bit for bit.  (The host function itself is checked against the exact
    oracle splitter in tests/test_seam_split.py and, frame by frame, in tests/test_python_api.py.)"""
    lon = np.ascontiguousarray(lon, dtype=np.float64)
    lat = np.ascontiguousarray(lat, dtype=np.float64)
    for kind, name in enumerate(("streamers", "overturnings", "cutoffs")):
        ev = got[name]
        ro, vo, gx, gy = (np.asarray(a) for a in got["geometry"][kind])
        po = np.asarray(ev["poly_offset"])
        n = len(po) - 1
        assert len(ro) == n + 1, (name, len(ro), n)
        if n == 0:
            continue
        polys = [np.c_[ev["poly_x"][po[k]:po[k + 1]], ev["poly_y"][po[k]:po[k + 1]]] for k in range(n)]
        host = ctx.transform_polygons(polys, lon, lat, lon.size)
        for k in range(n):
            rings = [np.c_[gx[vo[q]:vo[q + 1]], gy[vo[q]:vo[q + 1]]] for q in range(ro[k], ro[k + 1])]
            assert len(rings) == len(host[k]) and all(np.array_equal(a, b) for a, b in zip(rings, host[k])), (name, k)
            GEOMETRY_CHECKED["split"] += len(rings) != 1
        GEOMETRY_CHECKED["events"] += n


def compare_contours(got, ref_rows, levels, dlon):
    """got: contour table dict from the ABI; ref_rows: oracle rows (index coordinates)."""
    assert got["step"].size == len(ref_rows), "contour count %d != oracle %d" % (got["step"].size, len(ref_rows))
    levels = list(levels)
    for k, r in enumerate(ref_rows):
        o, n = int(got["offset"][k]), int(got["npts"][k])
        xy = np.c_[got["x"][o:o + n], got["y"][o:o + n]]
        assert got["step"][k] == r["_t"], k
        assert levels[got["level_idx"][k]] == r["level"], k
        assert np.array_equal(xy, r["geometry"]), "contour %d differs" % k
        assert bool(got["closed"][k]) == bool(r["closed"]), k
        assert got["nux"][k] * dlon == r["exp_lon"], k
        assert np.round(got["sum_y"][k] / n, 2) == r["mean_lat"], k


def compare_events(got, contours, ref_rows, kind, levels, rtol=1e-6):
    assert got["step"].size == len(ref_rows), "%s: %d events != oracle %d" % (kind, got["step"].size, len(ref_rows))
    levels = list(levels)
    for k, r in enumerate(ref_rows):
        assert got["step"][k] == r["_t"], (kind, k)
        assert levels[got["level_idx"][k]] == r["level"], (kind, k)
        c = int(got["contour"][k])
        o = int(contours["offset"][c])
        if kind == "overturnings":
            x0, y0, x1, y1 = got["box"][k]
            poly = np.array([[x1, y0], [x1, y1], [x0, y1], [x0, y0]])
            assert ("cyclonic" if got["orientation"][k] == 0 else "anticyclonic") == r["orientation"], k
        else:
            i1, i2 = int(got["ind1"][k]), int(got["ind2"][k])
            poly = np.c_[contours["x"][o + i1:o + i2 + 1], contours["y"][o + i1:o + i2 + 1]]
        assert np.array_equal(poly, r["_polygon"]), (kind, k)
        a, b = got["poly_offset"][k], got["poly_offset"][k + 1]
        assert np.array_equal(np.c_[got["poly_x"][a:b], got["poly_y"][a:b]], r["_polygon"]), (kind, k)
        # cells: bit-exact
        xs, ys = r["_cells"]
        assert got["ncells"][k] == len(xs), (kind, k)
        if "cell_offset" in got:
            a, b = got["cell_offset"][k], got["cell_offset"][k + 1]
            assert np.array_equal(got["cell_x"][a:b], xs) and np.array_equal(got["cell_y"][a:b], ys), (kind, k)
        assert bool(got["flags"][k] & _hip.EVENT_NOT_SIMPLE) == bool(r["_not_simple"]), (kind, k)
        ref = np.asarray(r["_sums"])
        s = got["sums"][k]
        # com = (lon[int(x_com / areas) % nlon], lat[int(y_com / areas)]) (index_utils.py:105-110): the truncated indices
        # are bit-exact against the reference's arithmetic (per-cell products, pandas' compensated group sum) ...
        assert int(s[3] / s[0]) == int(ref[3] / ref[0]) and int(s[4] / s[0]) == int(ref[4] / ref[0]), (kind, k, s, ref)
        # ... because every event whose quotient lies within the rounding slack of an integer is re-evaluated in the
        # reference's order: there the three sums themselves are equal to the last bit
        rows = int(ys.max() - ys.min()) + 1
        for q in (ref[3] / ref[0], ref[4] / ref[0]):
            if abs(q - np.rint(q)) <= (abs(q) + 1.0) * (2.0 * rows + 16.0) * 2.0 ** -53:
                assert s[0] == ref[0] and s[3] == ref[3] and s[4] == ref[4], (kind, k, s, ref)
                NEAR_INTEGER_EVENTS[kind] = NEAR_INTEGER_EVENTS.get(kind, 0) + 1
        assert np.allclose(s[[0, 3, 4]], ref[[0, 3, 4]], rtol=1e-12, atol=0.0), (kind, k, s, ref)
        # data sums: 1e-6 relative (north_star)
        assert np.allclose(s[1:3], ref[1:3], rtol=rtol, atol=1e-9 * abs(ref[0])), (kind, k, s, ref)


NEAR_INTEGER_EVENTS = {}     # events seen by compare_events whose com quotient sat within the slack of an integer


def full_parity(ctx, data, lat, lon, times, levels, intensity=None, periodic_add=120.0, **kw):
    """Everything the path produces, ABI vs oracle.  Returns (abi tables, oracle tables)."""
    got = run_abi(ctx, data, lat, lon, levels, intensity=intensity, periodic_add=periodic_add, **kw)
    # index parameters go to both sides (the oracle has ONE min_exp, as run_all calls the two index functions with the same)
    okw = {k: kw[k] for k in ("geo_dis", "cont_dis", "range_group") if k in kw}
    if "ot_min_exp" in kw or "co_min_exp" in kw:
        assert kw.get("ot_min_exp", 5.0) == kw.get("co_min_exp", 5.0)
        okw["min_exp"] = kw.get("ot_min_exp", 5.0)
    ref = oapi.run_all(data, levels, lat, lon, times, intensity=intensity, periodic_add=periodic_add, **okw)
    dlon = ocont.spatial_resolution(np.asarray(lon, dtype=np.float64), "lon")
    compare_contours(got["contours"], ref["contours"], levels, dlon)
    for kind in ("streamers", "overturnings", "cutoffs"):
        compare_events(got[kind], got["contours"], ref[kind], kind, levels)
        check_property_columns(got[kind], ref[kind], lat, lon)
    check_event_geometry(ctx, got, lat, lon)
    return got, ref


PROPS_CHECKED = {"events": 0}


def check_property_columns(got, ref_rows, lat, lon):
    """The frame's property columns as the DEVICE left them beside the sums (wb_k_pack_results -> csrc/wb_props.h:wb_props_columns;
    index_utils.py:104-120): bit for bit numpy's statement of those lines on the same sums (utils.index_utils.properties_from_sums,
    itself pinned to the reference's own lines by tests/golden/reference_rules.npz), and -- the truncated centre of mass exactly,
    the rounded columns wherever the 1e-6 slack of the sums does not straddle a rounding boundary -- the oracle's rows."""
    from wavebreaking_amd.utils.index_utils import properties_from_sums
    n = got["step"].size
    if not n:
        return
    com, mean_var, inten, area = properties_from_sums(got["sums"], np.asarray(lon), np.asarray(lat), len(lon))
    p = got["props"]
    assert p.shape == (n, 5)
    assert np.array_equal(p[:, :2], np.asarray(com)) and np.array_equal(p[:, 2], mean_var), "com / mean_var columns"
    assert np.array_equal(p[:, 3], inten) and np.array_equal(p[:, 4], area), "intensity / event_area columns"
    for k, r in enumerate(ref_rows):
        assert tuple(p[k, :2]) == tuple(r["com"]), (k, p[k], r["com"])
    PROPS_CHECKED["events"] += n


def symmetric_cutoff_case(ctx, dlon=1.0, nsteps=2):
    """Cutoffs whose polygons are mirror-symmetric in x (x_com / areas is an integer or a half-integer mathematically), cut
    from a contour frame handed in (decorator_contour_calculation, contour_index.py:197-249): the events whose
    centre-of-mass sums the rasterising kernel has to re-evaluate in the reference's order.  Returns (ABI tables,
    oracle rows, rechecks)."""
    from oracle import indices as oind
    from oracle import properties as oprop
    from oracle import synth
    lat, lon = synth.grid(dlon)
    nlon = lon.size
    rng = np.random.RandomState(5)
    data = (rng.standard_normal((nsteps, lat.size, nlon)) * 2).astype(np.float32)
    W = nlon + int(120 / dlon)
    rows = []
    for t in range(nsteps):
        # one contour across the whole extended grid (so that the closed ones are shorter than the maximum), then closed
        # shapes: rectangles, diamonds, octagons; centre column integer or half-integer
        xs = np.arange(W)
        rows.append(dict(step=t, closed=False, xy=np.c_[xs, np.full(W, lat.size // 2) + (xs % 2)]))
        for k in range(14):
            sc = max(1, int(round(1.0 / dlon)))
            cx2 = 2 * int(rng.randint(30 * sc, nlon - 30 * sc)) + (k % 2)   # twice the centre column
            cy = int(rng.randint(30 * sc, lat.size - 30 * sc))
            hw2 = 2 * int(rng.randint(3 * sc, 12 * sc)) + (k % 2)           # twice the half width (same parity as cx2)
            hh = int(rng.randint(2 * sc, 9 * sc))
            x0, x1 = (cx2 - hw2) // 2, (cx2 + hw2) // 2
            if k % 3 == 0:
                ring = [(x0, cy - hh), (x1, cy - hh), (x1, cy + hh), (x0, cy + hh)]
            elif k % 3 == 1:
                m = min(hh, (x1 - x0) // 2)
                ring = [(x0 + m, cy - hh), (x1 - m, cy - hh), (x1, cy - hh + m), (x1, cy + hh - m), (x1 - m, cy + hh),
                        (x0 + m, cy + hh), (x0, cy + hh - m), (x0, cy - hh + m)]
            else:
                h = (x1 - x0) // 2
                ring = [(x0, cy), (x0 + h, cy - h), (x1 - h, cy - h), (x1, cy), (x1 - h, cy + h), (x0 + h, cy + h)]
            ring = [p for i, p in enumerate(ring) if p != ring[i - 1]]
            dense = []                                                     # every lattice point along the ring, like a traced contour
            for i, (ax, ay) in enumerate(ring):
                bx, by = ring[(i + 1) % len(ring)]
                g = int(np.gcd(abs(bx - ax), abs(by - ay)))
                dense += [(ax + (bx - ax) // g * q, ay + (by - ay) // g * q) for q in range(g)]
            ring = dense
            rows.append(dict(step=t, closed=True, xy=np.asarray(ring)))
    npts = np.array([len(r["xy"]) for r in rows], dtype=np.int32)
    table = dict(step=np.array([r["step"] for r in rows], dtype=np.int32), level_idx=np.zeros(len(rows), np.int32),
                 closed=np.array([r["closed"] for r in rows], dtype=np.int32), npts=npts,
                 offset=(np.cumsum(npts) - npts).astype(np.int64),
                 x=np.concatenate([r["xy"][:, 0] for r in rows]).astype(np.int32),
                 y=np.concatenate([r["xy"][:, 1] for r in rows]).astype(np.int32))
    b = ctx.batch_from_contours(table, data, lat, lon, [2.0], dlon, dlon, intensity=data, want_cells=True)
    b.indices(streamers=False, overturnings=False)
    got = b.events(_hip.CUTOFFS)
    rechecks = b.com_rechecks()
    assert not b.status().any()
    b.close()
    tm = synth.times(nsteps)
    crow = [dict(date=tm[r["step"]], level=2.0, closed=r["closed"], exp_lon=len(set(r["xy"][:, 0].tolist())) * dlon,
                 geometry=r["xy"], _t=r["step"]) for r in rows]
    ev = oind.calculate_cutoffs(crow, 5)
    ref = oprop.calculate_properties(ev, data, data, lat, lon, 120)
    assert got["step"].size == len(ref) and len(ref) >= 10 * nsteps, (got["step"].size, len(ref))
    nnear = 0
    for k, r in enumerate(ref):
        xs, ys = r["_cells"]
        a, e = got["cell_offset"][k], got["cell_offset"][k + 1]
        assert np.array_equal(got["cell_x"][a:e], xs) and np.array_equal(got["cell_y"][a:e], ys), k
        s, q = got["sums"][k], np.asarray(r["_sums"])
        assert int(s[3] / s[0]) == int(q[3] / q[0]) and int(s[4] / s[0]) == int(q[4] / q[0]), (k, s, q)
        qx = q[3] / q[0]
        if abs(qx - np.rint(qx)) < 1e-9:
            nnear += 1
            assert s[0] == q[0] and s[3] == q[3] and s[4] == q[4], (k, s, q)
        assert np.allclose(s, q, rtol=1e-6)
    return got, ref, rechecks, nnear


def level_hit_goldens_case(ctx, golden_dir, fname="contours_levelhit.npz"):
    """Every field of tests/golden/contours_levelhit.npz (cells equal to the level; scikit-image's own outputs) through the
    ABI as a one-step batch without periodic columns: the contour table equals contour_index.py:103-156 applied to
    skimage's contours -- also where two contour ends compete for a vertex (STATUS_LEVEL_ORDER: the job is then joined by
    the sequential walk, STATUS_LEVEL_WALKED) -- or the job carries STATUS_LEVEL_KEYERROR exactly on the fields on which
    skimage raises KeyError.  Returns (fields equal to skimage, fields flagged)."""
    from oracle import marching
    g = np.load(os.path.join(golden_dir, fname))
    nequal = nflag = 0
    for name in g["names"]:
        a, lv = g[name + "__field"].astype(np.float64), float(g[name + "__level"])
        nlat, nlon = a.shape
        lat, lon = np.arange(nlat, dtype=np.float64), np.arange(nlon, dtype=np.float64)
        b = ctx.contours(a[None], lat, lon, [lv], 1.0, 1.0, periodic_add=0.0)
        got, st = b.contour_table(), int(b.status()[0])
        b.close()
        overwritten = marching.find_contours_vertex_machine(a, lv)[1]
        assert bool(st & _hip.STATUS_LEVEL_ORDER) == (overwritten > 0), (name, st, overwritten)
        if int(g[name + "__raises"]):
            # scikit-image raises KeyError on this field: the sequential walk of the flagged job finds the same missing entry
            assert st & _hip.STATUS_LEVEL_ORDER and st & _hip.STATUS_LEVEL_KEYERROR, (name, st)
            nflag += 1
            continue
        assert not (st & _hip.STATUS_LEVEL_KEYERROR), (name, st)
        if overwritten:
            assert st & _hip.STATUS_LEVEL_WALKED, (name, st)
            nflag += 1
        off, pts = g[name + "__off"], g[name + "__pts"]
        raw = [pts[off[i]:off[i + 1]] for i in range(len(off) - 1)]            # scikit-image's contours themselves
        items, closed = [], []
        for item in raw:                                                      # contour_index.py:108-117
            xy, cl = ocont.round_dedupe(item)
            if len(xy) >= 4:
                items.append(xy)
                closed.append(cl)
        drop = ocont.check_duplicates(items, nlon)                            # :119-147
        keep = [i for i in range(len(items)) if i not in drop]
        assert got["step"].size == len(keep), (name, got["step"].size, len(keep))
        for k, i in enumerate(keep):
            o, n = int(got["offset"][k]), int(got["npts"][k])
            assert np.array_equal(np.c_[got["x"][o:o + n], got["y"][o:o + n]], items[i]), (name, k)
            assert bool(got["closed"][k]) == bool(closed[i]), (name, k)
        nequal += 1
    return nequal, nflag


def level_hit_field(seed, dlon=1.0, eps=0.3, nhits=4, level=2.0, seam=True):
    """synthetic PV with cells set EQUAL to the level next to the contour (some adjacent pairs, one in the first columns
    so that it appears twice in the periodically extended field)"""
    from oracle import synth
    lat, lon = synth.grid(dlon)
    rng = np.random.RandomState(1000 + seed)
    pv = synth.pv_field(seed, 1, lat, lon, eps=eps)
    a = pv[0]
    hi = a > level
    m = np.zeros_like(hi)
    m[:-1] |= hi[:-1] != hi[1:]
    m[1:] |= hi[:-1] != hi[1:]
    m[:, :-1] |= hi[:, :-1] != hi[:, 1:]
    m[:, 1:] |= hi[:, :-1] != hi[:, 1:]
    c = np.argwhere(m)
    for _ in range(nhits):
        j, i = c[rng.randint(len(c))]
        a[j, i] = level
        if rng.rand() < 0.3:
            a[j, (i + 1) % lon.size] = level
    if seam:
        cc = c[c[:, 1] < 3]
        if len(cc):
            a[cc[0][0], cc[0][1]] = level
    return pv, lat, lon


def half_way_crossing_field(seed, dlon=2.5, dtype=np.float32, level=2.0):
    """synthetic PV whose cells next to the `level` contour sit at distances from the level that put the crossings AT one half of
    their edges (exact ties of np.round: half to even), a few float32 / float64 ulps beside it, and a few 1e-4 .. 1e-2 beside it --
    the band in which wb_k_trace_fused leaves its float32 pre-decision for the reference's float64 quotient.  No cell equals the level."""
    from oracle import synth
    lat, lon = synth.grid(dlon)
    rng = np.random.RandomState(5000 + seed)
    pv = synth.pv_field(seed, 2, lat, lon, eps=0.25).astype(np.float64)
    for a in pv:
        hi = a > level
        m = np.zeros_like(hi)
        m[:-1] |= hi[:-1] != hi[1:]
        m[1:] |= hi[:-1] != hi[1:]
        m[:, :-1] |= hi[:, :-1] != hi[:, 1:]
        m[:, 1:] |= hi[:, :-1] != hi[:, 1:]
        dist = rng.choice([0.5, 1.0, 1.5, 3.0], size=a.shape)      # |value - level| of the cells at the contour: equal distances on both sides = ties
        kind = rng.randint(0, 4, size=a.shape)
        eps_t = np.finfo(dtype).eps
        off = np.where(kind == 0, 0.0, np.where(kind == 1, rng.randint(-3, 4, size=a.shape) * eps_t,
                       np.where(kind == 2, rng.uniform(-4e-3, 4e-3, size=a.shape), rng.uniform(-4e-2, 4e-2, size=a.shape))))
        q = level + np.where(hi, 1.0, -1.0) * dist * (1.0 + off)
        a[m] = q[m]
    pv = pv.astype(dtype)
    assert not np.any(pv == level)
    return pv, lat, lon


def fused_growth_case(ctx, gpu=False):
    from oracle import synth
    lat, lon = synth.grid(2.5)
    smooth = synth.pv_field(3, 2, lat, lon, eps=0.0)
    noisy = synth.pv_field(7, 2, lat, lon, eps=0.0)
    rng = np.random.RandomState(5)                              # four rows of noise around the level: a hundred small loops per job of level 2
    noisy[:, 8:12, :] = (2.0 + rng.uniform(-0.5, 0.5, size=(2, 4, lon.size))).astype(noisy.dtype)

    def counters(pv):
        b = ctx.contours(pv, lat, lon, [-2.0, 2.0], 2.5, 2.5)
        c = (b.counter("fused_declined"), b.counter("fused_lds_bytes"), b.counter("fused_jobs_per_cu"))
        b.close()
        return c

    first, second = counters(smooth), counters(smooth)
    # the second launch is sized by what the first one saw (a few chains, forty splitters): less LDS, nothing declined
    assert first[0] == 0 and second[0] == 0 and second[1] < first[1], (first, second)
    if gpu:
        assert second[2] >= 4, second
    seen = [counters(noisy)]
    assert seen[0][0] > 0, seen                                 # more segments, splitters and chains than that launch admits
    full_parity(ctx, noisy, lat, lon, synth.times(2), [-2.0, 2.0])          # ... taken by the three-kernel chain: the oracle's tables
    for _ in range(8):
        seen.append(counters(noisy))
        if seen[-1][0] == 0:
            break
    assert seen[-1][0] == 0 and seen[-1][1] > second[1], seen  # the launch has grown until it holds them
    assert all(a[0] >= b[0] for a, b in zip(seen, seen[1:])), seen
    full_parity(ctx, noisy, lat, lon, synth.times(2), [-2.0, 2.0])
    assert counters(smooth)[0] == 0


def not_simple_polygon_case(ctx):
    """A cutoff whose polygon crosses itself (a bow tie handed in as a closed contour) next to a simple one: the event
    table flags the first, the oracle's winding numbers agree, cells follow the non-zero winding rule."""
    from oracle import geom
    from oracle import synth
    lat, lon = synth.grid(1.0)
    W = lon.size + 120
    data = np.zeros((1, lat.size, lon.size), dtype=np.float32)

    def dense(ring):
        out = []
        for i, (ax, ay) in enumerate(ring):
            bx, by = ring[(i + 1) % len(ring)]
            g = int(np.gcd(abs(bx - ax), abs(by - ay)))
            out += [(ax + (bx - ax) // g * q, ay + (by - ay) // g * q) for q in range(g)]
        return np.asarray(out)

    xs = np.arange(W)
    rows = [np.c_[xs, np.full(W, 90) + (xs % 2)],
            dense([(100, 40), (112, 52), (112, 40), (100, 52)]),          # bow tie: the two diagonals cross at (106, 46)
            dense([(200, 40), (212, 40), (212, 52), (200, 52)])]          # square
    # the bow tie visits (106, 46) twice: a traced contour never repeats a point, a frame handed in may
    npts = np.array([len(r) for r in rows], dtype=np.int32)
    table = dict(step=np.zeros(3, np.int32), level_idx=np.zeros(3, np.int32), closed=np.array([0, 1, 1], np.int32), npts=npts,
                 offset=(np.cumsum(npts) - npts).astype(np.int64), x=np.concatenate([r[:, 0] for r in rows]).astype(np.int32),
                 y=np.concatenate([r[:, 1] for r in rows]).astype(np.int32))
    b = ctx.batch_from_contours(table, data, lat, lon, [2.0], 1.0, 1.0, want_cells=True)
    b.indices(streamers=False, overturnings=False)
    got = b.events(_hip.CUTOFFS)
    b.close()
    assert got["step"].size == 2
    assert [int(f) & _hip.EVENT_NOT_SIMPLE for f in got["flags"]] == [1, 0]
    assert geom.polygon_not_simple(rows[1], W, lat.size) and not geom.polygon_not_simple(rows[2], W, lat.size)
    for k, r in enumerate(rows[1:]):
        cx, cy = geom.polygon_cells(r, 0.5, W, lat.size)
        a, e = got["cell_offset"][k], got["cell_offset"][k + 1]
        assert np.array_equal(got["cell_x"][a:e], cx) and np.array_equal(got["cell_y"][a:e], cy)
    return got


def reference_rules_case(ctx, golden_dir):
    """tests/golden/reference_rules.npz -- outputs of the REFERENCE'S OWN statements (make_golden_reference.py) -- through the
    C ABI: the contour table of every fixture field against contour_index.py:105-155 run on scikit-image's output; the
    overturning boxes of every fixture contour (handed in as a frame, contour_index.py:197-249) against
    overturning_index.py:123-188; the contour selection and the cutoff filter against streamer_index.py:106-108 /
    cutoff_index.py:89-93.  Returns the numbers of contours, boxes and frames compared."""
    g = np.load(os.path.join(golden_dir, "reference_rules.npz"))
    unpack = lambda cat, off: [cat[off[i]:off[i + 1]] for i in range(len(off) - 1)]
    ncont = nbox = nframes = 0
    # ---- contours
    for name in g["contours__names"]:
        a = g["contours__%s__field" % name]
        level, nlon, dlon, padd = g["contours__%s__par" % name]
        nlon = int(nlon)
        lat = np.linspace(-90, 90, a.shape[0])
        lon = np.arange(nlon) * dlon - 180.0
        b = ctx.contours(a[None], lat, lon, [float(level)], dlon, lat[1] - lat[0], periodic_add=float(padd))
        got, st = b.contour_table(), int(b.status()[0])
        b.close()
        assert not st & ~(_hip.STATUS_LEVEL_HIT | 4), (name, st)          # (4: the field has NaN rows)
        want = unpack(g["contours__%s__pts" % name], g["contours__%s__off" % name])
        assert got["step"].size == len(want), (name, got["step"].size, len(want))
        for k, w in enumerate(want):
            o, n = int(got["offset"][k]), int(got["npts"][k])
            assert np.array_equal(np.c_[got["x"][o:o + n], got["y"][o:o + n]], w), (name, k)
            assert bool(got["closed"][k]) == bool(g["contours__%s__closed" % name][k]), (name, k)
            assert got["nux"][k] * dlon == g["contours__%s__exp_lon" % name][k], (name, k)
            assert np.round(got["sum_y"][k] / n, 2) == g["contours__%s__mean_lat" % name][k], (name, k)
        ncont += len(want)

    def frame(contours, closed):
        npts = np.array([len(c) for c in contours], dtype=np.int32)
        return dict(step=np.zeros(len(contours), np.int32), level_idx=np.zeros(len(contours), np.int32),
                    closed=np.asarray(closed, dtype=np.int32), npts=npts, offset=(np.cumsum(npts) - npts).astype(np.int64),
                    x=np.concatenate([c[:, 0] for c in contours]).astype(np.int32),
                    y=np.concatenate([c[:, 1] for c in contours]).astype(np.int32))
    # ---- overturnings
    from oracle import synth
    for k in range(int(g["overturn__count"])):
        key = "overturn__o%03d" % k
        xy = g[key + "__xy"]
        dlon, nlon, range_group, min_exp = g[key + "__par"]
        lat, lon = synth.grid(dlon)
        data = np.zeros((1, lat.size, lon.size), dtype=np.float32)
        b = ctx.batch_from_contours(frame([xy], [0]), data, lat, lon, [2.0], dlon, dlon, range_group=range_group, ot_min_exp=min_exp)
        b.indices(streamers=False, cutoffs=False)
        got = b.events(_hip.OVERTURNINGS)
        b.close()
        want = g[key + "__out"]                                            # min_lon, max_lon, min_lat, max_lat
        assert np.array_equal(got["box"][:, [0, 2, 1, 3]], want), (key, got["box"], want)
        nbox += len(want)
    # ---- selection (exp_lon == max over the frame) and cutoff filter.  Row i is a zigzag over nux[i] columns whose first
    # column holds three points: with min_exp = 0 every SELECTED contour carries exactly one overturning box, so the
    # overturning table lists the selection; the cutoff table lists the filter's rows.
    for k in range(int(g["select__count"])):
        key = "select__%02d" % k
        nux, closed = g[key + "__nux"], g[key + "__closed"]
        dlon, min_exp = g[key + "__par"]
        nlon = int(360 / dlon)
        lon = np.arange(nlon) * dlon
        lat = np.arange(-20, 21) * dlon
        rows = []
        for n in nux:
            xs = np.arange(1, int(n))
            rows.append(np.concatenate([np.array([[0, 10], [0, 11], [0, 12], [0, 13]][:3 if n > 1 else 4]), np.c_[xs, 12 + xs % 2]]).astype(np.int64))
        data = np.zeros((1, lat.size, lon.size), dtype=np.float32)
        b = ctx.batch_from_contours(frame(rows, closed), data, lat, lon, [2.0], dlon, dlon, ot_min_exp=0.0, co_min_exp=min_exp)
        assert b.max_nux() == int(nux.max()), key
        b.indices(streamers=False)
        ot, cut = b.events(_hip.OVERTURNINGS), b.events(_hip.CUTOFFS)
        b.close()
        assert ot["contour"].tolist() == g[key + "__selected"].tolist(), key
        assert cut["contour"].tolist() == g[key + "__cutoffs"].tolist(), key
        nframes += 1
    return ncont, nbox, nframes


def check_deferred_results(ctx, nt=6, dlon=1.0):
    """What wb_indices leaves unfinished on purpose since round 6: the geometry kernels are not waited for (context option
    geometry_async: wb_get_event_geometry waits) and the polygons in grid indices stay on the device (index_polygons = 0:
    wb_get_events packs them, wb_get_event_rows never does).  Every combination of the two options gives the same tables, rings
    and polygons; the rows-only table differs from the full one in the vertices alone; a batch freed with its geometry still in
    flight and a second wb_indices on a batch whose first geometry nobody asked for are safe."""
    lat, lon = synth.grid(dlon)
    pv = synth.pv_field(30, nt, lat, lon, eps=0.2)
    out = {}
    try:
        for ga, ip, dma in ((1, 0, 1), (0, 0, 1), (1, 1, 0), (0, 1, 0), (1, 0, 0)):
            ctx.set_option("geometry_async", ga)
            ctx.set_option("index_polygons", ip)
            ctx.set_option("geometry_dma", dma)         # (the kernels write device arrays, a DMA copy moves the rings / they store into pinned memory)
            b = ctx.contours(pv, lat, lon, [-2.0, 2.0], dlon, dlon, periodic_add=120.0, intensity=pv)
            b.indices()
            rows = [b.events(k, polygons=False) for k in (0, 1, 2)]                # first: nothing packed yet when ip == 0
            geo = [b.event_geometry(k) for k in (0, 1, 2)]
            full = [b.events(k) for k in (0, 1, 2)]
            out[(ga, ip, dma)] = (rows, geo, full)
            b.indices(cutoffs=False)                                               # again on the same batch: the first call's leftovers are settled
            again = [b.events(k) for k in (0, 1, 2)]
            assert again[2]["step"].size == 0 and np.array_equal(again[0]["poly_x"], full[0]["poly_x"])
            b.indices()                                                            # geometry in flight, nobody asks: close() waits
            b.close()
    finally:
        ctx.set_option("geometry_async", 1)
        ctx.set_option("index_polygons", 0)
        ctx.set_option("geometry_dma", 1)
    ref_rows, ref_geo, ref_full = out[(0, 1, 0)]
    assert sum(r["step"].size for r in ref_rows) > 10
    for key, (rows, geo, full) in out.items():
        for k in (0, 1, 2):
            for col in ref_full[k]:
                assert np.array_equal(full[k][col], ref_full[k][col]), (key, k, col)
                if col in ("poly_x", "poly_y"):
                    assert rows[k][col].size == 0
                else:
                    assert np.array_equal(rows[k][col], ref_full[k][col]), (key, k, col)
            for a, c in zip(geo[k], ref_geo[k]):
                assert np.array_equal(a, c), (key, k)
    return out
