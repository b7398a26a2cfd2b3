"""Shared parity helpers: run the C ABI (HIP build on the GPU box, host-simulation build on CPU)
and the oracle on the same input and compare every table."""
import os
import subprocess

import numpy as np

from oracle import api as oapi
from oracle import contours as ocont
from wavebreaking_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTSIM_DIR = os.path.join(ROOT, "tests", "_hostsim")
HOSTSIM_LIB = os.path.join(HOSTSIM_DIR, "libwb_hostsim.so")
CSRC = os.path.join(ROOT, "wavebreaking_amd", "csrc")


def build_hostsim(force=False, variant="", defines=()):
    """g++ build of the kernel sources with -DWB_HOSTSIM (test infrastructure only).  ``variant`` / ``defines``
    build a second copy with other compile-time limits (e.g. a tiny WB_TRACE_LDS_N to force the big-job paths)."""
    lib = HOSTSIM_LIB if not variant else HOSTSIM_LIB.replace(".so", "_%s.so" % variant)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    srcs.append(os.path.join(ROOT, "include", "wavebreaking_hip.h"))
    if (not force and os.path.exists(lib)
            and os.path.getmtime(lib) >= max(os.path.getmtime(s) for s in srcs)):
        return lib
    os.makedirs(HOSTSIM_DIR, exist_ok=True)
    cmd = ["g++", "-O2", "-g", "-DWB_HOSTSIM", *["-D" + d for d in defines], "-x", "c++", "-std=c++17",
           "-ffp-contract=off", "-shared", "-fPIC", os.path.join(CSRC, "wb_api.hip"), "-o", lib]
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
               timings=b.timings())
    b.close()
    return out


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
        ref = np.asarray(r["_sums"])
        s = got["sums"][k]
        # areas / x_com / y_com share the summation order with the oracle: exact; data sums: rtol
        assert s[0] == ref[0] and s[3] == ref[3] and s[4] == ref[4], (kind, k, s, ref)
        assert np.allclose(s[1:3], ref[1:3], rtol=rtol, atol=1e-9 * abs(ref[0])), (kind, k, s, ref)


def full_parity(ctx, data, lat, lon, times, levels, intensity=None, periodic_add=120.0, **kw):
    """Everything the path produces, ABI vs oracle.  Returns (abi tables, oracle tables)."""
    got = run_abi(ctx, data, lat, lon, levels, intensity=intensity, periodic_add=periodic_add, **kw)
    ref = oapi.run_all(data, levels, lat, lon, times, intensity=intensity, periodic_add=periodic_add)
    dlon = ocont.spatial_resolution(np.asarray(lon, dtype=np.float64), "lon")
    compare_contours(got["contours"], ref["contours"], levels, dlon)
    for kind in ("streamers", "overturnings", "cutoffs"):
        compare_events(got[kind], got["contours"], ref[kind], kind, levels)
    return got, ref
