"""to_xarray's flag cube: oracle against hand-derived GEOS truth, the C ABI (host simulation; MI355X with -m gpu) against
the oracle on lattice polygons, MultiPolygons, overlapping events with values, and on the events of the pipeline."""
import numpy as np
import pytest

from oracle import flags as oflags
from tests import helpers
from tests.golden import geos_truth
from wavebreaking_amd import _hip


def _truth_cube(points, lat, lon):
    cube = np.zeros((1, lat.size, lon.size), dtype=np.int8)
    for x, y in points:
        cube[0, int(np.nonzero(lat == y)[0][0]), int(np.nonzero(lon == x)[0][0])] = 1
    return cube


@pytest.mark.parametrize("case", range(len(geos_truth.FLAGS)))
def test_oracle_against_hand_derived_truth(case):
    rings, points, why = geos_truth.FLAGS[case]
    lat, lon = np.arange(6.0), np.arange(8.0)
    got = oflags.flag_events(1, lat, lon, 0.5, [0], [[np.asarray(r, float) for r in rings]])
    np.testing.assert_array_equal(got, _truth_cube(points, lat, lon), err_msg=why)


def _random_events(rng, nt, lat, lon, nev):
    steps, rings, vals = [], [], []
    for _ in range(nev):
        steps.append(int(rng.integers(0, nt)))
        parts = []
        for _ in range(int(rng.integers(1, 3))):
            n = int(rng.integers(3, 9))
            cx, cy = rng.integers(2, lon.size - 2), rng.integers(2, lat.size - 2)
            ang = np.sort(rng.uniform(0, 2 * np.pi, n))
            rad = rng.uniform(1.0, 9.0, n)
            ii = np.clip(np.round(cx + rad * np.cos(ang)), 0, lon.size - 1).astype(int)
            jj = np.clip(np.round(cy + rad * np.sin(ang)), 0, lat.size - 1).astype(int)
            parts.append(np.c_[lon[ii], lat[jj]])
        rings.append(parts)
        vals.append(float(rng.integers(1, 100)))
    return steps, rings, vals


def _abi_checks(ctx):
    lat, lon = np.arange(6.0), np.arange(8.0)
    for rings, points, why in geos_truth.FLAGS:
        got = ctx.flag_events(1, lat, lon, 0.5, [0], [[np.asarray(r, float) for r in rings]])
        np.testing.assert_array_equal(got, _truth_cube(points, lat, lon), err_msg=why)
    rng = np.random.default_rng(31)
    for dlon, dlat in ((1.0, 1.0), (0.25, 0.25), (0.5, 0.25)):
        lat = np.arange(-10, 10 + dlat / 2, dlat)
        lon = np.arange(-20, 20, dlon)
        r = (dlon + dlat) / 4
        steps, rings, vals = _random_events(rng, 3, lat, lon, 25)
        want = oflags.flag_events(3, lat, lon, r, steps, rings)
        np.testing.assert_array_equal(ctx.flag_events(3, lat, lon, r, steps, rings), want)
        assert want.sum() > 0
        for dt in (np.float32, np.float64):
            wantv = oflags.flag_events(3, lat, lon, r, steps, rings, vals, dtype=dt)
            np.testing.assert_array_equal(ctx.flag_events(3, lat, lon, r, steps, rings, vals, dtype=dt), wantv)
    # no events, events with empty geometry
    np.testing.assert_array_equal(ctx.flag_events(2, lat, lon, r, [], []), np.zeros((2, lat.size, lon.size), np.int8))
    np.testing.assert_array_equal(ctx.flag_events(2, lat, lon, r, [1], [[]]), np.zeros((2, lat.size, lon.size), np.int8))
    with pytest.raises(RuntimeError):
        ctx.flag_events(2, lat, lon, r, [2], [[np.zeros((3, 2))]])           # time index out of range


def test_abi_hostsim():
    lib = helpers.hostsim_library()
    ctx = _hip.Context(0, lib)
    try:
        for rev in (0, 1):
            lib.lib.wb_hostsim_set_reverse(rev)
            _abi_checks(ctx)
    finally:
        lib.lib.wb_hostsim_set_reverse(0)
        ctx.close()


@pytest.mark.gpu
def test_abi_gpu():
    ctx = _hip.Context(0)
    try:
        _abi_checks(ctx)
    finally:
        ctx.close()
