"""oracle/spatial.py against the golden vectors of scipy.ndimage.convolve / numpy nanmean (tests/golden/spatial.npz)
and, when scipy is importable, against the live libraries on fresh random inputs."""
import os

import numpy as np
import pytest

from oracle import spatial

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "spatial.npz"))


@pytest.mark.parametrize("case", list(G["smooth_cases"]))
def test_smoothing_matches_scipy_golden(case):
    name, passes, mode = case.split("|")
    got = spatial.smooth_field(G[name + "_in"], int(passes), G[name + "_w"], mode)
    want = G[name + "_out"]
    assert got.dtype == want.dtype
    np.testing.assert_array_equal(got, want)            # bit-exact, NaN == NaN


@pytest.mark.parametrize("name", list(G["mflux_cases"]))
def test_momentum_flux_matches_numpy_golden(name):
    u, v = G[name + "_u"], G[name + "_v"]
    np.testing.assert_array_equal(spatial.zonal_mean(u), G[name + "_um"])
    np.testing.assert_array_equal(spatial.zonal_mean(v), G[name + "_vm"])
    got = spatial.momentum_flux(u, v)
    assert got.dtype == u.dtype
    np.testing.assert_array_equal(got, G[name + "_mflux"])


def test_pairwise_sum_live():
    rng = np.random.default_rng(3)
    for n in (1, 7, 8, 9, 127, 128, 129, 255, 256, 257, 719, 1440, 2881):
        for dt in (np.float32, np.float64):
            a = (rng.standard_normal(n) * 100).astype(dt)
            assert spatial.pairwise_sum(a) == np.sum(a)


def test_convolve_live_scipy():
    ndimage = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(4)
    for mode in ("wrap", "reflect", "mirror", "nearest", "constant"):
        for shape in ((3, 3), (5, 3), (3, 5), (1, 3), (4, 3)):
            a = rng.standard_normal((7, 6)).astype(np.float32)
            w = rng.standard_normal(shape)
            np.testing.assert_array_equal(spatial.convolve2d(a, w, mode), ndimage.convolve(a, weights=w, mode=mode))
