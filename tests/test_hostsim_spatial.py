"""Host logic of wb_smooth_field / wb_momentum_flux (tap order, boundary modes, pass chaining, the pairwise-sum plan)
through the C ABI of the host-simulation build, against the scipy / numpy golden vectors and the oracle."""
import os

import numpy as np
import pytest

from tests import helpers
from oracle import spatial
from wavebreaking_amd import _hip

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "spatial.npz"))


@pytest.fixture(scope="module")
def ctx():
    lib = helpers.hostsim_library()
    c = _hip.Context(0, lib)
    yield c
    c.close()


@pytest.mark.parametrize("case", list(G["smooth_cases"]))
def test_smoothing_golden(ctx, case):
    name, passes, mode = case.split("|")
    got = ctx.smooth_field(G[name + "_in"], int(passes), G[name + "_w"], mode)
    np.testing.assert_array_equal(got, G[name + "_out"])


@pytest.mark.parametrize("name", list(G["mflux_cases"]))
def test_momentum_flux_golden(ctx, name):
    got = ctx.momentum_flux(G[name + "_u"], G[name + "_v"])
    assert got.dtype == G[name + "_mflux"].dtype
    np.testing.assert_array_equal(got, G[name + "_mflux"])


def test_momentum_flux_race_free(ctx):
    rng = np.random.default_rng(5)
    u = rng.standard_normal((2, 3, 777)).astype(np.float32)
    v = rng.standard_normal((2, 3, 777)).astype(np.float32)
    want = spatial.momentum_flux(u, v)
    for rev in (0, 1):
        ctx.L.lib.wb_hostsim_set_reverse(rev)
        try:
            np.testing.assert_array_equal(ctx.momentum_flux(u, v), want)
        finally:
            ctx.L.lib.wb_hostsim_set_reverse(0)


def test_bad_arguments(ctx):
    a = np.zeros((1, 4, 4), np.float32)
    with pytest.raises(RuntimeError):
        ctx.smooth_field(a, 0, np.ones((3, 3)))
    with pytest.raises(RuntimeError):
        ctx.smooth_field(a, 1, np.ones((9, 3)))
    with pytest.raises(RuntimeError):
        ctx.smooth_field(a, 1, np.ones((3, 3)), mode="bogus")
    with pytest.raises(RuntimeError):
        ctx.smooth_field(np.zeros((1, 1, 4), np.float32), 1, np.ones((3, 3)))      # border rows exceed the axis


@pytest.mark.parametrize("S", [6.0, -6.0, 3.0, 16.0, 10.0, 0.875, 5.0 * 2.0 ** -20, float(2 ** 26 - 1), 12345678.0 * 2.0 ** 30])
def test_short_division_is_correctly_rounded(ctx, S):
    """wb_sm_div (q = a*y, r = fma(-S,q,a), q' = fma(r,y,q)) against the IEEE division for divisors of the admitted
    form k * 2^j: random magnitudes, values next to rounding boundaries of the quotient, and the special cases."""
    import ctypes as C
    rng = np.random.default_rng(int(abs(S) * 64) % 1000)
    n = 400000
    rnd = rng.standard_normal(n) * np.exp(rng.uniform(-40, 40, n))
    x = rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-30, 30, n)          # quotients; a = S * (x + half ulp) +- few ulps
    mid = x + np.spacing(x) / 2
    near = np.nextafter(mid * S, np.where(rng.integers(0, 2, n) == 1, np.inf, -np.inf))
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 1e-310, 2.0 ** -901, 2.0 ** -899, 2.0 ** 899,
                        2.0 ** 901, 1.7e308, -1.7e308, 6.0, 1.0, -1.0])
    a = np.concatenate([rnd, mid * S, near, np.round(rnd * 64) * S, special])
    fast, plain = np.empty_like(a), np.empty_like(a)
    dp = C.POINTER(C.c_double)
    ctx.L.lib.wb_hostsim_div(a.ctypes.data_as(dp), C.c_int64(a.size), C.c_double(S), fast.ctypes.data_as(dp),
                             plain.ctypes.data_as(dp))
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(plain, a / S)
    assert np.array_equal(fast.view(np.uint64)[~np.isnan(plain)], plain.view(np.uint64)[~np.isnan(plain)])   # bit patterns
    assert np.isnan(fast[np.isnan(plain)]).all()
