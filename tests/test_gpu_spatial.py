"""MI355X parity of wb_smooth_field (fused and one-launch-per-pass kernels) and wb_momentum_flux, through the C ABI,
against the scipy / numpy golden vectors and the CPU oracle (bit-exact)."""
import os

import numpy as np
import pytest

from oracle import spatial, synth
from wavebreaking_amd import _hip

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "spatial.npz"))
FIVE = np.array([[0, 1, 0], [1, 2, 1], [0, 1, 0]])


@pytest.fixture(scope="module")
def ctx():
    c = _hip.Context(0)
    assert c.L.backend == "hip"
    yield c
    c.close()


@pytest.mark.parametrize("per_pass", [False, True])
@pytest.mark.parametrize("case", list(G["smooth_cases"]))
def test_smoothing_golden(ctx, case, per_pass):
    name, passes, mode = case.split("|")
    got = ctx.smooth_field(G[name + "_in"], int(passes), G[name + "_w"], mode, per_pass=per_pass)
    np.testing.assert_array_equal(got, G[name + "_out"])


@pytest.mark.parametrize("passes", [1, 2, 3, 4, 5, 6, 7, 8, 9, 17])
def test_fused_passes_1deg(ctx, passes):
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(0, 2, lat, lon, eps=0.3, dtype=np.float32)
    want = spatial.smooth_field(pv, passes, FIVE)
    np.testing.assert_array_equal(ctx.smooth_field(pv, passes, FIVE), want)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fused_quarter_degree(ctx, dtype):
    lat, lon = synth.grid(0.25)
    pv = synth.pv_field(3, 2, lat, lon, eps=0.3, dtype=dtype)
    pv[1, 300:303, 700:702] = np.nan
    want = spatial.smooth_field(pv, 5, FIVE)
    got = ctx.smooth_field(pv, 5, FIVE)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(ctx.smooth_field(pv, 5, FIVE, per_pass=True), want)
    assert np.isnan(got[:, [0, 1, -2, -1]]).all() and np.isfinite(got[0, 2:-2]).all()


def test_fused_odd_longitudes_and_weights(ctx):
    rng = np.random.default_rng(11)
    pv = rng.standard_normal((2, 37, 251)).astype(np.float32)
    w = np.array([[0, 0.5, 0], [1.25, -3, 2], [0, 1e-17, 0]])          # plus-shaped, one tap below DBL_EPSILON
    np.testing.assert_array_equal(ctx.smooth_field(pv, 3, w), spatial.smooth_field(pv, 3, w))
    pv64 = rng.standard_normal((1, 9, 130))
    np.testing.assert_array_equal(ctx.smooth_field(pv64, 8, FIVE), spatial.smooth_field(pv64, 8, FIVE))


@pytest.mark.parametrize("name", list(G["mflux_cases"]))
def test_momentum_flux_golden(ctx, name):
    got = ctx.momentum_flux(G[name + "_u"], G[name + "_v"])
    assert got.dtype == G[name + "_mflux"].dtype
    np.testing.assert_array_equal(got, G[name + "_mflux"])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_momentum_flux_quarter_degree(ctx, dtype):
    rng = np.random.default_rng(12)
    u = (rng.standard_normal((2, 721, 1440)) * 15 + 8).astype(dtype)
    v = (rng.standard_normal((2, 721, 1440)) * 9).astype(dtype)
    u[0, 10, 5] = np.nan
    with np.errstate(all="ignore"):
        want = (u - np.nanmean(u, axis=2)[..., None]) * (v - np.nanmean(v, axis=2)[..., None])   # what xarray evaluates
    np.testing.assert_array_equal(ctx.momentum_flux(u, v), want)
    rows = [(0, 10), (1, 360)]
    for t, j in rows:                                                    # and the oracle on a few rows
        np.testing.assert_array_equal(spatial.momentum_flux(u[t:t + 1, j:j + 1], v[t:t + 1, j:j + 1])[0, 0], want[t, j])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fused_inputs_outside_the_short_division_range(ctx, dtype):
    """Infinities and magnitudes outside 2^-100 .. 2^100 make a strip fall back to the IEEE division; results stay exact."""
    rng = np.random.default_rng(13)
    pv = rng.standard_normal((3, 40, 300)).astype(dtype)
    pv[0, 7, 11] = np.inf
    pv[0, 30, 200] = -np.inf
    if dtype == np.float64:
        pv[1, 5, 5] = 3e-200
        pv[1, 20, 150:160] *= 1e-310          # subnormal
        pv[2, 9, 250] = 1e250
    else:
        pv[1, 5, 5] = 1e-44                    # float32 subnormal: inside the double range, stays on the short path
        pv[2, 9, 250] = 3e38
    with np.errstate(all="ignore"):
        want = spatial.smooth_field(pv, 6, FIVE)
    got = ctx.smooth_field(pv, 6, FIVE)
    np.testing.assert_array_equal(got, want)
    ok = ~np.isnan(want)
    assert np.array_equal(got[ok].view(np.uint64), want[ok].view(np.uint64))


def test_momentum_flux_row_lengths(ctx):
    rng = np.random.default_rng(14)
    for nlon in (1, 2, 8, 1536, 1537, 3600, 4096):
        u = rng.standard_normal((1, 3, nlon)).astype(np.float32)
        v = rng.standard_normal((1, 3, nlon))
        with np.errstate(all="ignore"):
            for a, b in ((u, v.astype(np.float32)), (u.astype(np.float64), v)):
                want = (a - np.nanmean(a, axis=2)[..., None]) * (b - np.nanmean(b, axis=2)[..., None])
                np.testing.assert_array_equal(ctx.momentum_flux(a, b), want)
    with pytest.raises(RuntimeError):
        ctx.momentum_flux(np.zeros((1, 1, 5000), np.float32), np.zeros((1, 1, 5000), np.float32))
