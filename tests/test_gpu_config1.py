"""BASELINE.json configs[1] at FULL size on the GPU: 1.0 degree global PV (360 x 181), one year of 6-hourly steps (1 460),
contour level [2], the streamer index, ONE batch on one MI355X -- status words, guard-band counters, batch invariance
(sampled steps run alone are bit-equal to their rows inside the year) and the oracle on sampled steps."""
import numpy as np
import pytest

from oracle import api as oapi
from oracle import synth
from tests import record_checks as rc
from wavebreaking_amd import _hip

pytestmark = pytest.mark.gpu


def test_one_degree_year_of_streamers_in_one_batch():
    nt, dlon, levels = 1460, 1.0, [2.0]
    lat, lon = synth.grid(dlon)
    nlat, nlon = lat.size, lon.size
    ctx = _hip.Context(0)
    bench = rc.load_bench()
    try:
        step_bytes = nlat * nlon * 4
        d_pv = ctx.alloc(nt * step_bytes)
        ctx.synth_fill(d_pv, np.float32, 0, nt, lat, lon, synth.wave_params(), 0.1, synth.SEED)     # (a slightly rough field: more small contours)
        b = ctx.contours(d_pv, lat, lon, levels, dlon, dlon, device_shape=(nt, nlat, nlon), device_dtype=np.float32)
        g = b.max_nux()
        assert g == nlon + 120                                   # a contour spans the whole extended grid
        b.indices(g, overturnings=False, cutoffs=False)
        assert not np.asarray(b.status()).any() and b.band_hits() == (0, 0)
        ev = b.events(_hip.STREAMERS)
        nev = int(ev["step"].size)
        assert nev > 2 * nt and np.all(np.diff(ev["step"]) >= 0) and np.unique(ev["step"]).size > 0.8 * nt
        assert ev["poly_offset"][-1] == ev["poly_x"].size and not (ev["flags"] & _hip.EVENT_NOT_SIMPLE).any()
        # batch invariance: a step alone (selected against the same maximum) gives exactly its rows of the year
        for t in (0, 777, nt - 1):
            one = ctx.contours(d_pv + t * step_bytes, lat, lon, levels, dlon, dlon, device_shape=(1, nlat, nlon), device_dtype=np.float32)
            one.indices(g, overturnings=False, cutoffs=False)
            e1 = one.events(_hip.STREAMERS)
            one.close()
            m = np.nonzero(ev["step"] == t)[0]
            assert e1["step"].size == m.size > 0
            for key in ("level_idx", "ind1", "ind2", "ncells", "flags"):
                assert np.array_equal(e1[key], ev[key][m]), (t, key)
            assert np.array_equal(e1["sums"], ev["sums"][m]), t
            a, z = ev["poly_offset"][m[0]], ev["poly_offset"][m[-1] + 1]
            assert np.array_equal(e1["poly_x"], ev["poly_x"][a:z]) and np.array_equal(e1["poly_y"], ev["poly_y"][a:z]), t
        # the oracle on sampled steps of the same device-generated fields: polygons, cell counts, truncated centre of mass bit
        # for bit, the data sums to 1e-6 relative (bench._compare_step)
        cmp = dict(events=0, mismatches=0, max_rel_err=0.0)
        sample = sorted(set(int(round(v)) for v in np.linspace(0, nt - 1, 10)))
        for t in sample:
            pv = ctx.to_host(d_pv, (1, nlat, nlon), np.float32, offset_bytes=t * step_bytes)
            ref = oapi.calculate_streamers(pv, levels, lat, lon, synth.times(1))
            bench._compare_step(ev, ref, t, levels, cmp)
        print("[config1] %d streamers in %d steps; oracle on %d steps: %s" % (nev, nt, len(sample), cmp))
        assert cmp["mismatches"] == 0 and cmp["events"] >= 15 and cmp["max_rel_err"] <= 1e-6
        b.close()
        ctx.free(d_pv)
    finally:
        ctx.close()
