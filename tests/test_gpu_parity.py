"""GPU: the HIP path through the C ABI against the oracle (bit-exact contours, event end points, cells;
1e-6 relative on the data sums) + size-independent properties at BASELINE.json's full grid size."""
import numpy as np
import pytest

from oracle import synth
from tests import helpers
from wavebreaking_amd import _hip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(request):
    lib = _hip.Library()
    assert lib.backend == "hip"
    c = _hip.Context(0, lib)
    c.set_option("scan_split", getattr(request, "param", 1))     # 1 (the library default): streaming scan + build; 0: the single kernel
    yield c
    c.close()


two_scans = pytest.mark.parametrize("ctx", [0, 1], indirect=True, ids=["scan", "scan_split"])


@two_scans
def test_full_parity_1deg(ctx):
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(10, 6, lat, lon, eps=0.3)
    inten = synth.pv_field(10, 6, lat, lon, eps=0.0, seed=20241)
    got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(6), [-2, 2], intensity=inten)
    assert len(ref["streamers"]) > 10 and got["band"] == (0, 0) and not got["status"].any()


def test_demo_shaped_config0(ctx):
    lat, lon = synth.grid(1.0, lat_min=-89.0, lat_max=89.0)
    pv = synth.pv_field(0, 1, lat, lon, eps=0.1)
    helpers.full_parity(ctx, pv, lat, lon, synth.times(1), [2], intensity=pv)


@two_scans
def test_nan_rows_f64_three_levels(ctx):
    lat, lon = synth.grid(2.5)
    pv = synth.pv_field(3, 3, lat, lon, eps=0.2).astype(np.float64)
    pv[:, [0, 1, -2, -1], :] = np.nan
    pv[1, 30:33, 20:26] = np.nan
    got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(3), [2, -2, 1.5])
    assert (got["status"] & _hip.STATUS_HAS_NAN).all()


@two_scans
@pytest.mark.parametrize("padd", [0, 57, 360])
def test_periodic_add_variants(ctx, padd):
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(5, 2, lat, lon, eps=0.3)
    helpers.full_parity(ctx, pv, lat, lon, synth.times(2), [2], periodic_add=padd)


def test_streaming_scan_float64_and_group_maximum_of_shifted_chords(ctx):
    # (1) the streaming scan's float64 instance on the 0.25 degree grid (six strips, scalar halo loads);
    # (2) this field holds a streamer whose group (check_groups, streamer_index.py:224-251) has two chords shifted by one point
    #     along the contour: their lengths on[ind1 : ind2 + 1].sum() differ in the last bits only and the reference's argmax
    #     turns on numpy's pairwise summation of the slices -- a prefix difference picked the other chord
    lat, lon = synth.grid(0.25)
    pv = synth.pv_field(40, 1, lat, lon, eps=0.3)
    for field in (pv, pv.astype(np.float64)):
        for eager in (0, 1):
            ctx.set_option("eager_f2", eager)
            try:
                helpers.full_parity(ctx, field, lat, lon, synth.times(1), [-2, 2, 1.5])
            finally:
                ctx.set_option("eager_f2", 0)


def test_group_maximum_shared_by_two_members(ctx):
    # check_groups with an exact tie of the largest length: the survivor follows the reference's CPython set order
    # (wb_k_streamer_ties); cases found by tools/parity_hunt.py
    lat, lon = synth.grid(1.0)
    ties = 0
    for seed, eps, t0, levels in ((41289, 0.0, 3601, [-2, 2]), (42352, 0.1, 2195, [1.0, 3.0])):
        pv = synth.pv_field(t0, 6, lat, lon, eps=eps, seed=1000 + seed)
        for eager in (0, 1):
            ctx.set_option("eager_f2", eager)
            try:
                helpers.full_parity(ctx, pv, lat, lon, synth.times(6), levels, intensity=pv)
            finally:
                ctx.set_option("eager_f2", 0)
        b = ctx.contours(pv, lat, lon, levels, 1.0, 1.0, intensity=pv)
        b.indices()
        ties += b.group_ties()
        b.close()
    assert ties >= 2


def test_eager_and_lazy_routine_2_agree(ctx):
    # the exhaustive formulation (chord test for every candidate pair, then routines 3 + 4) against the oracle as well
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(20, 4, lat, lon, eps=0.3)
    ctx.set_option("eager_f2", 1)
    try:
        helpers.full_parity(ctx, pv, lat, lon, synth.times(4), [-2, 2], intensity=pv)
    finally:
        ctx.set_option("eager_f2", 0)


@two_scans
def test_quarter_degree_parity(ctx):
    # BASELINE.json configs[2] grid (1440 x 721), both levels, all three indices; 2 steps keep the oracle in seconds
    lat, lon = synth.grid(0.25)
    pv = synth.pv_field(7, 2, lat, lon, eps=0.3)
    got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(2), [-2, 2], intensity=pv)
    assert max(len(c["geometry"]) for c in ref["contours"]) > 2000


def test_quarter_degree_streamer_filters_by_other_routes(ctx):
    # the same 0.25 degree contours (thousands of candidate pairs each) with routine 1 inside wb_k_streamer_f234 (its table over the
    # row buckets' LDS, in parts where the western pairs do not fit) and with a kernel LDS that holds 1024 points only (longer
    # contours: the row-after-row sweep in global memory)
    lat, lon = synth.grid(0.25)
    pv = synth.pv_field(7, 1, lat, lon, eps=0.3)
    for opts in ({"f1_fused": 1}, {"f234_lpts": 1024}, {"f1_fused": 1, "f234_lpts": 1024}):
        for k, v in opts.items():
            ctx.set_option(k, v)
        try:
            got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(1), [-2, 2])
        finally:
            ctx.set_option("f1_fused", 0)
            ctx.set_option("f234_lpts", 0)
        assert len(ref["streamers"]) > 3


def test_consecutive_launches_on_different_fields(ctx):
    # the streaming scan's workgroups hand each other halo columns through LDS, which keeps its contents from one launch
    # to the next: two rough fields of the same shape one after the other (a stale halo of the first gave the second 6-8
    # extra contours before the tags were cleared at kernel start), and the first again
    lat, lon = synth.grid(0.25)
    tm = synth.times(2)
    a = synth.pv_field(500, 2, lat, lon, eps=0.3, seed=7)
    b = synth.pv_field(900, 2, lat, lon, eps=0.3, seed=11)
    for pv in (a, b, a):
        helpers.full_parity(ctx, pv, lat, lon, tm, [-2, 2])


@two_scans
def test_capacity_retry_paths(ctx):
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(2, 2, lat, lon, eps=0.3)
    helpers.full_parity(ctx, pv, lat, lon, synth.times(2), [-2, 2], cap_seg=1024, cap_pairs=16)


def test_device_generator_and_batch_invariance(ctx):
    """Full-size property: a time step's events do not depend on what else is in the batch, and the
    device generator's field gives the same catalogue as the oracle run on the copied-back field."""
    lat, lon = synth.grid(0.25)
    nt = 24
    nbytes = nt * lat.size * lon.size * 4
    d = ctx.alloc(nbytes)
    ctx.synth_fill(d, np.float32, 100, nt, lat, lon, synth.wave_params(), 0.3, synth.SEED)
    b = ctx.contours(d, lat, lon, [-2, 2], 0.25, 0.25, device_shape=(nt, lat.size, lon.size), device_dtype=np.float32,
                     want_cells=False)
    b.indices()
    big = {k: b.events(i) for k, i in (("s", 0), ("o", 1), ("c", 2))}
    bigc = b.contour_table()
    assert not (b.status() & ~np.uint32(_hip.STATUS_HAS_NAN)).any()
    b.close()
    # steps 5 and 17 alone, from the copied-back field, through the ABI and through the oracle
    for t in (5, 17):
        host = ctx.to_host(d, (1, lat.size, lon.size), np.float32, offset_bytes=t * lat.size * lon.size * 4)
        synth.assert_no_level_hits(host, [-2, 2])
        got, ref = helpers.full_parity(ctx, host, lat, lon, synth.times(1), [-2, 2], want_cells=False)
        for key, kind in (("s", "streamers"), ("o", "overturnings"), ("c", "cutoffs")):
            m = big[key]["step"] == t
            assert m.sum() == got[kind]["step"].size
            assert np.array_equal(big[key]["sums"][m], got[kind]["sums"])
            assert np.array_equal(big[key]["ind1"][m], got[kind]["ind1"])
            assert np.array_equal(big[key]["box"][m], got[kind]["box"])
        m = bigc["step"] == t
        assert np.array_equal(bigc["npts"][m], got["contours"]["npts"])
    ctx.free(d)


def test_very_rough_field_takes_the_big_job_paths(ctx):
    # > 8192 segments per (step, level): 32-bit list ranking in global memory, hundreds of small contours
    lat, lon = synth.grid(0.25)
    pv = synth.pv_field(3, 1, lat, lon, eps=1.2)
    got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(1), [2.0], intensity=pv)
    assert len(ref["contours"]) > 100


def test_python_surface_with_time_chunking(ctx, monkeypatch):
    import wavebreaking_amd as wb
    from oracle import api as oapi
    from wavebreaking_amd import engine
    engine._contexts.clear()
    monkeypatch.setenv("WB_MAX_STEPS_PER_BATCH", "2")
    lat, lon = synth.grid(1.0)
    tm = synth.times(5, start="1959-06-01T00")
    pv = synth.pv_field(0, 5, lat, lon, eps=0.2)
    data = wb.GridField(pv, ("time", "lat", "lon"), {"time": tm, "lat": lat, "lon": lon})
    ref = oapi.run_all(pv, [-2, 2], lat, lon, tm)
    s = wb.calculate_streamers(data, [-2, 2])
    assert len(s) == len(ref["streamers"])
    for (_, row), r in zip(s.iterrows(), ref["streamers"]):
        assert np.datetime64(row["date"]) == r["date"] and row["level"] == r["level"] and row["event_area"] == r["event_area"]


def test_com_follows_the_reference_arithmetic(ctx):
    """>= 20 events whose x_com / areas is an integer mathematically (boxes with min_lon + max_lon even): the truncated
    centre-of-mass index and the three sums equal the reference's per-cell, pandas-summed values bit for bit."""
    helpers.NEAR_INTEGER_EVENTS.clear()
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(0, 12, lat, lon, eps=0.3)
    got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(12), [-2, 2], intensity=pv)
    assert helpers.NEAR_INTEGER_EVENTS.get("overturnings", 0) >= 20
    assert got["com_rechecks"][1] >= 20


@pytest.mark.parametrize("dlon,nsteps", [(1.0, 2), (0.25, 1)])
def test_com_of_symmetric_rasterised_events(ctx, dlon, nsteps):
    got, ref, rechecks, near = helpers.symmetric_cutoff_case(ctx, dlon, nsteps)
    assert near >= 5 * nsteps and rechecks[0] >= near and rechecks[1] == 0


def test_cells_equal_to_the_level_golden_fields(ctx, golden_dir):
    nequal, nflag = helpers.level_hit_goldens_case(ctx, golden_dir)
    assert nequal >= 35 and nflag >= 4


def test_order_dependent_joins_golden_fields(ctx, golden_dir):
    """two contour ends on one grid vertex: the flagged job is joined by the sequential walk (wb_k_trace_walk) and equals
    scikit-image on the 48 fields it returns contours on; STATUS_LEVEL_KEYERROR on the 6 it raises KeyError on"""
    nequal, nflag = helpers.level_hit_goldens_case(ctx, golden_dir, "contours_levelwalk.npz")
    assert nequal == 48 and nflag == 54


def test_link_merge_by_searches_equals_merge_by_row_tables(ctx):
    """option link_rows = 0 (the binary-search merge of wb_k_trace_link: grids whose row tables do not fit its LDS) gives the
    contour table of the row-table merge, at 0.25 degree and with a 360 degree periodic pad at 1 degree"""
    for dlon, padd, nt in ((0.25, 120.0, 2), (1.0, 360.0, 3)):
        lat, lon = synth.grid(dlon)
        pv = synth.pv_field(7, nt, lat, lon, eps=0.3)
        tabs = []
        for rows in (1, 0):
            ctx.set_option("link_rows", rows)
            try:
                b = ctx.contours(pv, lat, lon, [-2.0, 2.0], dlon, dlon, periodic_add=padd)
                tabs.append(b.contour_table())
                b.close()
            finally:
                ctx.set_option("link_rows", 1)
        assert tabs[0]["step"].size > 10
        for key in tabs[0]:
            assert np.array_equal(tabs[0][key], tabs[1][key]), (dlon, key)


def test_launch_options_change_no_result(ctx):
    """scan_priority (the field scan on a stream of the highest priority), persist_slack (places the persistent launches leave
    free), boxcom_per_cu / boxcom_after (when the boxes' sums run), CU-masked streams and the patch format of the scan (keys
    only, values from the field / key + ten values), routine 1 of the streamer index inside wb_k_streamer_f234 or as its own
    launch and the contour length that kernel's LDS holds (longer contours: the row-after-row sweep in global memory) only move
    work in time or place: same events, same sums"""
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(3, 4, lat, lon, eps=0.2)
    base = None
    for opts in ({}, {"scan_priority": 1}, {"persist_slack": 64, "boxcom_per_cu": 3}, {"boxcom_after": 0}, {"boxcom_after": 2},
                 {"scan_keys_only": 0}, {"cu_mask": 64}, {"cu_mask": 64128, "scan_keys_only": 0},
                 {"f1_fused": 1}, {"f234_lpts": 128}, {"f234_lpts": 64, "f1_fused": 1}):
        for k, v in opts.items():
            ctx.set_option(k, v)
        try:
            got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(4), [-2.0, 2.0], intensity=pv)
        finally:
            ctx.set_option("scan_priority", 0)
            ctx.set_option("persist_slack", 0)
            ctx.set_option("boxcom_per_cu", 12)
            ctx.set_option("boxcom_after", 1)
            ctx.set_option("scan_keys_only", 1)
            ctx.set_option("f1_fused", 0)
            ctx.set_option("f234_lpts", 0)
            if "cu_mask" in opts:
                ctx.set_option("cu_mask", 0)
        counts = tuple(int(got[kind]["step"].size) for kind in ("streamers", "overturnings", "cutoffs"))   # (full_parity has compared every column with the oracle)
        if base is None:
            base = counts
        assert counts == base and sum(counts) > 10


def test_reference_own_code_fixture_through_the_abi(ctx, golden_dir):
    """the HIP path against outputs of the REFERENCE'S OWN statements (tests/golden/reference_rules.npz, made in the build
    container by tests/golden/make_golden_reference.py): contour tables (contour_index.py:105-155 on scikit-image's output),
    overturning boxes (overturning_index.py:123-188), contour selection and cutoff filter (streamer_index.py:106-108,
    cutoff_index.py:89-93)"""
    ncont, nbox, nframes = helpers.reference_rules_case(ctx, golden_dir)
    assert ncont > 150 and nbox > 300 and nframes == 40


@pytest.mark.parametrize("seed,dlon", [(0, 1.0), (1, 1.0), (2, 1.0), (3, 1.0), (4, 0.25)])
def test_cells_equal_to_the_level_full_path(ctx, seed, dlon):
    pv, lat, lon = helpers.level_hit_field(seed, dlon=dlon, nhits=4 if dlon == 1.0 else 12)
    got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(1), [2.0], intensity=pv)
    assert got["status"][0] & _hip.STATUS_LEVEL_HIT and not (got["status"][0] & _hip.STATUS_LEVEL_ORDER)


def test_polygon_that_crosses_itself_is_flagged(ctx):
    helpers.not_simple_polygon_case(ctx)


def _contour_tables_gpu(ctx, pv, lat, lon, levels, d, padd, **opts):
    for k, v in opts.items():
        ctx.set_option(k, v)
    try:
        b = ctx.contours(pv, lat, lon, levels, d, d, periodic_add=padd)
        tab = b.contour_table()
        st = np.array(b.status())
        b.close()
    finally:
        for k, v in dict(trace_fused=1, trace_nmax=0, trace_cmax=0).items():
            ctx.set_option(k, v)
    return tab, st


def test_fused_trace_kernel_equals_the_three_kernel_chain(ctx):
    """wb_k_trace_fused (csrc/wb_trace_fused.h: one launch per job, LDS-resident, list ranking over splitters, longitude
    buckets) against link -> rank -> finish on the device: smooth / rough / NaN-bordered fields, three periodic pads, three levels,
    0.25 degree included, and launches whose LDS admits only some of the jobs (both paths fill one batch's tables)"""
    cases = []
    lat, lon = synth.grid(2.5)
    pv = synth.pv_field(7, 3, lat, lon, eps=0.4)
    pv[:, [0, 1, -2, -1], :] = np.nan
    cases.append((pv, lat, lon, 2.5))
    lat1, lon1 = synth.grid(1.0)
    cases.append((synth.pv_field(11, 4, lat1, lon1, eps=0.3), lat1, lon1, 1.0))
    cases.append((synth.pv_field(40, 4, lat1, lon1, eps=0.02).astype(np.float32), lat1, lon1, 1.0))
    latq, lonq = synth.grid(0.25)
    cases.append((synth.pv_field(300, 3, latq, lonq, eps=0.0).astype(np.float32), latq, lonq, 0.25))
    cases.append((synth.pv_field(900, 2, latq, lonq, eps=0.05).astype(np.float32), latq, lonq, 0.25))
    total = 0
    for pv, la, lo, d in cases:
        for padd in (0.0, 120.0, 360.0):
            ref, st0 = _contour_tables_gpu(ctx, pv, la, lo, [-2.0, 0.5, 2.0], d, padd, trace_fused=0)
            total += ref["step"].size
            for opts in (dict(trace_fused=1), dict(trace_fused=1, trace_nmax=1024 if d > 0.5 else 4096), dict(trace_fused=1, trace_cmax=8)):
                got, st1 = _contour_tables_gpu(ctx, pv, la, lo, [-2.0, 0.5, 2.0], d, padd, **opts)
                for key in ref:
                    assert np.array_equal(ref[key], got[key]), (d, padd, opts, key)
                assert np.array_equal(st0, st1), (d, padd, opts)
    assert total > 400


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_crossings_at_one_half_of_their_edges(ctx, dtype):
    """crossing fractions of exactly one half (np.round: half to even, contour_index.py:111), a few ulps beside it and a few 1e-3
    beside it: wb_k_trace_fused decides most lattice points from a float32 quotient (v_rcp_f32) and these from the reference's
    float64 one -- same tables as the three-kernel chain, which evaluates every crossing in float64, and as the oracle"""
    total = 0
    for seed in range(6):
        pv, lat, lon = helpers.half_way_crossing_field(seed, dlon=2.5 if seed < 4 else 1.0, dtype=dtype)
        d = 2.5 if seed < 4 else 1.0
        ref, st0 = _contour_tables_gpu(ctx, pv, lat, lon, [2.0], d, 120.0, trace_fused=0)
        got, st1 = _contour_tables_gpu(ctx, pv, lat, lon, [2.0], d, 120.0, trace_fused=1)
        for key in ref:
            assert np.array_equal(ref[key], got[key]), (seed, key)
        assert np.array_equal(st0, st1)
        total += ref["step"].size
        helpers.full_parity(ctx, pv, lat, lon, synth.times(2), [2.0])
    assert total > 30


def test_fused_trace_grows_with_what_the_batches_ask_for():
    """wb_k_trace_fused's LDS follows what the context's batches have asked for (four jobs per CU on the smooth bench field): a rough
    field behind a smooth one is declined where it asks for more chains, taken by the three-kernel chain (tables = the oracle's), and
    not declined the second time"""
    c = _hip.Context(0)
    try:
        helpers.fused_growth_case(c, gpu=True)
    finally:
        c.close()


def test_device_geometry_equals_host_geometry(ctx):
    """transform_polygons (index_utils.py:129-184): the rings wb_indices leaves in pinned host memory (cut at the last meridian,
    truncated and mapped to lon / lat by the device, csrc/wb_geom.h) against the host function on the same events -- ring for
    ring, bit for bit -- through the batch and through wb_transform_polygons_device on hand-made polygons"""
    from tests import test_seam_split as tss
    polys = [np.asarray(tss.HAND[k]) for k in sorted(tss.HAND)] * 5
    for ispace in (False, True):
        host = ctx.transform_polygons(polys, tss.LON, tss.LAT, tss.NLON, index_space=ispace)
        dev = ctx.transform_polygons(polys, tss.LON, tss.LAT, tss.NLON, index_space=ispace, device=True)
        assert tss._same_rings(host, dev)
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(20, 8, lat, lon, eps=0.2)
    rings = {}
    for flag in (1, 0):
        ctx.set_option("geometry", flag)
        try:
            b = ctx.contours(pv, lat, lon, [-2.0, 2.0], 1.0, 1.0, periodic_add=120.0, intensity=pv)
            b.indices()
            rings[flag] = [b.event_geometry(kind) for kind in (0, 1, 2)]
            b.close()
        finally:
            ctx.set_option("geometry", 1)
    nsplit = 0
    for kind in (0, 1, 2):
        for a, c in zip(rings[1][kind], rings[0][kind]):
            assert np.array_equal(np.asarray(a), np.asarray(c)), kind
        ro = np.asarray(rings[1][kind][0])
        nsplit += int((np.diff(ro) != 1).sum())
    assert sum(len(np.asarray(r[0])) for r in rings[1]) > 60 and nsplit > 0


def test_rows_without_vertices_lazy_polygons_and_deferred_geometry(ctx):
    """the same on the device, where the geometry kernels really are in flight when wb_indices returns"""
    helpers.check_deferred_results(ctx, nt=8)
    helpers.check_deferred_results(ctx, nt=4, dlon=0.25)


def test_float64_fields_take_the_fast_streaming_scan(ctx):
    """float64 fields -- what calculate_smoothed_field returns (numpy promotes) -- through the streaming scan's LDS halo exchange with
    patches as keys only (context option scan_f64_fast, round 6) and through the scalar-halo kernel of rounds 2-5: the same contour
    tables and status words, and full parity with the oracle, on grids of one, two and six strips, with NaN rows (the rows the
    smoothing leaves, spatial.py:106-107) and three levels"""
    cases = []
    for dlon, t0, nt, eps in ((0.25, 40, 2, 0.3), (1.0, 3, 4, 0.2), (0.5, 11, 2, 0.1), (2.5, 0, 3, 0.2)):
        lat, lon = synth.grid(dlon)
        pv = synth.pv_field(t0, nt, lat, lon, eps=eps).astype(np.float64)
        cases.append((pv, lat, lon, dlon, nt))
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(3, 3, lat, lon, eps=0.2).astype(np.float64)
    pv[:, :2] = np.nan; pv[:, -2:] = np.nan
    cases.append((pv, lat, lon, 1.0, 3))
    for pv, lat, lon, dlon, nt in cases:
        tabs = {}
        for fast in (1, 0):
            ctx.set_option("scan_f64_fast", fast)
            try:
                b = ctx.contours(pv, lat, lon, [-2.0, 1.5, 2.0], dlon, dlon, periodic_add=120.0)
                tabs[fast] = (b.contour_table(), b.status().copy(), dict(b.timings()))
                b.close()
            finally:
                ctx.set_option("scan_f64_fast", 1)
        for key in tabs[1][0]:
            assert np.array_equal(tabs[1][0][key], tabs[0][0][key]), (dlon, key)
        assert np.array_equal(tabs[1][1], tabs[0][1]) and tabs[1][0]["step"].size > 0
        helpers.full_parity(ctx, pv, lat, lon, synth.times(nt), [-2, 2])


def test_grids_wider_than_4096_columns(ctx):
    """nlon + periodic columns up to 16384 (until round 6: 4096): a 1/16 degree grid -- 5760 + 1920 columns -- whose 23 strips the
    streaming scan's workgroup does not hold (the single-kernel scan takes over) and whose jobs the fused trace may decline (the
    three-kernel chain): full parity with the oracle on a latitude band, float32 and float64, and the contour table of a global
    field of 2881 rows"""
    dl = 0.0625
    lon = -180.0 + dl * np.arange(5760)
    lat = 25.0 + dl * np.arange(641)
    pv = synth.pv_field(5, 2, lat, lon, eps=0.0)
    got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(2), [2.0], intensity=pv, want_cells=False)
    assert len(ref["contours"]) >= 4 and len(ref["overturnings"]) >= 2
    helpers.full_parity(ctx, pv[:1].astype(np.float64), lat, lon, synth.times(1), [2.0], want_cells=False)
    # the whole globe at that resolution: contours of both levels against the oracle's (a step of 16.6 M cells)
    from oracle import contours as ocont
    latg = -90.0 + dl * np.arange(2881)
    pvg = synth.pv_field(9, 1, latg, lon, eps=0.0)
    b = ctx.contours(pvg, latg, lon, [-2.0, 2.0], dl, dl, periodic_add=120.0)
    tab = b.contour_table()
    b.close()
    want = ocont.calculate_contours(pvg, [-2.0, 2.0], latg, lon, synth.times(1), periodic_add=120, original_coordinates=False)
    helpers.compare_contours(tab, want, [-2.0, 2.0], dl)
    assert len(want) >= 2 and tab["nux"].max() == 5760 + 1920


def test_property_columns_on_the_device_against_the_reference_own_lines(ctx, golden_dir):
    """index_utils.py:104-120 (the truncated centre of mass, the three .round(2)) as the DEVICE computes them for every event
    (wb_k_pack_results -> csrc/wb_props.h:wb_props_columns), on the sums of tests/golden/reference_rules.npz against the columns the
    reference's own statements gave for them -- bit for bit -- and on awkward quotients against numpy's statement of the same lines"""
    from wavebreaking_amd.utils.index_utils import properties_from_sums
    ref = np.load(golden_dir + "/reference_rules.npz")
    events = 0
    for k in range(int(ref["props__count"])):
        key = "props__p%02d" % k
        dlon = float(ref[key + "__field"][0])
        lat, lon = synth.grid(dlon)
        want = ref[key + "__sums"]
        dev = ctx.props_columns(want, lon, lat)
        assert np.array_equal(dev[:, :2], ref[key + "__com"]) and np.array_equal(dev[:, 2], ref[key + "__mean_var"])
        assert np.array_equal(dev[:, 3], ref[key + "__intensity"]) and np.array_equal(dev[:, 4], ref[key + "__event_area"])
        events += len(want)
    assert events > 100
    # quotients on and next to integers, halves at the second decimal (round half to even), negative means
    rng = np.random.RandomState(5)
    lat, lon = synth.grid(0.25)
    n = 20000
    a = rng.uniform(1e3, 1e9, n)
    s = np.c_[a, a * rng.uniform(-40, 40, n), a * rng.uniform(-5, 5, n), a * rng.uniform(0, lon.size + 400, n), a * rng.uniform(0, lat.size - 1, n)]
    s[:2000, 3] = a[:2000] * rng.randint(0, lon.size + 400, 2000)                      # x_com / areas an integer, up to the product's rounding
    s[2000:4000, 1] = a[2000:4000] * (rng.randint(-4000, 4000, 2000) + 0.5) / 100.0    # mean_var at a tie of the second decimal
    com, mean_var, inten, area = properties_from_sums(s, lon, lat, lon.size)
    dev = ctx.props_columns(s, lon, lat)
    assert np.array_equal(dev[:, :2], np.asarray(com)) and np.array_equal(dev[:, 2], mean_var)
    assert np.array_equal(dev[:, 3], inten) and np.array_equal(dev[:, 4], area)
