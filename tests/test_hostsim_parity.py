"""CPU: the kernel sources compiled for the host (-DWB_HOSTSIM, test infrastructure) driven through the
same C ABI and ctypes binding as the product, against the oracle.  Catches logic errors in the kernels,
the ABI marshalling and the retry paths without a GPU; WB_PFOR iteration order is also reversed to expose
order dependence (= races) inside a phase."""
import numpy as np
import pytest

from oracle import synth
from tests import helpers
from wavebreaking_amd import _hip


@pytest.fixture(scope="module")
def sim():
    lib = helpers.hostsim_library()
    assert lib.backend == "hostsim"
    ctx = _hip.Context(0, lib)
    yield lib, ctx
    lib.lib.wb_hostsim_set_reverse(0)
    ctx.close()


@pytest.mark.parametrize("eager", [0, 1], ids=["lazy_f2", "eager_f2"])
@pytest.mark.parametrize("reverse", [0, 1])
def test_full_parity_1deg_two_levels(sim, reverse, eager):
    lib, ctx = sim
    lib.lib.wb_hostsim_set_reverse(reverse)
    ctx.set_option("eager_f2", eager)       # 1: chord test for every candidate pair instead of the lazy sweep
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(10, 3, lat, lon, eps=0.3)
    inten = synth.pv_field(10, 3, lat, lon, eps=0.0, seed=20241)
    got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(3), [-2, 2], intensity=inten)
    assert len(ref["streamers"]) > 5 and len(ref["overturnings"]) > 5 and len(ref["cutoffs"]) > 5
    assert got["band"] == (0, 0) and not got["status"].any()
    ctx.set_option("eager_f2", 0)


@pytest.mark.parametrize("reverse", [0, 1])
def test_streamer_filters_on_contours_longer_than_their_lds(sim, reverse):
    # wb_k_streamer_f234 holds a contour of up to f234_lpts points in LDS (records of the rows' largest candidates tested a batch
    # at a time); a longer one is swept row after row in global memory: the same events either way
    lib, ctx = sim
    lib.lib.wb_hostsim_set_reverse(reverse)
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(10, 3, lat, lon, eps=0.3)
    for lpts, fused in ((64, 1), (256, 1), (0, 0), (64, 0), (0, 1)):
        ctx.set_option("f234_lpts", lpts)
        ctx.set_option("f1_fused", fused)      # routine 1 inside wb_k_streamer_f234 / as its own launch
        try:
            got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(3), [-2, 2], intensity=pv)
            assert len(ref["streamers"]) > 5
        finally:
            ctx.set_option("f234_lpts", 0)
            ctx.set_option("f1_fused", 0)


def test_demo_shaped_config0(sim):
    # BASELINE.json configs[0]: demo-data shape 360 x 179 (lat -89..89), one step, level 2, all three indices
    lib, ctx = sim
    lib.lib.wb_hostsim_set_reverse(0)
    lat, lon = synth.grid(1.0, lat_min=-89.0, lat_max=89.0)
    assert lat.size == 179 and lon.size == 360
    pv = synth.pv_field(0, 1, lat, lon, eps=0.1)
    helpers.full_parity(ctx, pv, lat, lon, synth.times(1), [2], intensity=pv)


def test_nan_rows_f64_and_three_levels(sim):
    lib, ctx = sim
    lat, lon = synth.grid(2.5)
    pv = synth.pv_field(3, 2, lat, lon, eps=0.2).astype(np.float64)
    pv[:, [0, 1, -2, -1], :] = np.nan          # calculate_smoothed_field leaves NaN border rows (spatial.py:106-107)
    pv[1, 30:33, 20:26] = np.nan
    got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(2), [2, -2, 1.5])
    assert (got["status"] & _hip.STATUS_HAS_NAN).all()


@pytest.mark.parametrize("padd", [0, 57, 360])
def test_periodic_add_variants(sim, padd):
    lib, ctx = sim
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(5, 2, lat, lon, eps=0.3)
    helpers.full_parity(ctx, pv, lat, lon, synth.times(2), [2], periodic_add=padd)


def test_quarter_degree_one_step(sim):
    lib, ctx = sim
    lat, lon = synth.grid(0.25)
    pv = synth.pv_field(7, 1, lat, lon, eps=0.3)
    got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(1), [2], intensity=pv)
    assert max(len(c["geometry"]) for c in ref["contours"]) > 2000


def test_capacity_retry_paths(sim):
    # tiny capacities force the segment- and pair-overflow retries; results must not change
    lib, ctx = sim
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(2, 2, lat, lon, eps=0.3)
    helpers.full_parity(ctx, pv, lat, lon, synth.times(2), [-2, 2], cap_seg=1024, cap_pairs=16)


def test_level_hit_is_flagged(sim):
    # a cell that equals the level AND sits on the contour puts a crossing on a grid vertex
    lib, ctx = sim
    lat, lon = synth.grid(2.5)
    pv = synth.pv_field(0, 1, lat, lon, eps=0.0)
    j, i = np.argwhere((pv[0, :, :-1] < 2.0) & (pv[0, :, 1:] > 2.0))[0]
    pv[0, j, i] = 2.0
    got = helpers.run_abi(ctx, pv, lat, lon, [2.0])
    assert got["status"][0] & _hip.STATUS_LEVEL_HIT
    # a hit far away from any crossing changes nothing and is not reported
    pv = synth.pv_field(0, 1, lat, lon, eps=0.0)
    pv[0, 36, 5] = 2.0
    pv[0, 35:38, 4:7] = np.minimum(pv[0, 35:38, 4:7], 2.0)
    got2 = helpers.run_abi(ctx, pv, lat, lon, [2.0])
    assert not (got2["status"][0] & _hip.STATUS_LEVEL_HIT)


def test_empty_field_gives_empty_tables(sim):
    lib, ctx = sim
    lat, lon = synth.grid(2.5)
    pv = np.zeros((2, lat.size, lon.size), dtype=np.float32)
    got = helpers.run_abi(ctx, pv, lat, lon, [2.0])
    assert got["contours"]["step"].size == 0
    for k in ("streamers", "overturnings", "cutoffs"):
        assert got[k]["step"].size == 0


def test_numpy_sum_restatement(sim):
    # wb_np_pairwise_sum (csrc/wb_core.h) against numpy's own a.sum(), bit for bit: leaves, splits, the 8192-element blocks
    import ctypes as C
    lib, _ = sim
    f = lib.lib.wb_hostsim_np_sum
    f.restype = C.c_double
    f.argtypes = [C.POINTER(C.c_double), C.c_int64]
    for seed in range(8):
        rng = np.random.default_rng(seed)
        for n in list(range(0, 24)) + [127, 128, 129, 255, 256, 257, 1000, 4095, 4097, 8191, 8192, 8193, 12000, 16385, 40000]:
            a = np.ascontiguousarray(rng.uniform(0, 30, n) * rng.choice([1.0, 1e-3, 1e3], n))
            assert f(a.ctypes.data_as(C.POINTER(C.c_double)), n) == float(a.sum()), (seed, n)
            if n > 40:                                           # a slice, as check_groups takes it
                b = a[7:n - 5]
                assert f(b.ctypes.data_as(C.POINTER(C.c_double)), b.size) == float(b.sum()), (seed, n)


def _reference_group_order(pairs):
    """combine_shared (index_utils.py:187-214) for the first group of `pairs`, on ndarray rows as the reference runs it --
    with the running interpreter's own sets"""
    first, *rest = np.asarray(pairs)
    first = set(first)
    lf = -1
    while len(first) > lf:
        lf = len(first)
        rest2 = []
        for r in rest:
            if len(first.intersection(set(r))) > 0:
                first |= set(r)
            else:
                rest2.append(r)
        rest = rest2
    return [int(v) for v in list(first)]


def test_cpython_set_order_restatement(sim):
    # wb_pyset_group_order (csrc/wb_core.h) against real sets: random connected groups of 2..40 small integers, their
    # crossing pairs in lexicographic order (self pairs included), table growth included
    import ctypes as C
    lib, _ = sim
    f = lib.lib.wb_hostsim_group_order
    f.restype = C.c_int
    ip = C.POINTER(C.c_int)
    rng = np.random.RandomState(0)
    for trial in range(1500):
        nm = rng.randint(2, 40)
        base = rng.randint(0, 200)
        members = np.sort(rng.choice(np.arange(base, base + rng.randint(nm, nm * 3 + 1)), nm, replace=False))
        edges = set((int(m), int(m)) for m in members)
        perm = rng.permutation(nm)
        for k in range(1, nm):
            a, b = members[perm[k]], members[perm[rng.randint(0, k)]]
            edges.add((int(min(a, b)), int(max(a, b))))
        for _ in range(rng.randint(0, nm)):
            a, b = rng.choice(members, 2)
            edges.add((int(min(a, b)), int(max(a, b))))
        pairs = sorted(edges)
        pp = np.ascontiguousarray([p for p, q in pairs], dtype=np.int32)
        pq = np.ascontiguousarray([q for p, q in pairs], dtype=np.int32)
        out = np.zeros(nm + 8, dtype=np.int32)
        cap = 8
        while cap < 8 * nm:
            cap *= 2
        n = f(pp.ctypes.data_as(ip), pq.ctypes.data_as(ip), len(pairs), cap, out.ctypes.data_as(ip))
        assert out[:n].tolist() == _reference_group_order(pairs), (trial, pairs[:8])


def test_group_maximum_shared_by_two_members(sim):
    # check_groups with an EXACT tie of the largest length (two chords shifted by one point): the reference keeps the member
    # that comes first in its CPython set -- here (15, 16) iterate as 16, 15.  Found by tools/parity_hunt.py (1 degree, 6 steps)
    lib, ctx = sim
    lat, lon = synth.grid(1.0)
    for seed, eps, t0, levels in ((41289, 0.0, 3601, [-2, 2]), (42352, 0.1, 2195, [1.0, 3.0])):
        pv = synth.pv_field(t0, 6, lat, lon, eps=eps, seed=1000 + seed)
        for eager in (0, 1):
            ctx.set_option("eager_f2", eager)
            try:
                helpers.full_parity(ctx, pv, lat, lon, synth.times(6), levels, intensity=pv)
            finally:
                ctx.set_option("eager_f2", 0)


def test_group_maximum_of_chords_shifted_along_the_contour(sim):
    # check_groups (streamer_index.py:224-251) keeps the member with the largest on[ind1 : ind2 + 1].sum(): this 0.25 degree
    # field holds a group with two chords shifted by one point, whose sums differ in the last bits only -- the argmax turns
    # on numpy's pairwise summation of the slices (a prefix difference picked the other chord); lazy and eager routine 2
    lib, ctx = sim
    lat, lon = synth.grid(0.25)
    pv = synth.pv_field(40, 1, lat, lon, eps=0.3)
    for eager in (0, 1):
        ctx.set_option("eager_f2", eager)
        try:
            helpers.full_parity(ctx, pv, lat, lon, synth.times(1), [2])
        finally:
            ctx.set_option("eager_f2", 0)


def test_bad_arguments_fail_loudly(sim):
    lib, ctx = sim
    lat, lon = synth.grid(2.5)
    pv = np.zeros((1, lat.size, lon.size), dtype=np.float32)
    with pytest.raises(RuntimeError):
        ctx.contours(pv, lat, lon, [1, 2, 3, 4, 5], 2.5, 2.5)          # more than 4 levels
    with pytest.raises(RuntimeError):
        ctx.contours(pv, lat, lon, [2], 2.5, 2.5, periodic_add=400)   # > 360 degrees


def test_packed_tables_in_global_memory():
    # packed (one-word) table entries, but LDS tables too small for any job: the global-memory variants of the packed paths
    lib = _hip.Library(helpers.build_hostsim(variant="noltab", defines=("WB_F1_LTAB=2", "WB_TRACE_LTAB=2")), allow_hostsim=True)
    ctx = _hip.Context(0, lib)
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(4, 2, lat, lon, eps=0.3)
    helpers.full_parity(ctx, pv, lat, lon, synth.times(2), [-2, 2], intensity=pv)
    ctx.close()


def test_big_job_paths_without_lds():
    # a build whose LDS threshold is tiny sends every job down the global-memory paths of the link / rank kernels
    lib = _hip.Library(helpers.build_hostsim(variant="nolds", defines=("WB_TRACE_LDS_N=16", "WB_TRACE_PACK=0", "WB_PROPS_LROWS=4", "WB_PROPS_PART=4", "WB_F1_LTAB=2", "WB_TRACE_LTAB=2")), allow_hostsim=True)
    ctx = _hip.Context(0, lib)
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(4, 2, lat, lon, eps=0.3)
    helpers.full_parity(ctx, pv, lat, lon, synth.times(2), [-2, 2], intensity=pv)
    ctx.close()


def test_streamer_sweep_with_a_short_record_list():
    # wb_k_streamer_f234 lists the records of the rows' bounds 256 at a time and goes on behind the last one when all of a list have
    # passed; a build that lists 4 at a time does that on every contour with more than four survivors
    lib = _hip.Library(helpers.build_hostsim(variant="f1parts_ch4", defines=("WB_F1_LTAB=256", "WB_F234_CH=4")), allow_hostsim=True)   # (one build for this and the next test)
    for reverse in (0, 1):
        lib.lib.wb_hostsim_set_reverse(reverse)
        ctx = _hip.Context(0, lib)
        try:
            lat, lon = synth.grid(1.0)
            pv = synth.pv_field(10, 3, lat, lon, eps=0.3)
            got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(3), [-2, 2], intensity=pv)
            assert len(ref["streamers"]) > 5
            lat, lon = synth.grid(0.25)
            pv = synth.pv_field(40, 1, lat, lon, eps=0.3)
            helpers.full_parity(ctx, pv, lat, lon, synth.times(1), [2])
        finally:
            ctx.close()
    lib.lib.wb_hostsim_set_reverse(0)


def test_routine1_in_parts_and_with_the_global_table():
    # check_duplicates (streamer_index.py:160-183): a contour with more western pairs than half the LDS table takes the table in
    # several parts (keys split by bits of their hash); a build with a 256-entry table sends the 0.25 degree contours there, one
    # with a 2-entry table (too small to split) to the table in global memory: the same pairs survive
    import ctypes
    lat, lon = synth.grid(0.25)
    pv = synth.pv_field(40, 1, lat, lon, eps=0.3)
    for variant, defines, path in (("f1parts_ch4", ("WB_F1_LTAB=256", "WB_F234_CH=4"), 0), ("nolds", ("WB_TRACE_LDS_N=16", "WB_TRACE_PACK=0", "WB_PROPS_LROWS=4", "WB_PROPS_PART=4", "WB_F1_LTAB=2", "WB_TRACE_LTAB=2"), 1)):
        lib = _hip.Library(helpers.build_hostsim(variant=variant, defines=defines), allow_hostsim=True)
        lib.lib.wb_hostsim_path_count.restype = ctypes.c_longlong
        lib.lib.wb_hostsim_path_count(0); lib.lib.wb_hostsim_path_count(1)
        ctx = _hip.Context(0, lib)
        try:
            got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(1), [2])
        finally:
            ctx.close()
        assert len(ref["streamers"]) > 3
        assert lib.lib.wb_hostsim_path_count(path) > 0, variant


def test_com_follows_the_reference_arithmetic(sim):
    """calculate_properties truncates x_com / areas (index_utils.py:105-110): overturning boxes with min_lon + max_lon even
    put that quotient ON an integer, so the sums must be pandas' compensated per-cell sums to the last bit."""
    lib, ctx = sim
    lib.lib.wb_hostsim_set_reverse(0)
    helpers.NEAR_INTEGER_EVENTS.clear()
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(0, 12, lat, lon, eps=0.3)
    got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(12), [-2, 2], intensity=pv)
    assert helpers.NEAR_INTEGER_EVENTS.get("overturnings", 0) >= 20       # compare_events asserted bit equality of the sums there
    assert got["com_rechecks"][1] >= 20


def test_com_of_symmetric_rasterised_events(sim):
    lib, ctx = sim
    got, ref, rechecks, near = helpers.symmetric_cutoff_case(ctx)
    assert near >= 10 and rechecks[0] >= near and rechecks[1] == 0


def test_cells_equal_to_the_level_golden_fields(sim, golden_dir):
    lib, ctx = sim
    nequal, nflag = helpers.level_hit_goldens_case(ctx, golden_dir)
    assert nequal >= 35 and nflag >= 4


def test_order_dependent_joins_golden_fields(sim, golden_dir):
    """two contour ends on one grid vertex: the flagged job is joined by the sequential walk (wb_k_trace_walk) and equals
    scikit-image on the 48 fields it returns contours on; STATUS_LEVEL_KEYERROR on the 6 it raises KeyError on"""
    lib, ctx = sim
    nequal, nflag = helpers.level_hit_goldens_case(ctx, golden_dir, "contours_levelwalk.npz")
    assert nequal == 48 and nflag == 54


def test_link_merge_by_searches_equals_merge_by_row_tables(sim):
    """wb_k_trace_link merges a job's strip lists through row tables in LDS; the binary-search merge (option link_rows = 0:
    the path of grids whose tables do not fit) gives the same contours -- with NaN rows, a periodic pad of 360 degrees (as many
    copy lists as base lists) and a rough field"""
    lib, ctx = sim
    lat, lon = synth.grid(2.5)
    pv = synth.pv_field(3, 3, lat, lon, eps=0.4)
    pv[:, [0, 1, -2, -1], :] = np.nan
    for padd in (120.0, 360.0):
        tabs = []
        for rows in (1, 0):
            ctx.set_option("link_rows", rows)
            try:
                b = ctx.contours(pv, lat, lon, [-2.0, 0.5, 2.0], 2.5, 2.5, periodic_add=padd)
                tabs.append(b.contour_table())
                b.close()
            finally:
                ctx.set_option("link_rows", 1)
        assert tabs[0]["step"].size > 20
        for key in tabs[0]:
            assert np.array_equal(tabs[0][key], tabs[1][key]), (padd, key)
    helpers.full_parity(ctx, pv[:, 2:-2], lat[2:-2], lon, synth.times(3), [-2, 2])


def _contour_tables(ctx, pv, lat, lon, levels, d, padd, **opts):
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


@pytest.mark.parametrize("reverse", [0, 1])
def test_fused_trace_kernel_equals_the_three_kernel_chain(sim, reverse):
    """wb_k_trace_fused (one launch per job, the job resident in LDS, list ranking over splitters, buckets by longitude instead
    of hash tables) against link -> rank -> finish: smooth and rough fields (hundreds of small closed loops: loops without a
    splitter, many chains), NaN rows, a periodic pad of 0 / 120 / 360 degrees, three levels -- and launches whose LDS admits
    only SOME of the jobs (trace_nmax / trace_cmax small), so that both paths fill one batch's tables"""
    lib, ctx = sim
    lib.lib.wb_hostsim_set_reverse(reverse)
    try:
        cases = []
        lat, lon = synth.grid(2.5)
        cases.append((synth.pv_field(3, 3, lat, lon, eps=0.0), lat, lon, 2.5))
        pv = synth.pv_field(7, 3, lat, lon, eps=0.4)
        pv[:, [0, 1, -2, -1], :] = np.nan
        cases.append((pv, lat, lon, 2.5))
        lat1, lon1 = synth.grid(1.0)
        cases.append((synth.pv_field(11, 2, lat1, lon1, eps=0.3), lat1, lon1, 1.0))
        cases.append((synth.pv_field(40, 2, lat1, lon1, eps=0.02).astype(np.float32), lat1, lon1, 1.0))
        total = 0
        for pv, la, lo, d in cases:
            for padd in (0.0, 120.0, 360.0):
                ref, st0 = _contour_tables(ctx, pv, la, lo, [-2.0, 0.5, 2.0], d, padd, trace_fused=0)
                total += ref["step"].size
                for opts in (dict(trace_fused=1), dict(trace_fused=1, trace_nmax=600), dict(trace_fused=1, trace_cmax=8),
                             dict(trace_fused=1, trace_nmax=512, trace_cmax=4)):
                    got, st1 = _contour_tables(ctx, pv, la, lo, [-2.0, 0.5, 2.0], d, padd, **opts)
                    for key in ref:
                        assert np.array_equal(ref[key], got[key]), (d, padd, opts, key)
                    assert np.array_equal(st0, st1), (d, padd, opts)
        assert total > 250
    finally:
        lib.lib.wb_hostsim_set_reverse(0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_crossings_at_one_half_of_their_edges(sim, dtype):
    """crossing fractions of exactly one half (np.round: half to even, contour_index.py:111), a few ulps beside it and a few 1e-3
    beside it: wb_k_trace_fused decides most lattice points from a float32 quotient and these from the reference's float64 one --
    same tables as the three-kernel chain (which evaluates every crossing in float64) and as the oracle"""
    lib, ctx = sim
    total = 0
    for seed in range(3):
        pv, lat, lon = helpers.half_way_crossing_field(seed, dtype=dtype)
        ref, st0 = _contour_tables(ctx, pv, lat, lon, [2.0], 2.5, 120.0, trace_fused=0)
        got, st1 = _contour_tables(ctx, pv, lat, lon, [2.0], 2.5, 120.0, trace_fused=1)
        for key in ref:
            assert np.array_equal(ref[key], got[key]), (seed, key)
        assert np.array_equal(st0, st1)
        total += ref["step"].size
        helpers.full_parity(ctx, pv, lat, lon, synth.times(2), [2.0])
    assert total > 20


def test_fused_trace_grows_with_what_the_batches_ask_for():
    """the fused trace kernel's LDS follows the context's batches: a rough field behind a smooth one asks for more chains than the
    launch admits -- those jobs are declined (counted), the three-kernel chain takes them, the tables equal the oracle's; the same
    field again finds a launch that has grown and is not declined"""
    helpers.fused_growth_case(_hip.Context(0, helpers.hostsim_library()))


def test_reference_own_code_fixture_through_the_abi(sim, golden_dir):
    """the contour table, the overturning boxes and the contour selection / cutoff filter against outputs of the reference's
    own statements (tests/golden/reference_rules.npz)"""
    lib, ctx = sim
    ncont, nbox, nframes = helpers.reference_rules_case(ctx, golden_dir)
    assert ncont > 150 and nbox > 300 and nframes == 40


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_cells_equal_to_the_level_full_path(sim, seed):
    """contours AND events of a field with level hits equal the oracle's, whose find_contours follows skimage's
    coordinate joins there (pinned by contours_levelhit.npz)"""
    lib, ctx = sim
    lib.lib.wb_hostsim_set_reverse(seed & 1)
    pv, lat, lon = helpers.level_hit_field(seed)
    got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(1), [2.0], intensity=pv)
    assert got["status"][0] & _hip.STATUS_LEVEL_HIT and not (got["status"][0] & _hip.STATUS_LEVEL_ORDER)
    lib.lib.wb_hostsim_set_reverse(0)


def test_box_com_one_lane_walk():
    # a build whose row-parallel box evaluation never applies: every box in the guard band is walked cell by cell by one
    # lane -- same sums
    lib = _hip.Library(helpers.build_hostsim(variant="boxlane", defines=("WB_BOXCOM_FORCE_WALK=1", "WB_BOXCOM_ROWS=3", "WB_BOXCOM_BOUND=0")), allow_hostsim=True)
    ctx = _hip.Context(0, lib)
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(0, 6, lat, lon, eps=0.3)
    got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(6), [-2, 2], intensity=pv)
    assert got["com_rechecks"][1] >= 8 and got["com_rechecks"][2] == got["com_rechecks"][1]
    ctx.close()


def test_box_com_in_chunks_of_two_rows():
    # the row-parallel evaluation with chunks of two rows and a single segment per row: state handed from chunk to chunk, wide
    # rows walked chunk-wise by one lane and the prediction taken up again after them -- same sums
    lib = _hip.Library(helpers.build_hostsim(variant="boxchunk", defines=("WB_BOXCOM_ROWS=2", "WB_BOXCOM_SEGS=1", "WB_BOXCOM_BOUND=0")), allow_hostsim=True)
    ctx = _hip.Context(0, lib)
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(0, 6, lat, lon, eps=0.3)
    got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(6), [-2, 2], intensity=pv)
    assert got["com_rechecks"][1] >= 8 and 0 < got["com_rechecks"][2] < got["com_rechecks"][1]
    ctx.close()


def test_box_sums_from_exact_totals_equal_the_chained_evaluation():
    """wb_k_box_com settles most queued boxes from the EXACT totals of their terms (no rounding boundary within the compensated
    sum's error bound: csrc/wb_props.h); a build with -DWB_BOXCOM_BOUND=2 evaluates every such box BOTH ways and counts the
    boxes on which the two differ in a single bit: none, on every grid of the suite; and the default build settles most boxes
    that way while the full parity (sums to the last bit where the centre of mass sits next to an integer) holds"""
    lib = _hip.Library(helpers.build_hostsim(variant="boxbound2", defines=("WB_BOXCOM_BOUND=2",)), allow_hostsim=True)
    ctx = _hip.Context(0, lib)
    settled = queued = 0
    try:
        for dlon, t0, nt, eps in ((1.0, 0, 12, 0.3), (1.0, 100, 12, 0.0), (2.5, 7, 16, 0.2), (0.5, 40, 4, 0.1), (0.25, 3, 1, 0.0)):
            lat, lon = synth.grid(dlon)
            pv = synth.pv_field(t0, nt, lat, lon, eps=eps)
            b = ctx.contours(pv, lat, lon, [-2.0, 2.0], dlon, dlon, periodic_add=120.0, want_cells=False)
            b.indices(streamers=False, cutoffs=False)
            assert b.counter("box_bound_mismatch") == 0, (dlon, t0)
            settled += b.counter("box_bound")
            queued += b.com_rechecks()[1]
            b.close()
    finally:
        ctx.close()
    assert queued >= 50 and settled >= 0.8 * queued, (settled, queued)


def test_polygon_that_crosses_itself_is_flagged(sim):
    lib, ctx = sim
    helpers.not_simple_polygon_case(ctx)


def _streamer_tables(ctx, pv, lat, lon, **opts):
    for k, v in opts.items():
        ctx.set_option(k, v)
    try:
        got = helpers.run_abi(ctx, pv, lat, lon, [-2, 2], want_cells=False)
    finally:
        for k in opts:
            ctx.set_option(k, 1)
    return got


def test_all_pairs_kernel_variants_agree(sim):
    """wb_k_streamer_pairs2 (contour in LDS, sin^2 tables of the regular grid) and the tiled kernel with its per-point table
    find the same streamers; a contour longer than the LDS holds takes the tiled kernel inside the same launch sequence."""
    lib, ctx = sim
    lib.lib.wb_hostsim_set_reverse(0)
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(21, 3, lat, lon, eps=0.3)
    a = _streamer_tables(ctx, pv, lat, lon, pairs_lds=1)
    b = _streamer_tables(ctx, pv, lat, lon, pairs_lds=0)
    assert a["streamers"]["step"].size > 5
    for key in ("step", "level_idx", "contour", "ind1", "ind2", "poly_x", "poly_y", "ncells"):
        assert np.array_equal(a["streamers"][key], b["streamers"][key]), key
    assert np.array_equal(a["streamers"]["sums"], b["streamers"]["sums"])
    assert a["band"] == (0, 0) and b["band"] == (0, 0)


def test_all_pairs_tables_on_a_jittered_coordinate_axis(sim):
    """coordinates off the regular grid by 1e-7 degrees (the reference's own check, data_utils.py:129-144, would not let
    such an axis through -- the C ABI takes what it is given): the guard band of the table-driven great-circle term widens
    by the deviation and the streamers stay those of the tiled kernel, which works from the coordinates as given"""
    lib, ctx = sim
    lat, lon = synth.grid(1.0)
    rng = np.random.RandomState(3)
    latj = lat + 1e-7 * rng.uniform(-1, 1, lat.size)
    pv = synth.pv_field(22, 2, lat, lon, eps=0.2)

    def run(flag):
        ctx.set_option("pairs_lds", flag)
        try:
            b = ctx.contours(pv, latj, lon, [-2, 2], 1.0, 1.0)
            b.indices()
            ev = b.events(_hip.STREAMERS)
            b.close()
        finally:
            ctx.set_option("pairs_lds", 1)
        return ev

    a, c = run(1), run(0)
    assert a["step"].size > 3
    for key in ("step", "level_idx", "ind1", "ind2", "poly_x", "poly_y"):
        assert np.array_equal(a[key], c[key]), key


@pytest.mark.parametrize("bands", [1, 2, 7])
def test_two_kernel_scan_simulation_equals_the_single_scan(sim, bands):
    """the host simulation of wb_k_scan_stream + wb_k_scan_build (patch capture per row band and strip in the field's own
    precision, then the list build) against the plain square-by-square scan: same contours and status words on a rough
    float32 field with NaN cells, a cell equal to a level and a non-default periodic_add, whatever the number of bands"""
    lib, ctx = sim
    lat, lon = synth.grid(1.0)
    pv = synth.pv_field(5, 2, lat, lon, eps=0.4)
    pv[0, 40:43, 100:104] = np.nan
    pv[1, 60, 359] = 2.0                                         # a level hit on the last column: the halo wraps to column 0
    out = []
    for split in (1, 0):
        ctx.set_option("scan_split", split)
        ctx.set_option("scan_grid_mult", bands)
        b = ctx.contours(pv, lat, lon, [-2.0, 2.0, 1.25], 1.0, 1.0, periodic_add=77.0)
        out.append((b.contour_table(), b.status().copy()))
        b.close()
    ctx.set_option("scan_split", 1)
    ctx.set_option("scan_grid_mult", 1)
    (ta, sa), (tb, sb) = out
    assert np.array_equal(sa, sb) and ta["step"].size > 6
    for key in ta:
        assert np.array_equal(ta[key], tb[key]), key


def test_rows_without_vertices_lazy_polygons_and_deferred_geometry(sim):
    lib, ctx = sim
    lib.lib.wb_hostsim_set_reverse(0)
    helpers.check_deferred_results(ctx, nt=3)
