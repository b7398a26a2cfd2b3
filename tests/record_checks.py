"""Checks on a whole synthetic record run through benchmarks/bench_record.py (BASELINE.json configs[3] / configs[4] on one
device): sampled steps against the oracle, one year re-run alone, tracking labels on a sub-catalogue that spans a year
boundary.  Used by tests/test_gpu_record.py at full length on the GPU and, at toy size on the host-simulation build, by
tests/test_record_hostsim.py."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench_record():
    spec = importlib.util.spec_from_file_location("bench_record", os.path.join(ROOT, "benchmarks", "bench_record.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_bench():
    spec = importlib.util.spec_from_file_location("wb_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def oracle_parity_on_steps(rec, steps, levels):
    """every event of the given ABSOLUTE steps (their year's tables must have been kept) against oracle.api.run_all on the
    regenerated fields: level, polygon vertices, cell count, truncated centre-of-mass indices bit for bit, the three data
    sums to 1e-6 relative (bench._compare_step).  Returns the comparison counters."""
    from oracle import api as oapi
    from oracle import synth
    bench = load_bench()
    cmp = dict(events=0, mismatches=0, max_rel_err=0.0)
    for t in steps:
        year, local = divmod(int(t), rec.nt)
        pv = rec.field_of_step(t, "pv")
        it = rec.field_of_step(t, "intensity")
        ref = oapi.run_all(pv, levels, rec.lat, rec.lon, synth.times(1), intensity=it)
        for k, name in enumerate(("streamers", "overturnings", "cutoffs")):
            bench._compare_step(rec.tables[year][k], ref[name], local, levels, cmp)
    return cmp


def tables_equal(a, b):
    """two lists of event tables (Batch.events dictionaries), every array bit for bit"""
    for ta, tb in zip(a, b):
        assert sorted(ta) == sorted(tb)
        for key in ta:
            assert ta[key].shape == tb[key].shape and np.array_equal(ta[key], tb[key]), key
    return True


def sub_catalogue(cat, lo, hi):
    """events with lo <= step < hi of one index's catalogue -> (row numbers, dates, rings per event as lattice arrays)"""
    rows = np.nonzero((cat["step"] >= lo) & (cat["step"] < hi))[0]
    rings = []
    for e in rows:
        rr = []
        for r in range(int(cat["ring_off"][e]), int(cat["ring_off"][e + 1])):
            a, b = int(cat["vert_off"][r]), int(cat["vert_off"][r + 1])
            rr.append(np.c_[cat["ix"][a:b], cat["iy"][a:b]].astype(np.int64))
        rings.append(rr)
    return rows, cat["dates"][rows], rings


def oracle_track_labels(dates, rings, time_range=6.0, overlap=0.0):
    """oracle.tracking's by_overlap labels; pairs whose bounding boxes are disjoint overlap by exactly 0 and skip the exact
    rational clipping (for overlap >= 0 such a pair is never linked, events.py:209-214)"""
    from oracle import tracking as otrack
    pairs = otrack.time_pairs(dates, time_range)
    box = []
    for rr in rings:
        if rr:
            xy = np.concatenate(rr)
            box.append((xy[:, 0].min(), xy[:, 0].max(), xy[:, 1].min(), xy[:, 1].max()))
        else:
            box.append((1, 0, 1, 0))
    check = np.zeros(len(pairs), dtype=bool)
    exact = 0
    for n, (i, j) in enumerate(pairs):
        a, b = box[i], box[j]
        if a[0] > b[1] or b[0] > a[1] or a[2] > b[3] or b[2] > a[3] or a[0] > a[1] or b[0] > b[1]:
            continue
        ok, _ = otrack.overlap_check(rings, pairs[n:n + 1], overlap)
        check[n] = bool(ok[0])
        exact += 1
    return otrack.labels_from_pairs(len(dates), pairs[check]), len(pairs), exact


def track_parity_across_boundary(rec, name, boundary_step, min_events, ctx=None):
    """labels of the events of one index in a window of steps centred on ``boundary_step`` (grown until it holds
    ``min_events``): the library on the sub-catalogue == the oracle on the sub-catalogue; and whatever the sub-catalogue
    links stays in one track of the whole record's labels.  Returns (events, pairs in range, pairs clipped exactly)."""
    cat = rec.catalogue[name]
    ctx = ctx or rec.pipe.contexts[0]
    half = 2
    last = int(cat["step"].max()) if cat["step"].size else 0
    while True:
        rows, dates, rings = sub_catalogue(cat, boundary_step - half, boundary_step + half)
        if rows.size >= min_events or (boundary_step - half <= 0 and boundary_step + half > last):
            break
        half *= 2
    assert rows.size >= min(min_events, cat["step"].size), (name, rows.size)
    assert (cat["step"][rows] < boundary_step).any() and (cat["step"][rows] >= boundary_step).any(), "window does not span the boundary"
    want, npairs, nexact = oracle_track_labels(dates, rings)
    got, _, stats = ctx.track_events(dates, "by_overlap", 6.0, overlap=0.0, rings_per_event=rings)
    assert stats["pairs_in_range"] == npairs, (stats, npairs)
    assert np.array_equal(got, want), "%s: labels differ on %d events" % (name, int((got != want).sum()))
    if "label" in cat:
        full = cat["label"][rows]
        for g in np.unique(want):
            assert np.unique(full[want == g]).size == 1, "a track of the window is split in the record's labels"
    return int(rows.size), npairs, nexact
