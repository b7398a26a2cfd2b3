"""BASELINE.json configs[3] / configs[4] on one GPU (benchmarks/bench_record.py): a short record as a smoke run, and the
FULL 40-year record (58 400 steps of 1440 x 721, all three indices + track_events) with its catalogue checked: sampled
steps against the oracle, one year re-run alone bit-equal, tracking labels across a year boundary against oracle.tracking."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import record_checks as rc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_record_benchmark_runs_on_a_short_record():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "bench_record.py"), "--years", "3", "--nt", "40"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["timesteps"] == 120 and out["status_or"] == 0
    for name in ("streamers", "overturnings", "cutoffs"):
        c = out["catalogue"][name]
        assert c["events"] > 0 and 0 < c["tracks"] <= c["events"] and c["pairs_in_range"] > 0


@pytest.mark.gpu
def test_forty_year_record_full_length():
    """configs[3] (40 years, all three indices) and configs[4] (+ track_events) in their one-GPU form."""
    br = rc.load_bench_record()
    years, nt = 40, 1460
    # steps sampled across the record (first and last step, both sides of the year boundary 19 | 20, a few in between)
    sample = [0, 5 * nt + 700, 11 * nt + 33, 17 * nt + 1200, 20 * nt - 1, 20 * nt, 26 * nt + 404, 33 * nt + 999, years * nt - 1]
    keep = sorted(set(t // nt for t in sample))
    out, rec = br.run_record(years, nt, 0.25, 0.0, keep_tables=keep)
    try:
        print("[record]", json.dumps(out))
        assert out["timesteps"] == 58400 and out["status_or"] == 0 and out["band_hits"] == [0, 0]
        for name in br.NAMES:
            c = out["catalogue"][name]
            assert c["events"] > 100000 and 0 < c["tracks"] < c["events"]
            cat = rec.catalogue[name]
            assert np.all(np.diff(cat["step"]) >= 0) and cat["step"][0] >= 0 and cat["step"][-1] < years * nt    # time order
            assert np.unique(cat["step"] // nt).size == years                                                      # every year holds events
        # (1) oracle parity on the sampled steps
        cmp = rc.oracle_parity_on_steps(rec, sample, br.LEVELS)
        print("[record] oracle parity:", cmp)
        assert cmp["mismatches"] == 0 and cmp["events"] >= 60 and cmp["max_rel_err"] <= 1e-6
        # (2) batch invariance: year 20 alone gives the rows it gave inside the record
        _, alone = br.run_record(years, nt, 0.25, 0.0, keep_tables=[20], track=False, only_years=[20], pipe=rec.pipe)
        assert rc.tables_equal(alone.tables[20], rec.tables[20])
        # (3) tracking across the year boundary 19 | 20: >= 2 000 overturnings, 150 streamers and cutoffs (the oracle clips in exact
        # rationals: ~0.1 s per pair of 0.25 degree streamers)
        for name, nmin in (("overturnings", 2000), ("streamers", 150), ("cutoffs", 150)):
            n, npairs, nexact = rc.track_parity_across_boundary(rec, name, 20 * nt, nmin)
            print("[record] tracking parity %s: %d events, %d pairs in range, %d clipped exactly" % (name, n, npairs, nexact))
            assert n >= nmin and npairs > n // 2
    finally:
        rec.close()
