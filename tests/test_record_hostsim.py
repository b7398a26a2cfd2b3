"""The record runner of benchmarks/bench_record.py and the checks tests/test_gpu_record.py applies to the 40-year record, at
toy size on the host-simulation build (CPU): 2.5 degree grid, 4 'years' of 6 steps."""
import numpy as np

from tests import helpers
from tests import record_checks as rc


def test_record_runner_and_checks_on_a_toy_record():
    br = rc.load_bench_record()
    years, nt = 4, 6
    out, rec = br.run_record(years, nt, 2.5, 0.0, keep_tables=range(years), library=helpers.hostsim_library())
    try:
        assert out["timesteps"] == years * nt and out["status_or"] == 0 and out["band_hits"] == [0, 0]
        total = 0
        for name in br.NAMES:
            cat = rec.catalogue[name]
            assert np.all(np.diff(cat["step"]) >= 0)
            assert out["catalogue"][name]["events"] == cat["step"].size
            total += cat["step"].size
        assert total > 0
        sample = [0, nt - 1, nt, 2 * nt + 3, years * nt - 1]
        cmp = rc.oracle_parity_on_steps(rec, sample, br.LEVELS)
        assert cmp["mismatches"] == 0 and cmp["events"] > 0
        _, alone = br.run_record(years, nt, 2.5, 0.0, keep_tables=[2], track=False, only_years=[2], pipe=rec.pipe)
        assert rc.tables_equal(alone.tables[2], rec.tables[2])
        for name in br.NAMES:
            if rec.catalogue[name]["step"].size >= 4:
                rc.track_parity_across_boundary(rec, name, 2 * nt, 40)
    finally:
        rec.close()
