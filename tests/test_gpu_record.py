"""benchmarks/bench_record.py (BASELINE.json configs[3] / configs[4] on one GPU) on a short record: runs, finds events in every
index, tracks them."""
import json
import os
import subprocess
import sys

import pytest

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
