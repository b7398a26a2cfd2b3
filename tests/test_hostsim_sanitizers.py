"""The kernel sources under AddressSanitizer + UndefinedBehaviorSanitizer (host-simulation build, CPU only): the irregular
kernels -- hash tables, LDS-sized arrays, capacity-overflow and retry paths -- run a parity case in a child process that
has libasan preloaded; any report fails the test."""
import os
import subprocess
import sys

from tests import helpers

CHILD = r"""
import numpy as np
from oracle import synth
from tests import helpers
from wavebreaking_amd import _hip
lib = _hip.Library(helpers.build_hostsim(sanitize=True), allow_hostsim=True)
ctx = _hip.Context(0, lib)
lat, lon = synth.grid(2.5)
for reverse in (0, 1):
    lib.lib.wb_hostsim_set_reverse(reverse)
    pv = synth.pv_field(3, 2, lat, lon, eps=0.3)
    helpers.full_parity(ctx, pv, lat, lon, synth.times(2), [-2, 2], intensity=pv)
lib.lib.wb_hostsim_set_reverse(0)
# capacity retries (segments, candidate pairs), cells equal to the level, a frame handed in, NaN rows
helpers.full_parity(ctx, synth.pv_field(2, 1, lat, lon, eps=0.3), lat, lon, synth.times(1), [-2, 2], cap_seg=1024, cap_pairs=16)
pv, la, lo = helpers.level_hit_field(1, dlon=2.5)
helpers.full_parity(ctx, pv, la, lo, synth.times(1), [2.0], intensity=pv)
helpers.symmetric_cutoff_case(ctx)
pv = synth.pv_field(3, 1, lat, lon, eps=0.2).astype(np.float64)
pv[:, [0, 1, -2, -1], :] = np.nan
helpers.full_parity(ctx, pv, lat, lon, synth.times(1), [2, -2, 1.5])
from oracle import spatial as ospatial
five = np.array([[0, 1, 0], [1, 2, 1], [0, 1, 0]])
assert np.array_equal(ctx.smooth_field(pv, 3, five), ospatial.smooth_field(pv, 3, five), equal_nan=True)
# track_events: row tables, exact overlaps, union-find (tests/test_tracking.py cases)
from tests import test_tracking as tt
tt._kernel_overlaps(ctx)
tt._kernel_tracking(ctx)
tt._kernel_thresholds(ctx)
ctx.close()
# round 5's rarer routes of the streamer filters: routine 1 in parts of the hash space (a 256-entry table), the record list four at a time,
# routine 1 inside wb_k_streamer_f234, a kernel LDS of 64 points (longer contours: the sweep in global memory)
lib2 = _hip.Library(helpers.build_hostsim(variant="f1parts_ch4", defines=("WB_F1_LTAB=256", "WB_F234_CH=4"), sanitize=True), allow_hostsim=True)
ctx2 = _hip.Context(0, lib2)
lat1, lon1 = synth.grid(1.0)
pv1 = synth.pv_field(10, 2, lat1, lon1, eps=0.3)
for opts in ({}, {"f1_fused": 1}, {"f234_lpts": 64}):
    for k, v in opts.items():
        ctx2.set_option(k, v)
    helpers.full_parity(ctx2, pv1, lat1, lon1, synth.times(2), [-2, 2])
    ctx2.set_option("f1_fused", 0); ctx2.set_option("f234_lpts", 0)
ctx2.close()
print("SANITIZED_OK")
"""


def test_parity_case_under_asan_ubsan():
    helpers.build_hostsim(sanitize=True)
    helpers.build_hostsim(variant="f1parts_ch4", defines=("WB_F1_LTAB=256", "WB_F234_CH=4"), sanitize=True)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    libubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=libasan + ":" + libubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", PYTHONPATH=helpers.ROOT)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=helpers.ROOT, capture_output=True, text=True, timeout=900)
    report = p.stdout[-3000:] + p.stderr[-6000:]
    assert p.returncode == 0 and "SANITIZED_OK" in p.stdout, report
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, report
