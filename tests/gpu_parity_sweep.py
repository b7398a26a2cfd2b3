"""Wider GPU parity sweep than the test suite (minutes, not seconds): three grid resolutions, smooth to very rough fields,
both scan variants, every table of the path against the oracle (tests/helpers.full_parity).  python tests/gpu_parity_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import synth
from tests import helpers
from wavebreaking_amd import _hip
ctx = _hip.Context(0)
t0 = time.time()
n = 0
for dl, nt, cases in ((0.25, 3, [(0.0, 100, 20240), (0.3, 500, 7), (0.6, 900, 11), (0.15, 1300, 99)]),
                      (1.0, 12, [(0.0, 0, 20240), (0.3, 40, 5), (0.8, 80, 6)]),
                      (2.5, 30, [(0.2, 0, 1), (0.5, 100, 2)])):
    lat, lon = synth.grid(dl)
    for eps, start, seed in cases:
        pv = synth.pv_field(start, nt, lat, lon, eps=eps, seed=seed)
        inten = synth.pv_field(start, nt, lat, lon, eps=0.0, seed=seed + 1)
        for split in (0, 1):
            ctx.set_option("scan_split", split)
            got, ref = helpers.full_parity(ctx, pv, lat, lon, synth.times(nt), [-2, 2], intensity=inten)
        n += 1
        print("ok dlon", dl, "eps", eps, "t0", start, {k: len(v) for k, v in ref.items()}, "status_or", int(np.bitwise_or.reduce(got["status"])), "band", got["band"], "%.0fs" % (time.time() - t0), flush=True)
print("sweep passed", n, "cases")
