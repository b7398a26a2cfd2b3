#!/usr/bin/env python
"""Throughput of the two pre-processing kernels on device-resident synthetic 0.25 degree fields (one JSON line).
Algorithmic bytes: smoothing = cells x (4 B float32 read + 8 B float64 written) whatever the pass count;
momentum flux = cells x 3 x 4 B."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=1460)
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from wavebreaking_amd import synthetic as synth
    from wavebreaking_amd import _hip
    ctx = _hip.Context(0)
    lat, lon = synth.grid(0.25)
    nt, nlat, nlon = a.nt, lat.size, lon.size
    cells = nt * nlat * nlon
    d_u = ctx.alloc(cells * 4)
    d_v = ctx.alloc(cells * 4)
    d_o = ctx.alloc(cells * 8)
    ctx.synth_fill(d_u, np.float32, 0, nt, lat, lon, synth.wave_params(), 0.3, synth.SEED)
    ctx.synth_fill(d_v, np.float32, 0, nt, lat, lon, synth.wave_params(20241), 0.3, 20241)
    five = np.array([[0, 1, 0], [1, 2, 1], [0, 1, 0]])
    out = {"nt": nt, "grid": [nlat, nlon], "passes": a.passes}
    for name, per_pass in (("smooth_fused", False), ("smooth_per_pass", True)):
        ms = []
        for _ in range(a.reps):
            ctx.smooth_field(d_u, a.passes, five, device_shape=(nt, nlat, nlon), device_dtype=np.float32, out_device=d_o,
                             per_pass=per_pass)
            ms.append(ctx.last_op_ms())
        t = float(np.median(ms[1:] or ms))
        out[name] = {"ms": t, "timesteps_per_s": nt / t * 1e3, "algorithmic_GBps": cells * 12 / t / 1e6,
                     "frac_of_8TBps": cells * 12 / t / 1e6 / 8000.0}
    ms = []
    for _ in range(a.reps):
        ctx.momentum_flux(d_u, d_v, device_shape=(nt, nlat, nlon), device_dtype=np.float32, out_device=d_o)
        ms.append(ctx.last_op_ms())
    t = float(np.median(ms[1:] or ms))
    out["momentum_flux"] = {"ms": t, "timesteps_per_s": nt / t * 1e3, "algorithmic_GBps": cells * 12 / t / 1e6,
                            "frac_of_8TBps": cells * 12 / t / 1e6 / 8000.0}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
