#!/usr/bin/env python
"""BASELINE.json configs[3] / configs[4] at full length on ONE MI355X: a 0.25 degree record of N years of six-hourly steps
(device-generated synthetic PV, a year = one batch of 1460 steps), all three indices for every year with two batches in
flight, the catalogue (event rows + lon / lat polygons) collected on the host, then ``track_events(method="by_overlap")``
over the whole catalogue of each index.  One JSON line.  (On 8 GPUs every rank runs the same loop over its share of the
years -- bench.py --gpus 8 measures that part -- and rank 0 tracks after the gather.)

    python benchmarks/bench_record.py --years 40
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--years", type=int, default=40)
    ap.add_argument("--nt", type=int, default=1460, help="time steps per year (batch)")
    ap.add_argument("--dlon", type=float, default=0.25)
    ap.add_argument("--overlap", type=float, default=0.0)
    a = ap.parse_args()
    from wavebreaking_amd import _hip
    from wavebreaking_amd import synthetic as synth
    from wavebreaking_amd.pipeline import BatchPipeline
    pipe = BatchPipeline(0, depth=2)
    lat, lon = synth.grid(a.dlon)
    nlat, nlon, nt = lat.size, lon.size, a.nt
    levels = [-2.0, 2.0]
    fbytes = nt * nlat * nlon * 4
    # every context owns the field + intensity buffers of the year it works on (filled on its own stream)
    bufs = [(c.alloc(fbytes), c.alloc(fbytes)) for c in pipe.contexts]
    par, pari = synth.wave_params(), synth.wave_params(20241)
    width = nlon + int(120.0 / a.dlon)

    gen = [0.0] * pipe.depth

    def contours(c, k):
        d_pv, d_int = bufs[k % pipe.depth]
        w = time.perf_counter()
        c.synth_fill(d_pv, np.float32, k * nt, nt, lat, lon, par, 0.0, synth.SEED)
        c.synth_fill(d_int, np.float32, k * nt, nt, lat, lon, pari, 0.0, 20241)
        gen[k % pipe.depth] += time.perf_counter() - w        # (synth_fill returns when the field is there)
        return c.contours(d_pv, lat, lon, levels, a.dlon, a.dlon, intensity=d_int, device_shape=(nt, nlat, nlon),
                          device_dtype=np.float32, want_cells=False)

    def indices(c, k, b, g):
        b.set_step_offset(k * nt)
        b.indices(g)
        out = []
        for kind in (0, 1, 2):
            ev = b.events(kind, copy=True)
            ring_off, vert_off, gx, gy = b.event_geometry(kind)
            out.append(dict(step=ev["step"].astype(np.int64) + k * nt, sums=ev["sums"], ring_off=ring_off.copy(), vert_off=vert_off.copy(),
                            gx=gx.copy(), gy=gy.copy(), status=int(np.bitwise_or.reduce(b.status())) if nt else 0))
        b.close()
        return out

    years = []
    tic = time.perf_counter()
    pipe.run(a.years, contours, lambda m: m, indices, lambda k, r: years.append(r), full_width=width)
    t_indices = time.perf_counter() - tic

    # the catalogue of each index: dates, lattice rings (every vertex of transform_polygons is a grid point) -> labels
    ctx = pipe.contexts[0]
    t0 = np.datetime64("1980-01-01T00", "ns")
    names = ("streamers", "overturnings", "cutoffs")
    res = {}
    t_track = 0.0
    for kind in (0, 1, 2):
        parts = [y[kind] for y in years]
        step = np.concatenate([p["step"] for p in parts])
        nring = np.cumsum([0] + [p["vert_off"].size - 1 for p in parts])
        nvert = np.cumsum([0] + [p["gx"].size for p in parts])
        ring_off = np.concatenate([p["ring_off"][:-1] + nring[i] for i, p in enumerate(parts)] + [nring[-1:]])
        vert_off = np.concatenate([p["vert_off"][:-1] + nvert[i] for i, p in enumerate(parts)] + [nvert[-1:]])
        gx = np.concatenate([p["gx"] for p in parts])
        gy = np.concatenate([p["gy"] for p in parts])
        ix = np.rint((gx - lon[0]) / a.dlon).astype(np.int32)
        iy = np.rint((gy - lat[0]) / a.dlon).astype(np.int32)
        dates = t0 + step * np.timedelta64(6 * 3600 * 10 ** 9, "ns")
        tic = time.perf_counter()
        label, area, stats = ctx.track_events(dates, "by_overlap", 6.0, overlap=a.overlap, rings_csr=(ring_off, vert_off, ix, iy))
        t_track += time.perf_counter() - tic
        res[names[kind]] = dict(events=int(step.size), polygon_vertices=int(gx.size), tracks=int(label.max()) + 1 if step.size else 0,
                                track_ms_device=ctx.last_op_ms(), **stats)
    nsteps = a.years * nt
    out = {"workload": "0.25deg global PV, %d years of six-hourly steps (%d steps), levels [-2, 2], all three indices + "
                       "track_events(by_overlap), one MI355X (BASELINE.json configs[3]/[4] at full length)" % (a.years, nsteps),
           "timesteps": nsteps, "indices_s": t_indices, "indices_timesteps_per_s": nsteps / t_indices,
           "indices_includes": "generation of every year's two synthetic fields on the device: %.2f s of worker time (two workers), "
                               "and the lon / lat polygons of every event (transform_polygons) + catalogue copies on the host" % sum(gen),
           "track_s": t_track, "catalogue": res,
           "status_or": int(np.bitwise_or.reduce([p["status"] for y in years for p in y]))}
    print(json.dumps(out), flush=True)
    pipe.close()


if __name__ == "__main__":
    main()
