#!/usr/bin/env python
"""BASELINE.json configs[3] / configs[4] at full length on ONE MI355X: a 0.25 degree record of N years of six-hourly steps
(device-generated synthetic PV, a year = one batch of 1460 steps), all three indices for every year with two batches in
flight, the catalogue (event rows + lon / lat polygons) collected on the host, then ``track_events(method="by_overlap")``
over the whole catalogue of each index.  One JSON line.  (On 8 GPUs every rank runs the same loop over its share of the
years -- bench.py --gpus 8 measures that part -- and rank 0 tracks after the gather.)

    python benchmarks/bench_record.py --years 40

``run_record`` is the same thing as a function: tests/test_gpu_record.py drives it at full length and checks the catalogue
(sampled steps against the oracle, one year re-run alone, tracking labels across a year boundary).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ("streamers", "overturnings", "cutoffs")
LEVELS = [-2.0, 2.0]
T0 = np.datetime64("1980-01-01T00", "ns")
SIX_HOURS = np.timedelta64(6 * 3600 * 10 ** 9, "ns")


class Record:
    """what ``run_record`` hands back next to the JSON line: the pipeline (open until ``close``), the grid, and per index the
    whole catalogue -- absolute step, lattice rings (CSR), labels -- plus the raw event tables of the years asked for"""

    def __init__(self):
        self.catalogue = {}
        self.tables = {}          # year -> [streamers, overturnings, cutoffs] event tables (Batch.events dictionaries)
        self.pipe = None
        self.lat = self.lon = None
        self.nt = 0

    def field_of_step(self, step, which="pv"):
        """one time step of the synthetic record on the host (regenerated on the device, copied back): (1, nlat, nlon) f32"""
        from wavebreaking_amd import synthetic as synth
        ctx = self.pipe.contexts[0]
        nbytes = self.lat.size * self.lon.size * 4
        d = ctx.alloc(nbytes)
        seed = synth.SEED if which == "pv" else 20241
        ctx.synth_fill(d, np.float32, int(step), 1, self.lat, self.lon, synth.wave_params(seed), 0.0, seed)
        out = ctx.to_host(d, (1, self.lat.size, self.lon.size), np.float32)
        ctx.free(d)
        return out

    def close(self):
        if self.pipe is not None:
            self.pipe.close()
            self.pipe = None


def _year_worker(pipe, bufs, lat, lon, nt, dlon, gen, keep_tables, want_geometry=True):
    from wavebreaking_amd import synthetic as synth
    par, pari = synth.wave_params(), synth.wave_params(20241)
    nlat, nlon = lat.size, lon.size

    def contours(c, k):
        d_pv, d_int = bufs[k % pipe.depth]
        w = time.perf_counter()
        c.synth_fill(d_pv, np.float32, k * nt, nt, lat, lon, par, 0.0, synth.SEED)
        c.synth_fill(d_int, np.float32, k * nt, nt, lat, lon, pari, 0.0, 20241)
        gen[k % pipe.depth] += time.perf_counter() - w        # (synth_fill returns when the field is there)
        return c.contours(d_pv, lat, lon, LEVELS, dlon, dlon, intensity=d_int, device_shape=(nt, nlat, nlon),
                          device_dtype=np.float32, want_cells=False)

    def indices(c, k, b, g):
        b.set_step_offset(k * nt)
        b.indices(g)
        out = []
        status = int(np.bitwise_or.reduce(b.status())) if nt else 0
        for kind in (0, 1, 2):
            ev = b.events(kind, copy=True)
            part = dict(step=ev["step"].astype(np.int64) + k * nt, sums=ev["sums"], status=status, band=b.band_hits())
            if want_geometry:
                ring_off, vert_off, gx, gy = b.event_geometry(kind)
                part.update(ring_off=ring_off.copy(), vert_off=vert_off.copy(), gx=gx.copy(), gy=gy.copy())
            if k in keep_tables:
                part["table"] = ev
            out.append(part)
        b.close()
        return out

    return contours, indices


def run_record(years=40, nt=1460, dlon=0.25, overlap=0.0, keep_tables=(), track=True, library=None, only_years=None, pipe=None):
    """The record, year by year, two batches in flight -> (JSON-able summary, Record).  ``keep_tables``: years whose raw event
    tables are kept in ``Record.tables``; ``only_years``: run just these years (each alone in its batch, same step
    offsets) instead of 0 .. years - 1; ``library``: a ``_hip.Library`` (the tests' host-simulation build), default the
    product library; ``pipe``: a ``BatchPipeline`` to reuse (its contexts keep their device scratch)."""
    from wavebreaking_amd import synthetic as synth
    from wavebreaking_amd.pipeline import BatchPipeline
    rec = Record()
    pipe = rec.pipe = pipe if pipe is not None else BatchPipeline(0, depth=2, library=library)
    lat, lon = synth.grid(dlon)
    rec.lat, rec.lon, rec.nt = lat, lon, nt
    nlat, nlon = lat.size, lon.size
    fbytes = nt * nlat * nlon * 4
    # every context owns the field + intensity buffers of the year it works on (filled on its own stream)
    bufs = [(c.alloc(fbytes), c.alloc(fbytes)) for c in pipe.contexts]
    width = nlon + int(120.0 / dlon)
    gen = [0.0] * pipe.depth
    which = list(range(years)) if only_years is None else list(only_years)
    contours, indices = _year_worker(pipe, bufs, lat, lon, nt, dlon, gen, set(keep_tables), want_geometry=True)
    done = []
    tic = time.perf_counter()
    # batch number j of the pipeline is year which[j]
    pipe.run(len(which), lambda c, j: contours(c, which[j]), lambda m: m, lambda c, j, b, g: indices(c, which[j], b, g),
             lambda j, r: done.append((which[j], r)), full_width=width)
    t_indices = time.perf_counter() - tic
    for c, (a, b) in zip(pipe.contexts, bufs):
        c.free(a)
        c.free(b)

    # the catalogue of each index: dates, lattice rings (every vertex of transform_polygons is a grid point) -> labels
    ctx = pipe.contexts[0]
    res = {}
    t_track = 0.0
    for kind in (0, 1, 2):
        parts = [r[kind] for _, r in done]
        step = np.concatenate([p["step"] for p in parts]) if parts else np.zeros(0, np.int64)
        nring = np.cumsum([0] + [p["vert_off"].size - 1 for p in parts])
        nvert = np.cumsum([0] + [p["gx"].size for p in parts])
        ring_off = np.concatenate([p["ring_off"][:-1] + nring[i] for i, p in enumerate(parts)] + [nring[-1:]])
        vert_off = np.concatenate([p["vert_off"][:-1] + nvert[i] for i, p in enumerate(parts)] + [nvert[-1:]])
        gx = np.concatenate([p["gx"] for p in parts])
        gy = np.concatenate([p["gy"] for p in parts])
        ix = np.rint((gx - lon[0]) / dlon).astype(np.int32)
        iy = np.rint((gy - lat[0]) / dlon).astype(np.int32)
        dates = T0 + step * SIX_HOURS
        cat = dict(step=step, dates=dates, ring_off=ring_off, vert_off=vert_off, ix=ix, iy=iy,
                   sums=np.concatenate([p["sums"] for p in parts]) if parts else np.zeros((0, 5)))
        entry = dict(events=int(step.size), polygon_vertices=int(gx.size))
        if track and step.size:
            tic = time.perf_counter()
            label, area, stats = ctx.track_events(dates, "by_overlap", 6.0, overlap=overlap, rings_csr=(ring_off, vert_off, ix, iy))
            t_track += time.perf_counter() - tic
            cat["label"] = label
            entry.update(tracks=int(label.max()) + 1, track_ms_device=ctx.last_op_ms(), **stats)
        rec.catalogue[NAMES[kind]] = cat
        res[NAMES[kind]] = entry
    for year, r in done:
        if "table" in r[0]:
            rec.tables[year] = [r[k]["table"] for k in (0, 1, 2)]
    nsteps = len(which) * nt
    band = np.sum([r[0]["band"] for _, r in done], axis=0) if done else np.zeros(2)
    out = {"workload": "0.25deg global PV, %d years of six-hourly steps (%d steps), levels [-2, 2], all three indices + "
                       "track_events(by_overlap), one MI355X (BASELINE.json configs[3]/[4] at full length)" % (len(which), nsteps),
           "timesteps": nsteps, "indices_s": t_indices, "indices_timesteps_per_s": nsteps / t_indices,
           "indices_includes": "generation of every year's two synthetic fields on the device: %.2f s of worker time (two workers), "
                               "and the lon / lat polygons of every event (transform_polygons) + catalogue copies on the host" % sum(gen),
           "track_s": t_track, "catalogue": res, "band_hits": [int(band[0]), int(band[1])],
           "status_or": int(np.bitwise_or.reduce([p["status"] for _, r in done for p in r])) if done else 0}
    return out, rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--years", type=int, default=40)
    ap.add_argument("--nt", type=int, default=1460, help="time steps per year (batch)")
    ap.add_argument("--dlon", type=float, default=0.25)
    ap.add_argument("--overlap", type=float, default=0.0)
    a = ap.parse_args()
    out, rec = run_record(a.years, a.nt, a.dlon, a.overlap)
    print(json.dumps(out), flush=True)
    rec.close()


if __name__ == "__main__":
    main()
