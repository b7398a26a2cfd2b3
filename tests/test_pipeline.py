"""BatchPipeline: several batches in flight (one context + one worker thread each), exchanges on the caller's thread in
batch order.  CPU: host-simulation library; results must equal the one-after-the-other results."""
import numpy as np

from oracle import synth
from tests import helpers
from wavebreaking_amd.pipeline import BatchPipeline


def test_pipeline_matches_sequential_and_keeps_order():
    lib = helpers.hostsim_library()
    lat, lon = synth.grid(2.5)
    fields = [synth.pv_field(3 * k, 3, lat, lon, eps=0.2) for k in range(5)]
    order = []

    def contours(ctx, k):
        return ctx.contours(fields[k], lat, lon, [-2.0, 2.0], 2.5, 2.5)

    def reduce_max(m):
        order.append(("max", m))
        return m

    def indices(ctx, k, b, g):
        b.indices(g)
        ev = [b.events(i) for i in (0, 1, 2)]
        b.close()
        return ev

    def collect(k, ev):
        order.append(("collect", k))
        return ev

    want = None
    for depth in (1, 2, 3):
        del order[:]
        pipe = BatchPipeline(0, depth=depth, library=lib)
        got = pipe.run(len(fields), contours, reduce_max, indices, collect)
        pipe.close()
        assert [o[1] for o in order if o[0] == "collect"] == list(range(len(fields)))
        if want is None:
            want = got
            assert sum(ev[0]["step"].size for ev in got) > 0
            continue
        for a, b in zip(got, want):
            for x, y in zip(a, b):
                for key in ("step", "level_idx", "ind1", "ind2", "box", "orientation", "sums", "poly_x", "poly_y"):
                    np.testing.assert_array_equal(x[key], y[key])
