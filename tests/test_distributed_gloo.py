"""N > 1 path on CPU: two processes (gloo), each running its time shard through the host-simulation
library, one all-reduce(max) + one gather; rank 0's concatenated catalogue must equal the single-process
catalogue of the whole record."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import synth
from tests import helpers
from wavebreaking_amd import _hip
from wavebreaking_amd import distributed as wd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import pytest
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from oracle import synth
from tests import helpers
from wavebreaking_amd import _hip
from wavebreaking_amd import distributed as wd

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
NT = 5
lat, lon = synth.grid(2.5)
first, count = wd.shard(NT, rank, world)
pv = synth.pv_field(first, count, lat, lon, eps=0.2)
if rank == 1:
    pv[:, 20:30, :] = 0.0          # rank 1's steps lose their southern full-span contour ...
ctx = _hip.Context(0, helpers.hostsim_library())
b = ctx.contours(pv, lat, lon, [-2.0, 2.0], 2.5, 2.5)
g = wd.global_max_nux(b.max_nux(), dist)
b.indices(g)
tabs = [wd.pack_events(b.events(k), step_offset=first) for k in (0, 1, 2)]
out = wd.gather_event_tables(tabs, dist, dst=0)
if rank == 0:
    np.savez(sys.argv[2], s=out[0], o=out[1], c=out[2], g=g)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_time_sharding_matches_single_process(tmp_path):
    helpers.build_hostsim()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "rank0.npz"
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(out)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)

    # single process over the whole record (same modification of the second shard)
    NT = 5
    lat, lon = synth.grid(2.5)
    pv = synth.pv_field(0, NT, lat, lon, eps=0.2)
    first1, _ = wd.shard(NT, 1, 2)
    pv[first1:, 20:30, :] = 0.0
    ctx = _hip.Context(0, helpers.hostsim_library())
    b = ctx.contours(pv, lat, lon, [-2.0, 2.0], 2.5, 2.5)
    assert int(got["g"]) == b.max_nux()
    b.indices()
    for key, kind in (("s", 0), ("o", 1), ("c", 2)):
        ref = wd.pack_events(b.events(kind))
        m = got[key]
        assert m.shape == ref.shape and m.shape[0] > 0
        cols = [c for c in range(18) if c != 2]           # contour row numbers are shard-local
        assert np.array_equal(m[:, cols], ref[:, cols])
    # round trip of the packing
    e = b.events(0)
    u = wd.unpack_events(wd.pack_events(e))
    for k in ("step", "ind1", "ind2", "box", "sums", "ncells"):
        assert np.array_equal(u[k], e[k])


def test_shard_covers_everything():
    for nt in (1, 7, 1460, 58440):
        for world in (1, 2, 3, 8):
            blocks = [wd.shard(nt, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == nt
            assert all(blocks[i][0] + blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))


WORKER_FRAMES = r'''
import os, sys, pickle
import numpy as np
import pytest
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
import wavebreaking_amd as wb
from oracle import synth
from tests import helpers
from wavebreaking_amd import _hip, engine
from wavebreaking_amd import distributed as wd

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
engine._contexts[0] = _hip.Context(0, helpers.hostsim_library())
NT = 5
lat, lon = synth.grid(2.5)
pv = synth.pv_field(0, NT, lat, lon, eps=0.2)
pv[wd.shard(NT, 1, 2)[0]:, 20:30, :] = 0.0          # the second shard has no southern full-span contour
data = wb.GridField(pv, ("time", "lat", "lon"), {"time": synth.times(NT), "lat": lat, "lon": lon}, name="PV")
out = {}
for name in ("calculate_contours", "calculate_streamers", "calculate_overturnings", "calculate_cutoffs"):
    kw = dict(intensity=data) if name != "calculate_contours" else dict(original_coordinates=False)
    out[name] = wd.calculate_sharded(getattr(wb, name), data, [-2, 2], dist=dist, **kw)
# a contour frame handed in (the whole record, on every rank): each rank keeps the rows of its own dates
frame = wb.calculate_contours(data, [-2, 2], original_coordinates=False)
out["streamers_from_frame"] = wd.calculate_sharded(wb.calculate_streamers, data, [-2, 2], dist=dist, contours=frame, intensity=data)
# a rank that fails still enters the collectives: nobody hangs, every rank raises
def failing(d, levels, **kw):
    if dist.get_rank() == 1:
        raise ValueError("boom")
    return wb.calculate_cutoffs(d, levels, **kw)
failing.__name__ = "calculate_cutoffs"
try:
    wd.calculate_sharded(failing, data, [-2, 2], dist=dist)
    raised = None
except Exception as e:
    raised = type(e).__name__ + ": " + str(e)
assert raised is not None and ("boom" in raised if dist.get_rank() == 1 else "another rank" in raised), raised
if dist.get_rank() == 0:
    pickle.dump(out, open(sys.argv[2], "wb"))
else:
    assert all(v is None for v in out.values())
dist.barrier()
dist.destroy_process_group()
'''


def test_sharded_calls_return_the_single_process_frames(tmp_path, monkeypatch):
    import pickle

    import wavebreaking_amd as wb
    from wavebreaking_amd import engine
    helpers.build_hostsim()
    script = tmp_path / "worker_frames.py"
    script.write_text(WORKER_FRAMES)
    out = tmp_path / "frames.pkl"
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(out)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = pickle.load(open(out, "rb"))

    ctx = _hip.Context(0, helpers.hostsim_library())
    monkeypatch.setitem(engine._contexts, 0, ctx)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    NT = 5
    lat, lon = synth.grid(2.5)
    pv = synth.pv_field(0, NT, lat, lon, eps=0.2)
    pv[wd.shard(NT, 1, 2)[0]:, 20:30, :] = 0.0
    data = wb.GridField(pv, ("time", "lat", "lon"), {"time": synth.times(NT), "lat": lat, "lon": lon}, name="PV")
    for name, frame in got.items():
        kw = dict(intensity=data) if name != "calculate_contours" else dict(original_coordinates=False)
        if name == "streamers_from_frame":
            cf = wb.calculate_contours(data, [-2, 2], original_coordinates=False)
            want = wb.calculate_streamers(data, [-2, 2], contours=cf, intensity=data)
        else:
            want = getattr(wb, name)(data, [-2, 2], **kw)
        assert frame.columns.to_list() == want.columns.to_list() and len(frame) == len(want) > 0, name
        for col in want.columns:
            if col == "geometry":
                for a, b in zip(frame[col], want[col]):
                    ra = [np.asarray(g.exterior.coords) for g in a.geoms] if hasattr(a, "geoms") else [np.asarray(a.coords)]
                    rb = [np.asarray(g.exterior.coords) for g in b.geoms] if hasattr(b, "geoms") else [np.asarray(b.coords)]
                    assert len(ra) == len(rb) and all(np.array_equal(x, y) for x, y in zip(ra, rb))
            else:
                assert frame[col].to_list() == want[col].to_list(), (name, col)
    ctx.close()


WORKER3 = r'''
import os, sys
import numpy as np
import pytest
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from oracle import synth
from tests import helpers
from wavebreaking_amd import _hip
from wavebreaking_amd import distributed as wd

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
NT = 2                                  # three ranks, two time steps: the last rank owns NO step
lat, lon = synth.grid(2.5)
first, count = wd.shard(NT, rank, world)
b = None
local_max = 0
if count:
    pv = synth.pv_field(first, count, lat, lon, eps=0.2)
    ctx = _hip.Context(0, helpers.hostsim_library())
    ctx.set_option("keep_device", 1)
    b = ctx.contours(pv, lat, lon, [-2.0, 2.0], 2.5, 2.5, intensity=pv)
    local_max = b.max_nux()
g = wd.global_max_nux(local_max, dist)
if b is not None:
    b.set_step_offset(first)
    b.indices(g)
out = wd.gather_device_tables(b, dist, dst=0)
if rank == 0:
    np.savez(sys.argv[2], g=g, **{"%s_%d" % (k, i): v for i, ev in enumerate(out) for k, v in ev.items()})
dist.barrier()
dist.destroy_process_group()
'''


def test_three_ranks_one_without_steps_gather_complete_catalogue(tmp_path):
    """exchange (2) as bench.py runs it for N > 1 (wavebreaking_amd.distributed.gather_device_tables): packed rows AND polygon
    vertices reach rank 0; a rank whose shard is empty takes part in both collectives"""
    helpers.build_hostsim()
    script = tmp_path / "worker3.py"
    script.write_text(WORKER3)
    out = tmp_path / "rank0.npz"
    port = _free_port()
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(out)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    lat, lon = synth.grid(2.5)
    pv = synth.pv_field(0, 2, lat, lon, eps=0.2)
    ctx = _hip.Context(0, helpers.hostsim_library())
    b = ctx.contours(pv, lat, lon, [-2.0, 2.0], 2.5, 2.5, intensity=pv)
    assert int(got["g"]) == b.max_nux()
    b.indices()
    total = 0
    for i in (0, 1, 2):
        ev = b.events(i)
        for k in ("step", "level_idx", "ind1", "ind2", "box", "orientation", "ncells", "flags", "poly_offset", "poly_x", "poly_y"):
            assert np.array_equal(got["%s_%d" % (k, i)], ev[k]), (i, k)
        assert np.array_equal(got["sums_%d" % i], ev["sums"])
        total += ev["step"].size
    assert total > 10
    b.close()


WORKER_RECORD = r'''
import os, sys
import numpy as np
import pytest
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from oracle import synth
from tests import helpers
from wavebreaking_amd import _hip
from wavebreaking_amd import distributed as wd
from wavebreaking_amd.pipeline import BatchPipeline, FullWidthNotReached

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
NT, PER = 8, 2                           # three ranks: two own four steps each = two batches of two, the third owns none
lat, lon = synth.grid(2.5)
full_w = lon.size + int(120 / 2.5)
first, count = (rank * 4, 4) if rank < 2 else (0, 0)
pv = synth.pv_field(first, max(count, 1), lat, lon, eps=0.2)
if rank == 1:
    pv[2:4] = 0.0                        # rank 1's second batch holds no contour at all: it does NOT reach the full width
pipe = BatchPipeline(0, depth=2, library=helpers.hostsim_library())
KEEP = int(sys.argv[3])                  # 1: rows + index polygons + rings travel; 2: rows + rings (what bench.py --gpus N gathers)
for c in pipe.contexts:
    c.set_option("keep_device", KEEP)
nb = 2

def phase1(c, k):
    return c.contours(pv[k * PER:(k + 1) * PER], lat, lon, [-2.0, 2.0], 2.5, 2.5, intensity=pv[k * PER:(k + 1) * PER])

def phase2(c, k, b, g):
    b.set_step_offset(first + k * PER)
    b.indices(g)
    return b

reduces = []
def reduce_max(m):
    reduces.append(m)
    return wd.global_max_nux(m, dist)

cats = {}
def gathered(k, cat):
    if cat is not None:
        cats[k] = tuple(np.array(a) for a in cat[:3]) + (tuple(np.array(a) for a in cat[3]),)       # (the buffers are reused by the next gather)
    return k

def record(field_ok):
    g = wd.TableGatherer(dist, comm=None, dst=0, on_result=gathered, geometry=True)
    done = []
    def collect(k, b):
        g.submit(k, b)
        pipe.close_batch(k, b)
    if count:
        pipe.run_record(nb, phase1, reduce_max, phase2, collect, full_w, before_reduce=lambda: done.extend(g.finish()))
    else:                                # a rank without time steps: the same collectives in the same order, nothing of its own
        for k in range(nb):
            g.submit(k, None)
        done.extend(g.finish())
        if reduce_max(0) != full_w:
            raise FullWidthNotReached("nothing")
    return done

done = record(True)
assert done == [0, 1] and len(reduces) == 1                    # ONE exchange of the maximum for the whole record
if rank == 0:
    rows = np.concatenate([cats[k][0] for k in (0, 1)])
    px = np.concatenate([cats[k][1] for k in (0, 1)])
    py = np.concatenate([cats[k][2] for k in (0, 1)])
    # the rings in lon / lat: batch after batch, offsets rebased
    ring_count = np.concatenate([np.diff(cats[k][3][0]) for k in (0, 1)])
    vert_count = np.concatenate([np.diff(cats[k][3][1]) for k in (0, 1)])
    gx = np.concatenate([cats[k][3][2] for k in (0, 1)])
    gy = np.concatenate([cats[k][3][3] for k in (0, 1)])
    np.savez(sys.argv[2], rows=rows, px=px, py=py, ring_count=ring_count, vert_count=vert_count, gx=gx, gy=gy)
# a record in which NO contour spans the extended grid: every rank learns so from the one exchange
pv[:, :, :12] = 0.0
try:
    record(False)
    raised = False
except FullWidthNotReached:
    raised = True
assert raised
pipe.close()
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("keep", [1, 2])
def test_record_with_one_deferred_maximum_and_gathers_off_the_critical_path(tmp_path, keep):
    """(keep = 2: the context option keep_device = 2 -- the rows travel without their polygons in grid indices, which stay
    descriptors on the ranks as they do on a single GPU; the rings in lon / lat are the catalogue's geometry.)
    what bench.py --gpus N runs: every batch selected against the full width at once (BatchPipeline.run_record), ONE
    all-reduce(max) per record that verifies it, every batch's tables staged and gathered by a thread of their own
    (distributed.TableGatherer) -- three ranks over gloo, one without time steps, one batch without any contour; rank 0's
    catalogue equals the single-process catalogue of the whole record"""
    helpers.build_hostsim()
    script = tmp_path / "worker_record.py"
    script.write_text(WORKER_RECORD)
    out = tmp_path / "rank0.npz"
    port = _free_port()
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(out), str(keep)], env=env))
    try:
        for p in procs:
            assert p.wait(timeout=240) == 0
    finally:
        for p in procs:                                   # (a rank stuck in a collective must not outlive the test)
            if p.poll() is None:
                p.kill()
    got = np.load(out)
    lat, lon = synth.grid(2.5)
    pv = synth.pv_field(0, 8, lat, lon, eps=0.2)
    pv[6:8] = 0.0
    ctx = _hip.Context(0, helpers.hostsim_library())
    b = ctx.contours(pv, lat, lon, [-2.0, 2.0], 2.5, 2.5, intensity=pv)
    assert b.max_nux() == lon.size + 48
    b.indices()
    # the gathered rows come batch by batch (rank order inside a batch): time order = a stable sort by step inside each index
    rows, px, py = got["rows"], got["px"], got["py"]
    npoly = rows[:, 18].astype(np.int64)
    start = np.cumsum(npoly) - npoly
    # ... and with them the rings in lon / lat, cut and mapped by every rank's own device (transform_polygons,
    # index_utils.py:129-184): row k of the gathered catalogue owns the next ring_count[k] rings
    ring_count, vert_count, ggx, ggy = got["ring_count"], got["vert_count"], got["gx"], got["gy"]
    assert ring_count.size == rows.shape[0] and vert_count.size == ring_count.sum() and ggx.size == vert_count.sum() == ggy.size
    ring_start = np.cumsum(ring_count) - ring_count
    vert_start = np.cumsum(vert_count) - vert_count
    nrings_checked = 0
    total = 0
    for kind in (0, 1, 2):
        ev = b.events(kind)
        m = np.nonzero(rows[:, 16] == kind)[0]
        m = m[np.argsort(rows[m, 0], kind="stable")]
        assert np.array_equal(rows[m, 0].astype(np.int32), ev["step"]), kind
        assert np.array_equal(rows[m, 1].astype(np.int32), ev["level_idx"])
        assert np.array_equal(rows[m, 10:15], ev["sums"])
        assert np.array_equal(rows[m, 5:9].astype(np.int32), ev["box"])
        if keep == 1:
            gx = np.concatenate([px[start[i]:start[i] + npoly[i]] for i in m]) if m.size else np.zeros(0, np.int32)
            gy = np.concatenate([py[start[i]:start[i] + npoly[i]] for i in m]) if m.size else np.zeros(0, np.int32)
            assert np.array_equal(gx, ev["poly_x"]) and np.array_equal(gy, ev["poly_y"])
        else:                                             # the rows still say how long each polygon is; no vertex travelled
            assert px.size == 0 and py.size == 0 and np.array_equal(rows[m, 18].astype(np.int64), np.diff(ev["poly_offset"]))
        ro, vo, wx, wy = b.event_geometry(kind)
        for q, i in enumerate(m):
            assert ring_count[i] == ro[q + 1] - ro[q], (kind, q)
            for j in range(int(ring_count[i])):
                a0 = vert_start[ring_start[i] + j]; a1 = a0 + vert_count[ring_start[i] + j]
                w0, w1 = vo[ro[q] + j], vo[ro[q] + j + 1]
                assert np.array_equal(ggx[a0:a1], wx[w0:w1]) and np.array_equal(ggy[a0:a1], wy[w0:w1]), (kind, q, j)
                nrings_checked += 1
        total += m.size
    assert total == rows.shape[0] > 20 and nrings_checked >= total
    b.close()
    ctx.close()


FAIL_WORKER = r'''
import os, sys, time
import numpy as np
import pytest
sys.path.insert(0, sys.argv[1])
import datetime
import torch.distributed as dist
from oracle import synth
from tests import helpers
from wavebreaking_amd import _hip
from wavebreaking_amd import distributed as wd
from wavebreaking_amd.pipeline import BatchPipeline

mode = sys.argv[2]
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]), timeout=datetime.timedelta(seconds=60))
rank, world = dist.get_rank(), dist.get_world_size()
lat, lon = synth.grid(2.5)
PER, NB = 1, 4
pv = synth.pv_field(rank * PER * NB, PER * NB, lat, lon, eps=0.2)
full_w = lon.size + int(120.0 / 2.5)
pipe = BatchPipeline(0, depth=2, library=helpers.hostsim_library())
for c in pipe.contexts:
    c.set_option("keep_device", 1)

def phase1(c, k):
    return c.contours(pv[k * PER:(k + 1) * PER], lat, lon, [-2.0, 2.0], 2.5, 2.5)

def phase2(c, k, b, g):
    if mode == "raise" and rank == 1 and k == 2:
        raise ValueError("rank 1 fails in batch 2")          # a worker raises: the record cannot be finished on this rank
    b.indices(g)
    return b

def on_result(k, cat):
    if mode == "poison" and rank == 1 and k == 1:
        raise ValueError("rank 1 fails behind the gather of batch 1")   # ... or the gather thread itself fails
    if mode == "kill" and rank == 1 and k == 1:
        os._exit(7)                                          # ... or the process dies mid-record
    return k

g = wd.TableGatherer(dist, comm=None, dst=0, on_result=on_result)
def collect(k, b):
    g.submit(k, b)
    pipe.close_batch(k, b)
t0 = time.time()
try:
    pipe.run_record(NB, phase1, lambda m: wd.global_max_nux(m, dist), phase2, collect, full_w, before_reduce=lambda: g.finish())
except BaseException as e:
    g.abort(e)
    print("rank", rank, "failed after %.1f s:" % (time.time() - t0), type(e).__name__, str(e)[:100], flush=True)
    os._exit(3)                                              # a failed gather group ends the process non-zero
print("rank", rank, "finished", flush=True)
os._exit(0)
'''


def _run_failing_record(tmp_path, mode):
    helpers.build_hostsim()
    script = tmp_path / "fail_worker.py"
    script.write_text(FAIL_WORKER)
    port = _free_port()
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   WB_PIPE_TIMEOUT="90")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            outs.append((p.wait(timeout=150), p.stdout.read()))
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append((None, "TIMEOUT " + p.stdout.read()))
    return outs


def test_a_rank_that_fails_behind_a_gather_poisons_the_later_slots_and_every_rank_raises(tmp_path):
    """rank 1's gather thread fails after batch 1: it still takes part in the exchange of the counts of batches 2 and 3, with -1
    for its own (GatherPoisoned on every rank, before any payload), so ranks 0 and 2 end with an error within seconds instead
    of waiting inside a collective for ever (until round 5 the failing rank drained its slots without collectives)"""
    outs = _run_failing_record(tmp_path, "poison")
    for rc, text in outs:
        assert rc == 3, outs
    assert "GatherPoisoned" in outs[0][1] and "GatherPoisoned" in outs[2][1], outs


def test_a_rank_whose_worker_raises_or_that_dies_mid_record_ends_the_others_non_zero(tmp_path):
    """the review's case: one of three ranks raises in a worker (its gatherer is aborted) or is killed mid-record; the others must
    exit non-zero within the time-out, not hang"""
    for mode in ("raise", "kill"):
        outs = _run_failing_record(tmp_path, mode)
        assert all(rc is not None and rc != 0 for rc, _ in outs), (mode, outs)
        if mode == "raise":
            # the rank that raised tells the others that it is leaving (one exchange of the counts with -2): they fail with
            # GatherAborted / GatherPoisoned within seconds, not at the process group's time-out (60 s here)
            import re
            for r in (0, 2):
                assert "Gather" in outs[r][1], outs
                assert float(re.search(r"failed after ([0-9.]+) s", outs[r][1]).group(1)) < 30.0, outs


WORKER_TRACK = r'''
import os, sys, pickle
import numpy as np
import pytest
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
import wavebreaking_amd as wb
from tests import helpers, test_tracking as tt
from wavebreaking_amd import _hip, engine
from wavebreaking_amd import distributed as wd

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
engine._contexts[0] = _hip.Context(0, helpers.hostsim_library())
out = {}
for case, (seed, n, steps, tr) in enumerate(((9, 60, 9, None), (21, 48, 7, 12.0), (33, 30, 12, None))):
    dates, polys, com = tt._tracking_case(seed, n, steps, shuffle=False)          # rows in time order, as calculate_* returns them
    lonlat = [[np.c_[-180 + 0.25 * r[:, 0], -90 + 0.25 * r[:, 1]] for r in rs] for rs in polys]
    # time-sharded: contiguous blocks of DATES (a date is never split between two ranks)
    ud = np.unique(dates)
    first, count = wd.shard(len(ud), rank, world)
    mine = np.isin(dates, ud[first:first + count])
    events = tt._frame(dates[mine], [p for p, m in zip(lonlat, mine) if m], com[mine])
    events.attrs["wavebreaking_grid"] = dict(lon=(-180.0, 0.25), lat=(-90.0, 0.25))
    for method, kw in (("by_overlap", dict(overlap=0.02)), ("by_distance", dict(distance=5000))):
        got = wd.track_events_sharded(events, dist, time_range=tr, method=method, **kw)
        if rank == 0:
            out[(case, method)] = (got["label"].tolist(), [str(d) for d in got["date"].values], [tuple(c) for c in got["com"]])
        else:
            assert got is None
if rank == 0:
    pickle.dump(out, open(sys.argv[2], "wb"))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_track_events_on_time_shards_equals_the_single_process_labels(tmp_path, monkeypatch, world):
    """track_events (events.py:113-241) on a time-sharded catalogue: every rank tracks its own block of dates plus a halo of the
    rows within time_range in front of it, the local labels are unified through the rows two ranks share -- the labels and the
    row order of the whole record are those of the single-process call, for both methods, with time_range given or taken from
    the smallest gap of the WHOLE record, with more dates than ranks and fewer rows than that"""
    import pickle

    import wavebreaking_amd as wb
    from tests import test_tracking as tt
    from wavebreaking_amd import engine
    helpers.build_hostsim()
    script = tmp_path / "worker_track.py"
    script.write_text(WORKER_TRACK)
    out = tmp_path / "track.pkl"
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(out)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = pickle.load(open(out, "rb"))
    ctx = _hip.Context(0, helpers.hostsim_library())
    monkeypatch.setitem(engine._contexts, 0, ctx)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    linked = 0
    for case, (seed, n, steps, tr) in enumerate(((9, 60, 9, None), (21, 48, 7, 12.0), (33, 30, 12, None))):
        dates, polys, com = tt._tracking_case(seed, n, steps, shuffle=False)
        lonlat = [[np.c_[-180 + 0.25 * r[:, 0], -90 + 0.25 * r[:, 1]] for r in rs] for rs in polys]
        events = tt._frame(dates, lonlat, com)
        events.attrs["wavebreaking_grid"] = dict(lon=(-180.0, 0.25), lat=(-90.0, 0.25))
        for method, kw in (("by_overlap", dict(overlap=0.02)), ("by_distance", dict(distance=5000))):
            want = wb.track_events(events, time_range=tr, method=method, **kw)
            labels, dts, coms = got[(case, method)]
            assert labels == want["label"].tolist(), (case, method)
            assert dts == [str(d) for d in want["date"].values] and coms == [tuple(c) for c in want["com"]], (case, method)
            linked += len(labels) - len(set(labels))
    assert linked > 20
    ctx.close()
