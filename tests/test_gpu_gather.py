"""The device-to-device gather of bench.py's N > 1 branch on real hardware: RCCL (backend "nccl") with a process group of
ONE rank on the one GPU of the test box -- the same code path as N ranks (size exchange, padded gather, pinned D2H), minus
the peers.  Runs in a child process (a process group per test process)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from oracle import synth
from wavebreaking_amd import _hip
from wavebreaking_amd import distributed as wd

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
lat, lon = synth.grid(1.0)
pv = synth.pv_field(0, 6, lat, lon, eps=0.2)
ctx = _hip.Context(0)
ctx.set_option("keep_device", 1)
b = ctx.contours(pv, lat, lon, [-2.0, 2.0], 1.0, 1.0, intensity=pv)
g = wd.global_max_nux(b.max_nux(), dist)
assert g == b.max_nux()
b.set_step_offset(100)
b.indices(g)
want = [b.events(k, copy=True) for k in (0, 1, 2)]
for rep in range(2):                                    # second call: the pinned buffers are reused
    rows, px, py = wd.gather_device_tables(b, dist, dst=0, as_rows=True)
    assert rows.shape == (sum(e["step"].size for e in want), 20) and rows.shape[0] > 0
    assert int(rows[:, 18].sum()) == px.size == py.size == sum(e["poly_x"].size for e in want)
    got = wd.rows_to_events(rows, px, py)
    for e, w in zip(got, want):
        assert np.array_equal(e["step"], w["step"] + 100) or np.array_equal(e["step"], w["step"])
        for key in ("level_idx", "ind1", "ind2", "box", "orientation", "ncells", "flags", "poly_offset", "poly_x", "poly_y"):
            assert np.array_equal(e[key], w[key]), key
        assert np.array_equal(e["sums"], w["sums"])
cat = wd.gather_device_tables(b, dist, dst=0)
assert [e["step"].size for e in cat] == [w["step"].size for w in want]
# the same gather INSIDE the library: wb_comm_init with the unique id carried by the process group, counts by ncclAllGather,
# payload by grouped send / recv (one rank: the device-to-device copy of its own part), pinned buffers reused
comm = wd.LibraryComm(ctx, dist)
assert comm.max_nux(b.max_nux()) == b.max_nux() and comm.max_nux(7) == 7
for rep in range(2):
    rows, px, py = comm.gather(b, dst=0)
    assert rows.shape == (sum(e["step"].size for e in want), 20) and int(rows[:, 18].sum()) == px.size == py.size
    got = wd.rows_to_events(rows, px, py)
    for e, w in zip(got, want):
        for key in ("level_idx", "ind1", "ind2", "box", "orientation", "ncells", "flags", "poly_offset", "poly_x", "poly_y"):
            assert np.array_equal(e[key], w[key]), key
        assert np.array_equal(e["sums"], w["sums"]) and np.array_equal(e["step"], w["step"] + 100)
rows, px, py = comm.gather(None, dst=0)                # a rank without time steps
assert rows.shape == (0, 20) and px.size == 0
# off the critical path (what bench.py --gpus N does): stage the tables, release the batch, gather the slots on a thread
seen = {}
def landed(k, cat):
    rows, px, py = cat
    got = wd.rows_to_events(np.array(rows), np.array(px), np.array(py))
    for e, w in zip(got, want if rows.shape[0] else []):
        for key in ("level_idx", "ind1", "ind2", "box", "orientation", "ncells", "flags", "poly_offset", "poly_x", "poly_y"):
            assert np.array_equal(e[key], w[key]), key
        assert np.array_equal(e["sums"], w["sums"]) and np.array_equal(e["step"], w["step"] + 100)
    seen[k] = rows.shape[0]
    return k
for use_comm in (comm, None):                            # RCCL inside the library, then torch.distributed
    seen.clear()
    g = wd.TableGatherer(dist, comm=use_comm, dst=0, on_result=landed)
    for k in range(6):                                   # more batches than slots: submit waits for the thread
        g.submit(k, b if k != 3 else None)
    assert g.finish() == list(range(6))
    assert seen[3] == 0 and all(seen[k] == sum(e["step"].size for e in want) for k in (0, 1, 2, 4, 5))
comm.close()
b.close()
dist.barrier()
dist.destroy_process_group()
print("GATHER_OK")
'''


@pytest.mark.gpu
def test_device_gather_over_rccl_single_rank(tmp_path):
    script = tmp_path / "worker_gather.py"
    script.write_text(WORKER)
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool only supports dmabuf IPC; with the legacy mode RCCL's
    # peer-to-peer set-up fails with `hipIpcGetMemHandle: invalid argument` (the image exports it; bench.py sets it too)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "GATHER_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
