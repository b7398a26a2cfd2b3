"""Time-step sharding across the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The path partitions by time step: every step's contours and events depend on that step's field only.
Exactly two exchanges exist: (1) all-reduce(max) of one int32 -- contour selection uses
``contours.exp_lon.max()`` over the WHOLE frame (streamer_index.py:106, overturning_index.py:105,
cutoff_index.py:90); (2) one gather of the event tables to rank 0 (counts first, then the payload),
concatenated in time order.
"""
import os
import sys

import numpy as np

EVENT_FIELDS = ("step", "level_idx", "contour", "ind1", "ind2", "box", "orientation", "sums", "ncells")


def shard(nt_total, rank, world):
    """Contiguous block of time steps owned by ``rank``: (first, count)."""
    base, rem = divmod(int(nt_total), int(world))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def _device(dist):
    """tensor device of the collectives: with RCCL the GPU of this rank's context -- LOCAL_RANK, the rule of
    engine.get_context -- whatever torch's current device happens to be"""
    import torch
    if dist.get_backend() != "nccl":
        return torch.device("cpu")
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", torch.cuda.current_device())))


_streams = {}


class _exchange_stream:
    """``with _exchange_stream(dev):`` -- the tensor work of an exchange runs on a stream of its own.  torch's default
    stream is the device's NULL stream, which is ordered against every blocking stream of the process: a 4-byte upload on
    it waits for all kernels the library's contexts have queued (a whole pass, ~8 ms measured) and would serialise the
    batches in flight.  torch's non-default streams are non-blocking.  Leaving the block waits for the stream: buffers
    of the batch may be freed, results may be read."""

    def __init__(self, dev):
        self.dev = dev
        self.cm = None

    def __enter__(self):
        if self.dev.type == "cuda":
            import torch
            key = (self.dev.index, __import__("threading").get_ident())
            if key not in _streams:
                _streams[key] = torch.cuda.Stream(device=self.dev)
            self.stream = _streams[key]
            self.cm = torch.cuda.stream(self.stream)
            self.cm.__enter__()
        return self

    def __exit__(self, *exc):
        if self.cm is not None:
            self.stream.synchronize()
            self.cm.__exit__(*exc)
        return False


def global_max_nux(local_max, dist=None):
    """exchange (1)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(local_max)
    import torch
    dev = _device(dist)
    with _exchange_stream(dev):
        t = torch.tensor([int(local_max)], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())


def pack_events(ev, step_offset=0, contour_offset=0):
    """One event table -> (n, 18) float64 matrix (all integer columns are exactly representable)."""
    n = ev["step"].size
    m = np.zeros((n, 18), dtype=np.float64)
    m[:, 0] = ev["step"] + step_offset
    m[:, 1] = ev["level_idx"]
    m[:, 2] = ev["contour"] + contour_offset
    m[:, 3] = ev["ind1"]
    m[:, 4] = ev["ind2"]
    m[:, 5:9] = ev["box"]
    m[:, 9] = ev["orientation"]
    m[:, 10:15] = ev["sums"]
    m[:, 15] = ev["ncells"]
    return m


def unpack_events(m):
    m = np.asarray(m, dtype=np.float64).reshape(-1, 18)
    i32 = lambda c: m[:, c].astype(np.int32)
    return dict(step=i32(0), level_idx=i32(1), contour=m[:, 2].astype(np.int64), ind1=i32(3), ind2=i32(4),
                box=m[:, 5:9].astype(np.int32), orientation=i32(9), sums=m[:, 10:15].copy(), ncells=i32(15))


def gather_event_tables(tables, dist=None, dst=0):
    """exchange (2): ``tables`` = list of packed (n_k, 18) matrices (one per index).  Returns on ``dst`` a list
    of per-index matrices concatenated in rank (= time) order, None elsewhere."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [np.asarray(t) for t in tables]
    import torch
    world, rank, dev = dist.get_world_size(), dist.get_rank(), _device(dist)
    width = 18
    with _exchange_stream(dev):
        counts = torch.tensor([t.shape[0] for t in tables], dtype=torch.int64, device=dev)
        all_counts = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(all_counts, counts)
        all_counts = torch.stack(all_counts).cpu().numpy()                 # (world, n_tables)
        rows = all_counts.sum(axis=1)
        mx = int(rows.max())
        buf = torch.zeros((mx, width), dtype=torch.float64, device=dev)
        mine = np.concatenate([np.asarray(t, dtype=np.float64).reshape(-1, width) for t in tables], axis=0)
        if mine.shape[0]:
            buf[: mine.shape[0]] = torch.from_numpy(mine).to(dev)
        out = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, out, dst=dst)
        if rank != dst:
            return None
        blocks = [out[r].cpu().numpy() for r in range(world)]
    res = [[] for _ in tables]
    for r in range(world):
        o = 0
        for k in range(len(tables)):
            c = int(all_counts[r, k])
            res[k].append(blocks[r][o:o + c])
            o += c
    return [np.concatenate(v, axis=0) if v else np.zeros((0, width)) for v in res]


# ------------------------------------------------------------------------------------------------ device-resident gather
class _DeviceBuffer:
    """a device allocation of the C ABI as a CUDA-array-interface object (zero-copy torch.as_tensor)"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _wrap(ptr, shape, np_dtype, on_gpu):
    """torch tensor over a buffer of the library: device memory of the HIP build (no copy), host memory of the
    host-simulation build used by the CPU tests"""
    import ctypes
    import torch
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return torch.zeros(tuple(shape), dtype=torch.from_numpy(np.zeros(0, np_dtype)).dtype, device="cuda" if on_gpu else "cpu")
    if on_gpu:
        return torch.as_tensor(_DeviceBuffer(ptr, shape, np.dtype(np_dtype).str), device="cuda")
    buf = (ctypes.c_char * (n * np.dtype(np_dtype).itemsize)).from_address(int(ptr))
    return torch.from_numpy(np.frombuffer(buf, dtype=np_dtype).reshape(shape).copy())


ROW_FIELDS = dict(step=0, level_idx=1, contour=2, ind1=3, ind2=4, orientation=9, ncells=15, kind=16, flags=17, npoly=18)


def rows_to_events(rows, poly_x, poly_y):
    """packed device rows (n, 20) + the vertices of the same events -> three event dictionaries with the keys of
    Batch.events() (a formatting step for consumers that want the per-index tables; no per-event Python).  One column
    differs in meaning: ``contour`` is the contour's number INSIDE its (step, level) job, as the rows carry it -- a row of
    the batch's contour table (what Batch.events() gives) means nothing once the shards of several ranks are concatenated."""
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, 20)
    poly_x, poly_y = np.asarray(poly_x), np.asarray(poly_y)
    npoly = rows[:, 18].astype(np.int64)
    if poly_x.size == 0:                                               # (context option keep_device = 2: the rows travelled without
        npoly = np.zeros_like(npoly)                                   #  their index polygons -- the rings in lon / lat are the geometry)
    start = np.cumsum(npoly) - npoly
    out = []
    for kind in (0, 1, 2):
        m = rows[:, 16] == kind
        r = rows[m]
        lens, first = npoly[m], start[m]
        own = np.cumsum(lens) - lens                                   # offsets inside this kind's vertex arrays
        vidx = np.repeat(first - own, lens) + np.arange(int(lens.sum()), dtype=np.int64)
        i32 = lambda c: r[:, c].astype(np.int32)
        out.append(dict(step=i32(0), level_idx=i32(1), contour=r[:, 2].astype(np.int64), ind1=i32(3), ind2=i32(4),
                        box=r[:, 5:9].astype(np.int32), orientation=i32(9), sums=r[:, 10:15].copy(), ncells=i32(15), flags=i32(17),
                        poly_offset=np.concatenate([[0], np.cumsum(lens)]).astype(np.int64),
                        poly_x=poly_x[vidx].astype(np.int32), poly_y=poly_y[vidx].astype(np.int32)))
    return out


_pinned = {}


def _pinned_host(name, shape, dtype):
    """a pinned host tensor of at least `shape` (first axis may be larger), kept for the next gather"""
    import torch
    t = _pinned.get(name)
    if t is None or t.shape[0] < shape[0] or tuple(t.shape[1:]) != tuple(shape[1:]):
        grown = (int(shape[0] * 1.25) + 1,) + tuple(shape[1:])
        t = torch.empty(grown, dtype=dtype, pin_memory=True)
        _pinned[name] = t
    return t


def _batch_tensors(batch, dist, single):
    """(rows (ne, 20), vx, vy) of a batch's device tables as torch tensors -- zero-copy views of the library's buffers"""
    if batch is None:
        ptr, n3, px, py, nv, on_gpu = 0, (0, 0, 0), 0, 0, 0, (not single and dist.get_backend() == "nccl")
    else:
        ptr, n3, px, py, nv = batch.device_tables()
        on_gpu = batch.ctx.L.backend == "hip"
    ne = sum(n3)
    return _wrap(ptr, (ne, 20), np.float64, on_gpu), _wrap(px, (nv,), np.int32, on_gpu), _wrap(py, (nv,), np.int32, on_gpu)


def gather_device_tables(batch, dist=None, dst=0, as_rows=False):
    """exchange (2) without host staging: the packed event rows and polygon vertices that ``Batch.indices`` left on the
    device (context option "keep_device") travel device-to-device in one gather each (sizes first); ``dst`` copies every
    rank's part to the host ONCE (into pinned buffers kept between calls) -- the complete catalogue including geometry,
    in rank (= time) order; the other ranks return None.  ``batch`` may be None on a rank without time steps.  When the
    call returns the batch's device buffers are no longer needed (the batch may be closed).

    Returns [streamers, overturnings, cutoffs] (the dictionaries of ``Batch.events``), or with ``as_rows`` the catalogue
    as it travelled: ``(rows (n, 20) float64, poly_x, poly_y)`` -- row k owns the next ``rows[k, 18]`` vertices, its index
    is ``rows[k, 16]`` (``rows_to_events`` converts; the arrays are views of the pinned buffers, valid until the next call)."""
    single = dist is None or not dist.is_initialized()      # an initialised group of ONE rank still takes the collective path
    done = (lambda r, x, y: (r, x, y)) if as_rows else rows_to_events
    if single:
        rows, vx, vy = _batch_tensors(batch, dist, True)
        return done(rows.cpu().numpy(), vx.cpu().numpy(), vy.cpu().numpy())
    dev = _device(dist)
    with _exchange_stream(dev):
        rows, vx, vy = _batch_tensors(batch, dist, False)
        out = _gather_tensors(rows, vx, vy, dist, dst, dev)
    return None if out is None else done(*out)


class GatherPoisoned(RuntimeError):
    """some rank took part in a gather's exchange of the counts with -1: it cannot deliver that slot (an error upstream);
    every rank's call for the slot raises this BEFORE the payload is exchanged, so no rank waits for data that never comes"""


class GatherAborted(GatherPoisoned):
    """some rank took part in the exchange of the counts with -2: it is LEAVING the group (its record failed) and issues no
    collective after this one.  Every rank sees that in the same exchange and stops issuing collectives as well, so nobody
    waits for a partner that is gone"""


def _gather_tensors(rows, vx, vy, dist, dst, dev, poison=False, leaving=False):
    """the collective part of gather_device_tables on tensors that already exist (inside an _exchange_stream block);
    (rows, vx, vy) numpy arrays on ``dst``, None elsewhere.  ``poison``: this rank has nothing to deliver for the slot
    (rows / vx / vy may be None): it sends -1 for its counts, and every rank raises GatherPoisoned.  ``leaving`` (with
    ``poison``): -2, the rank's last collective -- every rank raises GatherAborted."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    ne, nv = ((-2, -2) if leaving else (-1, -1)) if poison else (int(rows.shape[0]), int(vx.shape[0]))
    if not poison and dev.type != "cuda":               # dry runs of the N > 1 logic over gloo: CPU tensors
        rows, vx, vy = rows.cpu(), vx.cpu(), vy.cpu()
    counts = torch.tensor([ne, nv], dtype=torch.int64, device=dev)
    allc = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(allc, counts)
    allc = torch.stack(allc).cpu().numpy()
    if (allc == -2).any():
        raise GatherAborted("rank(s) %s left the gather group" % [int(r) for r in np.nonzero((allc == -2).any(axis=1))[0]])
    if (allc < 0).any():
        raise GatherPoisoned("rank(s) %s reported a failure for this slot of the gather" % [int(r) for r in np.nonzero((allc < 0).any(axis=1))[0]])
    me, mv = int(allc[:, 0].max()), int(allc[:, 1].max())
    rbuf = torch.zeros((max(me, 1), 20), dtype=torch.float64, device=dev)
    vbuf = torch.zeros((2, max(mv, 1)), dtype=torch.int32, device=dev)
    if ne:
        rbuf[:ne] = rows
    if nv:
        vbuf[0, :nv] = vx
        vbuf[1, :nv] = vy
    rout = [torch.zeros_like(rbuf) for _ in range(world)] if rank == dst else None
    vout = [torch.zeros_like(vbuf) for _ in range(world)] if rank == dst else None
    dist.gather(rbuf, rout, dst=dst)
    dist.gather(vbuf, vout, dst=dst)
    if rank != dst:
        return None                                     # (leaving the block waits for the staging copies and the sends)
    tot_e, tot_v = int(allc[:, 0].sum()), int(allc[:, 1].sum())
    if dev.type != "cuda":
        rcat = torch.cat([rout[r][: int(allc[r, 0])] for r in range(world)]).numpy()
        vcat = torch.cat([vout[r][:, : int(allc[r, 1])] for r in range(world)], dim=1).numpy()
        return rcat, vcat[0], vcat[1]
    # the one copy to the host: every rank's valid part straight into pinned memory
    hr = _pinned_host("rows", (max(tot_e, 1), 20), torch.float64)
    hx = _pinned_host("vx", (max(tot_v, 1),), torch.int32)
    hy = _pinned_host("vy", (max(tot_v, 1),), torch.int32)
    oe = ov = 0
    for r in range(world):
        k, v = int(allc[r, 0]), int(allc[r, 1])
        if k:
            hr[oe:oe + k].copy_(rout[r][:k], non_blocking=True)
        if v:
            hx[ov:ov + v].copy_(vout[r][0, :v], non_blocking=True)
            hy[ov:ov + v].copy_(vout[r][1, :v], non_blocking=True)
        oe += k
        ov += v
    torch.cuda.current_stream().synchronize()
    # rows are (streamers, overturnings, cutoffs) per rank, rank order = time order
    return hr[:tot_e].numpy(), hx[:tot_v].numpy(), hy[:tot_v].numpy()


def _geometry_tensors(batch, dist):
    """the rings of a batch's last indices() as torch tensors over the arrays the kernels wrote (device memory of the HIP build
    with the options keep_device / geometry_dma, no copy): (rings per event int64[ne], vertices per ring int64[nr], x float64[nv],
    y float64[nv]); a rank without a batch contributes empty ones"""
    import torch
    if batch is None:
        on_gpu = dist.get_backend() == "nccl"
        z = lambda dt: torch.zeros(0, dtype=dt, device="cuda" if on_gpu else "cpu")
        return z(torch.int64), z(torch.int64), z(torch.float64), z(torch.float64)
    ro, ne, vo, nr, x, y, nv, on_dev = batch.device_geometry()
    gpu = on_dev and batch.ctx.L.backend == "hip"
    if dist.get_backend() == "nccl" and not gpu:
        # (pinned host arrays -- the options were off: up to the device for the collective)
        dev = _device(dist)
        t = lambda p, n, dt: _wrap(p, (n,), dt, False).to(dev)
    else:
        t = lambda p, n, dt: _wrap(p, (n,), dt, gpu)
    ring_off, vert_off = t(ro, ne + 1, np.int64), t(vo, nr + 1, np.int64)
    return ring_off[1:] - ring_off[:-1], vert_off[1:] - vert_off[:-1], t(x, nv, np.float64), t(y, nv, np.float64)


def _gather_geometry(parts, dist, dst, dev):
    """the collective behind the geometry of a gathered catalogue: every rank's (rings per event, vertices per ring, x, y) to
    ``dst`` in rank (= time) order -- counts first, the payload padded to the largest rank's.  On ``dst``: (ring_offset
    int64[events + 1], vert_offset int64[rings + 1], x, y) of the whole catalogue, numpy; None elsewhere."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    rc, vc, x, y = parts
    if dev.type != "cuda":
        rc, vc, x, y = rc.cpu(), vc.cpu(), x.cpu(), y.cpu()
    counts = torch.tensor([int(rc.shape[0]), int(vc.shape[0]), int(x.shape[0])], dtype=torch.int64, device=dev)
    allc = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(allc, counts)
    allc = torch.stack(allc).cpu().numpy()
    me, mr, mv = (max(int(allc[:, k].max()), 1) for k in range(3))
    ibuf = torch.zeros(me + mr, dtype=torch.int64, device=dev)
    ibuf[: rc.shape[0]] = rc
    ibuf[me: me + vc.shape[0]] = vc
    fbuf = torch.zeros((2, mv), dtype=torch.float64, device=dev)
    fbuf[0, : x.shape[0]] = x
    fbuf[1, : y.shape[0]] = y
    iout = [torch.zeros_like(ibuf) for _ in range(world)] if rank == dst else None
    fout = [torch.zeros_like(fbuf) for _ in range(world)] if rank == dst else None
    dist.gather(ibuf, iout, dst=dst)
    dist.gather(fbuf, fout, dst=dst)
    if rank != dst:
        return None
    if dev.type != "cuda":
        rcs = torch.cat([iout[r][: int(allc[r, 0])] for r in range(world)]).numpy()
        vcs = torch.cat([iout[r][me: me + int(allc[r, 1])] for r in range(world)]).numpy()
        xs = torch.cat([fout[r][0, : int(allc[r, 2])] for r in range(world)]).numpy()
        ys = torch.cat([fout[r][1, : int(allc[r, 2])] for r in range(world)]).numpy()
    else:
        # the one copy to the host: every rank's valid part straight into pinned memory (views, valid until the next gather)
        te, tr, tv = (int(allc[:, k].sum()) for k in range(3))
        hrc = _pinned_host("g_rc", (max(te, 1),), torch.int64)
        hvc = _pinned_host("g_vc", (max(tr, 1),), torch.int64)
        hx = _pinned_host("g_x", (max(tv, 1),), torch.float64)
        hy = _pinned_host("g_y", (max(tv, 1),), torch.float64)
        oe = orr = ov = 0
        for r in range(world):
            ke, kr, kv = (int(allc[r, k]) for k in range(3))
            if ke:
                hrc[oe:oe + ke].copy_(iout[r][:ke], non_blocking=True)
            if kr:
                hvc[orr:orr + kr].copy_(iout[r][me:me + kr], non_blocking=True)
            if kv:
                hx[ov:ov + kv].copy_(fout[r][0, :kv], non_blocking=True)
                hy[ov:ov + kv].copy_(fout[r][1, :kv], non_blocking=True)
            oe += ke; orr += kr; ov += kv
        torch.cuda.current_stream().synchronize()
        rcs, vcs, xs, ys = hrc[:te].numpy(), hvc[:tr].numpy(), hx[:tv].numpy(), hy[:tv].numpy()
    return (np.concatenate([[0], np.cumsum(rcs)]).astype(np.int64), np.concatenate([[0], np.cumsum(vcs)]).astype(np.int64), xs, ys)


def gather_device_geometry(batch, dist, dst=0):
    """the lon / lat rings of every rank's batch (transform_polygons done by each rank's own device, index_utils.py:129-184) to
    rank ``dst``, in rank (= time) order and in the row order of ``gather_device_tables(..., as_rows=True)``: rank ``dst`` holds the
    catalogue WITH its geometry and transforms nobody's polygons.  (ring_offset, vert_offset, x, y) there, None elsewhere."""
    if dist is None or not dist.is_initialized():
        ro, ne, vo, nr, x, y, nv, on_dev = batch.device_geometry()
        t = _geometry_tensors(batch, _Single())
        return (np.concatenate([[0], np.cumsum(t[0].cpu().numpy())]).astype(np.int64), np.concatenate([[0], np.cumsum(t[1].cpu().numpy())]).astype(np.int64),
                t[2].cpu().numpy(), t[3].cpu().numpy())
    dev = _device(dist)
    with _exchange_stream(dev):
        return _gather_geometry(_geometry_tensors(batch, dist), dist, dst, dev)


class _Single:
    @staticmethod
    def get_backend():
        return "none"


class TableGatherer:
    """Exchange (2) off the critical path.  ``submit(k, batch)`` -- on the caller's thread, in batch order -- STAGES the
    batch's device tables (a device-to-device copy of a few MB into a slot that outlives the batch: the batch can be closed
    at once and its scratch goes back to the allocator) and hands the slot to ONE gather thread, which runs the collective
    for slot after slot while the caller's thread and the workers go on with the next batches.  Every rank submits its
    batches in the same order, so every rank's thread issues its collectives in the same order.  ``finish()`` waits for the
    last gather and returns ``on_result``'s values in batch order.  ``comm``: a LibraryComm (RCCL inside the library) or
    None for torch.distributed (``dist``).  ``on_result(k, (rows, poly_x, poly_y) or None)`` runs on the gather thread as
    soon as batch k's catalogue stands in rank ``dst``'s pinned host buffers (views, valid until the next gather)."""

    def __init__(self, dist, comm=None, dst=0, on_result=None, slots=None, geometry=False):
        import queue
        import threading
        from wavebreaking_amd import _hip
        self.dist, self.comm, self.dst = dist, comm, dst
        # geometry: the batches' lon / lat rings travel too (a second pair of gathers per slot, through torch.distributed on
        # either path: clones of the arrays the geometry kernels wrote); ``on_result`` then gets (rows, poly_x, poly_y, (ring_offset,
        # vert_offset, x, y))
        self.geometry = bool(geometry)
        self.gdev = _device(dist) if geometry else None
        self.on_result = on_result or (lambda k, cat: None)
        self.nslots = int(slots or _hip.COMM_SLOTS)
        self.free = threading.Semaphore(self.nslots)
        self.q = queue.Queue()
        self.results, self.error, self.next_slot = {}, None, 0
        self.reserved, self.aborted, self.leaving = False, False, False
        self.dev = _device(dist) if comm is None else None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def submit(self, k, batch):
        if self.error is not None:
            raise self.error
        if not self.reserved and self.comm is not None and self.comm.rank == self.dst and batch is not None:
            # the destination's buffers for the WHOLE group's catalogue, sized from the first batch and made now -- growing them
            # inside a gather is a hipFree / hipMalloc on the gather thread while the workers' kernels run
            _, n3, _, _, nv = batch.device_tables()
            self.comm.ctx.comm_reserve(int(1.5 * self.comm.world * (sum(n3) + 64)), int(1.5 * self.comm.world * (nv + 4096)))
            self.reserved = True
        self.free.acquire()                                  # a slot the gather thread is done with (it may lag by nslots batches)
        slot = self.next_slot
        self.next_slot = (slot + 1) % self.nslots
        geo = None
        if self.geometry:
            with _exchange_stream(self.gdev):                # (clones: the batch is released right after this call)
                geo = tuple(t.clone() for t in _geometry_tensors(batch, self.dist))
        if self.comm is not None:
            self.comm.ctx.comm_stage_tables(batch, slot)
            item = slot
        else:
            with _exchange_stream(self.dev):                 # clones outlive the batch (leaving the block waits for the copies)
                item = tuple(t.clone() for t in _batch_tensors(batch, self.dist, False))
        self.q.put((k, (item, geo)))

    def _gather(self, item, poison, leaving=False):
        if self.comm is not None:
            rows, px, py, seen = self.comm.ctx.comm_gather_staged(-1 if poison else item, self.dst)
            if seen != self.comm.world:
                raise RuntimeError("wb_comm_gather_staged saw %d ranks, the group has %d" % (seen, self.comm.world))
            return (rows, px, py) if self.comm.rank == self.dst else None
        with _exchange_stream(self.dev):
            if poison:
                return _gather_tensors(None, None, None, self.dist, self.dst, self.dev, poison=True, leaving=leaving)
            return _gather_tensors(item[0], item[1], item[2], self.dist, self.dst, self.dev)

    def _run(self):
        while True:
            job = self.q.get()
            if job is None:
                if self.leaving and not self.aborted:
                    # abort() on the torch.distributed path (there is no ncclCommAbort to wake the peers): ONE last exchange of
                    # the counts with -2 behind the slots this rank still owed; the peers' next gather -- whichever slot they
                    # are at -- raises GatherAborted on all of them at once, and nobody issues a collective after it
                    try:
                        self._gather(None, True, leaving=True)
                    except BaseException:
                        pass
                    self.aborted = True
                return
            k, (item, geo) = job
            try:
                if self.error is None:
                    cat = self._gather(item, False)
                    if self.geometry:
                        # (the same order on every rank's gather thread: the slot's tables, then its rings)
                        with _exchange_stream(self.gdev):
                            g = _gather_geometry(geo, self.dist, self.dst, self.gdev)
                        cat = None if cat is None else tuple(cat) + (g,)
                    self.results[k] = self.on_result(k, cat)
                elif not self.aborted:
                    # A failure here must not leave the peers inside a collective: this rank still takes part in the exchange of
                    # the counts of every slot it owes, with -1 for its own, so every rank's gather for the slot raises and the
                    # sequences of collectives stay matched (until round 5 the slots were drained WITHOUT collectives: the peers
                    # then waited in ncclAllGather for ever)
                    try:
                        self._gather(item, True)
                    except GatherAborted:
                        self.aborted = True                  # a peer has left: no collective after this one
                    except BaseException:
                        pass
            except GatherAborted as e:
                self.error, self.aborted = e, True
            except BaseException as e:                       # kept for the caller
                self.error = e
            finally:
                self.free.release()

    def abort(self, why=None):
        """the record cannot be finished on this rank (a worker raised).  With the library's communicator: no further
        collective is issued and the communicator is aborted (ncclCommAbort: peers blocked in a collective with this rank get
        an error instead of waiting for ever; wb_comm_abort is safe beside a gather that is running on the gather thread --
        the library looks at its `aborted` flag before every group of RCCL calls).  On the torch.distributed path nothing can
        wake a peer that waits inside a collective, so this rank stays in step instead: the slots it still owes are exchanged
        as poisoned (-1), then ONE exchange with -2 tells every rank that it is leaving (GatherAborted there; nobody issues a
        collective after it).  The gather thread is stopped either way.  The process should end non-zero; a gather group
        cannot be restarted in place."""
        if self.error is None:
            self.error = why if isinstance(why, BaseException) else RuntimeError("TableGatherer aborted: %s" % (why,))
        if self.comm is not None:
            self.aborted = True
            try:
                self.comm.ctx.comm_abort()
            except Exception:
                pass
        else:
            self.leaving = True
        self.q.put(None)
        self.thread.join(timeout=30)

    def finish(self):
        self.q.put(None)
        self.thread.join()
        if self.error is not None:
            raise self.error
        return [self.results[k] for k in sorted(self.results)]


# ------------------------------------------------------------------------------------------------ RCCL inside the library
class LibraryComm:
    """The two exchanges on RCCL INSIDE the library (wb_comm_init / wb_comm_allreduce_max / wb_comm_gather_tables,
    include/wavebreaking_hip.h): the launcher's only part is to carry the 128 bytes of the unique id from rank 0 to the
    others -- here through the torch.distributed group the ranks were started with (any backend; its store would do as
    well).  ``max_nux`` is exchange (1), ``gather`` exchange (2): counts by one all-gather, payload by grouped send / recv
    of the exact sizes into buffers that persist between passes, one copy into pinned host memory on ``dst``."""

    def __init__(self, ctx, dist):
        import torch
        self.ctx, self.rank, self.world = ctx, dist.get_rank(), dist.get_world_size()
        # every rank first finds out for itself whether it can use RCCL at all (making an id loads the library), and the
        # ranks agree on that BEFORE any of them enters the collective set-up: a rank that cannot would leave the others
        # waiting inside ncclCommInitRank -- or, if it is rank 0, inside the broadcast of the id
        uid, why = None, ""
        try:
            uid = ctx.comm_unique_id()
        except Exception as e:
            why = str(e)
        dev = _device(dist)
        with _exchange_stream(dev):
            ok = torch.tensor([1 if uid is not None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            ok = int(ok.item())
        if not ok:
            raise RuntimeError("RCCL is not usable on every rank" + (" (here: %s)" % why if why else ""))
        box = [uid if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctx.comm_init(box[0], self.rank, self.world)

    def max_nux(self, local_max):
        return self.ctx.comm_allreduce_max(local_max)

    def gather(self, batch, dst=0):
        """(rows, poly_x, poly_y) of the whole record on ``dst`` (views of pinned buffers, valid until the next call), None
        elsewhere"""
        rows, px, py, seen = self.ctx.comm_gather_tables(batch, dst)
        if seen != self.world:
            raise RuntimeError("wb_comm_gather_tables saw %d ranks, the group has %d" % (seen, self.world))
        return (rows, px, py) if self.rank == dst else None

    def close(self):
        self.ctx.comm_destroy()


# ------------------------------------------------------------------------------------------------ whole calls
def _time_block(data, time_name, first, count):
    """``data`` restricted to time steps [first, first + count)."""
    if hasattr(data, "isel"):                                   # xarray.DataArray
        return data.isel({time_name: slice(first, first + count)})
    from wavebreaking_amd import gridfield as _gf
    ax = data.dims.index(time_name)
    sl = [slice(None)] * len(data.dims)
    sl[ax] = slice(first, first + count)
    coords = dict(data.coords)
    c = data.coords[time_name]
    coords[time_name] = _gf.Coord(c.values[first:first + count], c.attrs, c.encoding)
    return _gf.GridField(data.values[tuple(sl)], data.dims, coords, data.attrs, data.name)


def gather_frames(frame, dist, dst=0):
    """exchange (2) for whole frames (rows AND geometries): every rank's frame travels as one byte buffer in a single
    gather (sizes first); ``dst`` gets the list of frames in rank (= time) order, the others None."""
    import pickle
    import torch
    world, rank, dev = dist.get_world_size(), dist.get_rank(), _device(dist)
    blob = np.frombuffer(pickle.dumps(frame, protocol=pickle.HIGHEST_PROTOCOL), dtype=np.uint8)
    with _exchange_stream(dev):
        size = torch.tensor([blob.size], dtype=torch.int64, device=dev)
        sizes = [torch.zeros_like(size) for _ in range(world)]
        dist.all_gather(sizes, size)
        sizes = [int(t.item()) for t in sizes]
        buf = torch.zeros(max(sizes), dtype=torch.uint8, device=dev)
        buf[: blob.size] = torch.from_numpy(blob.copy()).to(dev)
        out = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, out, dst=dst)
        if rank != dst:
            return None
        return [pickle.loads(out[r][: sizes[r]].cpu().numpy().tobytes()) for r in range(world)]


def calculate_sharded(func, data, contour_levels, *args, dist=None, dst=0, **kwargs):
    """Run ``func`` (calculate_contours / calculate_streamers / calculate_overturnings / calculate_cutoffs) time-sharded
    over the ranks of ``dist``: every rank passes the SAME full ``data`` (and ``intensity``), works on its contiguous
    block of time steps, the contour selection uses the maximum over all ranks (one all-reduce of an int32), and the
    frames are gathered to rank ``dst`` in time order -- which returns what the single-process call returns; the other
    ranks return None."""
    import pandas as pd
    from wavebreaking_amd import engine
    from wavebreaking_amd.utils import data_utils as du
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return func(data, contour_levels, *args, **kwargs)
    rank, world = dist.get_rank(), dist.get_world_size()
    tn = kwargs.get("time_name") or du.get_time_name(data)
    first, count = shard(len(data[tn]), rank, world)
    reduced = []                                                    # did this rank's call reach the all-reduce?

    def hook(g):
        reduced.append(True)
        return global_max_nux(g, dist)

    engine.set_global_max_hook(hook)
    error = None
    frame = pd.DataFrame()
    try:
        if count > 0:
            kw = dict(kwargs)
            if kw.get("intensity") is not None:
                kw["intensity"] = _time_block(kw["intensity"], tn, first, count)
            if kw.get("contours") is not None and len(kw["contours"]):
                # a contour frame handed in covers the whole record: every rank keeps the rows of its own dates
                block_times = set(engine._time_key(t) for t in np.asarray(_time_block(data, tn, first, count)[tn].values))
                keep = [engine._time_key(t) in block_times for t in kw["contours"]["date"]]
                kw["contours"] = kw["contours"][np.asarray(keep, dtype=bool)].reset_index(drop=True)
                # no contour row on this rank's dates: an empty shard, not the "contours ... is empty!" error of a call that was
                # handed an empty frame (the one-process call finds nothing on those dates either)
                count = count if len(kw["contours"]) else 0
            if count > 0:
                frame = func(_time_block(data, tn, first, count), contour_levels, *args, **kw)
    except Exception as e:                                           # a failing rank still enters the collectives below:
        error = e                                                    # the others would wait for it forever otherwise
        frame = pd.DataFrame()
    finally:
        engine.set_global_max_hook(None)
    if func.__name__ != "calculate_contours" and not reduced:
        global_max_nux(0, dist)                                      # empty shard / early failure: keep the collective matched
    failed = _any_rank_failed(error is not None, dist)
    frames = gather_frames(frame, dist, dst)
    if error is not None:
        raise error
    if failed:
        raise RuntimeError("calculate_sharded: the call failed on another rank")
    if frames is None:
        return None
    frames = [f for f in frames if len(f)]
    if not frames:
        return frame
    return pd.concat(frames).reset_index(drop=True)


def _any_rank_failed(mine, dist):
    import torch
    dev = _device(dist)
    with _exchange_stream(dev):
        t = torch.tensor([1 if mine else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(int(t.item()))


# ------------------------------------------------------------------------------------------------ tracking on shards
def track_events_sharded(events, dist=None, dst=0, time_range=None, **kwargs):
    """``track_events`` (processing/events.py:113-241) on a time-sharded catalogue WITHOUT moving it to one rank first
    ("a one-step halo of event polygons at shard boundaries"): every rank holds the frame of its own contiguous
    block of dates (what ``calculate_*`` returned for its shard, rows in time order) and

    1. the ranks exchange their TAILS -- the rows of each rank's last ``time_range`` hours, a step's worth of events -- and every
       rank puts in front of its rows those rows of the ranks before it that lie within ``time_range`` of its first date (its halo):
       every pair 0 < dt <= time_range (events.py:163-181) then has both members on the rank that owns the later one;
    2. every rank runs ``track_events`` on halo + own rows (the pair tests -- exact overlap areas or haversines -- on its own GPU);
    3. the local labels are unified: a halo row carries a label on the rank that owns it AND on the rank that borrowed it, so the
       (rank, label) classes that share a row are one component (union-find over a few dozen nodes per boundary on ``dst``);
       components are named after their smallest row of the whole record and renumbered 0, 1, 2, ... (events.py:224-238), the
       frame is sorted by label and date (:241).

    Returns on ``dst`` the tracked frame of the whole record -- row for row what ``track_events`` returns for the
    concatenated catalogue --, None elsewhere.  ``time_range`` None: the smallest positive gap between two rows' dates over ALL
    ranks (events.py:151-153)."""
    import pandas as pd
    from wavebreaking_amd.processing import events as pe
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return pe.track_events(events, time_range=time_range, **kwargs)
    rank, world = dist.get_rank(), dist.get_world_size()
    events = events.reset_index(drop=True) if len(events) else events
    n_own = len(events)
    dates = np.asarray(events["date"].values) if n_own else np.zeros(0, dtype="datetime64[ns]")
    is_time = n_own == 0 or np.issubdtype(dates.dtype, np.datetime64)
    key = (dates.astype("datetime64[ns]").astype(np.int64) if is_time else dates.astype(np.float64)) if n_own else np.zeros(0)
    unit = 1e9 * 3600.0 if is_time else 1.0
    # what the ranks tell each other: first / last date and row count; the smallest positive gap needs the neighbours' edges too
    info = [None] * world
    dist.all_gather_object(info, dict(n=n_own, first=(key.min() if n_own else None), last=(key.max() if n_own else None),
                                      uniq=np.unique(key) if n_own else np.zeros(0)))
    if time_range is None:
        allk = np.unique(np.concatenate([i["uniq"] for i in info])) if any(i["n"] for i in info) else np.zeros(0)
        gaps = np.diff(allk)
        gaps = gaps[gaps > 0]
        if gaps.size == 0:
            raise ValueError("No events detected in the time range: {}".format(np.nan))
        time_range = float(gaps.min() / unit)
    reach = time_range * unit
    # 1. tails: the rows of my last time_range (they can be somebody's halo)
    tail = events[key >= key.max() - reach] if n_own else events
    tail_rows = np.nonzero(key >= key.max() - reach)[0] if n_own else np.zeros(0, dtype=np.int64)
    tails = [None] * world
    dist.all_gather_object(tails, (tail, tail_rows))
    halo_frames, halo_ids = [], []                       # rows borrowed from the ranks before me, (owner rank, owner row) each
    if n_own:
        for q in range(rank):
            fq, rq = tails[q]
            if fq is None or not len(fq):
                continue
            kq = np.asarray(fq["date"].values)
            kq = kq.astype("datetime64[ns]").astype(np.int64) if is_time else kq.astype(np.float64)
            m = (kq >= key.min() - reach) & (kq < key.min())
            if m.any():
                halo_frames.append(fq[m])
                halo_ids.extend((q, int(r)) for r in np.asarray(rq)[m])
    # 2. local labels on halo + own rows
    error, local = None, np.zeros(0, dtype=np.int64)
    try:
        if n_own:
            both = pd.concat(halo_frames + [events]).reset_index(drop=True) if halo_frames else events
            both.attrs.update(events.attrs)
            if len(both) > 1 and np.unique(np.asarray(both["date"].values)).size > 1:
                try:
                    tr = pe.track_events(both, time_range=time_range, **kwargs)
                    local = np.empty(len(both), dtype=np.int64)
                    local[tr.index.to_numpy()] = np.asarray(tr["label"], dtype=np.int64)
                except ValueError as e:                       # "No events detected in the time range": no pair on this rank
                    if "No events detected" not in str(e):
                        raise
                    local = np.arange(len(both), dtype=np.int64)
            else:
                local = np.arange(len(both), dtype=np.int64)
    except Exception as e:
        error = e
    failed = _any_rank_failed(error is not None, dist)
    if error is not None:
        raise error
    if failed:
        raise RuntimeError("track_events_sharded: the call failed on another rank")
    nh = len(halo_ids)
    report = dict(own=local[nh:], halo=[(q, r, int(local[k])) for k, (q, r) in enumerate(halo_ids)])
    reports = [None] * world if rank == dst else None
    dist.gather_object(report, reports, dst=dst)
    frames = gather_frames(events, dist, dst)
    if rank != dst:
        return None
    # 3. one component per class of (rank, label) nodes that share a row
    parent = {}

    def find(a):
        while parent.setdefault(a, a) != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for r, rep in enumerate(reports):
        for q, row, lab in rep["halo"]:
            a, b = find((r, lab)), find((q, int(reports[q]["own"][row])))
            if a != b:
                parent[a] = b
    base = np.concatenate([[0], np.cumsum([i["n"] for i in info])])
    total = int(base[-1])
    if total == 0:
        return events
    if not any(len(rep["own"]) > 1 or rep["halo"] for rep in reports) and total < 2:
        raise ValueError("No events detected in the time range: {}".format(time_range))
    comp = np.empty(total, dtype=object)
    for r, rep in enumerate(reports):
        for i, lab in enumerate(rep["own"]):
            comp[base[r] + i] = find((r, int(lab)))
    first_row, label = {}, np.empty(total, dtype=np.int64)
    for g in range(total):                                 # components named after their smallest row, renumbered in that order
        label[g] = first_row.setdefault(comp[g], len(first_row))
    whole = pd.concat([f for f in frames if f is not None and len(f)]).reset_index(drop=True)
    wk = np.asarray(whole["date"].values)
    wk = wk.astype("datetime64[ns]").astype(np.int64) if is_time else wk.astype(np.float64)
    gaps = np.diff(np.unique(wk)) / unit
    if gaps.size == 0 or not (gaps.min() <= time_range):
        raise ValueError("No events detected in the time range: {}".format(time_range))      # events.py:179-181
    whole["label"] = label
    out = whole.iloc[np.lexsort((wk, label))]                                                # events.py:241
    return out
