"""Glue between the reference-shaped Python functions and the C ABI: one cached context per device,
array extraction, level chunking (the ABI takes up to 4 levels per pass), frame <-> table conversion."""
import logging
import os
import threading

import numpy as np

from wavebreaking_amd import _hip
from wavebreaking_amd.utils import data_utils as du

_contexts = {}
MAX_LEVELS = 4
_global_max_hook = None


def set_global_max_hook(hook):
    """``hook(local_max) -> global_max``: how the maximum distinct-longitude count is completed across the ranks of a
    time-sharded run (wavebreaking_amd.distributed installs an all-reduce(max) here); None = single process."""
    global _global_max_hook
    _global_max_hook = hook


def global_max(local_max):
    return int(local_max) if _global_max_hook is None else int(_global_max_hook(int(local_max)))


logger = logging.getLogger(__name__)

_FATAL = (
    (_hip.STATUS_OVERFLOW_SEG, "more contour segments than the segment capacity allows (after the automatic retries)"),
    (_hip.STATUS_HASH_FULL, "a hash table of the contour assembly ran full"),
    (_hip.STATUS_OVERFLOW_PAIRS, "more candidate base-point pairs, streamers or overturning groups per contour than the "
                                 "kernels hold; the event list would be truncated"),
    (_hip.STATUS_SCAN_TIMEOUT, "the field scan's halo exchange ran out of its spin budget (an internal error: please report the field)"),
)
LEVEL_ORDER_TEXT = ("cells equal to the contour level put two contour ends on one grid vertex and the order-dependent join of "
                    "scikit-image's find_contours (contour_index.py:103) was not reproduced for this job.")
LEVEL_KEYERROR_TEXT = ("cells equal to the contour level put two contour ends on one grid vertex: find_contours of scikit-image 0.18.3 "
                       "(contour_index.py:103; the release this package's assembly rules are pinned to -- the reference pins none, and "
                       "later releases rewrote the routine) raises KeyError on this field -- a piece end whose dictionary entry another "
                       "piece overwrote -- and so does this call.  Perturb the level by one ulp to avoid it.")


def check_batch(batch, what, times=None, levels=None, pos=None, t0=0):
    """Raise / warn on the per-(time step, level) status words and the guard-band counters of an ABI batch: a frame
    that may differ from the reference's is never handed out silently."""
    st = np.asarray(batch.status())
    nlev = max(1, int(batch.nlev))

    def where(jobs):
        out = []
        for j in jobs[:5]:
            step, li = int(j) // nlev + int(t0), int(j) % nlev
            date = times[step] if times is not None and pos is not None and step < len(times) else step
            lvl = levels[pos[li]] if levels is not None and pos is not None and li < len(pos) else li
            out.append("(%s, level %s)" % (date, lvl) if pos is not None else "(contour row %d)" % int(j))
        return ", ".join(out) + (" ... %d in all" % len(jobs) if len(jobs) > 5 else "")

    for bit, text in _FATAL:
        jobs = np.nonzero(st & bit)[0]
        if jobs.size:
            raise RuntimeError("wavebreaking_amd %s: %s at %s" % (what, text, where(jobs)))
    jobs = np.nonzero(st & _hip.STATUS_LEVEL_KEYERROR)[0]
    if jobs.size:                                       # what the reference's call does on such a field
        raise KeyError("wavebreaking_amd %s: %s  Affected: %s" % (what, LEVEL_KEYERROR_TEXT, where(jobs)))
    jobs = np.nonzero((st & _hip.STATUS_LEVEL_ORDER) != 0)[0]
    jobs = jobs[(st[jobs] & _hip.STATUS_LEVEL_WALKED) == 0]
    if jobs.size:                                       # (cannot happen: every flagged job is walked or raises)
        raise ValueError("wavebreaking_amd %s: %s  Affected: %s" % (what, LEVEL_ORDER_TEXT, where(jobs)))
    geo, cont = batch.band_hits()
    if geo or cont:
        logger.warning("wavebreaking_amd %s: %d great-circle and %d along-contour distance comparisons fell within the guard band "
                       "of their threshold (streamer_index.py:138) and were re-evaluated in the reference's operation order; "
                       "they may still depend on the last bit of the math library" % (what, geo, cont))


def get_context(device=None):
    """The process-wide context of ``device`` (default: LOCAL_RANK, else 0)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", 0))
    if device not in _contexts:
        _contexts[device] = _hip.Context(device)      # raises if libwb_hip.so is missing: no CPU fallback
    return _contexts[device]


def as_levels(contour_levels):
    """index_utils.py:247-250"""
    try:
        iter(contour_levels)
    except Exception:
        contour_levels = [contour_levels]
    return list(contour_levels)


class Prepared:
    """Oriented numpy views of a call's inputs."""

    def __init__(self, data, intensity, kwargs):
        data = du.correct_dimension_orientation(data, **kwargs)
        self.values, self.times, self.lat, self.lon = du.as_time_lat_lon(data, **kwargs)
        self.intensity = None
        if intensity is not None:
            if not isinstance(intensity, type(data)) and not hasattr(intensity, "dims"):
                raise TypeError("intensity has to be a " + str(type(data))[8:-2] + "!")
            intensity = du.correct_dimension_orientation(intensity, **kwargs)
            self.intensity = du.as_time_lat_lon(intensity, **kwargs)[0]
            if self.intensity.shape != self.values.shape:
                raise ValueError("intensity must have the shape of data")
        self.nlon, self.nlat = len(self.lon), len(self.lat)
        self.dlon, self.dlat = float(kwargs["dlon"]), float(kwargs["dlat"])


# ------------------------------------------------------------------------------------------------ inputs kept on the device
# The reference's way of not repeating work across calculate_streamers / _overturnings / _cutoffs on one cube is the
# ``contours=`` argument (contour_index.py:203-208); every call still reads the cube (index_utils.py:66-75).  Here the
# expensive part of a call on a host-resident cube is its trip over PCIe (4.15 MB per 0.25 degree step and cube), so the
# last cubes a context was given stay on the device, and so do the contour batches traced from them: a later call that
# presents the same buffer with the same bytes skips upload AND tracing.  "The same bytes" is VERIFIED, never assumed:
#   * every call fingerprints the WHOLE buffer (wb_fingerprint: all bytes, several host threads -- a fraction of the PCIe
#     trip it saves) and a cube whose fingerprint changed is uploaded again: an in-place edit of a single cell between two
#     calls is seen, as it is by the reference;
#   * cubes that cannot change under the caller -- ``arr.flags.writeable`` is False -- or that the caller promised not to
#     modify (``wavebreaking_amd.keep_inputs(data, ...)``, until ``release_inputs()``) are recognised by buffer address,
#     shape and dtype plus a strided 16 Ki-element sample (a tripwire, not the guarantee).
# An entry lives as long as the host array it mirrors (weakref) and goes when that array is collected.
# WB_INPUT_CACHE=0 switches the cache off; WB_INPUT_CACHE_GB bounds the device memory kept (default 48); a call whose own
# cubes do not fit beside each other under the bound uploads them for the call only.
INPUT_CACHE = os.environ.get("WB_INPUT_CACHE", "1") != "0"
# WB_PIPELINED_UPLOAD=1: the first call on a cube that is not on the device yet cuts the record into blocks and computes
# block k while block k + 1 is uploaded.  Off by default: measured on MI355X (365 steps of 0.25 degree, field + intensity)
# the four blocks' fixed costs -- four wb_contours / wb_indices calls per index function instead of one -- were as large as
# the 4 ms of computing they hid behind the 55 ms of PCIe (72.6 ms for the three calls without it, 84.6 ms with it on a
# slower box), and a later call with ``contours=`` needs a whole-record batch the allocator has not seen yet.
PIPELINED_UPLOAD = os.environ.get("WB_PIPELINED_UPLOAD", "0") == "1"
_tick = [0]
_promised = {}          # (address, nbytes) -> weakref of the owner: buffers registered with keep_inputs()
_dead = []              # (device, key) of cache entries whose host array was collected; purged by the next call
FINGERPRINT_STATS = {"full": 0, "sampled": 0, "seconds": 0.0}


def _caches(ctx):
    """(resident cubes: key -> dict(ptr, nbytes, tick), kept batches: key -> dict(batch, field, tick)) of a context; they
    live on the context object and go with it"""
    if not hasattr(ctx, "_wb_resident"):
        ctx._wb_resident, ctx._wb_kept = {}, {}
    return ctx._wb_resident, ctx._wb_kept


def _owner(arr):
    """the object that owns an array's memory (the end of its .base chain)"""
    o = arr
    while isinstance(o, np.ndarray) and o.base is not None:
        o = o.base
    return o


def _as_array(obj):
    """the numpy buffer behind a cube handed to keep_inputs(): ndarray, GridField / xarray.DataArray (.values)"""
    if isinstance(obj, np.ndarray):
        return obj
    v = getattr(obj, "values", None)
    if isinstance(v, np.ndarray):
        return v
    raise TypeError("keep_inputs: a numpy array, a GridField or an xarray.DataArray is expected")


def keep_inputs(*cubes):
    """Promise not to modify these cubes in place until ``release_inputs()``: calls on them then recognise the device copy
    by the buffer's address (plus a strided sample) instead of fingerprinting every byte of every cube on every call."""
    import weakref
    for c in cubes:
        if c is None:
            continue
        a = _as_array(c)
        o = _owner(a)
        try:
            ref = weakref.ref(o)
        except TypeError:                        # (an owner that cannot be weakly referenced: held until release_inputs())
            ref = (lambda obj: (lambda: obj))(o)
        _promised[(a.__array_interface__["data"][0], a.nbytes)] = ref


def _is_promised(arr):
    ref = _promised.get((arr.__array_interface__["data"][0], arr.nbytes))
    if ref is None:
        return False
    if ref() is None:                            # the registered array is gone; the address belongs to something else now
        _promised.pop((arr.__array_interface__["data"][0], arr.nbytes), None)
        return False
    return True


def _immutable(arr):
    """can nobody change these bytes?  A read-only VIEW of a writeable array can still change through its base: every array
    of the .base chain has to be read-only, and the chain has to end in an array or an immutable buffer (bytes)"""
    o = arr
    while isinstance(o, np.ndarray):
        if o.flags.writeable:
            return False
        if o.base is None:
            return True
        o = o.base
    return isinstance(o, bytes)


def _head(arr):
    return (arr.__array_interface__["data"][0], arr.shape, arr.dtype.str)


def _timed_fingerprint(lib, arr):
    import time
    tic = time.perf_counter()
    fp = lib.fingerprint(arr)                    # (releases the GIL; several host threads of its own)
    FINGERPRINT_STATS["full"] += 1
    FINGERPRINT_STATS["seconds"] += time.perf_counter() - tic
    return fp


def _array_key(arr, ctx=None, owner=None):
    """cache key of a cube: buffer address, shape, dtype and a fingerprint of its bytes -- of ALL of them unless the buffer
    cannot change (not writeable) or was registered with keep_inputs().  ``owner``: the thread identifier of the CALL the key
    is made for (resident_inputs makes the keys of a call's two cubes on pool threads: the pending fingerprints belong to the
    calling thread, which is the one that verifies or discards them); default: this thread"""
    import time
    if owner is None:
        owner = threading.get_ident()
    flat = arr.reshape(-1)                      # a view: the array is C-contiguous
    n = flat.size
    head = (arr.__array_interface__["data"][0], arr.shape, arr.dtype.str)
    if _immutable(arr) or _is_promised(arr):
        sample = np.ascontiguousarray(flat[:: max(1, n // 16384)])
        edge = np.concatenate([flat[:256], flat[-256:]])
        FINGERPRINT_STATS["sampled"] += 1
        return head + ("sampled", hash(sample.tobytes()), hash(edge.tobytes()))
    ctx = ctx or get_context()
    lib = ctx.L

    def fingerprint():
        tic = time.perf_counter()
        fp = lib.fingerprint(arr)                # (releases the GIL; several host threads of its own)
        FINGERPRINT_STATS["full"] += 1
        FINGERPRINT_STATS["seconds"] += time.perf_counter() - tic
        return fp
    if SPECULATE:
        # A cube of this address, shape and dtype is on the device already: go on with that copy while the fingerprint of the
        # bytes as they are NOW is taken on other threads; the call checks it before it returns anything (verify_pending) and
        # starts over with a fresh upload if the cube has changed -- the check costs the call nothing unless it fails.
        with _pending_lock:
            hit = _pending_by_head.get((ctx.device, owner, head))
            if hit is not None:
                return hit[0]
            resident, _ = _caches(ctx)
            for k in resident:
                if k[:3] == head and k[3] == "full":
                    fut = _fp_pool().submit(fingerprint)
                    _pending_by_head[(ctx.device, owner, head)] = (k, fut, ctx)
                    FINGERPRINT_STATS["speculated"] = FINGERPRINT_STATS.get("speculated", 0) + 1
                    return k
    return head + ("full", fingerprint())


class StaleInput(Exception):
    """a cube that a call took from the device cache turned out to have changed on the host (verify_pending)"""


# WB_INPUT_SPECULATE=0: every call waits for the fingerprints of its cubes before it touches the device copies
SPECULATE = os.environ.get("WB_INPUT_SPECULATE", "1") != "0"
_pending_by_head = {}    # (device, thread, (address, shape, dtype)) -> (speculated key, future of the fingerprint, context): the running calls', one per thread
_pending_lock = threading.Lock()
_fp_executor = [None]


def _fp_pool():
    if _fp_executor[0] is None:
        from concurrent.futures import ThreadPoolExecutor
        _fp_executor[0] = ThreadPoolExecutor(2)
    return _fp_executor[0]


def verify_pending():
    """the end of a call that used device copies on trust: wait for the fingerprints taken meanwhile; a cube whose bytes no
    longer match its copy is dropped from the cache (with the contour batches traced from it) and StaleInput is raised -- the
    caller runs the call again, which then uploads the cube as it is"""
    if _take_pending(True):
        raise StaleInput()


def _take_pending(strict):
    """the RUNNING call's entries (this thread's: calls on other contexts / threads keep theirs): wait for their
    fingerprints, drop the device copies whose cube has changed; True if there was one"""
    me = threading.get_ident()
    with _pending_lock:
        mine = [k for k in _pending_by_head if k[1] == me]
        items = [_pending_by_head.pop(k) for k in mine]
    stale = False
    for key, fut, ctx in items:
        try:
            fp = fut.result()
        except BaseException:
            if strict:
                raise
            continue
        if fp != key[4]:
            stale = True
            _drop_resident(ctx, key)
    return stale


def discard_pending():
    """a call that ends with an exception: let its fingerprints finish -- and still COMPARE them: the exception may be the
    old copy's (a cube whose cells equal to the level were edited away keeps raising KeyError from the stale copy
    otherwise).  True if a cube had changed (its copy is dropped: the caller runs the call once more)"""
    return _take_pending(False)


def _cacheable(arr):
    return (INPUT_CACHE and isinstance(arr, np.ndarray) and arr.ndim == 3 and arr.dtype in (np.float32, np.float64)
            and arr.flags.c_contiguous and arr.size > 0)


def _watch(ctx, arr, key):
    """drop the entry when the host array it mirrors is collected (at the next call: a context is driven by one thread)"""
    import weakref
    try:
        weakref.finalize(_owner(arr), _dead.append, (ctx.device, key))
    except TypeError:
        pass


def _purge_dead(ctx):
    resident, _ = _caches(ctx)
    keep = []
    while _dead:
        dev, key = _dead.pop()
        if dev == ctx.device:
            if key in resident:
                _drop_resident(ctx, key)
        else:
            keep.append((dev, key))
    _dead.extend(keep)


class _Upload:
    """One or two cubes on their way to the device, block of time steps after block (the field's block, then the
    intensity's), copied by a thread of its own through a context of its own (stream + pinned staging buffers; a context is
    driven by one thread at a time) into memory the calling context allocated.  ``wait(t1)``: until the steps below t1 of
    every cube are there.  The first call on a host cube computes block k while block k + 1 crosses PCIe."""

    def __init__(self, copier, jobs, nt, block):
        self.cv = threading.Condition()
        self.done, self.error, self.nt = 0, None, nt
        self.thread = threading.Thread(target=self._run, args=(copier, jobs, nt, block), daemon=True)
        self.thread.start()

    def _run(self, copier, jobs, nt, block):
        try:
            for t0 in range(0, nt, block):
                t1 = min(nt, t0 + block)
                for ptr, arr, step_bytes in jobs:
                    copier.copy_to_device(ptr + t0 * step_bytes, arr[t0:t1])
                with self.cv:
                    self.done = t1
                    self.cv.notify_all()
        except BaseException as e:                       # handed to whoever waits
            with self.cv:
                self.error = e
                self.cv.notify_all()

    def wait(self, t1=None):
        t1 = self.nt if t1 is None else min(int(t1), self.nt)
        with self.cv:
            while self.done < t1 and self.error is None:
                self.cv.wait()
            if self.error is not None:
                raise self.error


def _copier(ctx):
    """the context whose stream and staging buffers carry the background uploads of ``ctx``"""
    if getattr(ctx, "_wb_copier", None) is None:
        ctx._wb_copier = _hip.Context(ctx.device, ctx.L)
    return ctx._wb_copier


def _settle(ent):
    up = ent.get("upload")
    if up is not None:
        try:
            up.wait()
        finally:
            up.thread.join()
            ent["upload"] = None


def _drop_resident(ctx, key):
    resident, kept = _caches(ctx)
    ent = resident.pop(key, None)
    if ent is None:
        return
    try:
        _settle(ent)                                 # (nobody frees memory a copy is still writing to)
    except BaseException:
        pass
    for bk in [k for k, v in kept.items() if v["field"] == key or v["intensity"] == key]:
        kept.pop(bk)["batch"].close()
    if ctx.h:
        ctx.free(ent["ptr"])


def release_inputs():
    """give back the device copies of input cubes (and the contour batches traced from them) kept between calls, and forget
    the promises made with keep_inputs()"""
    _promised.clear()
    for ctx in _contexts.values():
        resident, kept = _caches(ctx)
        for bk in list(kept):
            kept.pop(bk)["batch"].close()
        for key in list(resident):
            _drop_resident(ctx, key)


# the fingerprint of a cube that is certainly not on the device is taken BESIDE its upload (WB_FINGERPRINT_BESIDE_UPLOAD=0: first)
FINGERPRINT_BESIDE_UPLOAD = os.environ.get("WB_FINGERPRINT_BESIDE_UPLOAD", "1") != "0"
# a call with two cubes of at least this many bytes takes their keys on two threads (tests lower it)
PARALLEL_KEYS_BYTES = 64 << 20


def _budget():
    return float(os.environ.get("WB_INPUT_CACHE_GB", 48)) * 2.0 ** 30


def _make_room(ctx, need, protect):
    """evict least-recently-used cubes -- never one of the running call's own (``protect``) -- until ``need`` more bytes fit
    under the bound; False: they do not fit even then"""
    resident, _ = _caches(ctx)
    budget = _budget()
    while sum(v["nbytes"] for v in resident.values()) + need > budget:
        victim = min((k for k in resident if k not in protect), key=lambda k: resident[k]["tick"], default=None)
        if victim is None:
            return False
        _drop_resident(ctx, victim)
    return True


def resident_inputs(ctx, field, intensity, block):
    """the entries of a call's cube(s).  Cubes that are not on the device yet are allocated here and uploaded in the
    background, ``block`` time steps at a time (``ent["upload"].wait(t1)`` before steps below t1 are used; ``ent["block"]`` is
    the block size the call should cut the record into so that computing overlaps the upload).  An entry with
    ``ent["transient"]`` belongs to this call alone (the call's cubes did not fit under WB_INPUT_CACHE_GB beside each
    other): the caller frees it with ``release_transient`` when the call is over."""
    _purge_dead(ctx)
    resident, _ = _caches(ctx)
    arrs = [a for a in (field, intensity) if a is not None]
    nt = field.shape[0]
    # Cubes that are CERTAINLY not on the device (no entry of this address, shape and dtype): their upload starts at once and the
    # fingerprint that becomes part of their cache key is taken beside it, on a pool thread (both read the cube once: 1.5 GB at
    # 55 GB/s over PCIe against 200-300 GB/s of hashing) -- until round 6 the call hashed first (10-15 ms for a year's two
    # cubes) and uploaded then.
    if FINGERPRINT_BESIDE_UPLOAD and not (PIPELINED_UPLOAD and block < nt):
        plan = []
        for a in arrs:
            head = _head(a)
            if not (_immutable(a) or _is_promised(a)) and not any(k[:3] == head for k in resident):
                plan.append((a, head, _fp_pool().submit(_timed_fingerprint, ctx.L, a)))
            else:
                plan.append((a, None, None))
        if any(f is not None for _, _, f in plan):
            keys = [None if f is not None else _array_key(a, ctx) for a, _, f in plan]
            ents = []
            for i, (a, head, fut) in enumerate(plan):
                mine = [k for k in keys if k is not None]
                if fut is None:
                    ents.append(resident_input(ctx, a, keys[i], protect=mine))
                    continue
                fits = _make_room(ctx, a.nbytes, set(mine))
                _tick[0] += 1
                ptr = ctx.to_device(a)
                key = head + ("full", fut.result())
                ent = dict(ptr=ptr, nbytes=a.nbytes, key=key, tick=_tick[0])
                if fits and key not in resident:
                    resident[key] = ent
                    _watch(ctx, a, key)
                else:
                    ent["transient"] = True
                keys[i] = key
                ents.append(ent)
            return ents[0], (ents[1] if intensity is not None else None)
    if len(arrs) == 2 and arrs[0].nbytes >= PARALLEL_KEYS_BYTES:
        # the two cubes' fingerprints side by side (wb_fingerprint releases the GIL; each uses its own host threads); the
        # pending entries they leave are the CALLING thread's -- it is the one that runs verify_pending / discard_pending
        from concurrent.futures import ThreadPoolExecutor
        me = threading.get_ident()
        with ThreadPoolExecutor(2) as pool:
            keys = list(pool.map(lambda a: _array_key(a, ctx, owner=me), arrs))
    else:
        keys = [_array_key(a, ctx) for a in arrs]
    fresh = [k not in resident for k in keys]
    if not any(fresh) or not PIPELINED_UPLOAD or block >= nt:
        ents = [resident_input(ctx, a, k, protect=keys) for a, k in zip(arrs, keys)]
        return ents[0], (ents[1] if intensity is not None else None)
    need = sum(a.nbytes for a, f in zip(arrs, fresh) if f)
    if not _make_room(ctx, need, keys):
        ents = [resident_input(ctx, a, k, protect=keys) for a, k in zip(arrs, keys)]
        return ents[0], (ents[1] if intensity is not None else None)
    jobs, ents, new = [], [], []
    for a, k in zip(arrs, keys):
        _tick[0] += 1
        ent = resident.get(k)
        if ent is None:
            ent = dict(ptr=ctx.alloc(a.nbytes), nbytes=a.nbytes, key=k, block=block, upload=None)
            resident[k] = ent
            _watch(ctx, a, k)
            jobs.append((ent["ptr"], a, a[0].nbytes))
            new.append(ent)
        elif ent not in new:
            _settle(ent)
        ent["tick"] = _tick[0]
        ents.append(ent)
    up = _Upload(_copier(ctx), jobs, nt, block)
    for ent in new:
        ent["upload"] = up
    return ents[0], (ents[1] if intensity is not None else None)


def resident_input(ctx, arr, key=None, protect=()):
    """device pointer entry of a host cube, uploading it unless this context still holds it.  ``protect``: keys of the
    running call's cubes, which are never evicted to make room (the field of a call must outlive the upload of its
    intensity).  A cube that does not fit under the bound beside them is uploaded for this call only (``transient``)."""
    resident, _ = _caches(ctx)
    if key is None:
        _purge_dead(ctx)
        key = _array_key(arr, ctx)
    _tick[0] += 1
    ent = resident.get(key)
    if ent is not None:
        _settle(ent)                                 # (a background upload of an earlier call: wait for all of it)
    if ent is None:
        fits = _make_room(ctx, arr.nbytes, set(protect) | {key})
        ent = dict(ptr=ctx.to_device(arr), nbytes=arr.nbytes, key=key)
        if fits:
            resident[key] = ent
            _watch(ctx, arr, key)
        else:
            ent["transient"] = True
    ent["tick"] = _tick[0]
    return ent


def _call_transients(ctx):
    if not hasattr(ctx, "_wb_transients"):
        ctx._wb_transients = []
    return ctx._wb_transients


def release_call_transients(ctx):
    """free the device copies that were made for the running call only (its cubes did not fit under the cache's bound);
    the index functions and calculate_contours call this when their batches are closed"""
    pending = _call_transients(ctx)
    while pending:
        e = pending.pop()
        if e.get("ptr") and ctx.h:
            ctx.free(e["ptr"])
        e["ptr"] = None


class _KeptBatch:
    """a contour batch owned by the cache: the index functions use it like any batch, their close() leaves it alone"""

    def __init__(self, batch):
        self._b = batch

    def __getattr__(self, name):
        return getattr(self._b, name)

    def close(self):
        pass


def max_steps_per_batch():
    """Time steps resident per ABI batch (device scratch grows with it: ~15 MB per step and level at 0.25 degree)."""
    return max(1, int(os.environ.get("WB_MAX_STEPS_PER_BATCH", 1460)))


def iter_contour_batches(prep, levels, periodic_add, params, want_cells=False):
    """yields (batch, level positions, first time step): one ABI batch per block of time steps and group of <= 4 levels; the
    consumer closes each batch (calculate_contours does so as soon as it has the batch's table, the index functions as soon
    as the block's events are on the host -- device memory holds one block at a time; WB_MAX_STEPS_PER_BATCH sets its size)"""
    ctx = get_context()
    nt, nlat, nlon = prep.values.shape
    chunk = max_steps_per_batch()
    keep = _cacheable(prep.values) and (prep.intensity is None or (_cacheable(prep.intensity) and prep.intensity.dtype == prep.values.dtype))
    if keep:
        # the cubes go to the device once (whole, whatever the block size) and stay there for the next call
        _, _kept = _caches(ctx)
        # (a record of at least 128 steps that still has to cross PCIe: in four blocks, so that three of them are computed
        # while the next one arrives; the kept contour batches are per block, later calls cut the record the same way)
        fe, ie = resident_inputs(ctx, prep.values, prep.intensity, min(chunk, max(64, -(-nt // 4))) if nt >= 128 else chunk)
        chunk = min(chunk, fe.get("block", chunk))
        # cubes uploaded for this call only (they did not fit under the cache's bound): the consumer frees them when the
        # call is over (release_call_transients), and the batches traced from them are not kept either
        transient = [e for e in (fe, ie) if e is not None and e.get("transient")]
        _call_transients(ctx).extend(transient)
        grid = hash((prep.lat.tobytes(), prep.lon.tobytes(), float(prep.dlon), float(prep.dlat)))
        step_bytes = nlat * nlon * prep.values.dtype.itemsize
    for t0 in range(0, nt, chunk):
        t1 = min(nt, t0 + chunk)
        inten = None if prep.intensity is None else prep.intensity[t0:t1]
        for lo in range(0, len(levels), MAX_LEVELS):
            pos = list(range(lo, min(lo + MAX_LEVELS, len(levels))))
            lev = [float(levels[i]) for i in pos]
            if keep:
                try:
                    for e in (fe, ie):
                        if e is not None and e.get("upload") is not None:
                            e["upload"].wait(t1)             # the block's steps are on the device (the later ones on their way)
                except BaseException:
                    for e in (fe, ie):                       # a copy failed: nothing of these cubes stays in the cache
                        if e is not None:
                            _drop_resident(ctx, e["key"])
                    raise
                d_int = ie["ptr"] + t0 * step_bytes if ie is not None else None
                bkey = (fe["key"], grid, tuple(lev), float(periodic_add), t0, t1, bool(want_cells))
                ent = _kept.get(bkey)
                _tick[0] += 1
                if ent is not None:                       # traced by an earlier call: new index parameters / intensity only
                    ent["tick"] = _tick[0]
                    ent["batch"].set_index_params(**params)
                    ent["batch"].set_intensity(d_int)
                    yield _KeptBatch(ent["batch"]), pos, t0
                    continue
                b = ctx.contours(fe["ptr"] + t0 * step_bytes, prep.lat, prep.lon, lev, prep.dlon, prep.dlat, periodic_add=periodic_add,
                                 intensity=d_int, want_cells=want_cells, device_shape=(t1 - t0, nlat, nlon),
                                 device_dtype=prep.values.dtype, **params)
            else:
                b = ctx.contours(prep.values[t0:t1], prep.lat, prep.lon, lev, prep.dlon, prep.dlat,
                                 periodic_add=periodic_add, intensity=inten, want_cells=want_cells, **params)
            try:
                check_batch(b, "contours", prep.times, levels, pos, t0)
            except Exception:
                b.close()
                raise
            if keep and not transient:
                while len(_kept) >= int(os.environ.get("WB_KEPT_BATCHES", 8)):
                    _kept.pop(min(_kept, key=lambda k: _kept[k]["tick"]))["batch"].close()
                _kept[bkey] = dict(batch=b, field=fe["key"], intensity=ie["key"] if ie is not None else None, tick=_tick[0])
                b = _KeptBatch(b)
            yield b, pos, t0


def _time_key(t):
    if hasattr(t, "to_datetime64"):          # pandas.Timestamp (what a datetime64 column hands back)
        t = t.to_datetime64()
    if isinstance(t, np.datetime64):
        return ("M", int(t.astype("datetime64[ns]").astype(np.int64)))
    return ("O", t.item() if hasattr(t, "item") else t)


def frame_to_table(contours, times, levels):
    """A reference-shaped contour frame (index coordinates) -> the ABI's contour table.  Column-wise: dates and levels are
    mapped through their distinct values, the only per-row Python step is reading a geometry's coordinate array."""
    tkey = {_time_key(t): i for i, t in enumerate(np.asarray(times))}
    dcol = np.asarray(contours["date"].values)
    if np.issubdtype(dcol.dtype, np.datetime64) or np.issubdtype(dcol.dtype, np.number):
        uniq, inv = np.unique(dcol, return_inverse=True)   # a frame holds a handful of rows per date
    else:
        keys = [_time_key(d) for d in dcol]
        index = {}
        inv = np.fromiter((index.setdefault(k, len(index)) for k in keys), dtype=np.int64, count=len(keys))
        uniq = np.empty(len(index), dtype=object)
        for d, k in zip(dcol, keys):
            uniq[index[k]] = d
    ustep = np.empty(len(uniq), dtype=np.int32)
    for i, d in enumerate(uniq):
        k = _time_key(d)
        if k not in tkey:
            raise ValueError("contour date {} not found in data".format(d))
        ustep[i] = tkey[k]
    step = ustep[inv].astype(np.int32)
    dates = dcol
    level_list = list(levels)
    lvl = np.empty(len(dates), dtype=np.int32)
    for i, v in enumerate(contours["level"]):
        if v not in level_list:
            level_list.append(v)
        lvl[i] = level_list.index(v)
    closed = (np.asarray(contours["closed"], dtype=bool).astype(np.int32) if "closed" in contours.columns
              else np.zeros(len(dates), dtype=np.int32))
    coords = [np.asarray(g.coords) for g in contours["geometry"]]
    npts = np.fromiter((len(c) for c in coords), dtype=np.int32, count=len(coords))
    xy = np.concatenate(coords).astype(np.int32) if coords else np.zeros((0, 2), np.int32)
    off = (np.cumsum(npts) - npts).astype(np.int64)
    table = dict(step=step, level_idx=lvl, closed=closed, npts=npts, offset=off,
                 x=np.ascontiguousarray(xy[:, 0]), y=np.ascontiguousarray(xy[:, 1]))
    return table, level_list
