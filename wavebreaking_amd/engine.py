"""Glue between the reference-shaped Python functions and the C ABI: one cached context per device,
array extraction, level chunking (the ABI takes up to 4 levels per pass), frame <-> table conversion."""
import logging
import os

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
)
LEVEL_ORDER_TEXT = ("cells equal to the contour level put two contour ends on one grid vertex and the order-dependent join of "
                    "scikit-image's find_contours (contour_index.py:103) was not reproduced for this job.")
LEVEL_KEYERROR_TEXT = ("cells equal to the contour level put two contour ends on one grid vertex: scikit-image's find_contours "
                       "(contour_index.py:103) raises KeyError on this field -- a piece end whose dictionary entry another piece "
                       "overwrote -- and so does this call.  Perturb the level by one ulp to avoid it.")


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
# ``contours=`` argument (contour_index.py:203-208); every call still reads the cube.  Here the expensive part of a call on
# a host-resident cube is its trip over PCIe (4.15 MB per 0.25 degree step and cube), so the last cubes a context was given
# stay on the device, and so do the contour batches traced from them: a second call on the same array -- same buffer,
# shape, dtype, and the same values in a strided sample of 16 Ki elements plus both ends -- skips upload AND tracing.
# In-place edits of single cells between two calls can slip through that sample: WB_INPUT_CACHE=0 (or release_inputs())
# uploads on every call.  WB_INPUT_CACHE_GB bounds the device memory kept (default 48).
INPUT_CACHE = os.environ.get("WB_INPUT_CACHE", "1") != "0"
_tick = [0]


def _caches(ctx):
    """(resident cubes: key -> dict(ptr, nbytes, tick), kept batches: key -> dict(batch, field, tick)) of a context; they
    live on the context object and go with it"""
    if not hasattr(ctx, "_wb_resident"):
        ctx._wb_resident, ctx._wb_kept = {}, {}
    return ctx._wb_resident, ctx._wb_kept


def _array_key(arr):
    flat = arr.reshape(-1)                      # a view: the array is C-contiguous
    n = flat.size
    sample = np.ascontiguousarray(flat[:: max(1, n // 16384)])
    edge = np.concatenate([flat[:256], flat[-256:]])
    return (arr.__array_interface__["data"][0], arr.shape, arr.dtype.str, hash(sample.tobytes()), hash(edge.tobytes()))


def _cacheable(arr):
    return (INPUT_CACHE and isinstance(arr, np.ndarray) and arr.ndim == 3 and arr.dtype in (np.float32, np.float64)
            and arr.flags.c_contiguous and arr.size > 0)


def _drop_resident(ctx, key):
    resident, kept = _caches(ctx)
    ent = resident.pop(key, None)
    if ent is None:
        return
    for bk in [k for k, v in kept.items() if v["field"] == key or v["intensity"] == key]:
        kept.pop(bk)["batch"].close()
    if ctx.h:
        ctx.free(ent["ptr"])


def release_inputs():
    """give back the device copies of input cubes (and the contour batches traced from them) kept between calls"""
    for ctx in _contexts.values():
        resident, kept = _caches(ctx)
        for bk in list(kept):
            kept.pop(bk)["batch"].close()
        for key in list(resident):
            _drop_resident(ctx, key)


def resident_input(ctx, arr):
    """device pointer entry of a host cube, uploading it unless this context still holds it"""
    resident, _ = _caches(ctx)
    key = _array_key(arr)
    _tick[0] += 1
    ent = resident.get(key)
    if ent is None:
        budget = float(os.environ.get("WB_INPUT_CACHE_GB", 48)) * 2.0 ** 30
        while resident and sum(v["nbytes"] for v in resident.values()) + arr.nbytes > budget:
            _drop_resident(ctx, min(resident, key=lambda k: resident[k]["tick"]))
        ent = dict(ptr=ctx.to_device(arr), nbytes=arr.nbytes, key=key)
        resident[key] = ent
    ent["tick"] = _tick[0]
    return ent


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
        fe = resident_input(ctx, prep.values)
        ie = resident_input(ctx, prep.intensity) if prep.intensity is not None else None
        grid = hash((prep.lat.tobytes(), prep.lon.tobytes(), float(prep.dlon), float(prep.dlat)))
        step_bytes = nlat * nlon * prep.values.dtype.itemsize
    for t0 in range(0, nt, chunk):
        t1 = min(nt, t0 + chunk)
        inten = None if prep.intensity is None else prep.intensity[t0:t1]
        for lo in range(0, len(levels), MAX_LEVELS):
            pos = list(range(lo, min(lo + MAX_LEVELS, len(levels))))
            lev = [float(levels[i]) for i in pos]
            if keep:
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
            if keep:
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
