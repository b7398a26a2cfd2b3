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
LEVEL_ORDER_TEXT = ("cells equal to the contour level put two contour ends on one grid vertex: scikit-image's "
                    "find_contours (contour_index.py:103) joins them in an order-dependent way there and may itself raise "
                    "KeyError; that case is not reproduced.  Perturb the level by one ulp or unpack the field to float32 / float64.")


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
    jobs = np.nonzero(st & _hip.STATUS_LEVEL_ORDER)[0]
    if jobs.size:
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


def max_steps_per_batch():
    """Time steps resident per ABI batch (device scratch grows with it: ~15 MB per step and level at 0.25 degree)."""
    return max(1, int(os.environ.get("WB_MAX_STEPS_PER_BATCH", 1460)))


def iter_contour_batches(prep, levels, periodic_add, params, want_cells=False):
    """yields (batch, level positions, first time step): one ABI batch per block of time steps and group of <= 4 levels; the
    consumer closes each batch (calculate_contours does so as soon as it has the batch's table, the index functions as soon
    as the block's events are on the host -- device memory holds one block at a time; WB_MAX_STEPS_PER_BATCH sets its size)"""
    ctx = get_context()
    nt = prep.values.shape[0]
    chunk = max_steps_per_batch()
    for t0 in range(0, nt, chunk):
        t1 = min(nt, t0 + chunk)
        inten = None if prep.intensity is None else prep.intensity[t0:t1]
        for lo in range(0, len(levels), MAX_LEVELS):
            pos = list(range(lo, min(lo + MAX_LEVELS, len(levels))))
            b = ctx.contours(prep.values[t0:t1], prep.lat, prep.lon, [float(levels[i]) for i in pos], prep.dlon, prep.dlat,
                             periodic_add=periodic_add, intensity=inten, want_cells=want_cells, **params)
            try:
                check_batch(b, "contours", prep.times, levels, pos, t0)
            except Exception:
                b.close()
                raise
            yield b, pos, t0


def _time_key(t):
    if hasattr(t, "to_datetime64"):          # pandas.Timestamp (what a datetime64 column hands back)
        t = t.to_datetime64()
    if isinstance(t, np.datetime64):
        return ("M", int(t.astype("datetime64[ns]").astype(np.int64)))
    return ("O", t.item() if hasattr(t, "item") else t)


def frame_to_table(contours, times, levels):
    """A reference-shaped contour frame (index coordinates) -> the ABI's contour table."""
    tkey = {_time_key(t): i for i, t in enumerate(np.asarray(times))}
    step, lvl, closed, npts, xs, ys = [], [], [], [], [], []
    level_list = list(levels)
    has_closed = "closed" in contours.columns
    for _, row in contours.iterrows():
        k = _time_key(row["date"])
        if k not in tkey:
            raise ValueError("contour date {} not found in data".format(row["date"]))
        step.append(tkey[k])
        if row["level"] not in level_list:
            level_list.append(row["level"])
        lvl.append(level_list.index(row["level"]))
        closed.append(bool(row["closed"]) if has_closed else False)
        c = np.asarray(row["geometry"].coords)
        xs.append(c[:, 0].astype(np.int32))
        ys.append(c[:, 1].astype(np.int32))
        npts.append(len(c))
    off = (np.cumsum(npts) - np.asarray(npts)) if npts else np.zeros(0, dtype=np.int64)
    table = dict(step=np.asarray(step, dtype=np.int32), level_idx=np.asarray(lvl, dtype=np.int32),
                 closed=np.asarray(closed, dtype=np.int32), npts=np.asarray(npts, dtype=np.int32),
                 offset=np.asarray(off, dtype=np.int64),
                 x=np.concatenate(xs) if xs else np.zeros(0, np.int32), y=np.concatenate(ys) if ys else np.zeros(0, np.int32))
    return table, level_list
