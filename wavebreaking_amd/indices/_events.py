"""Shared tail of the three index functions: run phase 2 on the batch(es), turn the ABI's event tables
into the reference's frame (calculate_properties + transform_polygons, index_utils.py:35-184)."""
import numpy as np

from wavebreaking_amd import _hip, engine
from wavebreaking_amd import gridfield as _gf
from wavebreaking_amd.utils import index_utils as iu

DIM_KEYS = ("time_name", "lon_name", "lat_name", "ntime", "nlon", "nlat", "dlon", "dlat")


def run_index(kind, data, contour_levels, contours, intensity, periodic_add, params, kwargs, global_max=None):
    """kind: _hip.STREAMERS / OVERTURNINGS / CUTOFFS.  Returns the reference-shaped frame."""
    dim_kwargs = {k: kwargs[k] for k in DIM_KEYS}
    prep = engine.Prepared(data, intensity, dim_kwargs)
    state = kwargs.get("_wb_state")
    ctx = engine.get_context()
    own = state is not None and contours is state.get("frame")     # the frame the decorator computed a moment ago
    if own and len(contours) == state["nrows"] and intensity is None:
        batches, levels = state["batches"], state["levels"]          # contours computed by the decorator: reuse
        for b, *_ in batches:
            b.set_index_params(**params)
    else:
        if state is not None:
            for b, *_ in state["batches"]:
                b.close()
        levels = engine.as_levels(contour_levels)
        if own:
            # same contours, but the index needs the intensity field on the device too
            batches = engine.contour_batches(prep, levels, periodic_add, params)
        else:
            table, levels = engine.frame_to_table(contours, prep.times, levels)
            b = ctx.batch_from_contours(table, prep.values, prep.lat, prep.lon, [float(v) for v in levels][:4] or [0.0],
                                        prep.dlon, prep.dlat, periodic_add=periodic_add, intensity=prep.intensity,
                                        **params)
            batches = [(b, None, 0)]
    g = max(b.max_nux() for b, *_ in batches) if batches else 0          # contours.exp_lon.max(), streamer_index.py:106
    if global_max is not None:
        g = max(g, int(global_max))
    g = engine.global_max(g)                                               # time-sharded runs: all-reduce(max) across ranks
    which = {_hip.STREAMERS: dict(streamers=True, overturnings=False, cutoffs=False),
             _hip.OVERTURNINGS: dict(streamers=False, overturnings=True, cutoffs=False),
             _hip.CUTOFFS: dict(streamers=False, overturnings=False, cutoffs=True)}[kind]
    rows = []
    batches = list(batches)
    for bi, (b, pos, t0) in enumerate(batches):
        b.indices(g, **which)
        try:
            engine.check_batch(b, "indices", prep.times, levels, pos, t0)
        except Exception:
            for ob, *_ in batches:
                ob.close()
            raise
        ev = b.events(kind)
        for k in range(ev["step"].size):
            lvl = ev["level_idx"][k]
            rows.append((int(ev["step"][k]) + t0 if pos is not None else k, pos[int(lvl)] if pos is not None else int(lvl), bi, k))
        batches[bi] = (b, pos, ev, t0)
    if len(batches) > 1:
        rows.sort(key=lambda r: (r[0], r[1]))          # stable: time-major, level-minor, event order inside kept
    cols = dict(date=[], level=[])
    polys, sums, orient = [], [], []
    for _, lpos, bi, k in rows:
        b, pos, ev, t0 = batches[bi]
        cols["date"].append(prep.times[int(ev["step"][k]) + t0])
        cols["level"].append(levels[lpos])
        a, e = int(ev["poly_offset"][k]), int(ev["poly_offset"][k + 1])
        polys.append(np.c_[ev["poly_x"][a:e], ev["poly_y"][a:e]])        # streamer_index.py:269-272 / box / cutoff_index.py:96
        if kind == _hip.OVERTURNINGS:
            orient.append("cyclonic" if ev["orientation"][k] == 0 else "anticyclonic")
        sums.append(ev["sums"][k])
    for b, *_ in batches:
        b.close()
    if len(rows) == 0:
        return _gf.empty_frame()                                          # streamer_index.py:278-279
    com, mean_var, inten, area = iu.properties_from_sums(np.asarray(sums), prep.lon, prep.lat, prep.nlon)
    cols.update(com=com, mean_var=mean_var, intensity=inten, event_area=area)   # index_utils.py:113-120
    if kind == _hip.OVERTURNINGS:
        cols["orientation"] = orient
    return _gf.make_frame(cols, iu.transform_polygons(polys, prep.lon, prep.lat, prep.nlon))
