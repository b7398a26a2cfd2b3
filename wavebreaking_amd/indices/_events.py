"""Shared tail of the three index functions: run phase 2 on the batch(es), turn the ABI's event tables
into the reference's frame (calculate_properties + transform_polygons, index_utils.py:35-184)."""
import numpy as np

from wavebreaking_amd import _hip, engine
from wavebreaking_amd import gridfield as _gf

DIM_KEYS = ("time_name", "lon_name", "lat_name", "ntime", "nlon", "nlat", "dlon", "dlat")


class LazyContours:
    """what the decorator hands to an index function in place of the contour frame when the caller gave none: the index
    functions trace the contours themselves, block of time steps by block, and nothing needs the frame (the reference's
    decorator builds it, contour_index.py:197-249, only to pass it on)."""
    empty = False

    def __len__(self):
        return 1


def run_index(kind, data, contour_levels, contours, intensity, periodic_add, params, kwargs, global_max=None):
    """kind: _hip.STREAMERS / OVERTURNINGS / CUTOFFS.  Returns the reference-shaped frame."""
    ctx = engine.get_context()                    # (once, in front of the try: a failing look-up must not mask the call's own exception)
    for attempt in (0, 1):
        try:
            out = _run_index(kind, data, contour_levels, contours, intensity, periodic_add, params, kwargs, global_max)
            engine.verify_pending()               # device copies used on trust: their cubes' fingerprints, taken meanwhile
            return out
        except engine.StaleInput:                 # a cube changed on the host since it was uploaded: once more, from its bytes as they are
            if attempt:
                raise RuntimeError("wavebreaking_amd: an input cube keeps changing while the call runs")
        except Exception:
            # the call raised on a device copy used on trust: if that cube has changed on the host, the error may be the OLD
            # bytes' (find_contours' KeyError on cells equal to the level that were edited away since) -- the reference rereads
            # its input on every call: once more from the fresh bytes; otherwise the exception stands
            if attempt or not engine.discard_pending():
                raise
        finally:
            engine.discard_pending()
            engine.release_call_transients(ctx)   # cubes uploaded for this call only (engine.resident_input)


def _run_index(kind, data, contour_levels, contours, intensity, periodic_add, params, kwargs, global_max):
    dim_kwargs = {k: kwargs[k] for k in DIM_KEYS}
    prep = engine.Prepared(data, intensity, dim_kwargs)
    ctx = engine.get_context()
    which = {_hip.STREAMERS: dict(streamers=True, overturnings=False, cutoffs=False),
             _hip.OVERTURNINGS: dict(streamers=False, overturnings=True, cutoffs=False),
             _hip.CUTOFFS: dict(streamers=False, overturnings=False, cutoffs=True)}[kind]
    parts = []
    pending = []                                   # batches whose contours exist and whose indices wait for the maximum

    def finish(g):
        """phase 2 for the pending batches against the selection maximum g: events + their geometry in lon / lat
        (transform_polygons for the whole table in the library); no per-event Python except wrapping a ring array into a
        geometry object.  Every batch is released as soon as its tables are on the host."""
        while pending:
            b, pos, t0 = pending.pop(0)
            try:
                b.indices(g, **which)
                engine.check_batch(b, "indices", prep.times, levels, pos, t0)
                ev = b.events(kind, polygons=False)          # (the frames are built from the lon / lat rings)
                n = ev["step"].size
                if n:
                    ring_off, vert_off, gx, gy = b.event_geometry(kind)
                    if pos is not None:
                        step = ev["step"].astype(np.int64) + t0
                        lpos = np.asarray(pos, dtype=np.int64)[ev["level_idx"]]
                    else:                       # a frame handed in: one job per contour row, rows keep the frame's order
                        step = ev["step"].astype(np.int64)
                        lpos = ev["level_idx"].astype(np.int64)
                    parts.append(dict(step=step, lpos=lpos, props=ev["props"], orientation=ev["orientation"],
                                      ring_off=ring_off, vert_off=vert_off, gx=gx, gy=gy, n=n))
            except Exception:
                for ob, *_ in pending:
                    ob.close()
                del pending[:]
                raise
            finally:
                b.close()

    if isinstance(contours, LazyContours):
        # contours traced here, one block of time steps (and group of <= 4 levels) at a time.  The reference selects the
        # contours with exp_lon == exp_lon.max() over the WHOLE frame (streamer_index.py:106): no contour can span more
        # than the full width of the periodically extended grid, so as soon as one does the maximum is known and every
        # block is finished and released right after its contours exist -- device memory then holds one block at a time.
        levels = engine.as_levels(contour_levels)
        full_width = prep.nlon + int(periodic_add / prep.dlon)
        g_local, nbatches = 0, 0
        try:
            for item in engine.iter_contour_batches(prep, levels, periodic_add, params):
                nbatches += 1
                pending.append(item)
                g_local = max(g_local, item[0].max_nux())
                if g_local >= full_width and (global_max is None or int(global_max) <= full_width):
                    finish(full_width)
        except Exception:
            for ob, *_ in pending:
                ob.close()
            raise
        g = engine.global_max(max(g_local, int(global_max)) if global_max is not None else g_local)   # time-sharded runs: ONE all-reduce(max) per call
        finish(g)
        several = nbatches > 1
    else:
        table, levels = engine.frame_to_table(contours, prep.times, engine.as_levels(contour_levels))
        if engine._cacheable(prep.values) and (prep.intensity is None or (engine._cacheable(prep.intensity) and prep.intensity.dtype == prep.values.dtype)):
            # the cubes stay on the device between calls (engine.resident_input): a frame handed in costs no upload either
            fe, ie = engine.resident_inputs(ctx, prep.values, prep.intensity, prep.values.shape[0])
            engine._call_transients(ctx).extend(e for e in (fe, ie) if e is not None and e.get("transient"))
            b = ctx.batch_from_contours(table, fe["ptr"], prep.lat, prep.lon, [float(v) for v in levels][:4] or [0.0],
                                        prep.dlon, prep.dlat, periodic_add=periodic_add, intensity=ie["ptr"] if ie is not None else None,
                                        device_shape=prep.values.shape, device_dtype=prep.values.dtype, **params)
        else:
            b = ctx.batch_from_contours(table, prep.values, prep.lat, prep.lon, [float(v) for v in levels][:4] or [0.0],
                                        prep.dlon, prep.dlat, periodic_add=periodic_add, intensity=prep.intensity, **params)
        g = b.max_nux()                                                        # contours.exp_lon.max(), streamer_index.py:106
        if global_max is not None:
            g = max(g, int(global_max))
        g = engine.global_max(g)
        pending.append((b, None, 0))
        finish(g)
        several = False
    if not parts:
        return _gf.empty_frame()                                          # streamer_index.py:278-279
    step = np.concatenate([p["step"] for p in parts])
    lpos = np.concatenate([p["lpos"] for p in parts])
    props = np.concatenate([p["props"] for p in parts])
    orient_code = np.concatenate([p["orientation"] for p in parts])
    # geometry objects, batch after batch (the three cases of index_utils.py:167-173 in make_polygon)
    geoms = []
    for p in parts:
        xy = np.c_[p["gx"], p["gy"]]
        ro, vo = p["ring_off"], p["vert_off"]
        for k in range(p["n"]):
            geoms.append(_gf.make_polygon([xy[vo[q]:vo[q + 1]] for q in range(ro[k], ro[k + 1])]))
    if len(parts) > 1 and several:
        # several batches (time blocks, level groups): time-major, level-minor, event order inside a (step, level) kept
        order = np.lexsort((np.arange(step.size), lpos, step))
        step, lpos, props, orient_code = step[order], lpos[order], props[order], orient_code[order]
        geoms = [geoms[i] for i in order]
    level_values = np.empty(len(levels), dtype=object)
    level_values[:] = list(levels)
    cols = dict(date=np.asarray(prep.times)[step], level=list(level_values[lpos]))
    # index_utils.py:104-120: the truncated centre of mass and the three rounded columns come from the device with the sums
    # (wb_k_pack_results; utils.index_utils.properties_from_sums is the same statement in numpy)
    cols.update(com=list(zip(props[:, 0].tolist(), props[:, 1].tolist())), mean_var=props[:, 2].copy(), intensity=props[:, 3].copy(),
                event_area=props[:, 4].copy())
    if kind == _hip.OVERTURNINGS:
        names = np.array(["cyclonic", "anticyclonic"], dtype=object)
        cols["orientation"] = list(names[orient_code])
    frame = _gf.make_frame(cols, geoms)
    # the grid the polygon vertices lie on: track_events(method="by_overlap") works on grid indices
    frame.attrs["wavebreaking_grid"] = dict(lon=(float(prep.lon[0]), float(prep.dlon)), lat=(float(prep.lat[0]), float(prep.dlat)))
    return frame
