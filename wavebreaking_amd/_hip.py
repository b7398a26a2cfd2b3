"""ctypes binding of the C ABI in include/wavebreaking_hip.h.

The product library is ``wavebreaking_amd/csrc/libwb_hip.so`` (hipcc, gfx950).  There is no CPU
fallback: if the library is missing or reports a backend other than "hip" this module raises.
(The test suite drives the same binding class against the host-simulation build of the same
sources by passing an explicit path and ``allow_hostsim=True``; nothing in the package does.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# WB_HIP_LIBRARY: another HIP build of the same sources (development: kernel variants side by side); never a CPU build
LIB_PATH = os.environ.get("WB_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libwb_hip.so")


class WbParams(C.Structure):
    _fields_ = [
        ("nt", C.c_int32), ("nlat", C.c_int32), ("nlon", C.c_int32), ("dtype", C.c_int32), ("on_device", C.c_int32),
        ("field", C.c_void_p), ("intensity", C.c_void_p), ("lat", C.POINTER(C.c_double)), ("lon", C.POINTER(C.c_double)),
        ("nlevels", C.c_int32), ("levels", C.POINTER(C.c_double)),
        ("periodic_add", C.c_double), ("dlon", C.c_double), ("dlat", C.c_double),
        ("geo_dis", C.c_double), ("cont_dis", C.c_double), ("range_group", C.c_double), ("ot_min_exp", C.c_double),
        ("co_min_exp", C.c_double), ("want_cells", C.c_int32), ("cap_seg", C.c_int32), ("cap_pairs", C.c_int32),
    ]


class WbContourTable(C.Structure):
    _fields_ = [
        ("n", C.c_int64), ("step", C.POINTER(C.c_int32)), ("level_idx", C.POINTER(C.c_int32)),
        ("closed", C.POINTER(C.c_int32)), ("nux", C.POINTER(C.c_int32)), ("sum_y", C.POINTER(C.c_int64)),
        ("npts", C.POINTER(C.c_int32)), ("offset", C.POINTER(C.c_int64)), ("npoints", C.c_int64),
        ("x", C.POINTER(C.c_int32)), ("y", C.POINTER(C.c_int32)),
    ]


class WbEventTable(C.Structure):
    _fields_ = [
        ("n", C.c_int64), ("step", C.POINTER(C.c_int32)), ("level_idx", C.POINTER(C.c_int32)),
        ("contour", C.POINTER(C.c_int64)), ("ind1", C.POINTER(C.c_int32)), ("ind2", C.POINTER(C.c_int32)),
        ("box", C.POINTER(C.c_int32)), ("orientation", C.POINTER(C.c_int32)), ("sums", C.POINTER(C.c_double)),
        ("ncells", C.POINTER(C.c_int32)), ("cell_offset", C.POINTER(C.c_int64)),
        ("cell_x", C.POINTER(C.c_int32)), ("cell_y", C.POINTER(C.c_int32)),
        ("poly_offset", C.POINTER(C.c_int64)), ("poly_x", C.POINTER(C.c_int32)), ("poly_y", C.POINTER(C.c_int32)),
        ("flags", C.POINTER(C.c_int32)), ("props", C.POINTER(C.c_double)),
    ]


class WbGeometry(C.Structure):
    _fields_ = [("n", C.c_int64), ("ring_offset", C.POINTER(C.c_int64)), ("nrings", C.c_int64),
                ("vert_offset", C.POINTER(C.c_int64)), ("x", C.POINTER(C.c_double)), ("y", C.POINTER(C.c_double))]


STREAMERS, OVERTURNINGS, CUTOFFS = 0, 1, 2
EVENT_NOT_SIMPLE = 1
ROW_WIDTH = 20
COMM_SLOTS = 4
STATUS_OVERFLOW_SEG, STATUS_LEVEL_HIT, STATUS_HAS_NAN, STATUS_HASH_FULL, STATUS_OVERFLOW_PAIRS, STATUS_LEVEL_ORDER = 1, 2, 4, 8, 16, 32
STATUS_LEVEL_KEYERROR, STATUS_LEVEL_WALKED, STATUS_SCAN_TIMEOUT = 64, 128, 256


def _arr(ptr, n, dtype, copy=True):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    a = np.ctypeslib.as_array(ptr, shape=(int(n),))
    return a.astype(dtype, copy=True) if copy else a


class Library:
    """One loaded copy of the C ABI."""

    def __init__(self, path=None, allow_hostsim=False):
        path = LIB_PATH if path is None else path
        if not os.path.exists(path):
            raise RuntimeError(
                "wavebreaking_amd: HIP library %s not found -- build it with `python -c 'import "
                "__graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        self.lib = lib = C.CDLL(path)
        lib.wb_backend.restype = C.c_char_p
        self.backend = lib.wb_backend().decode()
        if self.backend != "hip" and not allow_hostsim:
            raise RuntimeError("wavebreaking_amd: %s is a %r build, refusing to use it as the product path" % (path, self.backend))
        lib.wb_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.wb_destroy.argtypes = [C.c_void_p]
        lib.wb_last_error.argtypes = [C.c_void_p]
        lib.wb_last_error.restype = C.c_char_p
        lib.wb_contours.argtypes = [C.c_void_p, C.POINTER(WbParams), C.POINTER(C.c_void_p)]
        i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        lib.wb_batch_from_contours.argtypes = [C.c_void_p, C.POINTER(WbParams), C.c_int64, i32p, i32p, i32p, i64p, i32p,
                                               i32p, i32p, C.POINTER(C.c_void_p)]
        lib.wb_batch_max_nux.argtypes = [C.c_void_p]
        lib.wb_set_intensity.argtypes = [C.c_void_p, C.c_void_p]
        lib.wb_set_index_params.argtypes = [C.c_void_p] + [C.c_double] * 5
        lib.wb_indices.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.wb_get_contours.argtypes = [C.c_void_p, C.POINTER(WbContourTable)]
        lib.wb_get_events.argtypes = [C.c_void_p, C.c_int, C.POINTER(WbEventTable)]
        lib.wb_get_event_rows.argtypes = [C.c_void_p, C.c_int, C.POINTER(WbEventTable)]
        lib.wb_get_event_geometry.argtypes = [C.c_void_p, C.c_int, C.POINTER(WbGeometry)]
        lib.wb_transform_polygons.argtypes = [C.c_void_p, C.c_int64, i64p, i32p, i32p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                              C.c_int, C.c_int, C.POINTER(WbGeometry)]
        lib.wb_transform_polygons_device.argtypes = [C.c_void_p, C.c_int64, i64p, i32p, i32p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                     C.c_int, C.c_int, C.c_int, C.POINTER(WbGeometry)]
        lib.wb_get_status.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_int64)]
        lib.wb_get_band_hits.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.wb_probe_read_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        lib.wb_probe_read_ms.restype = C.c_double
        lib.wb_probe_streams_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
        lib.wb_probe_streams_ms.restype = C.c_double
        lib.wb_get_pair_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.wb_set_step_offset.argtypes = [C.c_void_p, C.c_int64]
        lib.wb_get_device_tables.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        lib.wb_get_group_ties.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        lib.wb_get_evaluated_pairs.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        lib.wb_get_com_rechecks.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.wb_get_timings.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int)]
        lib.wb_free_batch.argtypes = [C.c_void_p]
        lib.wb_dev_alloc.argtypes = [C.c_void_p, C.c_uint64]
        lib.wb_dev_alloc.restype = C.c_void_p
        lib.wb_dev_free.argtypes = [C.c_void_p, C.c_void_p]
        lib.wb_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        lib.wb_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        lib.wb_synth_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double,
                                      C.c_uint64]
        dp = C.POINTER(C.c_double)
        lib.wb_smooth_field.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, dp,
                                        C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_int]
        lib.wb_momentum_flux.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int,
                                         C.c_void_p, C.c_int]
        lib.wb_flag_events.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, dp, dp, C.c_double, C.c_int64, C.POINTER(C.c_int32),
                                       dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), dp, dp, C.c_int, C.c_void_p, C.c_int]
        i64p, i32p = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
        lib.wb_track_events.argtypes = [C.c_void_p, C.c_int64, i64p, dp, C.c_int, C.c_double, C.c_double, C.c_double, dp, i64p,
                                        i64p, i32p, i32p, i32p, dp, C.POINTER(C.c_uint64)]
        lib.wb_track_overlaps.argtypes = [C.c_void_p, C.c_int64, i64p, i64p, i32p, i32p, C.c_int64, i64p, i64p, dp, dp]
        lib.wb_track_events_buffered.argtypes = [C.c_void_p, C.c_int64, i64p, dp, C.c_double, C.c_double, C.c_double, C.c_double, i64p, i64p,
                                                 i32p, i32p, i32p, C.POINTER(C.c_uint64)]
        lib.wb_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        lib.wb_comm_unique_id.argtypes = [C.c_void_p, C.c_void_p]
        lib.wb_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.wb_comm_allreduce_max.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        lib.wb_comm_gather_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int64),
                                             C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int64),
                                             C.POINTER(C.c_int)]
        lib.wb_comm_stage_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.wb_comm_gather_staged.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int64),
                                             C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int64),
                                             C.POINTER(C.c_int)]
        lib.wb_comm_destroy.argtypes = [C.c_void_p]
        lib.wb_comm_destroy.restype = None
        lib.wb_comm_abort.argtypes = [C.c_void_p]
        lib.wb_comm_reserve.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        lib.wb_last_op_ms.argtypes = [C.c_void_p]
        lib.wb_last_op_ms.restype = C.c_float
        lib.wb_fingerprint.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        lib.wb_fingerprint.restype = C.c_uint64

    def fingerprint(self, arr, threads=0):
        """64-bit hash of every byte of a C-contiguous numpy array (wb_fingerprint: host threads, no device work)"""
        return int(self.lib.wb_fingerprint(C.c_void_p(arr.__array_interface__["data"][0]), int(arr.nbytes), int(threads)))


_default = None


def default_library():
    global _default
    if _default is None:
        _default = Library()
    return _default


class Context:
    """wb_ctx: one HIP stream on one device."""

    def __init__(self, device=0, library=None):
        self.L = default_library() if library is None else library
        h = C.c_void_p()
        if self.L.lib.wb_create(int(device), C.byref(h)) != 0:
            raise RuntimeError("wavebreaking_amd: wb_create(device=%d) failed -- no usable MI355X visible" % device)
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            self.L.lib.wb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def error(self):
        return self.L.lib.wb_last_error(self.h).decode()

    def set_option(self, name, value):
        if self.L.lib.wb_set_option(self.h, name.encode(), int(value)) != 0:
            raise RuntimeError("wb_set_option failed: " + self.error())

    # -- device memory helpers (bench / tests)
    def alloc(self, nbytes):
        p = self.L.lib.wb_dev_alloc(self.h, int(nbytes))
        if not p:
            raise MemoryError("wb_dev_alloc(%d) failed: %s" % (nbytes, self.error()))
        return p

    def free(self, p):
        self.L.lib.wb_dev_free(self.h, C.c_void_p(p))

    def probe_streams_ms(self, p, nbytes, chunk_bytes, grid=0, reps=3):
        """milliseconds of a read by `grid` workgroups each streaming chunks of chunk_bytes (best of reps)"""
        return float(self.L.lib.wb_probe_streams_ms(self.h, C.c_void_p(p), int(nbytes), int(chunk_bytes), int(grid), int(reps)))

    def probe_read_ms(self, p, nbytes, reps=3):
        """milliseconds of one streaming read of a device buffer (best of reps)"""
        return float(self.L.lib.wb_probe_read_ms(self.h, C.c_void_p(p), int(nbytes), int(reps)))

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        if self.L.lib.wb_memcpy_h2d(self.h, C.c_void_p(p), arr.ctypes.data_as(C.c_void_p), arr.nbytes) != 0:
            raise RuntimeError("h2d failed")
        return p

    def copy_to_device(self, p, arr):
        """host array -> device memory at p (any context's allocation on this device)"""
        arr = np.ascontiguousarray(arr)
        if self.L.lib.wb_memcpy_h2d(self.h, C.c_void_p(p), arr.ctypes.data_as(C.c_void_p), arr.nbytes) != 0:
            raise RuntimeError("h2d failed: " + self.error())

    def to_host(self, p, shape, dtype, offset_bytes=0):
        out = np.empty(shape, dtype=dtype)
        if self.L.lib.wb_memcpy_d2h(self.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(p + offset_bytes), out.nbytes) != 0:
            raise RuntimeError("d2h failed")
        return out

    def synth_fill(self, dev_ptr, dtype, t0, nt, lat, lon, wave_params, eps, seed):
        lat = np.ascontiguousarray(lat, dtype=np.float64)
        lon = np.ascontiguousarray(lon, dtype=np.float64)
        wp = np.ascontiguousarray(wave_params, dtype=np.float64)
        assert wp.shape == (7, 5)
        dp = C.POINTER(C.c_double)
        rc = self.L.lib.wb_synth_fill(self.h, C.c_void_p(dev_ptr), 0 if np.dtype(dtype) == np.float32 else 1, int(t0),
                                      int(nt), lat.size, lon.size, lat.ctypes.data_as(dp), lon.ctypes.data_as(dp),
                                      wp.ctypes.data_as(dp), float(eps), int(seed))
        if rc != 0:
            raise RuntimeError("wb_synth_fill failed: " + self.error())

    # -- pre-processing (processing/spatial.py)
    SMOOTH_MODES = {"reflect": 0, "grid-mirror": 0, "constant": 1, "grid-constant": 1, "nearest": 2, "mirror": 3,
                    "wrap": 4, "grid-wrap": 4}

    def smooth_field(self, field, passes, weights, mode="wrap", device_shape=None, device_dtype=None, out_device=None,
                     per_pass=False):
        """``passes`` x (scipy.ndimage.convolve / np.sum(weights)) per time step + NaN border rows -> float64 cube.
        ``field``: numpy (time, lat, lon) array or, with device_shape/device_dtype, a device pointer;
        ``out_device``: device pointer of a float64 cube to fill instead of returning a numpy array."""
        if mode not in self.SMOOTH_MODES:
            raise RuntimeError("wb_smooth_field: unknown boundary mode %r" % (mode,))
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if w.ndim != 2:
            raise RuntimeError("wb_smooth_field: weights must be two-dimensional")
        wsum = float(np.sum(weights))
        if device_shape is not None:
            nt, nlat, nlon = device_shape
            dt = np.dtype(device_dtype)
            src, on_dev = C.c_void_p(int(field)), 1
        else:
            field = np.ascontiguousarray(field)
            if field.dtype not in (np.float32, np.float64):
                field = field.astype(np.float64)
            nt, nlat, nlon = field.shape
            dt = field.dtype
            src, on_dev = field.ctypes.data_as(C.c_void_p), 0
        out = None
        if out_device is None:
            out = np.empty((nt, nlat, nlon), dtype=np.float64)
            dst, out_dev = out.ctypes.data_as(C.c_void_p), 0
        else:
            dst, out_dev = C.c_void_p(int(out_device)), 1
        rc = self.L.lib.wb_smooth_field(self.h, src, 0 if dt == np.float32 else 1, on_dev, int(nt), int(nlat), int(nlon),
                                        int(passes), w.ctypes.data_as(C.POINTER(C.c_double)), w.shape[0], w.shape[1], wsum,
                                        self.SMOOTH_MODES[mode], dst, out_dev, 1 if per_pass else 0)
        if rc != 0:
            raise RuntimeError("wb_smooth_field failed: " + self.error())
        return out

    def momentum_flux(self, u, v, device_shape=None, device_dtype=None, out_device=None):
        """(u - nanmean_lon u) * (v - nanmean_lon v) in u's dtype."""
        if device_shape is not None:
            nt, nlat, nlon = device_shape
            dt = np.dtype(device_dtype)
            pu, pv, on_dev = C.c_void_p(int(u)), C.c_void_p(int(v)), 1
        else:
            u = np.ascontiguousarray(u)
            if u.dtype not in (np.float32, np.float64):
                u = u.astype(np.float64)
            v = np.ascontiguousarray(v, dtype=u.dtype)
            if u.shape != v.shape or u.ndim != 3:
                raise RuntimeError("wb_momentum_flux: u and v must be (time, lat, lon) cubes of one shape")
            nt, nlat, nlon = u.shape
            dt = u.dtype
            pu, pv, on_dev = u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), 0
        out = None
        if out_device is None:
            out = np.empty((nt, nlat, nlon), dtype=dt)
            dst, out_dev = out.ctypes.data_as(C.c_void_p), 0
        else:
            dst, out_dev = C.c_void_p(int(out_device)), 1
        rc = self.L.lib.wb_momentum_flux(self.h, pu, pv, 0 if dt == np.float32 else 1, on_dev, int(nt), int(nlat), int(nlon),
                                         dst, out_dev)
        if rc != 0:
            raise RuntimeError("wb_momentum_flux failed: " + self.error())
        return out

    def flag_events(self, nt, lat, lon, radius, steps, rings_per_event, values=None, dtype=np.int8, out_device=None):
        """to_xarray's rule: cube (nt, nlat, nlon) with ``values[k]`` (1) at every grid point contained in event k's
        geometry buffered by ``radius``; ``rings_per_event[k]`` = list of (n, 2) lon/lat rings; lat / lon ascending."""
        lat = np.ascontiguousarray(lat, dtype=np.float64)
        lon = np.ascontiguousarray(lon, dtype=np.float64)
        nev = len(rings_per_event)
        ring_off = np.zeros(nev + 1, dtype=np.int64)
        vert_off = [0]
        xs, ys = [], []
        for k, rings in enumerate(rings_per_event):
            for ring in rings:
                ring = np.asarray(ring, dtype=np.float64).reshape(-1, 2)
                xs.append(ring[:, 0])
                ys.append(ring[:, 1])
                vert_off.append(vert_off[-1] + len(ring))
            ring_off[k + 1] = len(vert_off) - 1
        vert_off = np.asarray(vert_off, dtype=np.int64)
        vx = np.ascontiguousarray(np.concatenate(xs) if xs else np.zeros(0))
        vy = np.ascontiguousarray(np.concatenate(ys) if ys else np.zeros(0))
        st = np.ascontiguousarray(steps, dtype=np.int32)
        val = None if values is None else np.ascontiguousarray(values, dtype=np.float64)
        dt = np.dtype(dtype)
        code = {np.dtype(np.int8): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}[dt]
        out = None
        if out_device is None:
            out = np.empty((int(nt), lat.size, lon.size), dtype=dt)
            dst, on_dev = out.ctypes.data_as(C.c_void_p), 0
        else:
            dst, on_dev = C.c_void_p(int(out_device)), 1
        dp = C.POINTER(C.c_double)
        rc = self.L.lib.wb_flag_events(self.h, int(nt), lat.size, lon.size, lat.ctypes.data_as(dp), lon.ctypes.data_as(dp),
                                       float(radius), nev, st.ctypes.data_as(C.POINTER(C.c_int32)),
                                       val.ctypes.data_as(dp) if val is not None else None,
                                       ring_off.ctypes.data_as(C.POINTER(C.c_int64)), vert_off.ctypes.data_as(C.POINTER(C.c_int64)),
                                       vx.ctypes.data_as(dp), vy.ctypes.data_as(dp), code, dst, on_dev)
        if rc != 0:
            raise RuntimeError("wb_flag_events failed: " + self.error())
        return out

    @staticmethod
    def _lattice_rings(rings_per_event):
        """list (per event) of lists of (n, 2) integer rings -> ring_offset, vert_offset, vx, vy"""
        nev = len(rings_per_event)
        ring_off = np.zeros(nev + 1, dtype=np.int64)
        vert_off = [0]
        parts = []
        for k, rings in enumerate(rings_per_event):
            for ring in rings:
                ring = np.asarray(ring, dtype=np.int32).reshape(-1, 2)
                parts.append(ring)
                vert_off.append(vert_off[-1] + len(ring))
            ring_off[k + 1] = len(vert_off) - 1
        xy = np.concatenate(parts) if parts else np.zeros((0, 2), dtype=np.int32)
        return (ring_off, np.asarray(vert_off, dtype=np.int64), np.ascontiguousarray(xy[:, 0], dtype=np.int32),
                np.ascontiguousarray(xy[:, 1], dtype=np.int32))

    def track_events_buffered(self, dates, time_range, buffer, steps, rings_per_event=None, rings_csr=None):
        """wb_track_events_buffered: labels of track_events(method="by_overlap", buffer=buffer > 0, overlap=0) -- a pair in time range
        is linked when the planar distance of its polygons, in degrees, is below 2 * buffer.  ``steps`` = (degrees per grid step in
        x, in y); rings as for ``track_events``.  Returns (label int32[n], stats dict)."""
        d = np.asarray(dates)
        n = d.shape[0]
        i64p, i32p, dp = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)
        ns = val = None
        if np.issubdtype(d.dtype, np.datetime64):
            ns = np.ascontiguousarray(d.astype("datetime64[ns]").astype(np.int64))
        else:
            val = np.ascontiguousarray(d, dtype=np.float64)
        if rings_csr is not None:
            ring_off, vert_off, vx, vy = (np.ascontiguousarray(rings_csr[0], dtype=np.int64), np.ascontiguousarray(rings_csr[1], dtype=np.int64),
                                          np.ascontiguousarray(rings_csr[2], dtype=np.int32), np.ascontiguousarray(rings_csr[3], dtype=np.int32))
        else:
            ring_off, vert_off, vx, vy = self._lattice_rings(rings_per_event)
        assert ring_off.size == n + 1
        label = np.empty(n, dtype=np.int32)
        stats = np.zeros(6, dtype=np.uint64)
        rc = self.L.lib.wb_track_events_buffered(self.h, n, ns.ctypes.data_as(i64p) if ns is not None else None,
                                                 val.ctypes.data_as(dp) if val is not None else None, float(time_range), float(buffer),
                                                 float(steps[0]), float(steps[1]), ring_off.ctypes.data_as(i64p), vert_off.ctypes.data_as(i64p),
                                                 vx.ctypes.data_as(i32p), vy.ctypes.data_as(i32p), label.ctypes.data_as(i32p),
                                                 stats.ctypes.data_as(C.POINTER(C.c_uint64)))
        if rc != 0:
            raise RuntimeError("wb_track_events_buffered failed: " + self.error())
        return label, dict(pairs_in_range=int(stats[0]), pairs_boxes_within_reach=int(stats[1]), pairs_linked=int(stats[2]),
                           polygons_not_simple=int(stats[3]), pairs_linked_by_distance=int(stats[4]), pairs_in_arc_band=int(stats[5]))

    def track_events(self, dates, method, time_range, overlap=0.0, distance=1000.0, com=None, rings_per_event=None, rings_csr=None):
        """wb_track_events: labels of track_events (events.py:113-241) for events in the caller's row order.  ``dates``:
        datetime64 (any unit) or numbers; ``method``: "by_overlap" (``rings_per_event[k]`` = list of (n, 2) LATTICE rings, or
        ``rings_csr`` = (ring_offset[n + 1], vert_offset[rings + 1], x, y) with integer x, y -- the layout of
        ``Batch.event_geometry``) or "by_distance" (``com`` = (n, 2) degrees).  Returns (label int32[n], area float64[n] or
        None, stats dict)."""
        d = np.asarray(dates)
        n = d.shape[0]
        i64p, i32p, dp = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)
        ns = val = None
        if np.issubdtype(d.dtype, np.datetime64):
            ns = np.ascontiguousarray(d.astype("datetime64[ns]").astype(np.int64))
        else:
            val = np.ascontiguousarray(d, dtype=np.float64)
        label = np.empty(n, dtype=np.int32)
        stats = np.zeros(4, dtype=np.uint64)
        area = None
        if method == "by_overlap":
            if rings_csr is not None:
                ring_off = np.ascontiguousarray(rings_csr[0], dtype=np.int64)
                vert_off = np.ascontiguousarray(rings_csr[1], dtype=np.int64)
                vx = np.ascontiguousarray(rings_csr[2], dtype=np.int32)
                vy = np.ascontiguousarray(rings_csr[3], dtype=np.int32)
            else:
                ring_off, vert_off, vx, vy = self._lattice_rings(rings_per_event)
            assert ring_off.size == n + 1
            area = np.empty(n, dtype=np.float64)
            geo = (ring_off.ctypes.data_as(i64p), vert_off.ctypes.data_as(i64p), vx.ctypes.data_as(i32p), vy.ctypes.data_as(i32p))
            cm, code = None, 0
        elif method == "by_distance":
            cm = np.ascontiguousarray(com, dtype=np.float64).reshape(n, 2)
            geo, code = (None, None, None, None), 1
        else:
            raise ValueError("'{}' not supported as method!".format(method))
        rc = self.L.lib.wb_track_events(self.h, n, ns.ctypes.data_as(i64p) if ns is not None else None,
                                        val.ctypes.data_as(dp) if val is not None else None, code, float(time_range),
                                        float(overlap), float(distance), cm.ctypes.data_as(dp) if cm is not None else None,
                                        *geo, label.ctypes.data_as(i32p), area.ctypes.data_as(dp) if area is not None else None,
                                        stats.ctypes.data_as(C.POINTER(C.c_uint64)))
        if rc != 0:
            raise RuntimeError("wb_track_events failed: " + self.error())
        return label, area, dict(pairs_in_range=int(stats[0]), pairs_boxes_overlap=int(stats[1]), pairs_linked=int(stats[2]),
                                 polygons_not_simple=int(stats[3]))

    def track_overlaps(self, rings_per_event, pairs):
        """wb_track_overlaps: (intersection area of every pair, area of every event), index units"""
        ring_off, vert_off, vx, vy = self._lattice_rings(rings_per_event)
        n = ring_off.size - 1
        pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
        pi, pj = np.ascontiguousarray(pairs[:, 0]), np.ascontiguousarray(pairs[:, 1])
        inter = np.zeros(len(pairs), dtype=np.float64)
        area = np.empty(n, dtype=np.float64)
        i64p, i32p, dp = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)
        rc = self.L.lib.wb_track_overlaps(self.h, n, ring_off.ctypes.data_as(i64p), vert_off.ctypes.data_as(i64p),
                                          vx.ctypes.data_as(i32p), vy.ctypes.data_as(i32p), len(pairs), pi.ctypes.data_as(i64p),
                                          pj.ctypes.data_as(i64p), inter.ctypes.data_as(dp), area.ctypes.data_as(dp))
        if rc != 0:
            raise RuntimeError("wb_track_overlaps failed: " + self.error())
        return inter, area

    def last_op_ms(self):
        return float(self.L.lib.wb_last_op_ms(self.h))

    # -- time-sharded runs: the two exchanges on RCCL inside the library (wb_comm_*)
    COMM_ID_BYTES = 128

    def comm_unique_id(self):
        buf = C.create_string_buffer(self.COMM_ID_BYTES)
        if self.L.lib.wb_comm_unique_id(self.h, buf) != 0:
            raise RuntimeError("wb_comm_unique_id failed: " + self.error())
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        assert len(unique_id) == self.COMM_ID_BYTES
        if self.L.lib.wb_comm_init(self.h, C.c_char_p(bytes(unique_id)), int(rank), int(world)) != 0:
            raise RuntimeError("wb_comm_init failed: " + self.error())

    def comm_allreduce_max(self, value):
        v = C.c_int(int(value))
        if self.L.lib.wb_comm_allreduce_max(self.h, C.byref(v)) != 0:
            raise RuntimeError("wb_comm_allreduce_max failed: " + self.error())
        return int(v.value)

    def comm_gather_tables(self, batch, dst=0):
        """-> (rows (n, 20) float64, poly_x, poly_y, ranks seen): views of the context's pinned buffers on rank ``dst`` (valid
        until the next call), empty arrays elsewhere; ``batch`` may be None on a rank without time steps"""
        rows, px, py = C.POINTER(C.c_double)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        n, nv, seen = C.c_int64(), C.c_int64(), C.c_int()
        if self.L.lib.wb_comm_gather_tables(self.h, batch.h if batch is not None else None, int(dst), C.byref(rows), C.byref(n),
                                            C.byref(px), C.byref(py), C.byref(nv), C.byref(seen)) != 0:
            raise RuntimeError("wb_comm_gather_tables failed: " + self.error())
        r = _arr(rows, n.value * ROW_WIDTH, np.float64, copy=False).reshape(-1, ROW_WIDTH)
        return r, _arr(px, nv.value, np.int32, copy=False), _arr(py, nv.value, np.int32, copy=False), int(seen.value)

    def comm_stage_tables(self, batch, slot):
        """copy the batch's device tables (None: an empty shard) into slot ``slot`` (< COMM_SLOTS) of the communicator; the
        batch may be closed when this returns"""
        if self.L.lib.wb_comm_stage_tables(self.h, batch.h if batch is not None else None, int(slot)) != 0:
            raise RuntimeError("wb_comm_stage_tables failed: " + self.error())

    def comm_gather_staged(self, slot, dst=0):
        """comm_gather_tables for the tables staged in ``slot``; may run on another thread than the one driving the batches"""
        rows, px, py = C.POINTER(C.c_double)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        n, nv, seen = C.c_int64(), C.c_int64(), C.c_int()
        if self.L.lib.wb_comm_gather_staged(self.h, int(slot), int(dst), C.byref(rows), C.byref(n), C.byref(px), C.byref(py),
                                            C.byref(nv), C.byref(seen)) != 0:
            raise RuntimeError("wb_comm_gather_staged failed: " + self.error())
        r = _arr(rows, n.value * ROW_WIDTH, np.float64, copy=False).reshape(-1, ROW_WIDTH)
        return r, _arr(px, nv.value, np.int32, copy=False), _arr(py, nv.value, np.int32, copy=False), int(seen.value)

    def comm_abort(self):
        """ncclCommAbort: peers blocked in a collective with this rank fail instead of waiting for ever"""
        if self.h:
            self.L.lib.wb_comm_abort(self.h)

    def comm_reserve(self, rows, verts):
        """the destination rank's buffers for the whole group's catalogue, before the record's batches are in flight"""
        if self.L.lib.wb_comm_reserve(self.h, int(rows), int(verts)) != 0:
            raise RuntimeError("wb_comm_reserve failed: " + self.error())

    def comm_destroy(self):
        if self.h:
            self.L.lib.wb_comm_destroy(self.h)

    def props_columns(self, sums, lon, lat):
        """the frame's property columns from the five per-event sums, by the device function wb_indices runs for every event
        (index_utils.py:104-120): (n, 5) float64 -- com longitude, com latitude, mean_var, intensity, event_area"""
        sums = np.ascontiguousarray(sums, dtype=np.float64).reshape(-1, 5)
        lon = np.ascontiguousarray(lon, dtype=np.float64)
        lat = np.ascontiguousarray(lat, dtype=np.float64)
        out = np.zeros_like(sums)
        dp = C.POINTER(C.c_double)
        f = self.L.lib.wb_test_props_columns
        f.argtypes = [C.c_void_p, C.c_int64, dp, dp, C.c_int, dp, C.c_int, dp]
        if f(self.h, len(sums), sums.ctypes.data_as(dp), lon.ctypes.data_as(dp), int(lon.size), lat.ctypes.data_as(dp), int(lat.size), out.ctypes.data_as(dp)) != 0:
            raise RuntimeError("wb_test_props_columns failed: " + self.error())
        return out

    def transform_polygons(self, polygons, lon, lat, nlon, index_space=False, device=False):
        """wb_transform_polygons: index-lattice polygons (list of (n, 2) integer arrays) -> per polygon the list of its rings
        ((m, 2) float64 arrays) in (lon, lat) -- or, with ``index_space``, in index coordinates as they are cut at the seam.
        ``device``: through the kernels wb_indices runs for a pass's events (wb_transform_polygons_device) instead of the host
        function"""
        polys = [np.asarray(q).astype("int").reshape(-1, 2) for q in polygons]
        off = np.zeros(len(polys) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(q) for q in polys])
        xy = np.concatenate(polys) if polys else np.zeros((0, 2), dtype=np.int64)
        px = np.ascontiguousarray(xy[:, 0], dtype=np.int32)
        py = np.ascontiguousarray(xy[:, 1], dtype=np.int32)
        lon = np.ascontiguousarray(lon, dtype=np.float64)
        lat = np.ascontiguousarray(lat, dtype=np.float64)
        dp, i32p, i64p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        g = WbGeometry()
        if device:
            rc = self.L.lib.wb_transform_polygons_device(self.h, len(polys), off.ctypes.data_as(i64p), px.ctypes.data_as(i32p), py.ctypes.data_as(i32p),
                                                         lon.ctypes.data_as(dp), lat.ctypes.data_as(dp), int(nlon), int(lat.size),
                                                         1 if index_space else 0, C.byref(g))
        else:
            rc = self.L.lib.wb_transform_polygons(self.h, len(polys), off.ctypes.data_as(i64p), px.ctypes.data_as(i32p), py.ctypes.data_as(i32p),
                                                  lon.ctypes.data_as(dp), lat.ctypes.data_as(dp), int(nlon), 1 if index_space else 0,
                                                  C.byref(g))
        if rc != 0:
            raise RuntimeError("wb_transform_polygons failed: " + self.error())
        ro = _arr(g.ring_offset, g.n + 1, np.int64)
        vo = _arr(g.vert_offset, g.nrings + 1, np.int64)
        nv = int(vo[-1]) if vo.size else 0
        co = np.c_[_arr(g.x, nv, np.float64), _arr(g.y, nv, np.float64)]
        return [[co[vo[q]:vo[q + 1]] for q in range(ro[k], ro[k + 1])] for k in range(len(polys))]

    # -- the hot path
    def _params(self, field, lat, lon, levels, dlon, dlat, periodic_add, intensity, geo_dis, cont_dis, range_group,
                ot_min_exp, co_min_exp, want_cells, device_shape, device_dtype, cap_seg, cap_pairs):
        keep = []
        p = WbParams()
        if device_shape is not None:
            nt, nlat, nlon = device_shape
            dt = np.dtype(device_dtype)
            p.on_device = 1
            p.field = C.c_void_p(int(field))
            p.intensity = C.c_void_p(int(intensity)) if intensity is not None else None
        else:
            field = np.ascontiguousarray(field)
            if field.dtype not in (np.float32, np.float64):
                field = field.astype(np.float64)
            dt = field.dtype
            nt, nlat, nlon = field.shape
            keep.append(field)
            p.on_device = 0
            p.field = field.ctypes.data_as(C.c_void_p)
            if intensity is not None:
                intensity = np.ascontiguousarray(intensity, dtype=dt)
                assert intensity.shape == field.shape
                keep.append(intensity)
                p.intensity = intensity.ctypes.data_as(C.c_void_p)
        lat = np.ascontiguousarray(lat, dtype=np.float64)
        lon = np.ascontiguousarray(lon, dtype=np.float64)
        lev = np.ascontiguousarray(levels, dtype=np.float64)
        keep += [lat, lon, lev]
        assert lat.size == nlat and lon.size == nlon
        dp = C.POINTER(C.c_double)
        p.nt, p.nlat, p.nlon = int(nt), int(nlat), int(nlon)
        p.dtype = 0 if dt == np.float32 else 1
        p.lat, p.lon, p.levels = lat.ctypes.data_as(dp), lon.ctypes.data_as(dp), lev.ctypes.data_as(dp)
        p.nlevels = lev.size
        p.periodic_add, p.dlon, p.dlat = float(periodic_add), float(dlon), float(dlat)
        p.geo_dis, p.cont_dis = float(geo_dis), float(cont_dis)
        p.range_group, p.ot_min_exp, p.co_min_exp = float(range_group), float(ot_min_exp), float(co_min_exp)
        p.want_cells = 1 if want_cells else 0
        p.cap_seg, p.cap_pairs = int(cap_seg), int(cap_pairs)
        return p, keep, int(nt), lev.size

    def contours(self, field, lat, lon, levels, dlon, dlat, periodic_add=120.0, intensity=None, geo_dis=800.0,
                 cont_dis=1500.0, range_group=5.0, ot_min_exp=5.0, co_min_exp=5.0, want_cells=False,
                 device_shape=None, device_dtype=None, cap_seg=0, cap_pairs=0):
        """Phase 1.  ``field`` is a numpy array (time, lat, lon) or, with ``device_shape``/``device_dtype``,
        an integer device pointer.  Returns a Batch."""
        p, keep, nt, nlev = self._params(field, lat, lon, levels, dlon, dlat, periodic_add, intensity, geo_dis, cont_dis,
                                         range_group, ot_min_exp, co_min_exp, want_cells, device_shape, device_dtype,
                                         cap_seg, cap_pairs)
        h = C.c_void_p()
        if self.L.lib.wb_contours(self.h, C.byref(p), C.byref(h)) != 0:
            raise RuntimeError("wb_contours failed: " + self.error())
        return Batch(self, h, nt, nlev, want_cells)

    def batch_from_contours(self, table, field, lat, lon, levels, dlon, dlat, periodic_add=120.0, intensity=None,
                            geo_dis=800.0, cont_dis=1500.0, range_group=5.0, ot_min_exp=5.0, co_min_exp=5.0,
                            want_cells=False, device_shape=None, device_dtype=None, cap_pairs=0):
        """Phase 1 from a contour frame computed earlier: ``table`` has step, level_idx, closed, offset, npts, x, y."""
        p, keep, nt, nlev = self._params(field, lat, lon, levels, dlon, dlat, periodic_add, intensity, geo_dis, cont_dis,
                                         range_group, ot_min_exp, co_min_exp, want_cells, device_shape, device_dtype,
                                         0, cap_pairs)
        i32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)
        step, lvl, closed, npts = i32(table["step"]), i32(table["level_idx"]), i32(table["closed"]), i32(table["npts"])
        off = np.ascontiguousarray(table["offset"], dtype=np.int64)
        x, y = i32(table["x"]), i32(table["y"])
        i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        h = C.c_void_p()
        rc = self.L.lib.wb_batch_from_contours(self.h, C.byref(p), step.size, step.ctypes.data_as(i32p),
                                               lvl.ctypes.data_as(i32p), closed.ctypes.data_as(i32p),
                                               off.ctypes.data_as(i64p), npts.ctypes.data_as(i32p),
                                               x.ctypes.data_as(i32p), y.ctypes.data_as(i32p), C.byref(h))
        if rc != 0:
            raise RuntimeError("wb_batch_from_contours failed: " + self.error())
        return Batch(self, h, nt, nlev, want_cells)


class Batch:
    """wb_batch: the contours of one resident batch of time steps, then its index events."""

    def __init__(self, ctx, h, nt, nlev, want_cells):
        self.ctx, self.h, self.nt, self.nlev, self.want_cells = ctx, h, nt, nlev, want_cells

    def close(self):
        if self.h:
            if self.ctx.h:                      # a batch that outlived its context was released with it
                self.ctx.L.lib.wb_free_batch(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def max_nux(self):
        return int(self.ctx.L.lib.wb_batch_max_nux(self.h))

    def set_index_params(self, geo_dis=None, cont_dis=None, range_group=None, ot_min_exp=None, co_min_exp=None):
        """Parameters of the indices for a batch whose contours exist already (None keeps the current value)."""
        v = [float('nan') if x is None else float(x) for x in (geo_dis, cont_dis, range_group, ot_min_exp, co_min_exp)]
        self.ctx.L.lib.wb_set_index_params(self.h, *v)

    def set_intensity(self, dev_ptr):
        """intensity cube (device pointer, the field's shape and dtype, or None) for the next indices() of a batch whose
        field is device-resident"""
        if self.ctx.L.lib.wb_set_intensity(self.h, C.c_void_p(int(dev_ptr)) if dev_ptr else None) != 0:
            raise RuntimeError("wb_set_intensity failed: " + self.ctx.error())

    def indices(self, global_max_nux=None, streamers=True, overturnings=True, cutoffs=True):
        g = self.max_nux() if global_max_nux is None else int(global_max_nux)
        which = (1 if streamers else 0) | (2 if overturnings else 0) | (4 if cutoffs else 0)
        if self.ctx.L.lib.wb_indices(self.h, g, which) != 0:
            raise RuntimeError("wb_indices failed: " + self.ctx.error())

    def contour_table(self):
        t = WbContourTable()
        if self.ctx.L.lib.wb_get_contours(self.h, C.byref(t)) != 0:
            raise RuntimeError("wb_get_contours failed: " + self.ctx.error())
        n, m = t.n, t.npoints
        return dict(step=_arr(t.step, n, np.int32), level_idx=_arr(t.level_idx, n, np.int32),
                    closed=_arr(t.closed, n, np.int32).astype(bool), nux=_arr(t.nux, n, np.int32),
                    sum_y=_arr(t.sum_y, n, np.int64), npts=_arr(t.npts, n, np.int32), offset=_arr(t.offset, n, np.int64),
                    x=_arr(t.x, m, np.int32), y=_arr(t.y, m, np.int32))

    def events(self, kind, copy=True, polygons=True):
        """Event table of one index.  ``copy=False`` hands out views of the library's host tables (valid until the
        batch is closed or ``indices`` runs again) instead of private copies.  ``polygons=False``: without the polygon
        vertices in grid indices (``poly_x`` / ``poly_y`` empty, ``poly_offset`` still the vertex counts) -- what a caller
        wants who takes the polygons in lon / lat from ``event_geometry``: ``indices`` leaves the index polygons on the device,
        and asking for them (the default here) packs them into host memory first (wb_get_events / wb_get_event_rows)."""
        t = WbEventTable()
        get = self.ctx.L.lib.wb_get_events if polygons else self.ctx.L.lib.wb_get_event_rows
        if get(self.h, int(kind), C.byref(t)) != 0:
            raise RuntimeError("wb_get_events failed: " + self.ctx.error())
        n = t.n
        A = lambda p, m, dt: _arr(p, m, dt, copy)
        out = dict(step=A(t.step, n, np.int32), level_idx=A(t.level_idx, n, np.int32),
                   contour=A(t.contour, n, np.int64), ind1=A(t.ind1, n, np.int32), ind2=A(t.ind2, n, np.int32),
                   box=A(t.box, 4 * n, np.int32).reshape(-1, 4), orientation=A(t.orientation, n, np.int32),
                   sums=A(t.sums, 5 * n, np.float64).reshape(-1, 5), ncells=A(t.ncells, n, np.int32),
                   flags=A(t.flags, n, np.int32),
                   # the frame's property columns from the device (index_utils.py:104-120): com lon, com lat, mean_var, intensity, event_area
                   props=A(t.props, 5 * n, np.float64).reshape(-1, 5))
        poff = A(t.poly_offset, n + 1, np.int64)
        out["poly_offset"] = poff
        out["poly_x"] = A(t.poly_x, poff[-1] if (n and polygons) else 0, np.int32)
        out["poly_y"] = A(t.poly_y, poff[-1] if (n and polygons) else 0, np.int32)
        if self.want_cells and t.cell_offset:
            off = A(t.cell_offset, n + 1, np.int64)
            out["cell_offset"] = off
            out["cell_x"] = A(t.cell_x, off[-1] if n else 0, np.int32)
            out["cell_y"] = A(t.cell_y, off[-1] if n else 0, np.int32)
        return out

    def event_geometry(self, kind, copy=True):
        """transform_polygons for the whole table: (ring_offset per event, vert_offset per ring, lon, lat) arrays.  ``copy=False``
        hands out views of the library's host arrays -- the pinned blocks the device wrote the rings into during indices() --
        valid until the batch's next indices() / close()."""
        g = WbGeometry()
        if self.ctx.L.lib.wb_get_event_geometry(self.h, int(kind), C.byref(g)) != 0:
            raise RuntimeError("wb_get_event_geometry failed: " + self.ctx.error())
        voff = _arr(g.vert_offset, g.nrings + 1, np.int64, copy)
        nv = int(voff[-1]) if voff.size else 0
        return _arr(g.ring_offset, g.n + 1, np.int64, copy), voff, _arr(g.x, nv, np.float64, copy), _arr(g.y, nv, np.float64, copy)

    def status(self):
        p = C.POINTER(C.c_uint32)()
        n = C.c_int64()
        self.ctx.L.lib.wb_get_status(self.h, C.byref(p), C.byref(n))
        return _arr(p, n.value, np.uint32)

    def band_hits(self):
        g, c = C.c_int64(), C.c_int64()
        self.ctx.L.lib.wb_get_band_hits(self.h, C.byref(g), C.byref(c))
        return int(g.value), int(c.value)

    def group_ties(self):
        """check_groups groups of the last indices() call whose largest length was shared by several members exactly"""
        v = C.c_int64()
        self.ctx.L.lib.wb_get_group_ties(self.h, C.byref(v))
        return int(v.value)

    def com_rechecks(self):
        """(rasterised events, boxes, boxes walked by one lane) whose areas / x_com / y_com were re-evaluated cell by cell
        in the reference's order because x_com / areas or y_com / areas sat next to an integer (last indices() call)."""
        a, c, o = C.c_int64(), C.c_int64(), C.c_int64()
        self.ctx.L.lib.wb_get_com_rechecks(self.h, C.byref(a), C.byref(c), C.byref(o))
        return int(a.value), int(c.value), int(o.value)

    def counter(self, name):
        """a named counter (wb_get_counter) -- of the last indices() call: "box_bound", "box_bound_mismatch"; of contours():
        "fused_declined" (jobs the one-launch trace kernel left to the three-kernel chain), "fused_lds_bytes", "fused_jobs_per_cu" """
        v = C.c_int64()
        f = self.ctx.L.lib.wb_get_counter
        f.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
        if f(self.h, name.encode(), C.byref(v)) != 0:
            raise RuntimeError("wb_get_counter failed: " + self.ctx.error())
        return int(v.value)

    def pair_stats(self):
        """(base-point pairs covered by the all-pairs test, candidate pairs found) of the last indices() call."""
        a, c = C.c_int64(), C.c_int64()
        self.ctx.L.lib.wb_get_pair_stats(self.h, C.byref(a), C.byref(c))
        return int(a.value), int(c.value)

    def set_step_offset(self, first_step):
        """first time step of this batch in the whole record (added to the step column of the device rows)"""
        self.ctx.L.lib.wb_set_step_offset(self.h, int(first_step))

    def device_geometry(self):
        """(ring_offset pointer, events, vert_offset pointer, rings, x pointer, y pointer, vertices, on_device): the rings of the
        last indices() where the kernels wrote them (wb_get_device_geometry)"""
        ro, vo, x, y = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        ne, nr, nv, dev = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
        f = self.ctx.L.lib.wb_get_device_geometry
        f.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p), C.POINTER(C.c_int64)] * 2 + [C.POINTER(C.c_void_p)] * 2 + [C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        if f(self.h, C.byref(ro), C.byref(ne), C.byref(vo), C.byref(nr), C.byref(x), C.byref(y), C.byref(nv), C.byref(dev)) != 0:
            raise RuntimeError("wb_get_device_geometry failed: " + self.ctx.error())
        return ro.value or 0, int(ne.value), vo.value or 0, int(nr.value), x.value or 0, y.value or 0, int(nv.value), bool(dev.value)

    def device_tables(self):
        """(rows pointer, (n streamers, n overturnings, n cutoffs), poly_x pointer, poly_y pointer, vertices): the result
        of the last indices() as DEVICE buffers (context option "keep_device")"""
        rows, px, py = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n = (C.c_int64 * 3)()
        nv = C.c_int64()
        if self.ctx.L.lib.wb_get_device_tables(self.h, C.byref(rows), n, C.byref(px), C.byref(py), C.byref(nv)) != 0:
            raise RuntimeError("wb_get_device_tables: set the context option keep_device before indices()")
        return rows.value or 0, (int(n[0]), int(n[1]), int(n[2])), px.value or 0, py.value or 0, int(nv.value)

    def evaluated_pairs(self):
        """pairs whose great-circle term the all-pairs kernel actually evaluated (last indices() call)"""
        a = C.c_int64()
        self.ctx.L.lib.wb_get_evaluated_pairs(self.h, C.byref(a))
        return int(a.value)

    def timings(self):
        names = C.c_char_p()
        ms = C.POINTER(C.c_float)()
        n = C.c_int()
        self.ctx.L.lib.wb_get_timings(self.h, C.byref(names), C.byref(ms), C.byref(n))
        out = []
        addr = C.cast(names, C.c_void_p).value
        for i in range(n.value):
            s = C.string_at(addr)
            addr += len(s) + 1
            out.append((s.decode(), float(ms[i])))
        return out
