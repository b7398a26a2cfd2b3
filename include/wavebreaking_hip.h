/* wavebreaking_hip.h -- C ABI of the MI355X-native WaveBreaking hot path.
 *
 * The reference (skaderli/WaveBreaking 0.3.8) is pure Python and has no FFI layer; its boundary for
 * this path is the Python call surface
 *
 *     wavebreaking.calculate_contours      wavebreaking/indices/contour_index.py:45-51
 *     wavebreaking.calculate_streamers     wavebreaking/indices/streamer_index.py:44-57
 *     wavebreaking.calculate_overturnings  wavebreaking/indices/overturning_index.py:40-53
 *     wavebreaking.calculate_cutoffs       wavebreaking/indices/cutoff_index.py:33-45
 *
 * The functions below are what a ctypes binding inside those four functions calls instead of the
 * per-time-step Python loops (INTEGRATION.md shows the stub).  Plain pointers and sizes only; no C++
 * or torch types.  All functions return 0 on success, non-zero on error (text via wb_last_error);
 * nothing throws across the boundary.  A context owns one HIP stream and is single-threaded; use one
 * context per GPU / process.
 *
 * Conventions: field is (time, lat, lon), C-contiguous, lat and lon ASCENDING
 * (data_utils.py:196-213 is applied by the caller), float32 or float64; every decision is taken in
 * float64 like skimage.measure.find_contours does (contour_index.py:103).  Lattice coordinates are
 * int32 indices into the periodically extended grid (x in [0, nlon + int(periodic_add/dlon)),
 * y in [0, nlat)), exactly the reference's "index coordinates" (original_coordinates=False).
 */
#ifndef WAVEBREAKING_HIP_H
#define WAVEBREAKING_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wb_ctx wb_ctx;
typedef struct wb_batch wb_batch;

typedef struct wb_params {
    int32_t nt, nlat, nlon;
    int32_t dtype;            /* 0 = float32, 1 = float64 */
    int32_t on_device;        /* 1: field / intensity are device pointers, 0: host pointers (copied) */
    const void* field;        /* [nt][nlat][nlon] */
    const void* intensity;    /* same shape and dtype, or NULL (index_utils.py:77-89) */
    const double* lat;        /* host [nlat], degrees, ascending */
    const double* lon;        /* host [nlon], degrees, ascending */
    int32_t nlevels;          /* <= 4 */
    const double* levels;     /* host [nlevels] (index_utils.py:238-258) */
    double periodic_add;      /* degrees (contour_index.py:50) */
    double dlon, dlat;        /* data_utils.py:129-144 */
    double geo_dis, cont_dis;             /* streamer_index.py:51-52 */
    double range_group, ot_min_exp;       /* overturning_index.py:47-48 */
    double co_min_exp;                    /* cutoff_index.py:40 */
    int32_t want_cells;       /* also return the grid cells of every event */
    int32_t cap_seg;          /* 0 = automatic: per-(step,level) segment capacity, grown on overflow */
    int32_t cap_pairs;        /* 0 = automatic: per-contour candidate base-point pair capacity */
} wb_params;

/* contour table: contour_index.py:158-169 in index coordinates */
typedef struct wb_contour_table {
    int64_t n;                /* contours */
    const int32_t* step;      /* time index inside the batch */
    const int32_t* level_idx;
    const int32_t* closed;    /* contour_index.py:109 */
    const int32_t* nux;       /* len(set(x)); exp_lon = nux * dlon (contour_index.py:154) */
    const int64_t* sum_y;     /* mean_lat = round(sum_y / npts, 2) (contour_index.py:155) */
    const int32_t* npts;
    const int64_t* offset;    /* first point in x / y */
    int64_t npoints;
    const int32_t* x;
    const int32_t* y;
} wb_contour_table;

enum { WB_STREAMERS = 0, WB_OVERTURNINGS = 1, WB_CUTOFFS = 2 };

/* event table, one per index; rows in the reference's frame order (time, level, contour, event) */
typedef struct wb_event_table {
    int64_t n;
    const int32_t* step;
    const int32_t* level_idx;
    const int64_t* contour;   /* row of the contour table the event was cut from */
    const int32_t* ind1;      /* streamers: polygon = contour[ind1 .. ind2] (streamer_index.py:269-272) */
    const int32_t* ind2;      /* cutoffs: 0 .. npts-1 (cutoff_index.py:96) */
    const int32_t* box;       /* [n][4] overturnings: min_lon, min_lat, max_lon, max_lat (overturning_index.py:210-213) */
    const int32_t* orientation; /* overturnings: 0 cyclonic, 1 anticyclonic (overturning_index.py:199-202) */
    const double* sums;       /* [n][5]: sum(areas), sum(areas*data), sum(areas*intensity), sum(areas*x), sum(areas*y)
                                 (index_utils.py:94-102); com / mean_var / intensity / event_area follow on the host */
    const int32_t* ncells;
    const int64_t* cell_offset; /* into cell_x / cell_y, or NULL when want_cells == 0 */
    const int32_t* cell_x;
    const int32_t* cell_y;
    const int64_t* poly_offset; /* [n+1] polygon vertices in index coordinates (streamers, cutoffs; overturnings: */
    const int32_t* poly_x;      /*  the 4 box corners in shapely.geometry.box order), so that the event table   */
    const int32_t* poly_y;      /*  is self-contained and the contour table need not be fetched                 */
    const int32_t* flags;       /* bit 0 (WB_EVENT_NOT_SIMPLE): the event's polygon crosses itself -- its cells follow the non-zero
                                   winding rule; the reference relies on GEOS' make_valid there (index_utils.py:164), unpinned */
    const double* props;        /* [n][5] the frame's property columns, computed by the device from the sums exactly as
                                   index_utils.py:104-120 does: com longitude, com latitude (lon[int(x_com / areas) % nlon],
                                   lat[int(y_com / areas)]), mean_var, intensity, event_area (each .round(2)) */
} wb_event_table;
#define WB_EVENT_NOT_SIMPLE 1

#define WB_STATUS_OVERFLOW_SEG 1u
#define WB_STATUS_LEVEL_HIT 2u     /* a cell on a contour equals the level: crossings on grid vertices, joined as skimage joins them */
#define WB_STATUS_HAS_NAN 4u
#define WB_STATUS_HASH_FULL 8u
#define WB_STATUS_OVERFLOW_PAIRS 16u
#define WB_STATUS_LEVEL_ORDER 32u  /* ... and two contour ends compete for one vertex: skimage's sequential assembly is order dependent
                                      there; the job's pieces are then joined by a literal restatement of that walk ... */
#define WB_STATUS_LEVEL_KEYERROR 64u /* ... which found the dictionary entry skimage pops missing: find_contours raises KeyError on
                                      this field (contour_index.py:103), and so must the caller -- the job has no contours */
#define WB_STATUS_LEVEL_WALKED 128u  /* ... or finished: the contours are those of the sequential walk */
#define WB_STATUS_SCAN_TIMEOUT 256u  /* the field scan's halo exchange between the strips of a workgroup ran out of its (bounded) spin
                                        budget: an internal error -- the job's contours must not be used (the Python layer raises) */

int wb_create(int device_id, wb_ctx** ctx);
void wb_destroy(wb_ctx* ctx);
const char* wb_last_error(wb_ctx* ctx);
const char* wb_backend(void);                       /* "hip" for the product library */
/* tuning switches of a context; "scan_split" = 1: the field scan as two kernels (streaming capture + list build) */
int wb_set_option(wb_ctx* ctx, const char* name, int value);

/* phase 1: contours of every (time step, level) of the batch (contour_index.py:87-156) */
int wb_contours(wb_ctx* ctx, const wb_params* p, wb_batch** batch);
/* phase 1, alternative: a contour frame computed earlier and possibly filtered by the caller
   (decorator_contour_calculation, contour_index.py:197-249); row k has npts[k] points at x/y[offset[k]...],
   step[k] = index of its date in the data's time axis.  Events come back in frame-row order. */
int wb_batch_from_contours(wb_ctx* ctx, const wb_params* p, int64_t n, const int32_t* step, const int32_t* level_idx,
                           const int32_t* closed, const int64_t* offset, const int32_t* npts, const int32_t* x,
                           const int32_t* y, wb_batch** batch);
/* index parameters of an existing batch (its contours do not depend on them); NaN keeps the current value */
int wb_set_index_params(wb_batch* b, double geo_dis, double cont_dis, double range_group, double ot_min_exp,
                        double co_min_exp);
/* intensity cube of an existing batch whose field lives on the device (on_device = 1): a device pointer of the field's shape
   and dtype, or NULL -- a batch's contours do not depend on it, so the next wb_indices of a kept batch can serve another
   `intensity=` (index_utils.py:77-89) */
int wb_set_intensity(wb_batch* b, const void* dev_intensity);
int wb_batch_max_nux(wb_batch* b);                  /* local max of len(set(x)); all-reduce(max) it across ranks */
/* phase 2: indices.  global_max_nux plays contours.exp_lon.max() (streamer_index.py:106);
   which = bit mask 1 streamers | 2 overturnings | 4 cutoffs */
int wb_indices(wb_batch* b, int global_max_nux, int which);

/* transform_polygons (index_utils.py:129-184) for a whole event table: the polygons in the data's own coordinates,
   (lon[x % nlon], lat[y]); an event that reaches the periodic columns is cut at the last meridian into several rings
   (:148-173).  Event k owns rings ring_offset[k] .. ring_offset[k+1] (none: empty polygon, one: Polygon, several:
   MultiPolygon), ring q the vertices vert_offset[q] .. vert_offset[q+1] (not repeated at the end).  Same layout as the
   input of wb_flag_events.  With the context options "geometry" and "geometry_async" (both 1 by default) the device cuts and maps
   the rings during wb_indices into pinned host memory, and wb_indices does NOT wait for those kernels -- they are PCIe-bound
   (27 MB of doubles per year of 0.25 degree) and finish beside the next pass's field scan: wb_get_event_geometry waits for them
   (so do wb_free_batch and the batch's next wb_indices). */
typedef struct wb_geometry {
    int64_t n;
    const int64_t* ring_offset;
    int64_t nrings;
    const int64_t* vert_offset;
    const double* x;            /* degrees east */
    const double* y;            /* degrees north */
} wb_geometry;
int wb_get_event_geometry(wb_batch* b, int kind, wb_geometry* out);
/* the same host code for polygons given by the caller (index_utils.py:129-184 as a function of its own): polygon k = the
   index-lattice vertices poly_offset[k] .. poly_offset[k+1] of px / py.  index_space != 0: the rings as they are cut, in
   index coordinates, before the truncation to the lattice (:135) and the lon / lat look-up (lon / lat may be NULL then).
   The arrays of `out` belong to the context until its next call of this function. */
int wb_transform_polygons(wb_ctx* ctx, int64_t n, const int64_t* poly_offset, const int32_t* px, const int32_t* py,
                          const double* lon, const double* lat, int nlon, int index_space, wb_geometry* out);
/* ... and the same polygons through the DEVICE kernels that wb_indices runs for a pass's events (context option "geometry",
   default 1: the rings of wb_get_event_geometry are then cut, truncated and mapped to lon / lat by the device and written into
   pinned host memory beside the event tables; csrc/wb_geom.h): uploaded, transformed, read back -- ring for ring and bit for
   bit what wb_transform_polygons returns (the parity tests compare the two).  lat: [nlat]. */
int wb_transform_polygons_device(wb_ctx* ctx, int64_t n, const int64_t* poly_offset, const int32_t* px, const int32_t* py,
                                 const double* lon, const double* lat, int nlon, int nlat, int index_space, wb_geometry* out);
int wb_get_contours(wb_batch* b, wb_contour_table* out);
/* The event table of one index.  wb_get_events: complete, including the polygons in grid indices (poly_x / poly_y).  Since round 6
   wb_indices leaves those on the device as descriptors + contours (context option "index_polygons" 0, the default; 1: it packs
   them into pinned host memory itself, 13.5 MB per year of 0.25 degree over PCIe): the first wb_get_events of a pass then packs them
   -- it launches a kernel and waits, so it belongs to the thread that drives the context.  wb_get_event_rows: everything but the
   vertices (poly_x = poly_y = NULL; poly_offset still holds the vertex counts) -- what a caller needs who takes the polygons in
   lon / lat from wb_get_event_geometry, as the drop-in functions do (the reference's frames carry no index polygons,
   index_utils.py:129-184); it launches nothing and may be called from any thread.
   Threads: a context is DRIVEN by one thread (wb_contours, wb_indices, pass after pass).  A second thread may, at the same time,
   read a finished batch (wb_get_event_rows, wb_get_event_geometry, wb_get_status, the counters) and free it (wb_free_batch) while
   the driving thread is already in the next pass: the calls that launch take the context's lock, the allocators have their own. */
int wb_get_events(wb_batch* b, int kind, wb_event_table* out);
int wb_get_event_rows(wb_batch* b, int kind, wb_event_table* out);
int wb_get_status(wb_batch* b, const uint32_t** status, int64_t* njobs);   /* per (step, level) */
/* counters of the last wb_indices by name: "box_bound" -- overturning boxes whose centre-of-mass sums (index_utils.py:91-102,
   pandas' compensated sums) followed from the EXACT totals of their terms because no rounding boundary lay within the summation's
   error bound (csrc/wb_props.h, wb_k_box_com); the others took the chained evaluation in the reference's order.  Of wb_contours:
   "fused_declined" -- (step, level) jobs the one-launch trace kernel left to the three-kernel chain (a cell equal to the level, or
   more segments / contours than the launch's LDS admitted: a context sizes it by what its batches have asked for, and the next
   launch has grown), "fused_lds_bytes", "fused_jobs_per_cu" */
int wb_get_counter(wb_batch* b, const char* name, int64_t* value);
int wb_get_band_hits(wb_batch* b, int64_t* geo, int64_t* cont);            /* threshold guard-band re-evaluations */
/* calculate_properties truncates x_com / areas and y_com / areas (index_utils.py:105-110); events whose quotient lies
   within the rounding slack of an integer have their three sums re-evaluated cell by cell in the reference's order
   (pandas' compensated group sum over rows in (y, x) order, index_utils.py:91-102).  Counts of the last wb_indices:
   rasterised events (streamers, cutoffs) / overturning boxes that took that path / boxes among them whose row-parallel
   evaluation could not be verified and which were walked by one lane instead. */
int wb_get_com_rechecks(wb_batch* b, int64_t* events, int64_t* boxes, int64_t* boxes_one_lane);
/* streamer all-pairs test of the last wb_indices: base-point pairs covered (sum of N(N-1)/2 over the selected contours,
   streamer_index.py:130-157) and candidate pairs found */
int wb_get_pair_stats(wb_batch* b, int64_t* point_pairs, int64_t* candidate_pairs);
/* ... and the pairs whose great-circle term was actually evaluated (the others were excluded as whole blocks or by the
   |dx| / |dy| reach tests) */
int wb_get_evaluated_pairs(wb_batch* b, int64_t* evaluated);
/* groups of check_groups (streamer_index.py:224-251) whose largest along-contour length was shared by several members to the
   last bit in the last wb_indices call: there the survivor is the member that comes first in the reference's CPython set */
int wb_get_group_ties(wb_batch* b, int64_t* ties);
/* per-stage device times of the last wb_contours / wb_indices calls, measured with HIP events on the
   context's stream: names is a NUL-separated list.  Which stages are timed is the context option "timers": 1 (the default) the
   field scan only -- the other stages keep their names and read 0 ms --, 2 every stage (an event pair per stage: a dozen
   dependency gaps per pass, 3-5 % of it), 0 none */
int wb_get_timings(wb_batch* b, const char** names, const float** ms, int* n);
/* Time-sharded runs (one process per GPU): with the context option "keep_device" = 1, wb_indices also leaves its result ON
   THE DEVICE -- one row of WB_ROW_WIDTH doubles per event (step + the offset set by wb_set_step_offset, level, contour
   inside its (step, level), ind1, ind2, box[4], orientation, the five sums, cell count, kind, flags, polygon length, 0;
   streamers, then overturnings, then cutoffs: n[3]) and the polygon vertices -- so that the gather to rank 0 over RCCL
   moves device buffers (no host staging).  Pointers are device pointers, valid until the next wb_indices / wb_free_batch.
   "keep_device" = 2: the rows only (*nverts = 0) -- the polygons in grid indices stay descriptors on the rank that found them, as on
   a single GPU (context option "index_polygons"); the gathered catalogue's geometry are the rings of wb_get_device_geometry. */
#define WB_ROW_WIDTH 20
int wb_set_step_offset(wb_batch* b, int64_t first_step);
int wb_get_device_tables(wb_batch* b, const double** rows, int64_t n[3], const int32_t** poly_x, const int32_t** poly_y,
                         int64_t* nverts);
/* ... and the pass's rings in lon / lat (context option "geometry") where the kernels wrote them -- device arrays with the options
   geometry_dma and keep_device, else the pinned host arrays (*on_device says which): ring_offset[nev + 1] over the events in
   table order (streamers, overturnings, cutoffs), vert_offset[nrings + 1], x / y[nverts] degrees.  A time-sharded run gathers
   them beside the tables (wavebreaking_amd.distributed), so that rank dst holds the catalogue WITH its geometry and no rank
   transforms another rank's polygons (index_utils.py:129-184).  Waits for the geometry kernels. */
int wb_get_device_geometry(wb_batch* b, const int64_t** ring_offset, int64_t* nev, const int64_t** vert_offset, int64_t* nrings,
                           const double** x, const double** y, int64_t* nverts, int* on_device);
void wb_free_batch(wb_batch* b);

/* ---- the two exchanges of a time-sharded run (one process per GPU), on RCCL inside the library -------------------------
   The reference is single-process; its selection `contours.exp_lon.max()` (streamer_index.py:106, overturning_index.py:105,
   cutoff_index.py:90) runs over the whole frame, so ranks that each hold a block of time steps need (1) the maximum over
   all ranks and (2) the event tables on one rank.  wb_comm_unique_id: 128 bytes made on ONE rank, handed to all by the
   launcher; wb_comm_init: every rank, collectively.  wb_comm_allreduce_max: *value <- max over the ranks.
   wb_comm_gather_tables: what wb_indices left on the device (context option "keep_device") of every rank's batch (NULL: a
   rank without time steps) arrives on rank dst -- counts by one ncclAllGather, payload by grouped ncclSend / ncclRecv of the
   exact sizes -- in rank (= time) order: rows[nrows][WB_ROW_WIDTH], poly_x / poly_y[nverts], pinned host memory of the
   context, valid until its next call; the other ranks get nrows = nverts = 0.  ranks_seen (may be NULL) <- the number of
   ranks whose (rank, counts) triple arrived in its own place (fewer than the group's size is an error).
   wb_comm_stage_tables + wb_comm_gather_staged: the same exchange off the critical path -- stage copies the batch's tables
   (device to device) into slot 0 <= slot < WB_COMM_SLOTS of the communicator, after which the batch may be freed;
   gather_staged gathers a staged slot and frees it, and may run on ANOTHER host thread while the caller's thread goes on
   with the next batches (every rank must gather its slots in the same order).
   librccl is looked up at run time; without it wb_comm_init fails and nothing else changes. */
#define WB_COMM_ID_BYTES 128
int wb_comm_unique_id(wb_ctx* ctx, void* id /* [WB_COMM_ID_BYTES] */);
int wb_comm_init(wb_ctx* ctx, const void* id, int rank, int world);
int wb_comm_allreduce_max(wb_ctx* ctx, int* value);
int wb_comm_gather_tables(wb_ctx* ctx, wb_batch* b, int dst, const double** rows, int64_t* nrows, const int32_t** poly_x,
                          const int32_t** poly_y, int64_t* nverts, int* ranks_seen);
#define WB_COMM_SLOTS 4
int wb_comm_stage_tables(wb_ctx* ctx, wb_batch* b, int slot);
int wb_comm_gather_staged(wb_ctx* ctx, int slot, int dst, const double** rows, int64_t* nrows, const int32_t** poly_x,
                          const int32_t** poly_y, int64_t* nverts, int* ranks_seen);
/* Failure handling of a time-sharded run ("RCCL error -> abort").  A rank that cannot deliver a staged slot calls
   wb_comm_gather_staged with slot = -1: it still takes part in the exchange of the counts, with -1 for its own, so every
   rank's call for that slot FAILS ("... ranks in place") instead of waiting for a payload that never comes, and the ranks'
   sequences of collectives stay matched.  wb_comm_abort: ncclCommAbort -- peers blocked in a collective with this rank get an
   error from RCCL; every later exchange of this context fails at once.  It may be called from another host thread than the
   one that sits in wb_comm_gather_staged (the library looks at an atomic flag before every group of RCCL calls and holds a
   lock while it enqueues one).  wb_comm_reserve: the destination rank's device and
   pinned buffers for the whole group's catalogue, made before the record's batches are in flight (growing them inside a
   gather is a hipFree / hipMalloc beside the workers' kernels).  The views a gather handed out (*rows, *poly_x, *poly_y) are
   valid until the context's next gather OR reserve: a buffer that grows is freed. */
int wb_comm_abort(wb_ctx* ctx);
int wb_comm_reserve(wb_ctx* ctx, int64_t rows, int64_t polygon_vertices);
void wb_comm_destroy(wb_ctx* ctx);

/* device memory + synthetic ERA5-shaped PV for the benchmark */
void* wb_dev_alloc(wb_ctx* ctx, uint64_t bytes);
void wb_dev_free(wb_ctx* ctx, void* p);
int wb_memcpy_h2d(wb_ctx* ctx, void* dst, const void* src, uint64_t bytes);
int wb_memcpy_d2h(wb_ctx* ctx, void* dst, const void* src, uint64_t bytes);
int wb_synth_fill(wb_ctx* ctx, void* dev_field, int dtype, int64_t t0, int nt, int nlat, int nlon,
                  const double* lat, const double* lon, const double* wave_params /* [7][5] */, double eps,
                  uint64_t seed);

/* ---- pre-processing next to the hot path ------------------------------------------
   calculate_smoothed_field (processing/spatial.py:60-128): per time step `passes` times
   scipy.ndimage.convolve(temp, weights, mode) / np.sum(weights) (spatial.py:101), then the rows
   arange(-b, b), b = int(kh / 2 + 0.5), are NaN (spatial.py:104-107).  field: (nt, nlat, nlon) float32 (dtype 0) or
   float64 (1), host or device (on_device); weights: kh x kw row-major, 1..7 each; wsum = np.sum(weights) as the
   caller's numpy evaluates it; mode: 0 reflect, 1 constant (cval 0), 2 nearest, 3 mirror, 4 wrap; out: float64
   (nt, nlat, nlon) (numpy >= 2 promotion of float32 / int64), host or device; flags bit 0 = one launch per pass
   even where the fused kernel applies, bit 1 = always the hardware division (parity tests). */
int wb_smooth_field(wb_ctx* ctx, const void* field, int dtype, int on_device, int64_t nt, int nlat, int nlon, int passes,
                    const double* weights, int kh, int kw, double wsum, int mode, double* out, int out_on_device,
                    int flags);
/* calculate_momentum_flux (processing/spatial.py:27-57): (u - u.mean(lon)) * (v - v.mean(lon)) in u's dtype, the
   zonal mean being numpy's nanmean (pairwise row sum in the array dtype).  u, v, out: (nt, nlat, nlon). */
int wb_momentum_flux(wb_ctx* ctx, const void* u, const void* v, int dtype, int on_device, int64_t nt, int nlat, int nlon,
                     void* out, int out_on_device);
/* to_xarray (processing/events.py:37-110): out[step, j, i] = value (1 without `value`) for every grid point contained in
   event.geometry.buffer(radius), radius = (dlon + dlat) / 4 (events.py:69-79); later events overwrite earlier ones where
   they overlap (events.py:93-99).  lat / lon ascending and regular.  Event k: time index step[k], rings
   ring_offset[k] .. ring_offset[k+1] (parts of a MultiPolygon and holes alike: even-odd), ring q: vertices
   vert_offset[q] .. vert_offset[q+1] of vx (degrees east) / vy (degrees north).  out: (nt, nlat, nlon) of out_dtype
   0 int8 / 1 float32 / 2 float64, zero-filled here first; host or device. */
int wb_flag_events(wb_ctx* ctx, int64_t nt, int nlat, int nlon, const double* lat, const double* lon, double radius,
                   int64_t nev, const int32_t* step, const double* value, const int64_t* ring_offset,
                   const int64_t* vert_offset, const double* vx, const double* vy, int out_dtype, void* out,
                   int out_on_device);
/* track_events (processing/events.py:113-241).  n events in the caller's row order; dates either as int64 nanoseconds
   (date_ns; hours between two rows = difference / 1e9 / 3600 as pandas computes it, events.py:163) or as float64 numbers
   (date_val; |difference| is compared, events.py:165) -- exactly one of the two pointers is given.  Pairs with
   0 < dt <= time_range are linked when
     method 1 "by_distance": haversine(com_i, com_j) * 6371 < distance, com = [n][2] degrees read as the reference reads
                             them (events.py:189-198);
     method 0 "by_overlap":  area(A n B) / (area A + area B - area(A n B)) > overlap (events.py:209-214) for the polygons
                             given as LATTICE rings: event k owns rings ring_offset[k] .. ring_offset[k+1], ring q the
                             vertices vert_offset[q] .. vert_offset[q+1] of vx / vy (grid indices, |v| < 32768; closed or
                             not; parts of a MultiPolygon and holes alike: even-odd).  buffer = 0 only.  A ratio within 1e-13
                             (relative) of `overlap` counts as equal to it: not above.
   label[n] <- events.py:224-238 (components named after their smallest row, renumbered 0, 1, 2, ...).  area[n] (may be
   NULL; method 0) <- even-odd area of every event in index units.  stats[4] (may be NULL) <- pairs in range, pairs whose
   bounding boxes overlap (method 0), pairs linked, 0. */
int wb_track_events(wb_ctx* ctx, int64_t n, const int64_t* date_ns, const double* date_val, int method, double time_range,
                    double overlap, double distance, const double* com, const int64_t* ring_offset,
                    const int64_t* vert_offset, const int32_t* vx, const int32_t* vy, int32_t* label, double* area,
                    uint64_t* stats);
/* the overlap areas behind method 0 for an explicit list of pairs: inter[q] <- area(event pair_i[q] n event pair_j[q]),
   area[n] (may be NULL) <- area of every event; index units.  Exactly 0 when the interiors do not meet. */
/* track_events(method="by_overlap", buffer > 0, overlap = 0) (events.py:203-214): the reference buffers both geometries by
   `buffer` degrees (GEOS' round buffer) and links a pair whose buffered polygons overlap at all -- i.e. whose planar distance in
   degrees is below 2 * buffer (exact for discs; GEOS' inscribed arcs can fall short by 0.24 % of that at corners: such pairs
   are counted in stats[5] and decided by the disc; unpinned, GEOS is not in the image).  Rings as for wb_track_events on the
   grid with step_x_deg / step_y_deg degrees per index step.  stats[6]: the four of wb_track_events, [4] pairs linked by the
   distance alone (no overlap of the unbuffered polygons), [5] pairs inside the arc band. */
int wb_track_events_buffered(wb_ctx* ctx, int64_t n, const int64_t* date_ns, const double* date_val, double time_range,
                             double buffer_deg, double step_x_deg, double step_y_deg, const int64_t* ring_offset,
                             const int64_t* vert_offset, const int32_t* vx, const int32_t* vy, int32_t* label, uint64_t* stats);
int wb_track_overlaps(wb_ctx* ctx, int64_t n, const int64_t* ring_offset, const int64_t* vert_offset, const int32_t* vx,
                      const int32_t* vy, int64_t npairs, const int64_t* pair_i, const int64_t* pair_j, double* inter,
                      double* area);
/* device time (HIP events on the context's stream) of the kernels of the last wb_smooth_field / wb_momentum_flux /
   wb_flag_events / wb_track_events / wb_track_overlaps */
float wb_last_op_ms(wb_ctx* ctx);

/* 64-bit fingerprint of EVERY byte of a host buffer, computed by `threads` host threads (0: a default from the machine's core
   count; the value does not depend on it).  The Python layer keeps input cubes on the device between calls and asks this
   whether a cube it is handed again still holds the bytes it uploaded (the reference rereads its input on every call,
   index_utils.py:66-75; an in-place edit between two calls must be seen).  No device work. */
uint64_t wb_fingerprint(const void* host_buffer, uint64_t bytes, int threads);

#ifdef __cplusplus
}
#endif
#endif
