/* wavebreaking_hip_dev.h -- development yardsticks exported by libwb_hip.so beside the C ABI of wavebreaking_hip.h.
 * Not part of the drop-in boundary (nothing in the reference corresponds to them); used by tools/scan_placement_probe.py and
 * tools/stream_pattern_probe.py only. */
#ifndef WAVEBREAKING_HIP_DEV_H
#define WAVEBREAKING_HIP_DEV_H
#include "wavebreaking_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* milliseconds of one streaming read of a device buffer (best of reps; -1: failed).  No counterpart in the reference: on this
   platform an allocation's physical placement decides 15 % of the HBM throughput a kernel gets from it */
double wb_probe_read_ms(wb_ctx* ctx, const void* dev, uint64_t bytes, int reps);
/* ... and of a read by `grid` workgroups that each stream chunks of chunk_bytes from start to end (the scan's pattern: 256
   workgroups, 1.4 MB chunks); development yardstick */
double wb_probe_streams_ms(wb_ctx* ctx, const void* dev, uint64_t bytes, uint64_t chunk_bytes, int grid, int reps);
/* The two lattice predicates of the streamer index evaluated ON THE DEVICE by the very functions wb_k_streamer_f234 calls
   (csrc/wb_core.h: wb_chord_hits_segment -- `not chord.touches(contour)`, streamer_index.py:192-198 -- and
   wb_segments_intersect, :233-236), for the randomized differential tests against an independent exact-rational statement
   (tests/ogc_reference.py).  contour: n lattice points x / y; chord k joins the contour's vertices i1[k] and i2[k];
   touches[k] <- 1 when the open chord meets the contour's interior nowhere.  pairs: segment (ax, ay)-(bx, by) against
   (cx, cy)-(dx, dy), 8 ints each; intersects[k] <- 1 when the closed segments share a point. */
int wb_test_chord_touches(wb_ctx* ctx, int n, const int32_t* x, const int32_t* y, int nchords, const int32_t* i1, const int32_t* i2,
                          int32_t* touches);
int wb_test_segments_intersect(wb_ctx* ctx, int npairs, const int32_t* seg8, int32_t* intersects);
/* The frame's property columns (index_utils.py:104-120: the truncated centre of mass, the three .round(2)) evaluated ON THE DEVICE
   by the function wb_k_pack_results calls for every event (csrc/wb_props.h:wb_props_columns): sums[n][5] (areas, mean_var,
   intensity, x_com, y_com) -> out[n][5] (com longitude, com latitude, mean_var, intensity, event_area).  For the test against
   the reference's own statements (tests/golden/reference_rules.npz). */
int wb_test_props_columns(wb_ctx* ctx, int64_t n, const double* sums, const double* lon, int nlon, const double* lat, int nlat, double* out);
#ifdef __cplusplus
}
#endif
#endif
