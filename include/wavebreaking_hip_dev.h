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
#ifdef __cplusplus
}
#endif
#endif
