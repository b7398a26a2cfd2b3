// wb_flags.h -- events -> gridded flag cube.
//
// to_xarray (processing/events.py:37-110): a grid point (lon_i, lat_j) of time step t is flagged iff some event of that
// date CONTAINS it after `geometry.buffer((dlon + dlat) / 4)` (events.py:69-79), i.e. iff the point lies inside the
// (multi)polygon or closer than r to its boundary.  Geometries are the lon/lat polygons of transform_polygons
// (index_utils.py:129-184) -- lattice polygons, possibly split at the last meridian -- but nothing here assumes that:
// vertices are float64 degrees, every ring of an event (parts of a MultiPolygon, holes) takes part in one even-odd
// count.  Where several events cover a cell the one latest in the frame wins (events.py:93-99 assigns in index order).
//
// One workgroup per time step (events of a step are handled one after the other, so "latest wins" needs no atomics):
//   bbox of the event -> clear the toggle rows -> every edge toggles, on each grid row it spans (half-open rule
//   ymin <= lat_j < ymax), the cells left of its crossing -> running parity per row writes the interior -> every edge
//   flags the grid points closer than r to it (exact point-segment distance in float64).
// Points exactly on the boundary have an arbitrary parity but distance 0 < r, so they are always flagged.
#pragma once
#include "wb_spmd.h"

struct WbFlagArgs {
    long long nt;
    int nlat, nlon;
    const double* lat;            // ascending
    const double* lon;            // ascending
    double dlat, dlon, r;
    int nact;                     // time steps that have events
    const int* act_step;          // [nact]
    const long long* act_ev0;     // [nact+1] event range (events sorted by step, frame order inside a step)
    const long long* ev_ring0;    // [nev+1]
    const long long* ring_v0;     // [nrings+1]; a ring's vertices (closed or not: the closing edge is implied)
    const double* vx;
    const double* vy;
    const double* value;          // [nev] or null (= 1)
    int out_dtype;                // 0 int8, 1 float32, 2 float64
    void* out;                    // [nt][nlat][nlon], zeroed by the caller
    int* tog;                     // [grid][nlat*(nlon+1)] scratch
};

// smallest i in [0, n] with c[i] >= v (n if none); c ascending and regular (c0, step d)
static WB_HD inline int wb_fl_first_ge(const double* c, int n, double d, double v) {
    double g = (v - c[0]) / d;
    int i = g < 0.0 ? 0 : (g > (double)n ? n : (int)g);
    while (i > 0 && c[i - 1] >= v) --i;
    while (i < n && c[i] < v) ++i;
    return i;
}

static WB_HD inline void wb_fl_store(const WbFlagArgs& a, long long cell, double v) {
    if (a.out_dtype == 0) ((signed char*)a.out)[cell] = (signed char)v;
    else if (a.out_dtype == 1) ((float*)a.out)[cell] = (float)v;
    else ((double*)a.out)[cell] = v;
}

WB_KERNEL(wb_k_flag_events, WbFlagArgs a) {
    WB_SHARED(int, box, 4);       // i0, i1, j0, j1 (inclusive) of the current event
    int* tog = a.tog + (size_t)WB_BLOCK_ID * a.nlat * (a.nlon + 1);
    const double r2 = a.r * a.r;
    for (int s = WB_BLOCK_ID; s < a.nact; s += WB_GRID_DIM) {
        const long long plane = (long long)a.act_step[s] * a.nlat * a.nlon;
        for (long long e = a.act_ev0[s]; e < a.act_ev0[s + 1]; ++e) {
            const long long v0 = a.ring_v0[a.ev_ring0[e]], v1 = a.ring_v0[a.ev_ring0[e + 1]];
            const long long r0 = a.ev_ring0[e], r1 = a.ev_ring0[e + 1];
            const double val = a.value ? a.value[e] : 1.0;
            WB_SINGLE
                box[0] = a.nlon; box[1] = -1; box[2] = a.nlat; box[3] = -1;
            WB_END_SINGLE
            WB_PFOR(k, v1 - v0)
                const double x = a.vx[v0 + k], y = a.vy[v0 + k];
                const int i0 = wb_fl_first_ge(a.lon, a.nlon, a.dlon, x - a.r), i1 = wb_fl_first_ge(a.lon, a.nlon, a.dlon, x + a.r);
                const int j0 = wb_fl_first_ge(a.lat, a.nlat, a.dlat, y - a.r), j1 = wb_fl_first_ge(a.lat, a.nlat, a.dlat, y + a.r);
                wb_atomic_min(&box[0], i0 > 0 ? i0 - 1 : 0);
                wb_atomic_max(&box[1], i1 < a.nlon ? i1 : a.nlon - 1);
                wb_atomic_min(&box[2], j0 > 0 ? j0 - 1 : 0);
                wb_atomic_max(&box[3], j1 < a.nlat ? j1 : a.nlat - 1);
            WB_END_PFOR
            const int bi0 = box[0], bi1 = box[1], bj0 = box[2], bj1 = box[3];
            WB_SYNC();                                                    // everyone has read the box before it is reset
            if (bi1 < bi0 || bj1 < bj0) continue;                        // block-uniform
            const int bw = bi1 - bi0 + 2;                                 // one extra column: toggles past the last cell
            WB_PFOR(q, (long long)(bj1 - bj0 + 1) * bw)
                tog[q] = 0;
            WB_END_PFOR
            // interior: crossings of every ring edge with the grid rows
            for (long long rg = r0; rg < r1; ++rg) {
                const long long w0 = a.ring_v0[rg], nv = a.ring_v0[rg + 1] - w0;
                WB_PFOR(k, nv)
                    const long long k2 = k + 1 < nv ? k + 1 : 0;
                    const double x1 = a.vx[w0 + k], y1 = a.vy[w0 + k], x2 = a.vx[w0 + k2], y2 = a.vy[w0 + k2];
                    const double ymin = y1 < y2 ? y1 : y2, ymax = y1 < y2 ? y2 : y1;
                    if (ymin < ymax) {
                        for (int j = wb_fl_first_ge(a.lat, a.nlat, a.dlat, ymin); j <= bj1 && a.lat[j] < ymax; ++j) {
                            if (j < bj0) continue;
                            const double xc = x1 + (a.lat[j] - y1) * (x2 - x1) / (y2 - y1);
                            int ic = wb_fl_first_ge(a.lon, a.nlon, a.dlon, xc);     // cells with lon < xc are left of the crossing
                            if (ic > bi1 + 1) ic = bi1 + 1;
                            if (ic > bi0) {
                                wb_atomic_add(&tog[(size_t)(j - bj0) * bw], 1);
                                wb_atomic_add(&tog[(size_t)(j - bj0) * bw + (ic - bi0)], 1);
                            }
                        }
                    }
                WB_END_PFOR
            }
            WB_PFOR(jr, bj1 - bj0 + 1)
                int par = 0;
                const int* row = tog + (size_t)jr * bw;
                for (int i = bi0; i <= bi1; ++i) {
                    par ^= row[i - bi0] & 1;
                    if (par) wb_fl_store(a, plane + (long long)(bj0 + jr) * a.nlon + i, val);
                }
            WB_END_PFOR
            // rim: grid points closer than r to an edge
            for (long long rg = r0; rg < r1; ++rg) {
                const long long w0 = a.ring_v0[rg], nv = a.ring_v0[rg + 1] - w0;
                WB_PFOR(k, nv)
                    const long long k2 = k + 1 < nv ? k + 1 : 0;
                    const double ax = a.vx[w0 + k], ay = a.vy[w0 + k], bx = a.vx[w0 + k2], by = a.vy[w0 + k2];
                    const double dx = bx - ax, dy = by - ay, len2 = dx * dx + dy * dy;
                    const double xlo = (ax < bx ? ax : bx) - a.r, xhi = (ax < bx ? bx : ax) + a.r;
                    const double ylo = (ay < by ? ay : by) - a.r, yhi = (ay < by ? by : ay) + a.r;
                    const int i0 = wb_fl_first_ge(a.lon, a.nlon, a.dlon, xlo), j0 = wb_fl_first_ge(a.lat, a.nlat, a.dlat, ylo);
                    for (int j = j0; j < a.nlat && a.lat[j] <= yhi; ++j)
                        for (int i = i0; i < a.nlon && a.lon[i] <= xhi; ++i) {
                            const double px = a.lon[i], py = a.lat[j];
                            double t = 0.0;
                            if (len2 > 0.0) {
                                t = ((px - ax) * dx + (py - ay) * dy) / len2;
                                t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
                            }
                            const double cx = ax + t * dx, cy = ay + t * dy;
                            const double ex = px - cx, ey = py - cy;
                            if (ex * ex + ey * ey < r2) wb_fl_store(a, plane + (long long)j * a.nlon + i, val);
                        }
                WB_END_PFOR
            }
        }
    }
}
