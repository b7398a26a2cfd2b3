// wb_props.h -- event cells and area-weighted sums (one workgroup per event, persistent over events).
//
// Restates wavebreaking/utils/index_utils.py:35-102 (calculate_properties): the cells of an event are
// the lattice points contained in polygon.buffer((dlon+dlat)/4) (:37-51); on a lattice polygon that is
// {winding != 0} U {distance to the boundary < r} (oracle/geom.py).  The polygon is
// rasterised by scanline crossing counts (signed, suffix-summed per row), the rim by exact
// point-segment distance tests around every edge; sums follow :58-102.
//
// com = (lon[int(x_com / areas) % nlon], lat[int(y_com / areas)]) TRUNCATES the ratio of two sums (:105-110), so the
// sums must be the reference's to the last bit whenever the ratio sits next to an integer -- and it sits exactly ON one
// for every overturning box with min_lon + max_lon even.  The reference's sums: one frame row per cell in the join's row
// order (grid points numbered y-major, x-minor, :37-41), areas = weight_lat[y] (:63), x_com = x * areas, y_com = y * areas
// (:91-92, one rounding each), summed by pandas' groupby sum (:94-102) = sequential Kahan-compensated addition
// (pandas _libs/groupby.pyx group_sum; pinned by tests/golden/props_pandas.npz).  That chain is sequential by nature, so:
//   * fast path: per-row integer counts, areas += w * cnt etc. over rows -- within (2 rows + 8) ulp of the reference's value;
//   * when x_com / areas or y_com / areas lies within that bound (times a safety factor) of an integer, the event's three
//     sums are re-evaluated cell by cell exactly as above (WB_KAHAN) and counted (wb_get_com_rechecks): one lane walking
//     the event's raster here, one lane per box in wb_k_box_com (boxes need no raster).
// mean_var / intensity sums are compared to 1e-6 relative and keep the row-wise order.
#pragma once
#include "wb_core.h"
#include <stdio.h>

#if !defined(WB_HOSTSIM)
#pragma clang fp contract(off)
#endif

enum { WB_EV_STREAMER = 0, WB_EV_OVERTURNING = 1, WB_EV_CUTOFF = 2 };
#ifndef WB_PROPS_LROWS
#define WB_PROPS_LROWS 160
#endif
#ifndef WB_PROPS_PART
#define WB_PROPS_PART 192
#endif

struct WbPropsArgs {
    int nslots, nlat, nlon, W, cap_seg, nlev, dtype, want_cells, pad;
    WbCtl* ctl;                 // ctl->NE events; com_rechecks, cell_count
    double r2;                  // buffer radius squared, ((dlon+dlat)/4)^2 in index units
    const int* ev;              // [NE][8]: kind | orientation << 8, job, contour, a0..a3, overturnings: slot of the box in ot_sums
    const double* ot_sums;      // [slots][3] areas, x_com, y_com of the overturning boxes (wb_k_box_com)
    const int* c_off;
    const int* c_len;
    const uint32_t* cpts;
    const void* data;           // [nt][nlat][nlon]
    const void* intensity;      // or NULL
    const int* job_step;        // [njobs] time index of a job, or NULL: job / nlev
    const double* wlat;         // [nlat] cos(radians(lat)) * area_cell   (:58-62)
    int* cells;                 // [nslots][W*nlat]
    double* rowacc;             // [nslots][4][nlat]
    double* ev_sums;            // [nev][5]: areas, mean_var, intensity, x_com, y_com  (sums, :94-102)
    int* ev_ncells;             // [nev]
    int* ev_flags;              // [nev] bit 0: the polygon crosses itself (winding numbers other than 0 and one constant +-1)
    unsigned long long cell_cap;
    uint32_t* cell_pool;        // (x | y<<16)
    long long* ev_cell_off;     // [nev]
};

// The frame's property columns of one event from its five sums (index_utils.py:104-120), exactly as numpy / pandas compute them
// there: com = (lon[int(x_com / areas) % nlon], lat[int(y_com / areas)]) -- .astype("int") truncates, % takes the divisor's sign --
// and .round(2) = rint(v * 100) / 100 (numpy's round: multiply, round half to even, divide; the quotient v / areas is rounded to
// double first).  out[5]: com_lon, com_lat, mean_var, intensity, event_area.  The build runs with -ffp-contract=off: no product
// is fused into the division's fix-up.  Pinned by tests/golden/reference_rules.npz (the reference's own statements on the oracle's
// sums) through wb_test_props_columns.
WB_HD inline void wb_props_columns(const double* s, const double* lon, const double* lat, int nlon, int nlat, double* out) {
    const double a = s[0];
    const double qx = s[3] / a, qy = s[4] / a;
    long long ix = (qx == qx && fabs(qx) < 9.0e18) ? (long long)qx : 0, iy = (qy == qy && fabs(qy) < 9.0e18) ? (long long)qy : 0;
    ix %= (long long)nlon;
    if (ix < 0) ix += nlon;
    if (iy < 0) iy += nlat;                          // (numpy counts a negative index from the end)
    iy = iy < 0 ? 0 : (iy >= nlat ? nlat - 1 : iy);  // (beyond the axis numpy raises; an event's centre of mass lies inside its rows)
    out[0] = lon[ix];
    out[1] = lat[iy];
    out[2] = rint((s[1] / a) * 100.0) / 100.0;
    out[3] = rint((s[2] / a) * 100.0) / 100.0;
    out[4] = rint(a * 100.0) / 100.0;
}

WB_HD inline double wb_load_field(const void* p, size_t idx, int dtype) {
    return dtype == 0 ? (double)((const float*)p)[idx] : ((const double*)p)[idx];
}

WB_DEV inline uint32_t wb_event_vertex(const int* d, const uint32_t* pts, int k) {
    const int kind = d[0] & 0xFF;
    if (kind == WB_EV_STREAMER) return pts[d[3] + k];
    if (kind == WB_EV_CUTOFF) return pts[k];
    // shapely.geometry.box(minx, miny, maxx, maxy): (maxx,miny),(maxx,maxy),(minx,maxy),(minx,miny)
    const int x = (k == 0 || k == 1) ? d[5] : d[3];
    const int y = (k == 1 || k == 2) ? d[6] : d[4];
    return (uint32_t)x | ((uint32_t)y << 16);
}

// lattice point (x, y) closer than r to the segment a-b (exact integer projection, one division)
WB_DEV inline bool wb_near_segment(int x, int y, int ax, int ay, int bx, int by, double r2) {
    const long long dx = bx - ax, dy = by - ay, L = dx * dx + dy * dy;
    const long long t = (long long)(x - ax) * dx + (long long)(y - ay) * dy;
    if (L == 0 || t <= 0) return (double)((long long)(x - ax) * (x - ax) + (long long)(y - ay) * (y - ay)) < r2;
    if (t >= L) return (double)((long long)(x - bx) * (x - bx) + (long long)(y - by) * (y - by)) < r2;
    const long long cr = dx * (long long)(y - ay) - dy * (long long)(x - ax);
    return ((double)cr * (double)cr) / (double)L < r2;
}

// one step of pandas' group_sum: s += v with Kahan compensation c (groupby.pyx: y = v - c; t = s + y; c = t - s - y)
#define WB_KAHAN(s, c, v)                          \
    do {                                           \
        const double wb_y_ = (v) - (c);            \
        const double wb_t_ = (s) + wb_y_;          \
        (c) = (wb_t_ - (s)) - wb_y_;               \
        if ((c) != (c)) (c) = 0.0;                 \
        (s) = wb_t_;                               \
    } while (0)

#define WB_PROPS_LONG 16
// sum over the member cells of the box [x0, x0 + cols) x [y0, y0 + rows) of weight(y) * (data, intensity): the lanes run
// ALONG the rows (adjacent lanes on adjacent cells: a load instruction touches two or three cache lines, not 64), eight
// cells' loads in flight per lane; BM false: every cell is a member.  The lane's two partial sums go to out[2 * lane].
// (These sums are compared to 1e-6 relative: the order of summation is free.)
// The loads are UNCONDITIONAL (a cell that is not a member is read all the same -- its address lies inside the field -- and
// dropped afterwards) and the loop has no branch between them: written as `in ? load : 0` every load sat in a block of
// its own with an s_waitcnt vmcnt(0) behind it (the float -> double conversion inside the branch needs the value), the
// "eight loads in flight" were one, and the sums were 70 % of the kernel (seen in the ISA, round 3).
template <class T, bool BM>
WB_DEV inline void wb_props_field_sums_t(const WbPropsArgs& a, size_t tbase, int x0, int y0, int rows, int cols, const uint32_t* bm, int wb,
                                         double* out_v, double* out_i) {
    const T* data = (const T*)a.data + tbase;
    const bool has_i = a.intensity != nullptr;
    const T* inten = has_i ? (const T*)a.intensity + tbase : data;    // (no intensity: the same cells again, dropped below)
    const int total = rows * cols, dq = WB_NT / cols, dr = WB_NT - dq * cols;
    double accv = 0.0, acci = 0.0;
    int i = WB_TID, rr = i / cols, xx = i - rr * cols;
    while (i < total) {
        T dv[8], di[8];
        double wq[8];
        unsigned member = 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool live = i < total;
            const int rq = live ? rr : 0, xq = live ? xx : 0;
            bool in = live;
            if (BM) in = in && ((bm[rq * wb + (xq >> 5)] >> (xq & 31)) & 1u);
            int gx = x0 + xq;
            gx = gx < a.W ? gx : a.W - 1;
            const int col = gx >= a.nlon ? gx - a.nlon : gx;                   // W <= 2 nlon
            const int y = y0 + rq;
            const size_t idx = (size_t)y * a.nlon + (size_t)col;
            dv[q] = data[idx];
            di[q] = inten[idx];
            wq[q] = a.wlat[y];
            member |= (in ? 1u : 0u) << q;
            i += WB_NT; xx += dr; rr += dq;
            const bool wrap = xx >= cols;
            xx = wrap ? xx - cols : xx;
            rr = wrap ? rr + 1 : rr;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double pv = wq[q] * (double)dv[q], pi = wq[q] * (double)di[q];
            accv += ((member >> q) & 1u) ? pv : 0.0;
            acci += (((member >> q) & 1u) && has_i) ? pi : 0.0;
        }
    }
    *out_v = accv;                     // (the lane's own partial sums: they stay in its registers -- as a table in LDS they were 1 KB of the
    *out_i = acci;                     //  wave's 11, and at 10 KB a CU holds the sixteen waves the launch is sized for instead of fourteen)
}
WB_DEV inline void wb_props_field_sums(const WbPropsArgs& a, size_t tbase, int x0, int y0, int rows, int cols, const uint32_t* bm, int wb,
                                       double* out_v, double* out_i) {
    if (a.dtype == 0) {
        if (bm) wb_props_field_sums_t<float, true>(a, tbase, x0, y0, rows, cols, bm, wb, out_v, out_i);
        else wb_props_field_sums_t<float, false>(a, tbase, x0, y0, rows, cols, bm, wb, out_v, out_i);
    } else {
        if (bm) wb_props_field_sums_t<double, true>(a, tbase, x0, y0, rows, cols, bm, wb, out_v, out_i);
        else wb_props_field_sums_t<double, false>(a, tbase, x0, y0, rows, cols, bm, wb, out_v, out_i);
    }
}
// The raster of an event in LDS: per row of its bounding box a short list of signed scanline crossings (sorted by the
// row's lane, swept into winding numbers) and one bit per cell.  Events taller than WB_PROPS_LROWS rows, with more cells
// than the bitmap holds or with a row crossed more than WB_PROPS_K times keep the raster in global memory (one int per
// cell of the slot's W x nlat scratch: 1.2 GB fetched + 0.4 GB written per 0.25 degree year, rocprofv3 --pmc, before).
#ifndef WB_PROPS_K
#define WB_PROPS_K 12
#endif
#ifndef WB_PROPS_CHUNK
#define WB_PROPS_CHUNK 4             // events a slot takes from the queue at a time
#endif
#ifndef WB_PROPS_BM
#define WB_PROPS_BM 768              // bitmap words (24 576 cells)
#endif
WB_DEV inline bool wb_bm_test(const uint32_t* bm, int wb, int rr, int xx) { return (bm[rr * wb + (xx >> 5)] >> (xx & 31)) & 1u; }
#if defined(WB_PROPS_VGPRS) && !defined(WB_HOSTSIM)
__attribute__((amdgpu_waves_per_eu(WB_PROPS_VGPRS, WB_PROPS_VGPRS)))   // development: registers for that many waves per SIMD (the rest spills)
#endif
WB_KERNEL(wb_k_props, WbPropsArgs a) {
    WB_SHARED(unsigned long long, xlist8, (WB_PROPS_LROWS * WB_PROPS_K + 3) / 4 > 6 * 64 ? (WB_PROPS_LROWS * WB_PROPS_K + 3) / 4 : 6 * 64);
    unsigned short* xlist = (unsigned short*)xlist8;                  // per row: crossing column (relative to x0) | 0x8000 for a downward edge
    double* lrow6 = (double*)xlist8;                                  // ... and, once the raster stands, the lanes' partial sums over the rows
    WB_SHARED(int, xcnt, WB_PROPS_LROWS);
    WB_SHARED(uint32_t, bm, WB_PROPS_BM);
    WB_SHARED(int, ovf, 1);
    WB_SHARED(int, longe, WB_PROPS_LONG);
    WB_SHARED(int, nlong, 1);
    WB_SHARED(int, bb, 4);
    WB_SHARED(int, wseen, 1);           // winding numbers met: bit 0 positive, bit 1 negative, bit 2 beyond +-1
    WB_SHARED(long long, sh_off, 1);
    // per-row partial sums of the current event: in LDS when the event spans at most WB_PROPS_LROWS rows (the final
    // accumulation over rows is sequential by contract; from global memory every row costs a dependent round trip)
    WB_SHARED(double, lacc, 2 * WB_PROPS_LROWS);       // cells and sum of x per row (boxes and bitmap events; the global raster keeps four sums per row in rowacc)
    const int slot = WB_BLOCK_ID;
    int* cells = a.cells + (size_t)slot * a.W * a.nlat;
    double* rowacc = a.rowacc + (size_t)slot * 4 * a.nlat;
    const int nev = a.ctl->NE;
    // events are handed out one by one (streamers, the expensive ones, come first in the table): with a fixed assignment a
    // slot that drew several long streamers finished long after the others
    WB_SHARED(int, cur, 1);
    WB_RSTAMP_BEGIN                              // development stamps: 0 queue + box, 1 crossings, 2 sweep, 3 rim, 4 sums, 5 tail; +6 for boxes; 11 global raster
    // (WB_PROPS_CHUNK events per visit of the counter: 19 500 returning atomics on ONE word from 3 000 waves are served one after
    //  the other by the L2 -- a quarter of this kernel's wave time was spent between asking for an event and having its box)
    for (;;) {
        WB_SINGLE
        cur[0] = wb_atomic_add(&a.ctl->props_cursor, WB_PROPS_CHUNK);
        WB_END_SINGLE
        const int e0 = cur[0];
        if (e0 >= nev) break;
        for (int e = e0; e < e0 + WB_PROPS_CHUNK && e < nev; ++e) {
        const int* d = a.ev + (size_t)e * 8;
        const int kind = d[0] & 0xFF, job = d[1], cont = d[2];
        const size_t jb = (size_t)job * a.cap_seg;
        const uint32_t* pts = a.cpts + 2 * jb + a.c_off[jb + cont];
        const int n = (kind == WB_EV_STREAMER) ? (d[4] - d[3] + 1) : (kind == WB_EV_CUTOFF ? a.c_len[jb + cont] : 4);
        const size_t tbase = (size_t)(a.job_step ? a.job_step[job] : job / a.nlev) * a.nlat * a.nlon;
        double lane_v = 0.0, lane_i = 0.0;                      // this lane's share of the field sums (wb_props_field_sums)

        WB_SINGLE
        bb[0] = 0x7FFFFFFF; bb[1] = 0x7FFFFFFF; bb[2] = -1; bb[3] = -1;
        nlong[0] = 0;
        wseen[0] = 0;
        WB_END_SINGLE
        {   // a lane's vertices first, then one round of LDS atomics per lane (64 lanes hammering four words serialise)
            int mx0 = 0x7FFFFFFF, my0 = 0x7FFFFFFF, mx1 = -1, my1 = -1;
            for (int k = WB_TID; k < n; k += WB_NT) {
                const uint32_t v = wb_event_vertex(d, pts, k);
                const int x = wb_pt_x(v), y = wb_pt_y(v);
                mx0 = x < mx0 ? x : mx0; my0 = y < my0 ? y : my0; mx1 = x > mx1 ? x : mx1; my1 = y > my1 ? y : my1;
            }
            if (mx1 >= 0) { wb_atomic_min(&bb[0], mx0); wb_atomic_min(&bb[1], my0); wb_atomic_max(&bb[2], mx1); wb_atomic_max(&bb[3], my1); }
            WB_SYNC();
        }
        // An overturning event is an axis-parallel box (overturning_index.py:211): with r <= 1 its cells are exactly the
        // lattice points of the closed box (every other lattice point is at least 1 away), so nothing is rasterised
        const bool full = kind == WB_EV_OVERTURNING && a.r2 <= 1.0;
        const int epad = full ? 0 : a.pad;
        const int x0 = bb[0] - epad < 0 ? 0 : bb[0] - epad;
        const int y0 = bb[1] - epad < 0 ? 0 : bb[1] - epad;
        const int x1 = bb[2] + epad > a.W - 1 ? a.W - 1 : bb[2] + epad;
        const int y1 = bb[3] + epad > a.nlat - 1 ? a.nlat - 1 : bb[3] + epad;
        const int cols = x1 - x0 + 1, rows = y1 - y0 + 1;
        bool lrow = rows <= WB_PROPS_LROWS;                              // per-row counts in LDS (not for the global raster: decided below)
#define WB_ACC(q, y) (lrow ? lacc[(q) * WB_PROPS_LROWS + ((y) - y0)] : rowacc[(q) * a.nlat + (y)])
        WB_RSTAMP(a.ctl, 0);
        const int wb = (cols + 31) >> 5;                                  // bitmap words per row
        bool use_bm = !full && lrow && (long long)rows * wb <= WB_PROPS_BM && cols <= 0x7FFF;
        if (use_bm) {
            // A': signed crossings per row.  Edge a->b spanning row y (half-open in y) covers all x < x_cross.
            WB_PFOR(q, rows)
                xcnt[q] = 0;
                if (q == 0) ovf[0] = 0;
            WB_END_PFOR
            WB_PFOR(q, rows * wb)
                bm[q] = 0u;
            WB_END_PFOR
            WB_PFOR(k, n)
                const uint32_t va = wb_event_vertex(d, pts, (int)k), vb = wb_event_vertex(d, pts, (int)((k + 1) % n));
                const int ax = wb_pt_x(va), ay = wb_pt_y(va), bx = wb_pt_x(vb), by = wb_pt_y(vb);
                if (ay != by) {
                    const int ylo = ay < by ? ay : by, yhi = ay < by ? by : ay;
                    for (int y = ylo; y < yhi; ++y) {
                        int num = (y - ay) * (bx - ax), den = by - ay;        // lattice coordinates are < 2^15: exact in 32 bits
                        if (den < 0) { num = -num; den = -den; }
                        const int cdiv = num >= 0 ? (num + den - 1) / den : -((-num) / den);
                        int ecol = ax + cdiv - 1;                            // last lattice x strictly left of the crossing
                        if (ecol >= x0) {
                            if (ecol > x1) ecol = x1;
                            const int slot = wb_atomic_add(&xcnt[y - y0], 1);
                            if (slot < WB_PROPS_K) xlist[(y - y0) * WB_PROPS_K + slot] = (unsigned short)((ecol - x0) | (by > ay ? 0 : 0x8000));
                            else ovf[0] = 1;
                        }
                    }
                }
            WB_END_PFOR
            if (ovf[0]) use_bm = false;                                  // a row crossed more often than its list holds
            WB_RSTAMP(a.ctl, 1);
        }
        if (use_bm) {
            // B': a lane per row sorts its crossings by column (descending) and sweeps: cells left of a crossing column take its
            //     sign; winding number != 0 -> member
            WB_PFOR(rr, rows)
                const int c = xcnt[rr];
                unsigned short* e = xlist + (int)rr * WB_PROPS_K;          // the row's own slots: sorted in place
                for (int i = 1; i < c; ++i) {                              // insertion sort, descending by column
                    const unsigned short v = e[i];
                    int j = i - 1;
                    while (j >= 0 && (e[j] & 0x7FFF) < (v & 0x7FFF)) { e[j + 1] = e[j]; --j; }
                    e[j + 1] = v;
                }
                int run = 0, seen = 0;
                uint32_t* row = bm + (int)rr * wb;
                for (int i = 0; i < c;) {
                    const int col = e[i] & 0x7FFF;
                    while (i < c && (e[i] & 0x7FFF) == col) { run += (e[i] & 0x8000) ? -1 : 1; ++i; }
                    const int lo = i < c ? (e[i] & 0x7FFF) + 1 : 0;         // cells lo .. col carry this winding number
                    if (run != 0) {
                        seen |= (run > 0 ? 1 : 2) | ((run > 1 || run < -1) ? 4 : 0);
                        for (int w = lo >> 5; w <= col >> 5; ++w) {
                            const int b0 = w == lo >> 5 ? lo & 31 : 0, b1 = w == col >> 5 ? col & 31 : 31;
                            row[w] |= (b1 == 31 ? 0xFFFFFFFFu : ((1u << (b1 + 1)) - 1u)) & ~((1u << b0) - 1u);
                        }
                    }
                }
                if (seen & ~wseen[0]) wb_atomic_or(&wseen[0], seen);
            WB_END_PFOR
            WB_RSTAMP(a.ctl, 2);
            // C': rim: lattice points closer than r to an edge (boundary points included: d = 0)
            WB_PFOR(k, n)
                const uint32_t va = wb_event_vertex(d, pts, (int)k), vb = wb_event_vertex(d, pts, (int)((k + 1) % n));
                const int ax = wb_pt_x(va), ay = wb_pt_y(va), bx = wb_pt_x(vb), by = wb_pt_y(vb);
                const long long dx = bx - ax, dy = by - ay, L = dx * dx + dy * dy;
                if ((double)L * a.r2 <= 1.0 && a.r2 <= 1.0) {
                    // short edge: only the lattice points ON it count (gcd + 1 of them, end points included)
                    int g = (int)(dx < 0 ? -dx : dx), h = (int)(dy < 0 ? -dy : dy);
                    while (h) { const int t = g % h; g = h; h = t; }
                    if (g == 0) g = 1;
                    const int sx = (int)dx / g, sy = (int)dy / g;
                    for (int t = 0; t <= g; ++t) {
                        const int x = ax + t * sx, y = ay + t * sy;
                        if (x >= x0 && x <= x1 && y >= y0 && y <= y1) wb_atomic_or(&bm[(y - y0) * wb + ((x - x0) >> 5)], 1u << ((x - x0) & 31));
                    }
                    continue;
                }
                // long edge (a streamer's closing chord): queued and tested by the whole workgroup below; only if the queue is
                // full does this lane walk the box itself
                const int slot_e = wb_atomic_add(&nlong[0], 1);
                if (slot_e < WB_PROPS_LONG) { longe[slot_e] = (int)k; continue; }
                int ex0 = (ax < bx ? ax : bx) - a.pad, ex1 = (ax < bx ? bx : ax) + a.pad;
                int ey0 = (ay < by ? ay : by) - a.pad, ey1 = (ay < by ? by : ay) + a.pad;
                if (ex0 < x0) ex0 = x0;
                if (ex1 > x1) ex1 = x1;
                if (ey0 < y0) ey0 = y0;
                if (ey1 > y1) ey1 = y1;
                for (int y = ey0; y <= ey1; ++y)
                    for (int x = ex0; x <= ex1; ++x)
                        if (wb_near_segment(x, y, ax, ay, bx, by, a.r2)) wb_atomic_or(&bm[(y - y0) * wb + ((x - x0) >> 5)], 1u << ((x - x0) & 31));
            WB_END_PFOR
            {
                const int nq = nlong[0] < WB_PROPS_LONG ? nlong[0] : WB_PROPS_LONG;
                for (int qe = 0; qe < nq; ++qe) {
                    const int k = longe[qe];
                    const uint32_t va = wb_event_vertex(d, pts, k), vb = wb_event_vertex(d, pts, (k + 1) % n);
                    const int ax = wb_pt_x(va), ay = wb_pt_y(va), bx = wb_pt_x(vb), by = wb_pt_y(vb);
                    int ex0 = (ax < bx ? ax : bx) - a.pad, ex1 = (ax < bx ? bx : ax) + a.pad;
                    int ey0 = (ay < by ? ay : by) - a.pad, ey1 = (ay < by ? by : ay) + a.pad;
                    if (ex0 < x0) ex0 = x0;
                    if (ex1 > x1) ex1 = x1;
                    if (ey0 < y0) ey0 = y0;
                    if (ey1 > y1) ey1 = y1;
                    const int ew = ex1 - ex0 + 1, eh = ey1 - ey0 + 1;
                    if (ew <= 0 || eh <= 0) continue;                        // block-uniform
                    WB_PFOR(c, (int64_t)ew * eh)
                        const int x = ex0 + (int)(c / eh), y = ey0 + (int)(c % eh);
                        if (wb_near_segment(x, y, ax, ay, bx, by, a.r2)) wb_atomic_or(&bm[(y - y0) * wb + ((x - x0) >> 5)], 1u << ((x - x0) & 31));
                    WB_END_PFOR
                }
            }
        }
        if (use_bm) WB_RSTAMP(a.ctl, 3);
        if (!full && !use_bm) {
        lrow = false;
        WB_SINGLE
        nlong[0] = 0;                                                     // (the LDS attempt may have queued long edges)
        wseen[0] = 0;
        WB_END_SINGLE
        WB_PFOR(idx, (int64_t)rows * cols)
            cells[idx] = 0;
        WB_END_PFOR
        // A: signed crossing counts.  Edge a->b spanning row y (half-open in y) covers all x < x_cross.
        WB_PFOR(k, n)
            const uint32_t va = wb_event_vertex(d, pts, (int)k), vb = wb_event_vertex(d, pts, (int)((k + 1) % n));
            const int ax = wb_pt_x(va), ay = wb_pt_y(va), bx = wb_pt_x(vb), by = wb_pt_y(vb);
            if (ay != by) {
                const int sgn = by > ay ? 1 : -1;
                const int ylo = ay < by ? ay : by, yhi = ay < by ? by : ay;
                for (int y = ylo; y < yhi; ++y) {
                    // lattice coordinates are < 2^15: 32-bit arithmetic is exact here
                    int num = (y - ay) * (bx - ax), den = by - ay;
                    if (den < 0) { num = -num; den = -den; }
                    const int cdiv = num >= 0 ? (num + den - 1) / den : -((-num) / den);
                    int ecol = ax + cdiv - 1;                            // last lattice x strictly left of the crossing
                    if (ecol >= x0) {
                        if (ecol > x1) ecol = x1;
                        wb_atomic_add(&cells[(size_t)(ecol - x0) * rows + (y - y0)], sgn);
                    }
                }
            }
        WB_END_PFOR
        // B: suffix sums -> winding number != 0.  Cells are stored column-major (x outer, y inner): the lane = row
        //    sweeps below are coalesced, and the 8-wide chunks let the loads of a chunk issue back to back
        WB_PFOR(rr, rows)
            int run = 0, seen = 0;
            int* col = cells + rr;
            for (int x = cols - 1; x >= 0; x -= 8) {
                int v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (x - q >= 0) ? col[(size_t)(x - q) * rows] : 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    run += v[q];
                    seen |= (run > 0 ? 1 : 0) | (run < 0 ? 2 : 0) | ((run > 1 || run < -1) ? 4 : 0);
                    if (x - q >= 0) col[(size_t)(x - q) * rows] = run != 0 ? 1 : 0;
                }
            }
            if (seen & ~wseen[0]) wb_atomic_or(&wseen[0], seen);
        WB_END_PFOR
        // C: rim: lattice points closer than r to an edge (boundary points included: d = 0)
        WB_PFOR(k, n)
            const uint32_t va = wb_event_vertex(d, pts, (int)k), vb = wb_event_vertex(d, pts, (int)((k + 1) % n));
            const int ax = wb_pt_x(va), ay = wb_pt_y(va), bx = wb_pt_x(vb), by = wb_pt_y(vb);
            const long long dx = bx - ax, dy = by - ay, L = dx * dx + dy * dy;
            if ((double)L * a.r2 <= 1.0 && a.r2 <= 1.0) {
                // short edge: a lattice point off its line is at least 1/sqrt(L) >= r away, and one beyond its ends at
                // least 1 >= r: only the lattice points ON the edge count (gcd + 1 of them, end points included)
                int g = (int)(dx < 0 ? -dx : dx), h = (int)(dy < 0 ? -dy : dy);
                while (h) { const int t = g % h; g = h; h = t; }
                if (g == 0) g = 1;
                const int sx = (int)dx / g, sy = (int)dy / g;
                for (int t = 0; t <= g; ++t) {
                    const int x = ax + t * sx, y = ay + t * sy;
                    if (x >= x0 && x <= x1 && y >= y0 && y <= y1) cells[(size_t)(x - x0) * rows + (y - y0)] = 1;
                }
                if (L == 0 && ax >= x0 && ax <= x1 && ay >= y0 && ay <= y1) cells[(size_t)(ax - x0) * rows + (ay - y0)] = 1;
                continue;
            }
            // long edge (a streamer's closing chord): its box can hold hundreds of lattice points -- queued and tested
            // by the whole workgroup below; only if the queue is full does this lane walk the box itself
            const int slot_e = wb_atomic_add(&nlong[0], 1);
            if (slot_e < WB_PROPS_LONG) { longe[slot_e] = (int)k; continue; }
            int ex0 = (ax < bx ? ax : bx) - a.pad, ex1 = (ax < bx ? bx : ax) + a.pad;
            int ey0 = (ay < by ? ay : by) - a.pad, ey1 = (ay < by ? by : ay) + a.pad;
            if (ex0 < x0) ex0 = x0;
            if (ex1 > x1) ex1 = x1;
            if (ey0 < y0) ey0 = y0;
            if (ey1 > y1) ey1 = y1;
            for (int y = ey0; y <= ey1; ++y)
                for (int x = ex0; x <= ex1; ++x)
                    if (wb_near_segment(x, y, ax, ay, bx, by, a.r2)) cells[(size_t)(x - x0) * rows + (y - y0)] = 1;
        WB_END_PFOR
        {
            const int nq = nlong[0] < WB_PROPS_LONG ? nlong[0] : WB_PROPS_LONG;
            for (int qe = 0; qe < nq; ++qe) {
                const int k = longe[qe];
                const uint32_t va = wb_event_vertex(d, pts, k), vb = wb_event_vertex(d, pts, (k + 1) % n);
                const int ax = wb_pt_x(va), ay = wb_pt_y(va), bx = wb_pt_x(vb), by = wb_pt_y(vb);
                int ex0 = (ax < bx ? ax : bx) - a.pad, ex1 = (ax < bx ? bx : ax) + a.pad;
                int ey0 = (ay < by ? ay : by) - a.pad, ey1 = (ay < by ? by : ay) + a.pad;
                if (ex0 < x0) ex0 = x0;
                if (ex1 > x1) ex1 = x1;
                if (ey0 < y0) ey0 = y0;
                if (ey1 > y1) ey1 = y1;
                const int ew = ex1 - ex0 + 1, eh = ey1 - ey0 + 1;
                if (ew <= 0 || eh <= 0) continue;                        // block-uniform
                WB_PFOR(c, (int64_t)ew * eh)
                    const int x = ex0 + (int)(c / eh), y = ey0 + (int)(c % eh);
                    if (wb_near_segment(x, y, ax, ay, bx, by, a.r2)) cells[(size_t)(x - x0) * rows + (y - y0)] = 1;
                WB_END_PFOR
            }
        }
        }   // !full
        if (full) {
            // D for boxes: every cell counts, so the integer sums are closed forms; the field sums with the lanes along the rows
            // (a kernel of their own for these sums, on the side stream, was tried in round 3: it sat behind the persistent
            // all-pairs kernel there and then ran beside routine 1 -- 0.3 ms MORE per pass)
            WB_PFOR(rr, rows)
                WB_ACC(0, y0 + (int)rr) = (double)cols;
                WB_ACC(1, y0 + (int)rr) = (double)((long long)cols * x0 + (long long)cols * (cols - 1) / 2);
            WB_END_PFOR
            wb_props_field_sums(a, tbase, x0, y0, rows, cols, nullptr, 0, &lane_v, &lane_i);
        } else if (use_bm) {
            // D': integer counts per row from the bitmap (a lane per row) ...
            WB_PFOR(rr, rows)
                const uint32_t* row = bm + (int)rr * wb;
                long long cnt = 0, sx = 0;
                for (int w = 0; w < wb; ++w) {
                    uint32_t v = row[w];
                    const int pc = wb_popc32(v);
                    cnt += pc;
                    sx += (long long)pc * (x0 + 32 * w);
                    while (v) { sx += wb_ctz32(v); v &= v - 1u; }
                }
                WB_ACC(0, y0 + (int)rr) = (double)cnt;
                WB_ACC(1, y0 + (int)rr) = (double)sx;
            WB_END_PFOR
            // ... and the field sums with the lanes along the rows.  (Over the box: its cells are read whether they are members or not.
            // Round 6 measured what that wastes -- the members are 50 % of a streamer's box, 65 % of a cutoff's, all of an overturning
            // box: 62 M member cells of 78 M box cells per 0.25 degree year -- and tried sums over the rows' member spans instead:
            // they never held less than half a box, and the search for a cell's row cost more than the loads it saved.)
            wb_props_field_sums(a, tbase, x0, y0, rows, cols, bm, wb, &lane_v, &lane_i);
        } else {
        // D: per-row counts and sums
        WB_PFOR(rr, rows)
            const int y = y0 + (int)rr;
            const int* col = cells + rr;
            long long cnt = 0, sx = 0;
            double sv = 0.0, si = 0.0;
            for (int x = 0; x < cols; x += 8) {
                int fl[8];
                double dv[8], di[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) fl[q] = (x + q < cols) ? (full ? 1 : col[(size_t)(x + q) * rows]) : 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {                       // field loads of the whole chunk, in flight together
                    const int gx = x0 + x + q;
                    const size_t idx = tbase + (size_t)y * a.nlon + (size_t)((gx < a.W ? gx : a.W - 1) % a.nlon);
                    dv[q] = fl[q] ? wb_load_field(a.data, idx, a.dtype) : 0.0;
                    di[q] = (fl[q] && a.intensity) ? wb_load_field(a.intensity, idx, a.dtype) : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (fl[q]) {
                        ++cnt;
                        sx += x0 + x + q;
                        sv += dv[q];
                        si += di[q];
                    }
                }
            }
            WB_ACC(0, y) = (double)cnt;
            WB_ACC(1, y) = (double)sx;
            WB_ACC(2, y) = sv;
            WB_ACC(3, y) = si;
        WB_END_PFOR
        }   // full / generic D
        WB_RSTAMP(a.ctl, full ? 10 : (use_bm ? 4 : 11));
        // the rows' products: every lane adds up its rows (one lane walking them all, a load of wlat per row, was a fifth of the
        // kernel), the lanes' partial sums are then added in lane order.  (The row-wise sums are the fast path: within
        // wb_com_ulps(rows) of the reference's compensated cell-wise sums whatever their order; the truncation test below
        // decides whether that is enough.)
        const bool lane_sums = full || use_bm;                             // (data, intensity) came from wb_props_field_sums
        {
            double pa = 0.0, px = 0.0, py = 0.0, pv = 0.0, pi = 0.0, pn = 0.0;
            for (int y = y0 + WB_TID; y <= y1; y += WB_NT) {
                const double cnt = WB_ACC(0, y);
                if (cnt == 0.0) continue;
                const double w = a.wlat[y];
                pa = pa + w * cnt;
                px = px + w * WB_ACC(1, y);
                py = py + w * ((double)y * cnt);
                if (!lane_sums) { pv = pv + w * WB_ACC(2, y); pi = pi + w * WB_ACC(3, y); }
                pn += cnt;
            }
            if (lane_sums) { pv = lane_v; pi = lane_i; }
            WB_SYNC();
            // the lanes' shares added by butterfly (a fixed order: the same bits on every run); lane 0 adding 64 x 6 values
            // from LDS one after the other was 10 us per event
            pa = wb_block_sum_f64(pa, lrow6); px = wb_block_sum_f64(px, lrow6); py = wb_block_sum_f64(py, lrow6);
            pv = wb_block_sum_f64(pv, lrow6); pi = wb_block_sum_f64(pi, lrow6); pn = wb_block_sum_f64(pn, lrow6);
            if (WB_TID == 0) { lrow6[0] = pa; lrow6[1] = px; lrow6[2] = py; lrow6[3] = pv; lrow6[4] = pi; lrow6[5] = pn; }
            WB_SYNC();
        }
        WB_SINGLE
        double areas = lrow6[0], xc = lrow6[1], yc = lrow6[2], mv = lrow6[3], it = lrow6[4];
        long long nc = (long long)lrow6[5];
        // boxes: areas / x_com / y_com come from wb_k_box_com (which ran beside the streamer kernels)
        if (full) {
            a.ev_sums[(size_t)e * 5 + 0] = a.ot_sums[(size_t)d[7] * 3 + 0];
            a.ev_sums[(size_t)e * 5 + 3] = a.ot_sums[(size_t)d[7] * 3 + 1];
            a.ev_sums[(size_t)e * 5 + 4] = a.ot_sums[(size_t)d[7] * 3 + 2];
        } else {
            const double ulps = wb_com_ulps(rows);
            if (nc > 0 && (wb_trunc_unsafe(xc, areas, ulps) || wb_trunc_unsafe(yc, areas, ulps))) {
                // the reference's order: cells (y, x) ascending, one product per cell, compensated sums
                double sa = 0.0, ca = 0.0, sx = 0.0, cx = 0.0, sy = 0.0, cy = 0.0;
                for (int y = y0; y <= y1; ++y) {
                    if (WB_ACC(0, y) == 0.0) continue;
                    const double w = a.wlat[y];
                    const double yw = (double)y * w;
                    const int* col = cells + (y - y0);
                    for (int x = 0; x < cols; ++x) {
                        if (use_bm ? !wb_bm_test(bm, wb, y - y0, x) : !col[(size_t)x * rows]) continue;
                        const double xw = (double)(x0 + x) * w;
                        WB_KAHAN(sa, ca, w);
                        WB_KAHAN(sx, cx, xw);
                        WB_KAHAN(sy, cy, yw);
                    }
                }
                areas = sa; xc = sx; yc = sy;
                wb_atomic_add(&a.ctl->com_rechecks[0], 1);
            }
            a.ev_sums[(size_t)e * 5 + 0] = areas;
            a.ev_sums[(size_t)e * 5 + 3] = xc;
            a.ev_sums[(size_t)e * 5 + 4] = yc;
        }
        a.ev_sums[(size_t)e * 5 + 1] = mv;
        a.ev_sums[(size_t)e * 5 + 2] = it;
        // a simple polygon has ONE non-zero winding number, +1 or -1 (events whose polygon is not simple are
        // flagged; the reference hands such polygons to GEOS' make_valid, index_utils.py:164, which is unpinned here)
        a.ev_flags[e] = (!full && ((wseen[0] & 4) || (wseen[0] & 3) == 3)) ? 1 : 0;
        a.ev_ncells[e] = (int)nc;
        long long off = -1;
        if (a.want_cells) {
            off = (long long)wb_atomic_add(&a.ctl->cell_count, (unsigned long long)nc);
            if ((unsigned long long)off + (unsigned long long)nc > a.cell_cap) off = -1;
            // per-row start offsets (rows ascending, x ascending inside a row)
            long long run = off;
            for (int y = y0; y <= y1; ++y) {
                const double cnt = WB_ACC(0, y);
                WB_ACC(0, y) = (double)run;
                run += (long long)cnt;
            }
            a.ev_cell_off[e] = off;
        }
        sh_off[0] = off;
        WB_END_SINGLE
        if (a.want_cells && sh_off[0] >= 0) {
            WB_PFOR(rr, rows)
                const int y = y0 + (int)rr;
                const int* col = cells + rr;
                long long o = (long long)WB_ACC(0, y);
                for (int x = 0; x < cols; ++x)
                    if (full || (use_bm ? wb_bm_test(bm, wb, (int)rr, x) : col[(size_t)x * rows] != 0)) a.cell_pool[o++] = (uint32_t)(x0 + x) | ((uint32_t)y << 16);
            WB_END_PFOR
        }
#undef WB_ACC
        WB_RSTAMP(a.ctl, 5);
        }
    }
    WB_RSTAMP_FLUSH(a.ctl);
}

// areas / x_com / y_com of the overturning boxes (the events wb_k_props does not rasterise): every cell of the closed box
// counts, so the row-wise sums are closed forms -- wb_k_overturn evaluates them.  A box with min_lon + max_lon even has
// x_com / areas = an integer mathematically -- half of all boxes need the reference-order re-evaluation, a chain of
// rows x cols dependent compensated additions (up to 10^5 cells): wb_k_overturn puts those on a queue, and this kernel's
// persistent single-wave workgroups take them one at a time (round 4; until then two workgroups per contour walked that
// contour's boxes, most of them finding none in need: 1.05 ms, 0.55 now).  One wave per box makes the chain row-parallel
// and still bit-exact:
//   * a step of pandas' compensated sum (y = v - c; t = s + y; c = t - s - y) is EXACT -- s - c grows by exactly v -- unless
//     v - c needs rounding, which happens where the terms step up into a higher binade (c still carries the finer bits of
//     the previous terms); between such steps the state is a function of the exact running total E alone: s = RN(E), c = s - E;
//   * pass 1 (lane per row): the row's terms fl(x * w) cut into segments at every binade step-up; first term and exact
//     128-bit fixed-point total of the rest of each segment;
//   * pass 2 (one lane per chain, O(1) per segment): state from E, the REAL step for the segment's first term, E := s - c
//     exactly, E += rest; this predicts the state at the start of every row;
//   * pass 3 (lane per row): the real recurrence over the whole row from the predicted state; its end state must equal the
//     state predicted for the next row bit for bit.  All rows verified => the chained states are exactly those of the
//     sequential walk (induction from the first row, whose start state 0, 0 is certain).  The rows are taken in chunks of
//     WB_BOXCOM_ROWS; any mismatch in a chunk (a rounding tie, an inexact step elsewhere, more than WB_BOXCOM_SEGS segments
//     in a row) => one lane walks THAT CHUNK's cells from the chunk's true start state, and the next chunk is predicted again
//     (boxes with such a chunk are counted in com_rechecks[2]); correctness never rests on the prediction.  (Until round 3 a
//     box taller than 128 rows, wider than 4 binades or with one mismatch was walked whole: 16 of the bench batch's 6 391
//     boxes, the largest of them 1.6 ms on one lane -- the kernel's whole duration.)
#ifndef WB_BOXCOM_ROWS
#define WB_BOXCOM_ROWS 64            // rows of a box per chunk (a lane each; 64 / WB_BOXCOM_ROWS boxes share a wave: 16 rows x 4 boxes
                                     // ran pass 2 on twelve lanes instead of three and the kernel in 0.93 ms instead of 0.55 -- a quarter of
                                     // the waves, each three times as long in passes 1 and 3: the kernel is short of waves, not of lanes)
#endif
#ifndef WB_BOXCOM_FORCE_WALK
#define WB_BOXCOM_FORCE_WALK 0       // 1 (a test build): no chunk is predicted, one lane walks them all
#endif
#ifndef WB_BOXCOM_SEGS
#define WB_BOXCOM_SEGS 6             // binade steps of x * w along a row the prediction follows (x < 4096 admits 12: a row with more is walked;
                                     // 6 halve the kernel's LDS -- 13 KB a wave -- and what it takes from the all-pairs kernel it runs beside)
#endif
struct WbBoxComArgs {
    int nlat, W;
    WbCtl* ctl;                   // n_boxq queued boxes, boxq_cursor; com_rechecks[2]: boxes with a chunk walked by one lane
    const int* boxq;              // [n_boxq] slot (selected contour * WB_OT_CAP + box) of a queued box (wb_k_overturn)
    const int* ot_box;            // [nsel][WB_OT_CAP][4]
    const double* wlat;
    double* ot_sums;              // [nsel][WB_OT_CAP][3] areas, x_com, y_com
};
WB_DEV inline int wb_dbl_exp(double v) { unsigned long long m; int e; wb_dbl_split(v, &m, &e); return m ? e : -100000; }

// WB_BOXCOM_G boxes at a time per workgroup (one wave), WB_BOXCOM_ROWS rows of each per chunk: passes 1 and 3 have a lane per
// (box, row), pass 2 -- the sequential one, three quarters of this kernel's wave time when a wave held ONE box and ran it on
// three lanes -- a lane per (box, chain).
#define WB_BOXCOM_G (64 / WB_BOXCOM_ROWS > 0 ? 64 / WB_BOXCOM_ROWS : 1)
#define WB_BOXCOM_NL (WB_BOXCOM_G * WB_BOXCOM_ROWS)
#define WB_BOXCOM_ST(q, g, r) st[((q) * WB_BOXCOM_G + (g)) * (WB_BOXCOM_ROWS + 1) + (r)]
WB_KERNEL(wb_k_box_com, WbBoxComArgs a) {
    WB_SHARED(double, segv, WB_BOXCOM_SEGS * WB_BOXCOM_NL);                // x chain: first term of a row's segment (index: segment, lane)
    WB_SHARED(unsigned long long, seglo, (WB_BOXCOM_SEGS > 5 ? WB_BOXCOM_SEGS : 5) * WB_BOXCOM_NL);   // ... exact total of the segment's other terms
    WB_SHARED(unsigned long long, seghi, (WB_BOXCOM_SEGS > 5 ? WB_BOXCOM_SEGS : 5) * WB_BOXCOM_NL);   // (at least five words per lane: the lanes' totals of the bound's pre-pass)
    WB_SHARED(int, nseg, WB_BOXCOM_NL);
    WB_SHARED(double, st, 6 * WB_BOXCOM_G * (WB_BOXCOM_ROWS + 1));   // per chain and box: state s, c at the start of every row of the chunk (+ after its last)
    WB_SHARED(int, bx, 12 * WB_BOXCOM_G);                         // per box: x0, y0, x1, y1, rows, first row of the chunk, rows of the chunk, unit exponent, largest exponent, whole box by one lane, this chunk by one lane, a chunk was walked
    WB_SHARED(int, ctl, 2);                                      // [0] the first queue entry taken, [1] some box has a chunk left

    WB_BSTAMP_BEGIN                                              // development stamps: 0 closed forms, 1 .. 3 the passes, 4 the one-lane walks / output
    // persistent: a workgroup (one wave) takes the next WB_BOXCOM_G queued boxes until the queue is empty -- only the boxes that
    // need the reference-order sums are here (wb_k_overturn queued them), whatever contour they belong to
    for (;;) {
    WB_SINGLE
    ctl[0] = wb_atomic_add(&a.ctl->boxq_cursor, WB_BOXCOM_G);
    WB_END_SINGLE
    const int q0 = ctl[0], nq = a.ctl->n_boxq;
    if (q0 >= nq) break;
    WB_PFOR(g, WB_BOXCOM_G)
        int* b = bx + 12 * g;
        int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
        if (q0 + (int)g < nq) {
            const int* d = a.ot_box + (size_t)(a.boxq[q0 + (int)g] & 0x3FFFFFFF) * 4;
            x0 = d[0]; y0 = d[1]; x1 = d[2]; y1 = d[3];
            if (x0 < 0) x0 = 0;
            if (y0 < 0) y0 = 0;
            if (x1 > a.W - 1) x1 = a.W - 1;
            if (y1 > a.nlat - 1) y1 = a.nlat - 1;
        }
        b[0] = x0; b[1] = y0; b[2] = x1; b[3] = y1;
        b[4] = (q0 + (int)g < nq) ? y1 - y0 + 1 : 0;             // rows (0: no box in this place, or an empty one)
        b[5] = 0; b[6] = 0; b[7] = 4096; b[8] = -4096; b[9] = 0; b[10] = 0; b[11] = 0;
    WB_END_PFOR
    WB_PFOR(l, WB_BOXCOM_NL)                                     // exponent range of the rows' weights
        const int g = (int)l / WB_BOXCOM_ROWS;
        int* b = bx + 12 * g;
        for (int r = (int)l % WB_BOXCOM_ROWS; r < b[4]; r += WB_BOXCOM_ROWS) {
            unsigned long long m; int ex;
            wb_dbl_split(a.wlat[b[1] + r], &m, &ex);
            if (m) { wb_atomic_min(&b[7], ex); wb_atomic_max(&b[8], ex); }
        }
    WB_END_PFOR
    WB_PFOR(g, WB_BOXCOM_G)
        int* b = bx + 12 * g;
        const int emin = b[7], emax = b[8];
        // every term is below max(x1, y1) * w_max < 2^(emax + 53 + 16); the total of < 2^32 cells must stay below 2^127
        b[9] = (emin > emax || (emax - emin) + 53 + 16 + 32 > 126) ? 1 : 0;
        for (int q = 0; q < 6; ++q) WB_BOXCOM_ST(q, g, 0) = 0.0;            // the sums start at s = c = 0: certain
    WB_END_PFOR
    WB_BSTAMP(a.ctl, 0);
    // (Most boxes never come here: wb_k_overturn settles a box whose three sums are decided by the EXACT totals of their terms --
    //  the value pandas returns is s after the last step, and s = RN(E) with E the running total as the compensated sum carries it:
    //  E grows by exactly the term v at every step except where y = v - c is rounded, and then by v + e1 with |e1| <= ulp(y) / 2
    //  (t - s and (t - s) - y are exact: c is the exact error of s + y).  y = v - c can need rounding only (i) where v's unit is
    //  coarser than c's -- c is a multiple of the unit of the previous term's y, so: where the sequence of terms steps up into a
    //  higher binade -- or (ii) where v - c leaves v's binade (v within |c| <= ulp(S) / 2 of an edge of it).  So S = RN(T + d), T the
    //  exact total of the rounded terms, |d| <= B = the sum of ulp(v) over the steps of kind (i), and T, B and "no term of kind
    //  (ii)" are sums / minima over the cells: no chain.  Where RN(T - B) == RN(T + B) for each of the three sums that IS the
    //  reference's value, bit for bit (wb_core.h: wb_box_row_terms, wb_box_decide); the boxes for which it is not -- a rounding
    //  boundary within B of T: 1.5 % of them at 0.25 degree, 6 % at 1 degree -- are queued for the chained passes below.  A build
    //  with -DWB_BOXCOM_BOUND=2 queues EVERY box, marked with what the totals said, and counts the boxes on which the chained
    //  evaluation differs in a single bit: none of 23 217 on the grids of profiles/r06_boxbound_check.txt.)
    // The boxes in chunks of WB_BOXCOM_ROWS rows.  A chunk starts from the TRUE state of the three sums (verified by the
    // previous chunk, or walked) in st[..][0]; if it verifies, its end state is true as well and becomes the next start; if
    // not (a rounding tie, an inexact step the prediction missed, more binade steps in a row than WB_BOXCOM_SEGS), one lane
    // walks the chunk's cells in the reference's order from that start -- and the chunk after it is predicted again.
    for (;;) {
        WB_SINGLE
        ctl[1] = 0;
        WB_END_SINGLE
        WB_PFOR(g, WB_BOXCOM_G)
            int* b = bx + 12 * g;
            const int left = b[9] ? 0 : b[4] - b[5];
            b[6] = left < WB_BOXCOM_ROWS ? (left > 0 ? left : 0) : WB_BOXCOM_ROWS;
            b[10] = WB_BOXCOM_FORCE_WALK;
            if (b[6] > 0) ctl[1] = 1;
        WB_END_PFOR
        if (!ctl[1]) break;
        // pass 1: segments of the x chain, row by row
        WB_PFOR(l, WB_BOXCOM_NL)
            const int g = (int)l / WB_BOXCOM_ROWS, r = (int)l % WB_BOXCOM_ROWS;
            int* b = bx + 12 * g;
            if (r < b[6] && !b[10]) {
                const int x0 = b[0], x1 = b[2], ue = b[7];
                const double w = a.wlat[b[1] + b[5] + r];
                int ns = 0, eprev = 0;
                wb_u128 rest = 0;
                bool over = false;
                for (int x = x0; x <= x1; ++x) {
                    const double p = (double)x * w;
                    const int ep = wb_dbl_exp(p);
                    if (x == x0 || ep > eprev) {                     // row start or a step up into a higher binade: new segment
                        if (ns > 0) { seglo[(ns - 1) * WB_BOXCOM_NL + l] = (unsigned long long)rest; seghi[(ns - 1) * WB_BOXCOM_NL + l] = (unsigned long long)(rest >> 64); }
                        if (ns == WB_BOXCOM_SEGS) { over = true; break; }
                        segv[ns * WB_BOXCOM_NL + l] = p;
                        rest = 0;
                        ++ns;
                    } else {
                        rest += wb_fix128(p, ue);
                    }
                    eprev = ep;
                }
                if (ns > 0) { seglo[(ns - 1) * WB_BOXCOM_NL + l] = (unsigned long long)rest; seghi[(ns - 1) * WB_BOXCOM_NL + l] = (unsigned long long)(rest >> 64); }
                nseg[l] = ns;
                if (over) b[10] = 1;
            }
        WB_END_PFOR
        WB_BSTAMP(a.ctl, 1);
        // pass 2: predicted state at every row start; chains 0 areas (terms w), 1 x_com (x * w), 2 y_com (y * w)
        WB_PFOR(l, WB_BOXCOM_NL)
            const int g = (int)l / WB_BOXCOM_ROWS;
            int* b = bx + 12 * g;
            const int nr = b[6];
            for (int ch = (int)l % WB_BOXCOM_ROWS; ch < 3 && nr > 0 && !b[10]; ch += WB_BOXCOM_ROWS) {   // (a lane per chain; chunks of fewer than three rows: a lane takes several)
                const int ue = b[7], cols = b[2] - b[0] + 1;
                wb_u128 E = 0;
                bool ok = true;
                for (int r = 0; r < nr && ok; ++r) {
                    const int y = b[1] + b[5] + r, lr = g * WB_BOXCOM_ROWS + r;
                    const double w = a.wlat[y];
                    const int ns = ch == 1 ? nseg[lr] : 1;
                    for (int k = 0; k < ns && ok; ++k) {
                        double sv, cv;
                        if (r == 0 && k == 0) {                      // the chunk's start: the true state, as it is
                            sv = WB_BOXCOM_ST(2 * ch, g, 0); cv = WB_BOXCOM_ST(2 * ch + 1, g, 0);
                        } else {
                            ok = wb_state_from_exact(E, ue, &sv, &cv);
                            if (k == 0) { WB_BOXCOM_ST(2 * ch, g, r) = sv; WB_BOXCOM_ST(2 * ch + 1, g, r) = cv; }
                        }
                        double v;
                        wb_u128 rest;
                        if (ch == 1) {
                            v = segv[k * WB_BOXCOM_NL + lr];
                            rest = ((wb_u128)seghi[k * WB_BOXCOM_NL + lr] << 64) | seglo[k * WB_BOXCOM_NL + lr];
                        } else {
                            v = ch == 0 ? w : (double)y * w;
                            unsigned long long m; int ex;
                            wb_dbl_split(v, &m, &ex);
                            rest = m ? (((wb_u128)m * (unsigned)(cols - 1)) << (ex - ue)) : (wb_u128)0;
                        }
                        WB_KAHAN(sv, cv, v);                          // the real step for the segment's first term
                        E = wb_fix128(sv, ue);                         // s - c exactly
                        if (cv > 0.0) { const wb_u128 f = wb_fix128(cv, ue); if (f > E) ok = false; else E -= f; }
                        else if (cv < 0.0) E += wb_fix128(-cv, ue);
                        E += rest;
                    }
                }
                double sv = 0.0, cv = 0.0;
                ok = ok && wb_state_from_exact(E, ue, &sv, &cv);
                WB_BOXCOM_ST(2 * ch, g, nr) = sv;
                WB_BOXCOM_ST(2 * ch + 1, g, nr) = cv;
                if (!ok) b[10] = 1;
            }
        WB_END_PFOR
        WB_BSTAMP(a.ctl, 2);
        // pass 3: the real recurrence over every row from its predicted state; the end must be the next row's start
        WB_PFOR(l, WB_BOXCOM_NL)
            const int g = (int)l / WB_BOXCOM_ROWS, r = (int)l % WB_BOXCOM_ROWS;
            int* b = bx + 12 * g;
            if (r < b[6] && !b[10]) {
                const int x0 = b[0], x1 = b[2], y = b[1] + b[5] + r;
                const double w = a.wlat[y], yw = (double)y * w;
                double sa = WB_BOXCOM_ST(0, g, r), ca = WB_BOXCOM_ST(1, g, r);
                double sx = WB_BOXCOM_ST(2, g, r), cx = WB_BOXCOM_ST(3, g, r);
                double sy = WB_BOXCOM_ST(4, g, r), cy = WB_BOXCOM_ST(5, g, r);
                for (int x = x0; x <= x1; ++x) {
                    const double xw = (double)x * w;
                    WB_KAHAN(sa, ca, w);
                    WB_KAHAN(sx, cx, xw);
                    WB_KAHAN(sy, cy, yw);
                }
                if (!(sa == WB_BOXCOM_ST(0, g, r + 1) && ca == WB_BOXCOM_ST(1, g, r + 1) &&
                      sx == WB_BOXCOM_ST(2, g, r + 1) && cx == WB_BOXCOM_ST(3, g, r + 1) &&
                      sy == WB_BOXCOM_ST(4, g, r + 1) && cy == WB_BOXCOM_ST(5, g, r + 1))) b[10] = 1;
            }
        WB_END_PFOR
        WB_BSTAMP(a.ctl, 3);
        WB_PFOR(g, WB_BOXCOM_G)                                      // a lane per box: the chunk's end state becomes the next chunk's start
            int* b = bx + 12 * g;
            const int nr = b[6];
            if (nr > 0) {
                double sa, ca, sx, cx, sy, cy;
                if (b[10]) {                                         // one lane, the chunk's cells in the reference's order, from the true start
                    b[11] = 1;
                    sa = WB_BOXCOM_ST(0, g, 0); ca = WB_BOXCOM_ST(1, g, 0);
                    sx = WB_BOXCOM_ST(2, g, 0); cx = WB_BOXCOM_ST(3, g, 0);
                    sy = WB_BOXCOM_ST(4, g, 0); cy = WB_BOXCOM_ST(5, g, 0);
                    for (int y = b[1] + b[5]; y < b[1] + b[5] + nr; ++y) {
                        const double w = a.wlat[y];
                        const double yw = (double)y * w;
                        for (int x = b[0]; x <= b[2]; ++x) {
                            const double xw = (double)x * w;
                            WB_KAHAN(sa, ca, w);
                            WB_KAHAN(sx, cx, xw);
                            WB_KAHAN(sy, cy, yw);
                        }
                    }
                } else {
                    sa = WB_BOXCOM_ST(0, g, nr); ca = WB_BOXCOM_ST(1, g, nr);
                    sx = WB_BOXCOM_ST(2, g, nr); cx = WB_BOXCOM_ST(3, g, nr);
                    sy = WB_BOXCOM_ST(4, g, nr); cy = WB_BOXCOM_ST(5, g, nr);
                }
                WB_BOXCOM_ST(0, g, 0) = sa; WB_BOXCOM_ST(1, g, 0) = ca;
                WB_BOXCOM_ST(2, g, 0) = sx; WB_BOXCOM_ST(3, g, 0) = cx;
                WB_BOXCOM_ST(4, g, 0) = sy; WB_BOXCOM_ST(5, g, 0) = cy;
                b[5] += nr;
            }
        WB_END_PFOR
        WB_BSTAMP(a.ctl, 4);
    }
    WB_PFOR(g, WB_BOXCOM_G)
        const int* b = bx + 12 * g;
        if (q0 + (int)g < nq) {
            double* out = a.ot_sums + (size_t)(a.boxq[q0 + (int)g] & 0x3FFFFFFF) * 3;
            double sa, sx, sy;
            if (b[9]) {                                              // exponents out of the fixed-point range: one lane walks every cell
                double ca = 0.0, cx = 0.0, cy = 0.0;
                sa = 0.0; sx = 0.0; sy = 0.0;
                for (int y = b[1]; y <= b[3]; ++y) {
                    const double w = a.wlat[y];
                    const double yw = (double)y * w;
                    for (int x = b[0]; x <= b[2]; ++x) {
                        const double xw = (double)x * w;
                        WB_KAHAN(sa, ca, w);
                        WB_KAHAN(sx, cx, xw);
                        WB_KAHAN(sy, cy, yw);
                    }
                }
            } else {
                sa = WB_BOXCOM_ST(0, g, 0);
                sx = WB_BOXCOM_ST(2, g, 0);
                sy = WB_BOXCOM_ST(4, g, 0);
            }
            if (b[9] || b[11]) wb_atomic_add(&a.ctl->com_rechecks[2], 1);
#if WB_BOXCOM_BOUND == 2
            // test build: bit 30 of the queue entry marks a box wb_k_overturn settled from the exact totals (its sums stand in out[])
            if ((a.boxq[q0 + (int)g] & 0x40000000) && !(out[0] == sa && out[1] == sx && out[2] == sy)) wb_atomic_add((unsigned long long*)&a.ctl->dbg[23], 1ull);
#endif
            out[0] = sa;
            out[1] = sx;
            out[2] = sy;
        }
    WB_END_PFOR
    WB_BSTAMP(a.ctl, 4);
    }   // the queue
    WB_BSTAMP_FLUSH(a.ctl);
}
