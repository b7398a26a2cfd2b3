// wb_track.h -- temporal tracking of events: track_events, processing/events.py:113-241.
//
// Events whose dates differ by 0 < dt <= time_range are candidates (events.py:156-181); a candidate pair is LINKED when
//   "by_distance": haversine(com_i, com_j) * 6371 < distance   (events.py:189-200; sklearn's haversine reads the first
//                  coordinate as the latitude and the reference hands it (lon, lat) -- kept as is),
//   "by_overlap":  area(A n B) / (area(A) + area(B) - area(A n B)) > overlap   (events.py:205-214);
// the connected components of the links get the labels (events.py:220-238).
//
// Overlap areas.  Every event polygon the reference makes has its vertices ON the grid (transform_polygons truncates them
// to grid indices, index_utils.py:135), and the ratio above does not change under the affine map index -> degrees, so
// the polygons arrive here as integer lattice rings.  The plane is cut into unit rows k <= y <= k + 1.  A ring edge that
// crosses a row does so from its bottom to its top line (vertices have integer y), edges of a valid polygon do not cross
// inside a row, so a row holds the polygon as a left-to-right list of edge pieces and the inside lies between pieces
// 0-1, 2-3, ... (even-odd).  wb_k_track_build writes that list for every row of every event once.  The overlap of two
// events is the sum over their common rows, over pairs of such trapezoids, of
//     integral over t in [0, 1] of max(0, min(rA, rB)(t) - max(lA, lB)(t)) dt,
// whose integrand is concave and piecewise linear.  WHETHER it is positive somewhere is decided in integers
// (cross-multiplied rationals, 128-bit products), so polygons that only share an edge or a corner overlap by exactly 0;
// HOW MUCH is summed in float64 from non-negative terms only (no cancellation).  Per-row terms are stored and added in
// row order by one thread: the result does not depend on the launch geometry.
#pragma once
#include "wb_spmd.h"

#define WB_TRACK_THREADS 64
#define WB_TRACK_MAXROWS 4096         // rows of one pair's common band that the LDS term buffer holds per sweep

struct WbTrackArgs {
    long long n;                  // events, in date order (stable)
    int method;                   // 0 by_overlap, 1 by_distance
    int date_kind;                // 0: date_ns (int64 nanoseconds), 1: date_val (float64, |difference| is compared)
    const long long* date_ns;     // [n] sorted
    const double* date_val;       // [n] sorted
    const int* orig;              // [n] row of the caller's frame
    double time_range, overlap, distance;
    const double* com;            // [n][2] degrees, (lon, lat) as the reference stores it
    // geometry (by_overlap)
    const long long* edge0;       // [n+1] edges of event e (horizontal edges removed)
    const short* ex0; const short* ey0; const short* ex1; const short* ey1;
    const int* ymin; const int* ymax; const int* xmin; const int* xmax;     // [n] bounding boxes
    const long long* row0;        // [n+1] first row slot of event e (ymax - ymin slots)
    const long long* piece0;      // [n+1] first piece slot of event e
    int* row_off;                 // [rows total + 1 per event is not needed: row_off[row0[e] + k] is relative to piece0[e]]
    int* row_cnt;                 // [rows total]
    unsigned long long* piece;    // [pieces total] packed x0 | y0 << 16 | x1 << 32 | y1 << 48 (as unsigned 16 bit, biased)
    double* row_area;             // [rows total] scratch
    double* area;                 // [n] out: even-odd area of event e
    // union-find and results
    int* parent;                  // [n] indexed by caller's row
    int* flag;                    // [n] scratch: 1 where a row is its component's smallest row
    int* label;                   // [n] out
    unsigned long long* stats;    // [0] pairs in range, [1] pairs whose boxes overlap, [2] pairs linked, [3] events whose polygon crosses itself,
                                  // [4] pairs linked by the reach test alone (buffered tracking), [5] pairs inside the arc band of that test
    // buffered tracking (events.py:203-214 with buffer > 0 and overlap == 0): two events are linked when their buffered polygons
    // share interior points, i.e. when the planar distance of the polygons in degrees is below reach = 2 * buffer
    double reach, sx, sy;         // reach in degrees (0: off), degrees per grid step in x and y
    const long long* fedge0;      // [n+1] ALL ring edges of event e, the horizontal ones included
    const short* fx0; const short* fy0; const short* fx1; const short* fy1;
};

// ----------------------------------------------------------------------------------------------------- union-find
// parent[] always points to a smaller or equal row, a root is the smallest row of its set
static WB_HD inline int wb_uf_find(const int* parent, int x) {
    int p = ((volatile const int*)parent)[x];
    while (p != x) { x = p; p = ((volatile const int*)parent)[x]; }
    return x;
}
static WB_DEV inline void wb_uf_union(int* parent, int a, int b) {
    for (;;) {
        a = wb_uf_find(parent, a);
        b = wb_uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }                 // hook the larger root under the smaller
        if (wb_atomic_cas(&parent[a], a, b) == a) return;
    }
}

// ----------------------------------------------------------------------------------------------------- pieces
struct WbPiece { long long n0, n1, den; };       // x(k) = n0 / den, x(k + 1) = n1 / den, den > 0

static WB_HD inline unsigned long long wb_piece_pack(int x0, int y0, int x1, int y1) {
    return (unsigned long long)(unsigned short)x0 | ((unsigned long long)(unsigned short)y0 << 16) |
           ((unsigned long long)(unsigned short)x1 << 32) | ((unsigned long long)(unsigned short)y1 << 48);
}
static WB_HD inline WbPiece wb_piece_at(unsigned long long p, int k) {
    const long long x0 = (short)(p & 0xffff), y0 = (short)((p >> 16) & 0xffff), x1 = (short)((p >> 32) & 0xffff),
                    y1 = (short)((p >> 48) & 0xffff);
    WbPiece q;
    q.den = y1 - y0;
    q.n0 = x0 * q.den + (x1 - x0) * (k - y0);
    q.n1 = q.n0 + (x1 - x0);
    if (q.den < 0) { q.den = -q.den; q.n0 = -q.n0; q.n1 = -q.n1; }
    return q;
}
// left-to-right order inside a row: by the mid-line abscissa, then by the abscissa on the bottom line
static WB_HD inline bool wb_piece_before(const WbPiece& a, const WbPiece& b) {
    const long long sa = (a.n0 + a.n1) * b.den, sb = (b.n0 + b.n1) * a.den;
    if (sa != sb) return sa < sb;
    return a.n0 * b.den < b.n0 * a.den;
}

struct WbFrac { long long num, den; };           // den > 0
static WB_HD inline bool wb_frac_less(const WbFrac& a, const WbFrac& b) {
    return (__int128)a.num * b.den < (__int128)b.num * a.den;
}
static WB_HD inline double wb_piece_x(const WbPiece& p, double t) {
    return ((double)p.n0 + (double)(p.n1 - p.n0) * t) / (double)p.den;
}
// where (u - v)(t) changes sign inside (0, 1): the kink of max(u, v) / min(u, v); false if there is none
static WB_HD inline bool wb_piece_kink(const WbPiece& u, const WbPiece& v, double* t) {
    const long long p = u.n0 * v.den - v.n0 * u.den, q = u.n1 * v.den - v.n1 * u.den;
    if ((p > 0) == (q > 0) || p == q) return false;
    if ((p == 0) || (q == 0)) return false;                            // touches at an end of the row: no kink inside
    *t = (double)p / (double)(p - q);
    return true;
}

// area of (trapezoid lA..rA) n (trapezoid lB..rB) inside one unit row; exactly 0 unless their interiors meet
static WB_HD inline double wb_span_overlap(const WbPiece& lA, const WbPiece& rA, const WbPiece& lB, const WbPiece& rB) {
    WbFrac lo = {0, 1}, hi = {1, 1};
    const WbPiece* R[2] = {&rA, &rB};
    const WbPiece* L[2] = {&lA, &lB};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            const WbPiece& r = *R[i];
            const WbPiece& l = *L[j];
            const long long p = r.n0 * l.den - l.n0 * r.den, q = r.n1 * l.den - l.n1 * r.den;   // (r - l) at t = 0, 1, times r.den * l.den
            if (p > 0 && q > 0) continue;
            if (p <= 0 && q <= 0) return 0.0;
            if (p > 0) { const WbFrac f = {p, p - q}; if (wb_frac_less(f, hi)) hi = f; }         // positive before its zero
            else { const WbFrac f = {-p, q - p}; if (wb_frac_less(lo, f)) lo = f; }              // positive after its zero
        }
    if (!wb_frac_less(lo, hi)) return 0.0;
    // magnitude: the integrand is linear between lo, the kinks of max(l) and min(r), and hi
    double t[4];
    int nt = 0;
    t[nt++] = (double)lo.num / (double)lo.den;
    const double thi = (double)hi.num / (double)hi.den;
    double tk;
    if (wb_piece_kink(lA, lB, &tk) && tk > t[0] && tk < thi) t[nt++] = tk;
    if (wb_piece_kink(rA, rB, &tk) && tk > t[0] && tk < thi) {
        if (nt == 2 && tk < t[1]) { t[2] = t[1]; t[1] = tk; ++nt; }
        else t[nt++] = tk;
    }
    t[nt++] = thi;
    double area = 0.0, gprev = 0.0;
    for (int i = 0; i < nt; ++i) {
        const double xa = wb_piece_x(rA, t[i]), xb = wb_piece_x(rB, t[i]), ya = wb_piece_x(lA, t[i]), yb = wb_piece_x(lB, t[i]);
        double g = (xa < xb ? xa : xb) - (ya > yb ? ya : yb);
        g = g > 0.0 ? g : 0.0;
        if (i > 0) area += (t[i] - t[i - 1]) * (g + gprev) * 0.5;
        gprev = g;
    }
    return area;
}

// ----------------------------------------------------------------------------------------------------- row tables
// one workgroup per event: pieces per row counted, offsets by a block scan, pieces placed, every row sorted; the
// even-odd area of the event falls out of the sorted rows
WB_KERNEL(wb_k_track_build, WbTrackArgs a) {
    WB_SHARED(int, xflag, 1);
    WB_SINGLE
        xflag[0] = 0;
    WB_END_SINGLE
    for (long long e = WB_BLOCK_ID; e < a.n; e += WB_GRID_DIM) {
        const long long e0 = a.edge0[e], ne = a.edge0[e + 1] - e0;
        const int y0 = a.ymin[e], rows = a.ymax[e] - y0;
        int* cnt = a.row_cnt + a.row0[e];
        int* off = a.row_off + a.row0[e];
        unsigned long long* pc = a.piece + a.piece0[e];
        double* rarea = a.row_area + a.row0[e];
        WB_PFOR(k, rows)
            off[k] = 0;
        WB_END_PFOR
        WB_PFOR(q, ne)
            const int ya = a.ey0[e0 + q], yb = a.ey1[e0 + q];
            for (int k = (ya < yb ? ya : yb); k < (ya < yb ? yb : ya); ++k) wb_atomic_add(&off[k - y0], 1);
        WB_END_PFOR
        WB_PFOR(k, rows)
            cnt[k] = off[k];
        WB_END_PFOR
        wb_block_scan_excl(off, rows);
        WB_SYNC();
        WB_PFOR(k, rows)
            cnt[k] += off[k];                    // cursor counts DOWN from the end of the row's slot range
        WB_END_PFOR
        WB_PFOR(q, ne)
            const int ya = a.ey0[e0 + q], yb = a.ey1[e0 + q];
            const unsigned long long packed = wb_piece_pack(a.ex0[e0 + q], ya, a.ex1[e0 + q], yb);
            for (int k = (ya < yb ? ya : yb); k < (ya < yb ? yb : ya); ++k) pc[wb_atomic_add(&cnt[k - y0], -1) - 1] = packed;
        WB_END_PFOR
        // cnt[k] is back at off[k]; row k holds pieces off[k] .. off[k + 1] (the last row ends at the event's piece count)
        const int total = (int)(a.piece0[e + 1] - a.piece0[e]);
        WB_PFOR(k, rows)
            const int b = off[k], c = (k + 1 < rows ? off[k + 1] : total);
            for (int i = b + 1; i < c; ++i) {                            // insertion sort: a row holds a handful of pieces
                const unsigned long long pi = pc[i];
                const WbPiece qi = wb_piece_at(pi, y0 + (int)k);
                int j = i - 1;
                while (j >= b && wb_piece_before(qi, wb_piece_at(pc[j], y0 + (int)k))) { pc[j + 1] = pc[j]; --j; }
                pc[j + 1] = pi;
            }
            double s = 0.0;
            for (int i = b; i + 1 < c; i += 2) {
                const WbPiece l = wb_piece_at(pc[i], y0 + (int)k), r = wb_piece_at(pc[i + 1], y0 + (int)k);
                s += 0.5 * (((double)r.n0 + (double)r.n1) / (double)r.den - ((double)l.n0 + (double)l.n1) / (double)l.den);
            }
            rarea[k] = s;
            // two pieces that swap sides between the row's bottom and top line cross inside it: the polygon is not simple (the
            // reference repairs such geometries with buffer(0).make_valid(), events.py:203-204 -- GEOS, not restated: the event
            // is counted and the caller warned; its region here is the even-odd one)
            bool cross = false;
            for (int i = b; i < c && !cross; ++i) {
                const WbPiece qi = wb_piece_at(pc[i], y0 + (int)k);
                for (int j = i + 1; j < c && j <= i + 8; ++j) {
                    const WbPiece qj = wb_piece_at(pc[j], y0 + (int)k);
                    const long long d0 = qi.n0 * qj.den - qj.n0 * qi.den, d1 = qi.n1 * qj.den - qj.n1 * qi.den;
                    if ((d0 < 0 && d1 > 0) || (d0 > 0 && d1 < 0)) { cross = true; break; }
                }
            }
            if (cross) xflag[0] = 1;
        WB_END_PFOR
        WB_SINGLE
            double s = 0.0;
            for (int k = 0; k < rows; ++k) s += rarea[k];
            a.area[e] = s;
            if (xflag[0]) { if (a.stats) wb_atomic_add(&a.stats[3], 1ull); xflag[0] = 0; }
        WB_END_SINGLE
    }
}

// ----------------------------------------------------------------------------------------------------- pairs
static WB_HD inline double wb_track_hours(const WbTrackArgs& a, long long i, long long j) {
    if (a.date_kind == 0) return (double)(a.date_ns[j] - a.date_ns[i]) / 1e9 / 3600.0;      // events.py:163 (pandas total_seconds)
    const double d = a.date_val[j] - a.date_val[i];
    return d < 0.0 ? -d : d;                                                                  // events.py:165
}

// sklearn's haversine on (com[0], com[1]) in radians, first coordinate in the latitude's role (events.py:189-195)
static WB_HD inline double wb_track_haversine(double a0d, double a1d, double b0d, double b1d) {
    const double rad = 3.141592653589793 / 180.0;                     // numpy.radians: x * (pi / 180)
    const double a0 = a0d * rad, a1 = a1d * rad, b0 = b0d * rad, b1 = b1d * rad;
    const double s0 = sin(0.5 * (a0 - b0)), s1 = sin(0.5 * (a1 - b1));
    return 2.0 * asin(sqrt(s0 * s0 + cos(a0) * cos(b0) * s1 * s1));
}

// area of (event i) n (event j), all threads of the workgroup call it; term = WB_TRACK_MAXROWS doubles of LDS
static WB_DEV inline double wb_track_pair_overlap(const WbTrackArgs& a, long long i, long long j, double* term, bool* boxes) {
    const int ay0 = a.ymin[i], ay1 = a.ymax[i], by0 = a.ymin[j], by1 = a.ymax[j];
    const int y0 = ay0 > by0 ? ay0 : by0, y1 = ay1 < by1 ? ay1 : by1;
    const int x0 = a.xmin[i] > a.xmin[j] ? a.xmin[i] : a.xmin[j], x1 = a.xmax[i] < a.xmax[j] ? a.xmax[i] : a.xmax[j];
    *boxes = y0 < y1 && x0 < x1;                                       // the boxes share interior points (block-uniform)
    if (!*boxes) return 0.0;
    const int* aoff = a.row_off + a.row0[i];
    const int* boff = a.row_off + a.row0[j];
    const unsigned long long* apc = a.piece + a.piece0[i];
    const unsigned long long* bpc = a.piece + a.piece0[j];
    const int atot = (int)(a.piece0[i + 1] - a.piece0[i]), arows = ay1 - ay0;
    const int btot = (int)(a.piece0[j + 1] - a.piece0[j]), brows = by1 - by0;
    double inter = 0.0;
    for (int base = y0; base < y1; base += WB_TRACK_MAXROWS) {
        const int nrow = y1 - base < WB_TRACK_MAXROWS ? y1 - base : WB_TRACK_MAXROWS;
        WB_PFOR(r, nrow)
            const int k = base + (int)r, ka = k - ay0, kb = k - by0;
            const int pa0 = aoff[ka], pa1 = ka + 1 < arows ? aoff[ka + 1] : atot;
            const int pb0 = boff[kb], pb1 = kb + 1 < brows ? boff[kb + 1] : btot;
            double s = 0.0;
            for (int u = pa0; u + 1 < pa1; u += 2) {
                const WbPiece lA = wb_piece_at(apc[u], k), rA = wb_piece_at(apc[u + 1], k);
                for (int v = pb0; v + 1 < pb1; v += 2)
                    s += wb_span_overlap(lA, rA, wb_piece_at(bpc[v], k), wb_piece_at(bpc[v + 1], k));
            }
            term[r] = s;
        WB_END_PFOR
        WB_SINGLE
            double s = 0.0;
            for (int r = 0; r < nrow; ++r) s += term[r];                // row order: the same sum whatever the launch geometry
            term[0] = s;
        WB_END_SINGLE
        inter += term[0];
        WB_SYNC();
    }
    return inter;
}

// the overlap areas of an explicit list of pairs (positions in date order): what the tests compare with the exact oracle
WB_KERNEL(wb_k_track_overlap_list, WbTrackArgs a, long long npairs, const long long* pi, const long long* pj, double* out) {
    WB_SHARED(double, term, WB_TRACK_MAXROWS);
    for (long long q = WB_BLOCK_ID; q < npairs; q += WB_GRID_DIM) {
        bool boxes = false;
        const double inter = wb_track_pair_overlap(a, pi[q], pj[q], term, &boxes);
        WB_SINGLE
            out[q] = inter;
        WB_END_SINGLE
    }
}

// ----------------------------------------------------------------------------------------------------- buffered tracking
// events.py:203-214 with buffer > 0: geom.buffer(buffer) is GEOS' round buffer (geopandas: 16 segments per quadrant), and
// with overlap == 0 the test `inter.area / union.area > 0` asks whether the two buffered polygons share interior points --
// for discs of radius `buffer`: whether the planar distance of the two (closed) polygons, in degrees, is BELOW 2 * buffer.
// GEOS' arcs are inscribed polygons: at a corner they fall short of the disc by at most buffer * (1 - cos(pi / 64)), so a
// pair whose distance lies within 2 * that of the threshold could go either way there; such pairs are counted (stats[5])
// and decided by the disc.  GEOS itself is not in the image: **unpinned**, like the other GEOS pieces.
#define WB_TRACK_ARC_SLACK 0.0024091                         // 2 * (1 - cos(pi / 64))
static WB_HD inline long long wb_tr_orient(long long ax, long long ay, long long bx, long long by, long long cx, long long cy) {
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
}
static WB_HD inline bool wb_tr_on(long long ax, long long ay, long long bx, long long by, long long px, long long py) {   // p on [a, b], given collinear
    return (px >= (ax < bx ? ax : bx)) && (px <= (ax < bx ? bx : ax)) && (py >= (ay < by ? ay : by)) && (py <= (ay < by ? by : ay));
}
// do the closed segments [a, b] and [c, d] share a point? (exact: lattice coordinates)
static WB_HD inline bool wb_tr_meet(int ax, int ay, int bx, int by, int cx, int cy, int dx, int dy) {
    const long long o1 = wb_tr_orient(ax, ay, bx, by, cx, cy), o2 = wb_tr_orient(ax, ay, bx, by, dx, dy);
    const long long o3 = wb_tr_orient(cx, cy, dx, dy, ax, ay), o4 = wb_tr_orient(cx, cy, dx, dy, bx, by);
    if (((o1 > 0 && o2 < 0) || (o1 < 0 && o2 > 0)) && ((o3 > 0 && o4 < 0) || (o3 < 0 && o4 > 0))) return true;
    if (o1 == 0 && wb_tr_on(ax, ay, bx, by, cx, cy)) return true;
    if (o2 == 0 && wb_tr_on(ax, ay, bx, by, dx, dy)) return true;
    if (o3 == 0 && wb_tr_on(cx, cy, dx, dy, ax, ay)) return true;
    if (o4 == 0 && wb_tr_on(cx, cy, dx, dy, bx, by)) return true;
    return false;
}
// squared distance of the point p to the segment [a, b], coordinates in degrees
static WB_HD inline double wb_tr_point_seg_d2(double px, double py, double ax, double ay, double bx, double by) {
    const double ux = bx - ax, uy = by - ay, wx = px - ax, wy = py - ay;
    const double L = ux * ux + uy * uy;
    double t = L > 0.0 ? (wx * ux + wy * uy) / L : 0.0;
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    const double ex = wx - t * ux, ey = wy - t * uy;
    return ex * ex + ey * ey;
}
// squared distance of two lattice segments in degrees (sx, sy degrees per grid step); exactly 0 when they share a point
static WB_HD inline double wb_tr_seg_seg_d2(int ax, int ay, int bx, int by, int cx, int cy, int dx, int dy, double sx, double sy) {
    if (wb_tr_meet(ax, ay, bx, by, cx, cy, dx, dy)) return 0.0;
    const double Ax = ax * sx, Ay = ay * sy, Bx = bx * sx, By = by * sy, Cx = cx * sx, Cy = cy * sy, Dx = dx * sx, Dy = dy * sy;
    double d = wb_tr_point_seg_d2(Ax, Ay, Cx, Cy, Dx, Dy);
    const double d2 = wb_tr_point_seg_d2(Bx, By, Cx, Cy, Dx, Dy), d3 = wb_tr_point_seg_d2(Cx, Cy, Ax, Ay, Bx, By), d4 = wb_tr_point_seg_d2(Dx, Dy, Ax, Ay, Bx, By);
    d = d2 < d ? d2 : d; d = d3 < d ? d3 : d; d = d4 < d ? d4 : d;
    return d;
}
// smallest squared distance between the rings of events i and j, in degrees^2 (all threads of the workgroup call it; best: one
// LDS word holding the bit pattern of a non-negative double, which orders like the number)
static WB_DEV inline double wb_track_pair_distance2(const WbTrackArgs& a, long long i, long long j, unsigned long long* best) {
    const long long a0 = a.fedge0[i], na = a.fedge0[i + 1] - a0, b0 = a.fedge0[j], nb = a.fedge0[j + 1] - b0;
    WB_SINGLE
        const double inf = 1e300;
        memcpy(best, &inf, 8);
    WB_END_SINGLE
    // (the other event's box, widened by the reach: an edge that lies outside it cannot be within reach of anything)
    const double rx = a.reach / a.sx, ry = a.reach / a.sy;
    const double jx0 = a.xmin[j] - rx, jx1 = a.xmax[j] + rx, jy0 = a.ymin[j] - ry, jy1 = a.ymax[j] + ry;
    WB_PFOR(q, na * nb)
        const long long ea = a0 + q / nb, eb = b0 + q % nb;
        const int ax = a.fx0[ea], ay = a.fy0[ea], bx = a.fx1[ea], by = a.fy1[ea];
        if (!((ax < jx0 && bx < jx0) || (ax > jx1 && bx > jx1) || (ay < jy0 && by < jy0) || (ay > jy1 && by > jy1))) {
            const double d2 = wb_tr_seg_seg_d2(ax, ay, bx, by, a.fx0[eb], a.fy0[eb], a.fx1[eb], a.fy1[eb], a.sx, a.sy);
            unsigned long long bits;
            memcpy(&bits, &d2, 8);
            if (bits < *((volatile unsigned long long*)best)) wb_atomic_min(best, bits);
        }
    WB_END_PFOR
    double out;
    memcpy(&out, best, 8);
    return out;
}

// one workgroup per event i: its partners are the later events j (date order) with 0 < dt <= time_range
WB_KERNEL(wb_k_track_pairs, WbTrackArgs a) {
    WB_SHARED(long long, jr, 2);
    WB_SHARED(double, term, WB_TRACK_MAXROWS);
    WB_SHARED(unsigned long long, cnts, 3);
    WB_SHARED(unsigned long long, dbest, 1);
    for (long long i = WB_BLOCK_ID; i < a.n; i += WB_GRID_DIM) {
        WB_SINGLE
            long long lo = i + 1, hi = a.n;                            // first j with dt > 0
            while (lo < hi) { const long long m = (lo + hi) >> 1; if (wb_track_hours(a, i, m) > 0.0) hi = m; else lo = m + 1; }
            jr[0] = lo;
            hi = a.n;                                                  // first j with dt > time_range
            while (lo < hi) { const long long m = (lo + hi) >> 1; if (wb_track_hours(a, i, m) <= a.time_range) lo = m + 1; else hi = m; }
            jr[1] = lo;
            cnts[0] = (unsigned long long)(jr[1] - jr[0]); cnts[1] = 0; cnts[2] = 0;
        WB_END_SINGLE
        const long long j0 = jr[0], j1 = jr[1];
        if (a.method == 1) {
            WB_PFOR(q, j1 - j0)
                const long long j = j0 + q;
                const double d = wb_track_haversine(a.com[2 * i], a.com[2 * i + 1], a.com[2 * j], a.com[2 * j + 1]);
                if (d * 6371.0 < a.distance) {                         // events.py:198
                    wb_uf_union(a.parent, a.orig[i], a.orig[j]);
                    wb_atomic_add(&cnts[2], 1ull);
                }
            WB_END_PFOR
        } else if (a.reach > 0.0) {
            for (long long j = j0; j < j1; ++j) {
                // boxes further apart than the reach: nothing to do (block-uniform)
                const int gx = (a.xmin[i] > a.xmin[j] ? a.xmin[i] : a.xmin[j]) - (a.xmax[i] < a.xmax[j] ? a.xmax[i] : a.xmax[j]);
                const int gy = (a.ymin[i] > a.ymin[j] ? a.ymin[i] : a.ymin[j]) - (a.ymax[i] < a.ymax[j] ? a.ymax[i] : a.ymax[j]);
                const double bx = gx > 0 ? gx * a.sx : 0.0, by = gy > 0 ? gy * a.sy : 0.0;
                if (bx * bx + by * by >= a.reach * a.reach) continue;
                bool boxes = false;
                const double inter = wb_track_pair_overlap(a, i, j, term, &boxes);        // one inside the other: no edge need come near
                const double d2 = inter > 0.0 ? 0.0 : wb_track_pair_distance2(a, i, j, dbest);
                WB_SINGLE
                    cnts[1] += 1;
                    const double r2 = a.reach * a.reach, lo = a.reach * (1.0 - WB_TRACK_ARC_SLACK);
                    if (d2 < r2) {                                     // the buffered polygons share interior points
                        wb_uf_union(a.parent, a.orig[i], a.orig[j]);
                        cnts[2] += 1;
                        if (!(inter > 0.0)) wb_atomic_add(&a.stats[4], 1ull);
                        if (d2 >= lo * lo) wb_atomic_add(&a.stats[5], 1ull);
                    }
                WB_END_SINGLE
            }
        } else {
            for (long long j = j0; j < j1; ++j) {
                bool boxes = false;
                const double inter = wb_track_pair_overlap(a, i, j, term, &boxes);
                WB_SINGLE
                    if (boxes) cnts[1] += 1;
                    // events.py:210-214: inter / (area_j + area_i - inter) > overlap, decided without the division and with a
                    // guard of 1e-13 relative: a ratio that EQUALS the threshold in exact arithmetic (lattice polygons give
                    // 1/2, 1/5, ... on the dot) is not above it, whatever the rounding of the trapezoid sum says
                    const double uni = a.area[j] + a.area[i] - inter, rhs = a.overlap * uni;
                    if (uni > 0.0 && inter > rhs && inter - rhs > 1e-13 * (fabs(inter) + fabs(rhs))) {
                        wb_uf_union(a.parent, a.orig[i], a.orig[j]);
                        cnts[2] += 1;
                    }
                WB_END_SINGLE
            }
        }
        WB_SINGLE
            wb_atomic_add(&a.stats[0], cnts[0]);
            if (cnts[1]) wb_atomic_add(&a.stats[1], cnts[1]);
            if (cnts[2]) wb_atomic_add(&a.stats[2], cnts[2]);
        WB_END_SINGLE
    }
}

// ----------------------------------------------------------------------------------------------------- labels
WB_KERNEL(wb_k_track_init, WbTrackArgs a) {
    const long long per = (a.n + WB_GRID_DIM - 1) / WB_GRID_DIM, b0 = per * WB_BLOCK_ID;
    WB_PFOR(q, (b0 + per <= a.n ? per : (a.n > b0 ? a.n - b0 : 0)))
        a.parent[b0 + q] = (int)(b0 + q);
    WB_END_PFOR
}
// events.py:224-238: a component is named after its smallest row, the names are then renumbered 0, 1, 2, ... ascending
WB_KERNEL(wb_k_track_roots, WbTrackArgs a) {
    const long long per = (a.n + WB_GRID_DIM - 1) / WB_GRID_DIM, b0 = per * WB_BLOCK_ID;
    WB_PFOR(q, (b0 + per <= a.n ? per : (a.n > b0 ? a.n - b0 : 0)))
        const int x = (int)(b0 + q), r = wb_uf_find(a.parent, x);
        a.label[x] = r;
        a.flag[x] = r == x ? 1 : 0;
    WB_END_PFOR
}
WB_KERNEL(wb_k_track_rank, WbTrackArgs a) {                     // one workgroup
    wb_block_scan_excl(a.flag, (int)a.n);
}
WB_KERNEL(wb_k_track_labels, WbTrackArgs a) {
    const long long per = (a.n + WB_GRID_DIM - 1) / WB_GRID_DIM, b0 = per * WB_BLOCK_ID;
    WB_PFOR(q, (b0 + per <= a.n ? per : (a.n > b0 ? a.n - b0 : 0)))
        a.label[b0 + q] = a.flag[a.label[b0 + q]];
    WB_END_PFOR
}
