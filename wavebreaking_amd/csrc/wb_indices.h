// wb_indices.h -- phase 2 kernels: contour selection, overturning index, streamer index,
// cutoff selection.  Host-simulable SPMD bodies (wb_spmd.h).
//
// Reference restated (file:line under the reference tree):
//   selection      streamer_index.py:104-108, overturning_index.py:103-107, cutoff_index.py:87-93
//   overturning    overturning_index.py:120-218
//   streamer       streamer_index.py:119-272 (N x N test :130-139, |dx|>120 drop :157, four filters :160-264)
#pragma once
#include "wb_core.h"
#include "wb_trace.h"

// ------------------------------------------------------------------------------------------------
// selection
struct WbSelectArgs {
    int njobs, cap_seg, gmax;      // gmax: global max of the distinct-x count (exp_lon.max() / dlon)
    int want_cut;                  // cutoffs requested
    double dlon, co_min_exp;
    const int* ncont;              // [njobs]
    const int* c_nux;              // [njobs][cap_seg]
    const int* c_closed;
    int* nsel;                     // [njobs] number of contours with nux == gmax
    int* ncut;                     // [njobs] number of cutoff contours
    // second pass
    const int* selbase;            // [njobs] exclusive scans of the above
    const int* cutbase;
    int* sel_job;                  // [total sel]
    int* sel_cont;                 // [total sel] contour index inside the job
    int* cut_job;
    int* cut_cont;
    // the phase-2 part of the batch's control block, zeroed by the first workgroup of wb_k_select_count (zero_words 32-bit words)
    uint32_t* zero_at;
    int zero_words;
};

WB_HD inline bool wb_is_cutoff(int nux, int closed, int gmax, double dlon, double min_exp) {
    // cutoff_index.py:89-93: exp_lon < max & exp_lon >= min_exp & closed, exp_lon = len(set(x)) * dlon
    return nux < gmax && ((double)nux * dlon >= min_exp) && closed;
}

WB_KERNEL(wb_k_select_count, WbSelectArgs a) {
    const int job = WB_BLOCK_ID;
    if (job >= a.njobs) return;
    const size_t jb = (size_t)job * a.cap_seg;
    if (job == 0) {          // (nothing in this kernel reads those counters; the next kernel is the first to write one)
        WB_PFOR(w, a.zero_words)
            a.zero_at[w] = 0u;
        WB_END_PFOR
    }
    WB_SINGLE
    int s = 0, c = 0;
    for (int k = 0; k < a.ncont[job]; ++k) {
        if (a.c_nux[jb + k] == a.gmax) ++s;
        if (a.want_cut && wb_is_cutoff(a.c_nux[jb + k], a.c_closed[jb + k], a.gmax, a.dlon, a.co_min_exp)) ++c;
    }
    a.nsel[job] = s;
    a.ncut[job] = c;
    WB_END_SINGLE
}

WB_KERNEL(wb_k_scan_jobs, int n, int* a0, int* a1, int* totals) {
    // single workgroup: in-place exclusive scans over the per-job counts
    const int t0 = wb_block_scan_excl(a0, n);
    const int t1 = wb_block_scan_excl(a1, n);
    WB_SINGLE
    totals[0] = t0;
    totals[1] = t1;
    WB_END_SINGLE
}

WB_KERNEL(wb_k_select_fill, WbSelectArgs a) {
    const int job = WB_BLOCK_ID;
    if (job >= a.njobs) return;
    const size_t jb = (size_t)job * a.cap_seg;
    WB_SINGLE
    int s = a.selbase[job], c = a.cutbase[job];
    for (int k = 0; k < a.ncont[job]; ++k) {
        if (a.c_nux[jb + k] == a.gmax) { a.sel_job[s] = job; a.sel_cont[s] = k; ++s; }
        if (a.want_cut && wb_is_cutoff(a.c_nux[jb + k], a.c_closed[jb + k], a.gmax, a.dlon, a.co_min_exp)) {
            a.cut_job[c] = job; a.cut_cont[c] = k; ++c;
        }
    }
    WB_END_SINGLE
}

// ------------------------------------------------------------------------------------------------
// overturning index, one workgroup per selected contour
#define WB_OT_CAP 256        // groups per contour (>= W / (range_group/dlon + 1))
#define WB_MAXW 4100

struct WbOverturnArgs {
    int cap_seg, nlon, W;
    WbCtl* ctl;                // ctl->nsel selected contours; n_boxq / com_rechecks[1]: boxes queued for wb_k_box_com
    double rg_over_dlon;       // range_group / dlon
    double minexp_over_dlon;   // min_exp / dlon
    const int* sel_job;
    const int* sel_cont;
    const int* c_off;
    const int* c_len;
    const uint32_t* cpts;      // [njobs][2*cap_seg]
    const double* lat;         // [nlat] degrees
    // out, per selected contour
    int* ot_n;                 // [nsel]
    int* ot_box;               // [nsel][WB_OT_CAP][4] min_lon, min_lat, max_lon, max_lat (index units)
    int* ot_orient;            // [nsel][WB_OT_CAP]    0 cyclonic, 1 anticyclonic
    uint32_t* status;          // [njobs]
    // centre-of-mass sums of the boxes (index_utils.py:91-110; every cell of the closed box counts): closed forms here, and
    // the boxes whose truncated quotient could turn on the last bits of the reference's cell-by-cell sums go on a queue
    int nlat;
    const double* wlat;        // [nlat] area weight of a latitude row (null: no sums, the property kernel rasterises the boxes)
    double* ot_sums;           // [nsel][WB_OT_CAP][3] areas, x_com, y_com
    int* boxq;                 // [nsel * WB_OT_CAP] slot (contour * WB_OT_CAP + box) of a queued box
};

// set(range(a0, b0+1) % nlon) subset of set(range(a1, b1+1) % nlon)   (overturning_index.py:141-151)
WB_HD inline bool wb_arc_subset(int a0, int b0, int a1, int b1, int nlon) {
    const int l0 = b0 - a0 + 1, l1 = b1 - a1 + 1;
    if (l1 >= nlon) return true;
    if (l0 > l1) return false;  // also covers l0 >= nlon > l1
    int off = (a0 - a1) % nlon;
    if (off < 0) off += nlon;
    return off + l0 <= l1;
}

WB_KERNEL(wb_k_overturn, WbOverturnArgs a) {
    const int s = WB_BLOCK_ID;
    if (s >= a.ctl->nsel) return;
    const int job = a.sel_job[s], k = a.sel_cont[s];
    const size_t jb = (size_t)job * a.cap_seg;
    const uint32_t* pts = a.cpts + 2 * jb + a.c_off[jb + k];
    const int N = a.c_len[jb + k];
    WB_DYN_SHARED_BEGIN
    WB_DYN_SHARED(int, hist, a.W);               // visits per longitude, then 0 / 1 visited three times or more / 2 ... and first of its group
    WB_DYN_SHARED(int, prevm, a.W);              // the last such longitude west of x, then the groups that start west of x
    WB_SHARED(int, gmin, WB_OT_CAP);
    WB_SHARED(int, gmax, WB_OT_CAP);
    WB_SHARED(int, gdrop, WB_OT_CAP);
    WB_SHARED(int, ylo, WB_OT_CAP);
    WB_SHARED(int, yhi, WB_OT_CAP);
    WB_SHARED(int, ifirst, WB_OT_CAP);
    WB_SHARED(int, ilast, WB_OT_CAP);
    WB_SHARED(int, gcount, 2);
    WB_SHARED(int, need, WB_OT_CAP);             // boxes whose truncated centre of mass turns on the last bits of the reference's sums
    WB_SHARED(int, elim, 2);                     // exponent range of a box's row weights
    WB_SHARED(unsigned long long, bacc, 16);     // a box's exact totals: 3 x 4 limbs of 32 bits (added up as 64-bit words), 3 bounds, the edge distance
    int* box = a.ot_box + (size_t)s * WB_OT_CAP * 4;
    int* orient = a.ot_orient + (size_t)s * WB_OT_CAP;

    WB_PFOR(x, a.W)
        hist[x] = 0;
    WB_END_PFOR
    WB_PFOR(i, N)
        wb_atomic_add(&hist[wb_pt_x(pts[i])], 1);
    WB_END_PFOR
    // (1)-(3) longitudes visited >= 3 times, labelled by gaps > range_group/dlon, min/max per label: two block scans over the
    // longitudes (the last marked longitude to the west; the groups that start to the west) -- one lane walking the W
    // longitudes, a dependent LDS read each, was 80 us of a workgroup's 150
    WB_PFOR(x, a.W)
        prevm[x] = hist[x] >= 3 ? (int)x : -1;
    WB_END_PFOR
    wb_block_scan_excl_max(prevm, a.W, -1);
    WB_PFOR(x, a.W)
        const int pm = prevm[x];
        const int code = hist[x] >= 3 ? ((pm < 0 || (double)((int)x - pm) > a.rg_over_dlon) ? 2 : 1) : 0;
        hist[x] = code;
        prevm[x] = code == 2 ? 1 : 0;
    WB_END_PFOR
    const int Gall = wb_block_scan_excl(prevm, a.W);
    WB_PFOR(g, Gall < WB_OT_CAP ? Gall : WB_OT_CAP)
        gmax[g] = -1;
    WB_END_PFOR
    WB_PFOR(x, a.W)
        const int code = hist[x];
        if (code) {
            const int g = prevm[x] + (code == 2 ? 1 : 0) - 1;
            if (g < WB_OT_CAP) {
                if (code == 2) gmin[g] = (int)x;
                wb_atomic_max(&gmax[g], (int)x);
            }
        }
    WB_END_PFOR
    WB_SINGLE
    gcount[0] = Gall < WB_OT_CAP ? Gall : WB_OT_CAP;
    if (Gall > WB_OT_CAP) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_OVERFLOW_PAIRS);
    WB_END_SINGLE
    int G = gcount[0];
    // check_duplicates (:136-162)
    WB_PFOR(g, G)
        gdrop[g] = 0;
    WB_END_PFOR
    WB_PFOR(pair, (int64_t)G * G)
        const int g0 = (int)(pair / G), g1 = (int)(pair - (int64_t)g0 * G);
        if (g0 != g1 && wb_arc_subset(gmin[g0], gmax[g0], gmin[g1], gmax[g1], a.nlon)) {
            const int l0 = gmax[g0] - gmin[g0] + 1, l1 = gmax[g1] - gmin[g1] + 1;
            const int d = (l0 == l1) ? (g0 > g1 ? g0 : g1) : (l0 < l1 ? g0 : g1);
            gdrop[d] = 1;
        }
    WB_END_PFOR
    // compaction; check_expansion (:164-166) only runs while rows remain (:185-188) -- same result
    WB_SINGLE
    int m = 0;
    for (int g = 0; g < G; ++g) {
        if (gdrop[g]) continue;
        if (!((double)(gmax[g] - gmin[g]) >= a.minexp_over_dlon)) continue;
        gmin[m] = gmin[g];
        gmax[m] = gmax[g];
        ++m;
    }
    gcount[1] = m;
    WB_END_SINGLE
    G = gcount[1];
    // find_lat_expansion (:168-180) and the two orientation probes (:192-197)
    WB_PFOR(g, G)
        ylo[g] = 0x7FFFFFFF;
        yhi[g] = -1;
        ifirst[g] = 0x7FFFFFFF;
        ilast[g] = -1;
    WB_END_PFOR
    WB_PFOR(i, N)
        const int x = wb_pt_x(pts[i]), y = wb_pt_y(pts[i]);
        for (int g = 0; g < G; ++g) {
            if (x >= gmin[g] && x <= gmax[g]) {
                wb_atomic_min(&ylo[g], y);
                wb_atomic_max(&yhi[g], y);
                if (x == gmin[g]) wb_atomic_min(&ifirst[g], (int)i);
                if (x == gmax[g]) wb_atomic_max(&ilast[g], (int)i);
            }
        }
    WB_END_PFOR
    WB_PFOR(g, G)
        const double lat_west = a.lat[wb_pt_y(pts[ifirst[g]])];
        const double lat_east = a.lat[wb_pt_y(pts[ilast[g]])];
        orient[g] = (fabs(lat_west) <= fabs(lat_east)) ? 0 : 1;
        box[4 * g + 0] = gmin[g];
        box[4 * g + 1] = ylo[g];
        box[4 * g + 2] = gmax[g];
        box[4 * g + 3] = yhi[g];
        if (a.wlat) {
            // areas / x_com / y_com of the box row by row in closed form.  A box with min_lon + max_lon even has
            // x_com / areas = an integer mathematically: when the quotient is that close to one, int() (index_utils.py:105-110)
            // turns on the last bits of the reference's compensated cell-by-cell sums, and the box is queued for wb_k_box_com
            int x0 = gmin[g], y0 = ylo[g], x1 = gmax[g], y1 = yhi[g];
            if (x0 < 0) x0 = 0;
            if (y0 < 0) y0 = 0;
            if (x1 > a.W - 1) x1 = a.W - 1;
            if (y1 > a.nlat - 1) y1 = a.nlat - 1;
            const int cols = x1 - x0 + 1, rows = y1 - y0 + 1;
            const double cnt = (double)cols;
            const double sxr = (double)((long long)cols * x0 + (long long)cols * (cols - 1) / 2);
            double areas = 0.0, xc = 0.0, yc = 0.0;
            for (int y = y0; y <= y1; ++y) {
                const double w = a.wlat[y];
                areas = areas + w * cnt;
                xc = xc + w * sxr;
                yc = yc + w * ((double)y * cnt);
            }
            const double ulps = wb_com_ulps(rows);
            double* out = a.ot_sums + ((size_t)s * WB_OT_CAP + g) * 3;
            need[g] = 0;
            if (cols > 0 && rows > 0 && (wb_trunc_unsafe(xc, areas, ulps) || wb_trunc_unsafe(yc, areas, ulps))) {
#if WB_BOXCOM_BOUND >= 1
                need[g] = 1;                                    // settled below from the exact totals of its terms, or queued
#else
                a.boxq[wb_atomic_add(&a.ctl->n_boxq, 1)] = s * WB_OT_CAP + (int)g;
#endif
                wb_atomic_add(&a.ctl->com_rechecks[1], 1);
            } else {
                out[0] = areas;
                out[1] = xc;
                out[2] = yc;
            }
        }
    WB_END_PFOR
#if WB_BOXCOM_BOUND >= 1
    // The reference's sums of such a box are decided by the EXACT totals of their terms unless a rounding boundary lies within
    // the compensated summation's error bound of a total (the argument: wb_props.h in front of wb_k_box_com; the pieces:
    // wb_core.h): a thread per row adds the row's terms up in 128-bit fixed point, the workgroup's totals go through LDS as 32-bit
    // limbs, one thread decides.  98 % of the boxes of a 0.25 degree year end here; the rest -- a few dozen per pass -- are queued
    // for wb_k_box_com's chained evaluation (until round 6 ALL of them were: 3 193 single-wave chains of ~10^5 instructions each
    // beside the streamer kernels).
    if (a.wlat) {
    for (int g = 0; g < G; ++g) {
        if (!need[g]) continue;                                 // (workgroup-uniform: LDS)
        int x0 = gmin[g], y0 = ylo[g], x1 = gmax[g], y1 = yhi[g];
        if (x0 < 0) x0 = 0;
        if (y0 < 0) y0 = 0;
        if (x1 > a.W - 1) x1 = a.W - 1;
        if (y1 > a.nlat - 1) y1 = a.nlat - 1;
        const int rows = y1 - y0 + 1;
        WB_SINGLE
        elim[0] = 4096; elim[1] = -4096;
        for (int q = 0; q < 15; ++q) bacc[q] = 0ull;
        bacc[15] = ~0ull;
        WB_END_SINGLE
        WB_PFOR(r, rows)
            unsigned long long m; int ex;
            wb_dbl_split(a.wlat[y0 + (int)r], &m, &ex);
            if (m) { wb_atomic_min(&elim[0], ex); wb_atomic_max(&elim[1], ex); }
        WB_END_PFOR
        const int emin = elim[0], emax = elim[1];
        // every term is below max(x1, y1) * w_max < 2^(emax + 53 + 16); the total of < 2^32 cells must stay below 2^127
        const bool range_ok = !(emin > emax || (emax - emin) + 53 + 16 + 32 > 126);
        if (range_ok) {
            WB_PFOR(r, rows)
                WbBoxRows o;
                wb_box_rows_clear(&o);
                wb_box_row_terms(a.wlat, y0 + (int)r, r == 0, x0, x1, emin, &o);
                for (int q = 0; q < 3; ++q) {
                    for (int limb = 0; limb < 4; ++limb) {
                        const unsigned long long part = (unsigned long long)(o.T[q] >> (32 * limb)) & 0xFFFFFFFFull;
                        if (part) wb_atomic_add(&bacc[4 * q + limb], part);
                    }
                    if (o.B[q]) wb_atomic_add(&bacc[12 + q], o.B[q]);
                }
                wb_atomic_min(&bacc[15], o.edge);
            WB_END_PFOR
        }
        WB_SINGLE
        bool sure = false;
        double sv[3] = {0.0, 0.0, 0.0};
        if (range_ok) {
            wb_u128 T[3], B[3];
            for (int q = 0; q < 3; ++q) {
                T[q] = 0;
                for (int limb = 0; limb < 4; ++limb) T[q] += (wb_u128)bacc[4 * q + limb] << (32 * limb);
                B[q] = bacc[12 + q];
            }
            sure = wb_box_decide(T, B, bacc[15], emin, sv);
        }
        if (sure) {
            double* out = a.ot_sums + ((size_t)s * WB_OT_CAP + g) * 3;
            out[0] = sv[0]; out[1] = sv[1]; out[2] = sv[2];
            wb_atomic_add(&a.ctl->box_bound, 1);
#if WB_BOXCOM_BOUND == 2
            a.boxq[wb_atomic_add(&a.ctl->n_boxq, 1)] = (s * WB_OT_CAP + g) | 0x40000000;     // (test build: the chained passes all the same, and the two compared)
#endif
        } else {
            a.boxq[wb_atomic_add(&a.ctl->n_boxq, 1)] = s * WB_OT_CAP + g;
        }
        WB_END_SINGLE
    }
    }
#endif
    WB_SINGLE
    a.ot_n[s] = G;
    WB_END_SINGLE
}

// ------------------------------------------------------------------------------------------------
// streamer index
struct WbStreamerArgs {
    int cap_seg, nlon, nlat, W, cap_pairs, cap_ev, hash_size;  // hash_size: pow2 >= 2*cap_pairs
    WbCtl* ctl;                // ctl->nsel selected contours; counters of the pass
    double geo_dis, cont_dis;
    double hthr;               // sin^2(geo_dis / (2*6371))
    const int* sel_job;
    const int* sel_cont;
    const int* c_off;
    const int* c_len;
    const uint32_t* cpts;
    // per-latitude / per-longitude tables (host libm): radians, sin(phi/2), cos(phi/2), cos(phi), ...
    const double* lat_rad;     // [nlat]
    const double* lon_rad;     // [nlon]
    const double* sphi2;       // [nlat] sin(lat_rad/2)
    const double* cphi2;       // [nlat] cos(lat_rad/2)
    const double* cphi;        // [nlat] cos(lat_rad)
    const double* slam2;       // [nlon] sin(lon_rad/2)
    const double* clam2;       // [nlon] cos(lon_rad/2)
    // per selected contour, packed by point: contour s owns [poff[s], poff[s+1])
    long long* poff;           // [nsel+1] first point of a selected contour in the packed arrays (device scan)
    int* item_off;             // [nsel+1] first work item of a contour in the all-pairs kernel's list
    int* item_sel;             // [items] contour / first 16-row tile of a work item
    int* item_tile;
    int* f2_off;               // [nsel+1] first 256-pair block of a contour in routine 2's work list
    double* on;                // on[k] = d(k-1, k) * 6371, on[0] = 0
    const double* on_tab;      // [nlat][nlon][4]: unit steps E, N, NE, NW from the host's math library (or null)
    double* pre;               // inclusive prefix of on
    double* ptab;              // [5][npts_total] sin(phi/2), cos(phi/2), cos(phi), sin(lam/2), cos(lam/2) per point
    long long npts_total;      // capacity of the packed per-point arrays (plane stride of ptab)
    int* bbx;                  // per block of WB_BLK points (+1: the segment leaving the block): xmin, xmax, ymin, ymax;
                               // contour s owns blocks [(poff[s] >> 5) + s, ...)
    int* tbx;                  // per tile of WB_PAIR_ROWS rows: xmin, xmax, ymin, ymax; contour s owns tiles [(poff[s] >> 4) + s, ...)
    int dy_max;                // |y_i - y_j| above this cannot be within geo_dis (R * dphi <= geo)
    const int* dxlim;          // [nlat] |x_i - x_j| above this cannot be within geo_dis for a base point in row y (<= 120)
    unsigned long long* pairs; // [nsel][cap_pairs] (ind1 << 32 | ind2)
    int* alive;                // [nsel][cap_pairs]
    int* npairs;               // [nsel]
    int* stage;                // [nsel] how many of the four routines have been applied
    int* nalive;               // [nsel] rows left after the last applied routine
    unsigned long long* hkey;  // [nsel][hash_size]
    unsigned long long* hval;  // [nsel][hash_size]
    int* rowmax;               // packed by point like `on`
    int* rowaux;
    int* rowmx;
    int* ev_ind;               // [nsel][cap_ev][2]
    int* tie_m;                // [nsel] 0, or the survivor count of a contour whose check_groups met an exact tie (wb_streamer_resolve_ties)
    int eager;                 // 1: the survivors were laid out by wb_k_streamer_f34, 0: by wb_k_streamer_f234
    int* ev_n;                 // [nsel]
    uint32_t* status;          // [njobs]
    // wb_k_streamer_pairs2 (wb_pairs.h): the all-pairs test with the contour in LDS, regular grids
    const double* sl_tab;      // [dy_max + 1] sin^2(k dlat / 2), host math library
    const double* sx_tab;      // [121]        sin^2(k dlon / 2)
    double geo_band;           // ABSOLUTE guard band on h around hthr (WB_GEO_BAND * hthr, widened by the grid's irregularity)
    int p2_cap;                // contour points its LDS holds; 0: kernel not used (every contour in the tiled kernel's list)
    int p2_split;              // workgroups per contour
    int f234_cap;              // wb_k_streamer_f234: contour points its dynamic LDS holds (12 bytes each + a bit)
    int f1_fused;              // 1: wb_k_streamer_f234 runs routine 1 itself (no wb_k_streamer_f1 launch)
    int* order;                // [nsel] the selected contours, those with the most candidate pairs first (wb_k_pairs_order)
    int* recheck;              // [recheck_cap][4] pairs inside a guard band: contour, i, j, 0 (wb_k_streamer_recheck decides them)
    int recheck_cap;
};

// sklearn's haversine on radians, float64 (streamer_index.py:130): 2 asin(sqrt(sin^2 + cos cos sin^2))
WB_HD inline double wb_haversine(double la1, double lo1, double la2, double lo2) {
    const double s0 = sin(0.5 * (la1 - la2));
    const double s1 = sin(0.5 * (lo1 - lo2));
    return 2.0 * asin(sqrt(s0 * s0 + cos(la1) * cos(la2) * s1 * s1));
}

#define WB_BLK 32
#define WB_GEO_BAND 1e-11     // relative guard band on h = sin^2(d/2R) around its threshold
#define WB_CONT_BAND 1e-6     // absolute guard band (km) on the along-contour distance

// per selected contour: on[] and its prefix (streamer_index.py:133), per-point trigonometric table, block boxes
#ifndef WB_PREP_OCC
#define WB_PREP_OCC 4              // waves per SIMD the register allocation leaves room for (4: 128 registers; the kernel is launched with 256 threads)
#endif
WB_KERNEL_LB(wb_k_streamer_prep, 256, WB_PREP_OCC, WbStreamerArgs a) {
    const int s = WB_BLOCK_ID;
    if (s >= a.ctl->nsel) return;
    const int job = a.sel_job[s], k = a.sel_cont[s];
    const size_t jb = (size_t)job * a.cap_seg;
    const uint32_t* pts = a.cpts + 2 * jb + a.c_off[jb + k];
    const int N = a.c_len[jb + k];
    const long long po = a.poff[s];
    double* on = a.on + po;
    double* pre = a.pre + po;
    double* tab = a.ptab + po;
    int* bbx = a.bbx + 4 * ((po >> 5) + s);
    const int nblk = (N + WB_BLK - 1) / WB_BLK;
    // (WB_PFOR_B: four points with their predecessors requested, then -- straight-line, behind them -- the four table values;
    //  written one point per iteration the loop waited twice per point, nine times over for a contour of 2 400 points)
    uint32_t p1a[WB_PFOR_B_WIDTH], p0a[WB_PFOR_B_WIDTH];
    double da[WB_PFOR_B_WIDTH];
    WB_PFOR_B(i, ic, u, N)
        const uint32_t p1 = pts[ic], p0 = pts[ic > 0 ? ic - 1 : 0];
        p1a[u] = p1;
        p0a[u] = p0;
        const int y1 = wb_pt_y(p1), y0 = wb_pt_y(p0), dy = y1 - y0, dx = wb_pt_x(p1) - wb_pt_x(p0);        // dx in the extended coordinates
        // a step to a neighbouring grid point: the host's value (bit for bit the reference's), looked up from the
        // lower of the two points (the lower-left one of a row step)
        const bool step = a.on_tab && dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1 && (dx | dy);
        const int yb = dy < 0 ? y1 : y0;                                         // base row
        const int xb = dy < 0 ? wb_pt_x(p1) : (dy > 0 ? wb_pt_x(p0) : (dx > 0 ? wb_pt_x(p0) : wb_pt_x(p1)));
        const int dxs = dy < 0 ? -dx : dx;                                       // column step seen from the base point
        const int kind = dy == 0 ? 0 : (dxs == 0 ? 1 : (dxs > 0 ? 2 : 3));
        const size_t idx = step ? ((size_t)yb * a.nlon + (xb % a.nlon)) * 4 + kind : 0;
        da[u] = a.on_tab ? a.on_tab[idx] : 0.0;
    WB_PFOR_B_USE(i, u)
        const uint32_t p1 = p1a[u], p0 = p0a[u];
        const int y1 = wb_pt_y(p1), x1 = wb_pt_x(p1) % a.nlon;
        double d = 0.0;
        if (i > 0) {
            const int y0 = wb_pt_y(p0), dy = y1 - y0, dx = wb_pt_x(p1) - wb_pt_x(p0);
            if (a.on_tab && dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1 && (dx | dy)) d = da[u];
            else d = wb_haversine(a.lat_rad[y0], a.lon_rad[wb_pt_x(p0) % a.nlon], a.lat_rad[y1], a.lon_rad[x1]) * 6371.0;
        }
        on[i] = d;
        if (a.ptab) {                                         // per-point table of the tiled all-pairs kernel (absent when every
            tab[0 * a.npts_total + i] = a.sphi2[y1];          // contour goes through wb_k_streamer_pairs2)
            tab[1 * a.npts_total + i] = a.cphi2[y1];
            tab[2 * a.npts_total + i] = a.cphi[y1];
            tab[3 * a.npts_total + i] = a.slam2[x1];
            tab[4 * a.npts_total + i] = a.clam2[x1];
        }
    WB_END_PFOR_B
    // the boxes: a lane requests all points of its tile / block before it compares any (an index past the contour's end is
    // clamped to its last point, which belongs to the box that asks for it)
    int* tbx = a.tbx + 4 * ((po >> 4) + s);
    WB_PFOR(tl, (N + 15) / 16)
        const int lo = (int)tl * 16;
        uint32_t q[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) q[e] = pts[lo + e < N ? lo + e : N - 1];
        int x0 = 0x7FFFFFFF, x1 = -1, y0 = 0x7FFFFFFF, y1 = -1;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int x = wb_pt_x(q[e]), y = wb_pt_y(q[e]);
            x0 = x < x0 ? x : x0; x1 = x > x1 ? x : x1;
            y0 = y < y0 ? y : y0; y1 = y > y1 ? y : y1;
        }
        tbx[4 * tl + 0] = x0; tbx[4 * tl + 1] = x1; tbx[4 * tl + 2] = y0; tbx[4 * tl + 3] = y1;
    WB_END_PFOR
    WB_PFOR(b, nblk)
        const int lo = (int)b * WB_BLK;                        // points lo .. lo + WB_BLK: one point past the block
        uint32_t q[WB_BLK + 1];
#pragma unroll
        for (int e = 0; e <= WB_BLK; ++e) q[e] = pts[lo + e < N ? lo + e : N - 1];
        int x0 = 0x7FFFFFFF, x1 = -1, y0 = 0x7FFFFFFF, y1 = -1;
#pragma unroll
        for (int e = 0; e <= WB_BLK; ++e) {
            const int x = wb_pt_x(q[e]), y = wb_pt_y(q[e]);
            x0 = x < x0 ? x : x0; x1 = x > x1 ? x : x1;
            y0 = y < y0 ? y : y0; y1 = y > y1 ? y : y1;
        }
        bbx[4 * b + 0] = x0; bbx[4 * b + 1] = x1; bbx[4 * b + 2] = y0; bbx[4 * b + 3] = y1;
    WB_END_PFOR
    // Running sum of `on` (streamer_index.py:133-135).  Only DIFFERENCES of it are used, against cont_dis with a guard band of
    // 1e-6 km inside which the pair's distance is summed again term by term in the reference's order (wb_k_streamer_recheck,
    // wb_k_streamer_pairs) -- so the prefix itself may be formed as a parallel scan forms it (a few ulp of 4e4 km off the
    // left-to-right sum).  A lane walking the contour's 2 500 terms alone set this kernel's duration before.
    WB_SHARED(double, stage, 2048);
    double carry = 0.0;
    for (int c0 = 0; c0 < N; c0 += 2048) {
        const int cn = N - c0 < 2048 ? N - c0 : 2048;
        WB_PFOR(i, cn)
            stage[i] = on[c0 + i];
        WB_END_PFOR
        carry = wb_block_cumsum_f64(stage, cn, carry);
        WB_PFOR(i, cn)
            pre[c0 + i] = stage[i];
        WB_END_PFOR
    }
    WB_SINGLE
    a.npairs[s] = 0;
    a.stage[s] = 0;
    a.ev_n[s] = 0;
    WB_END_SINGLE
}

// all-pairs test (streamer_index.py:130-157); grid = nsel * tiles, a workgroup owns WB_PAIR_ROWS rows i.
// Blocks of WB_BLK candidate points j are skipped as a whole when their bounding box cannot hold a partner:
// |x_i - x_j| <= 120 (:157), |y_i - y_j| <= dy_max (geo < geo_dis needs R*|dphi| < geo_dis), and the block's
// largest along-contour distance must exceed cont_dis.  What survives is evaluated exactly as before.
#define WB_MAX_BLOCKS 512       // contours up to 16384 points; longer ones fall back to sweeping all blocks

// six waves per SIMD (80 VGPRs): the kernel waits on memory 60 % of its wave cycles; 5 waves (96) are no faster, 7-8 spill
WB_KERNEL_LB(wb_k_streamer_pairs, 64, 6, WbStreamerArgs a) {
    // work list built on the device (wb_k_pair_items): workgroup w (ONE wave) sweeps WB_PAIR_TPB tiles from item_tile[w] of
    // contour item_sel[w]; single-wave workgroups make the phase barriers free and keep 32 tiles in flight per CU
    if (WB_BLOCK_ID >= a.ctl->n_pair_items) return;
    const int tiles_per_block = WB_PAIR_TPB;
    const int* blk_sel = a.item_sel;
    const int* blk_tile0 = a.item_tile;
    long long evaluated = 0;                    // pairs whose great-circle term is evaluated by this lane
    const int s = blk_sel[WB_BLOCK_ID];
    const int job = a.sel_job[s], k = a.sel_cont[s];
    const size_t jb = (size_t)job * a.cap_seg;
    const uint32_t* pts = a.cpts + 2 * jb + a.c_off[jb + k];
    const int N = a.c_len[jb + k];
    const long long po = a.poff[s];
    const double* on = a.on + po;
    const double* pre = a.pre + po;
    const double* t_sp = a.ptab + 0 * a.npts_total + po;
    const double* t_cp = a.ptab + 1 * a.npts_total + po;
    const double* t_c = a.ptab + 2 * a.npts_total + po;
    const double* t_sl = a.ptab + 3 * a.npts_total + po;
    const double* t_cl = a.ptab + 4 * a.npts_total + po;
    const int* bbx = a.bbx + 4 * ((po >> 5) + s);
    const int nblk = (N + WB_BLK - 1) / WB_BLK;
    unsigned long long* pairs = a.pairs + (size_t)s * a.cap_pairs;
    WB_SHARED(int, blist, WB_MAX_BLOCKS);
    WB_SHARED(int, nlist, 1);
    WB_SHARED(int, tb, 4);
    WB_SHARED(double, ri, WB_PAIR_ROWS * 6);
    WB_SHARED(int, rx, WB_PAIR_ROWS);
    WB_SHARED(int, ry, WB_PAIR_ROWS);
    WB_SHARED(int, rl, WB_PAIR_ROWS);

    const int tile_begin = blk_tile0[WB_BLOCK_ID];
    WB_PSTAMP_BEGIN
    for (int tile = tile_begin; tile < tile_begin + tiles_per_block; ++tile) {
    const int i0 = tile * WB_PAIR_ROWS;
    if (i0 >= N - 1) break;
    const int i1 = i0 + WB_PAIR_ROWS < N - 1 ? i0 + WB_PAIR_ROWS : N - 1;   // rows i0 .. i1-1
    WB_SINGLE
    const int* tb4 = a.tbx + 4 * ((po >> 4) + s) + 4 * tile;     // box of rows i0 .. i0+15 (a superset of i0 .. i1-1)
    // dxlim grows with |latitude|: over the tile's rows it is largest at one of the two ends of their y range
    const int ya = tb4[2] < 0 ? 0 : (tb4[2] > a.nlat - 1 ? a.nlat - 1 : tb4[2]), yb = tb4[3] < 0 ? 0 : (tb4[3] > a.nlat - 1 ? a.nlat - 1 : tb4[3]);
    const int ml = a.dxlim[ya] > a.dxlim[yb] ? a.dxlim[ya] : a.dxlim[yb];
    tb[0] = tb4[0] - ml; tb[1] = tb4[1] + ml; tb[2] = tb4[2] - a.dy_max; tb[3] = tb4[3] + a.dy_max;
    nlist[0] = 0;
    WB_END_SINGLE
    WB_PSTAMP(a.ctl, 0);
    const bool use_list = nblk <= WB_MAX_BLOCKS;
    if (use_list) {
        const double pre_i0 = pre[i0];
        WB_PFOR(b, nblk)
            const int jlo = (int)b * WB_BLK, jhi = jlo + WB_BLK - 1 < N - 1 ? jlo + WB_BLK - 1 : N - 1;
            if (jhi <= i0) continue;                                        // only j > i
            const int* bb = bbx + 4 * b;
            if (bb[0] > tb[1] || bb[1] < tb[0] || bb[2] > tb[3] || bb[3] < tb[2]) continue;
            if (pre[jhi] - pre_i0 <= a.cont_dis - WB_CONT_BAND) continue;   // cont(i,j) <= pre[jhi] - pre[i0]
            blist[wb_atomic_add(&nlist[0], 1)] = (int)b;
        WB_END_PFOR
    }
    const int nb = use_list ? nlist[0] : nblk;
    WB_PSTAMP(a.ctl, 1);
    // the tile's rows live in LDS (read as broadcasts); a lane owns ONE candidate point j, keeps its data in
    // registers and sweeps the rows: per evaluated pair that is 0.4 global loads instead of 14
    const int nrows = i1 - i0;
    WB_PFOR(ii, nrows)
        const int i = i0 + (int)ii;
        ri[6 * ii + 0] = t_sp[i]; ri[6 * ii + 1] = t_cp[i]; ri[6 * ii + 2] = t_c[i];
        ri[6 * ii + 3] = t_sl[i]; ri[6 * ii + 4] = t_cl[i]; ri[6 * ii + 5] = pre[i];
        rx[ii] = wb_pt_x(pts[i]);
        ry[ii] = wb_pt_y(pts[i]);
        rl[ii] = a.dxlim[wb_pt_y(pts[i])];
    WB_END_PFOR
    WB_PSTAMP(a.ctl, 2);
#if defined(WB_STAMPS) && WB_STAMPS == 2
    if (threadIdx.x == 0) { atomicAdd((unsigned long long*)&a.ctl->dbg[12], 1ull); atomicAdd((unsigned long long*)&a.ctl->dbg[13], (unsigned long long)nb); }
#endif
    WB_PFOR(idx, (int64_t)nb * WB_BLK)
        const int bsel = (int)(idx / WB_BLK), jj = (int)(idx % WB_BLK);
        const int j = (use_list ? blist[bsel] : bsel) * WB_BLK + jj;
        if (j <= i0 || j >= N) continue;
        const int xj = wb_pt_x(pts[j]), yj = wb_pt_y(pts[j]);
        const double spj = t_sp[j], cpj = t_cp[j], cj = t_c[j], slj = t_sl[j], clj = t_cl[j], prej = pre[j];
        uint32_t passed = 0;                                          // bit ii: pair (i0 + ii, j) is a candidate (WB_PAIR_ROWS <= 32)
        for (int ii = 0; ii < nrows; ++ii) {
            const int i = i0 + ii;
            if (j <= i) break;                                        // only j > i
            int dx = rx[ii] - xj;
            if (dx < 0) dx = -dx;
            if (dx > rl[ii]) continue;                                // streamer_index.py:157 (<= 120) and the row's geo_dis reach
            int dy = ry[ii] - yj;
            if (dy < 0) dy = -dy;
            if (dy > a.dy_max) continue;                              // R * |dphi| alone already exceeds geo_dis
            // sin((phi_i - phi_j)/2), sin((lam_i - lam_j)/2) by the subtraction formulas
            const double sa = ri[6 * ii + 0] * cpj - ri[6 * ii + 1] * spj;
            const double sb = ri[6 * ii + 3] * clj - ri[6 * ii + 4] * slj;
            const double h = sa * sa + ri[6 * ii + 2] * cj * (sb * sb);
            ++evaluated;
            bool near_geo;
            if (fabs(h - a.hthr) <= WB_GEO_BAND * a.hthr) {            // guard band: evaluate as sklearn does
                wb_atomic_add(&a.ctl->band[0], 1);
                const uint32_t pi = pts[i], pj = pts[j];
                const int yi = wb_pt_y(pi), yj = wb_pt_y(pj), xim = wb_pt_x(pi) % a.nlon, xjm = wb_pt_x(pj) % a.nlon;
                const double geo = wb_haversine(a.lat_rad[yi], a.lon_rad[xim], a.lat_rad[yj], a.lon_rad[xjm]) * 6371.0;
                near_geo = geo < a.geo_dis;
            } else {
                near_geo = h < a.hthr;
            }
            if (!near_geo) continue;
            double cont = prej - ri[6 * ii + 5];
            if (fabs(cont - a.cont_dis) <= WB_CONT_BAND) {            // guard band: sequential row cumsum (:135)
                wb_atomic_add(&a.ctl->band[1], 1);
                double acc = 0.0;
                for (int m = i + 1; m <= j; ++m) acc = acc + on[m];
                cont = acc;
            }
            if (!(cont > a.cont_dis)) continue;
            passed |= 1u << ii;
        }
        // the rows that passed are appended after the sweep, one returning atomic per wave (inside the row loop every
        // row with a pass stalled the wave for a trip to L2: 16 trips per tile where streamers are)
        int slot = wb_append_slots(&a.npairs[s], wb_popc32(passed));
        while (passed) {
            const int ii = wb_ctz32(passed);
            passed &= passed - 1u;
            if (slot < a.cap_pairs) pairs[slot] = ((unsigned long long)(i0 + ii) << 32) | (unsigned long long)j;
            ++slot;
        }
    WB_END_PFOR
    WB_PSTAMP(a.ctl, 3);
    }   // tiles
    WB_PSTAMP_FLUSH(a.ctl);
    // one global atomic per workgroup (the lanes' counts meet in LDS first: 64 lanes adding to one global word serialise)
    WB_SHARED(unsigned long long, evsum, 1);
    WB_SINGLE
    evsum[0] = 0ull;
    WB_END_SINGLE
    if (evaluated) wb_atomic_add(&evsum[0], (unsigned long long)evaluated);
    WB_SYNC();
    WB_SINGLE
    if (evsum[0]) wb_atomic_add((unsigned long long*)&a.ctl->evaluated_pairs, evsum[0]);
    WB_END_SINGLE
}

#ifndef WB_F1_LTAB
#define WB_F1_LTAB 4096            // entries of routine 1's LDS table (32 KiB)
#endif
WB_HD inline int wb_bits_for_i(int n) { int b = 0; while ((1 << b) < n) ++b; return b; }

// routine 1: check_duplicates (streamer_index.py:160-183); also fixes the stage bookkeeping of :261-264.  Every thread of the
// workgroup calls the body.  (It was tried behind the sweep of wb_k_streamer_pairs2, where the contour's points are in LDS
// already: the 16 waves and 64 KiB of that workgroup then sit through this routine's latency-bound passes over the pair list
// -- 2.3 ms for the fused kernel against 1.0 + 0.4 apart.  `ltab_only` is what remains of that: no fall-back to the tables
// in global memory; false is returned when the western pairs do not fit the LDS table.)
// A sweep over a contour's candidate pairs, four per lane and step: the four pairs are requested together, then their eight
// end points together, and only then the body runs (p, pr = pairs[p], pi, pj = the end points).  Pair after pair -- a load,
// a dependent gather, the body -- the one contour with 50 000 pairs spent 49 steps x 2 round trips per sweep in each of the
// three sweeps of routine 1, and it alone set the kernel's duration.
#define WB_PAIRS_FOR4(P, pairs, pts)                                                                                  \
    WB_PFOR(wb_g4, ((P) + 3) / 4)                                                                                     \
        unsigned long long wb_pr4[4];                                                                                 \
        uint32_t wb_pi4[4], wb_pj4[4];                                                                                \
        _Pragma("unroll") for (int wb_u = 0; wb_u < 4; ++wb_u) {                                                      \
            const int wb_pp = 4 * (int)wb_g4 + wb_u;                                                                  \
            wb_pr4[wb_u] = (pairs)[wb_pp < (P) ? wb_pp : (P) - 1];                                                    \
        }                                                                                                             \
        _Pragma("unroll") for (int wb_u = 0; wb_u < 4; ++wb_u) {                                                      \
            wb_pi4[wb_u] = (pts)[(int)(wb_pr4[wb_u] >> 32)];                                                          \
            wb_pj4[wb_u] = (pts)[(int)(wb_pr4[wb_u] & 0xFFFFFFFFull)];                                                \
        }                                                                                                             \
        _Pragma("unroll") for (int wb_u = 0; wb_u < 4; ++wb_u) {                                                      \
            const int p = 4 * (int)wb_g4 + wb_u;                                                                      \
            if (p < (P)) {                                                                                            \
                const unsigned long long pr = wb_pr4[wb_u];                                                           \
                const uint32_t pi = wb_pi4[wb_u], pj = wb_pj4[wb_u];                                                  \
                (void)pr;
#define WB_PAIRS_END4 \
            }         \
        }             \
    WB_END_PFOR

WB_DEV inline bool wb_streamer_f1_body(const WbStreamerArgs& a, int s, int job, const uint32_t* pts, unsigned long long* ltab,
                                       int ltab_cap, bool ltab_only WB_DSTAMP_ARGS) {
    unsigned long long* pairs = a.pairs + (size_t)s * a.cap_pairs;
    int* alive = a.alive + (size_t)s * a.cap_pairs;
    unsigned long long* hkey = a.hkey + (size_t)s * a.hash_size;
    unsigned long long* hval = a.hval + (size_t)s * a.hash_size;
    WB_SHARED(int, cnt, 1);
    WB_SHARED(int, nwest, 1);
    int P = a.npairs[s];
    if (P > a.cap_pairs) {
        WB_SINGLE
        wb_atomic_max(&a.ctl->max_pairs, P);
        wb_atomic_add((unsigned long long*)&a.ctl->cand_pairs, (unsigned long long)P);
        wb_atomic_or(&a.status[job], (uint32_t)WB_ST_OVERFLOW_PAIRS);
        a.nalive[s] = 0;
        a.stage[s] = 4;
        WB_END_SINGLE
        return true;
    }
    const int padd = a.W - a.nlon;
    if (P <= 1) {
        WB_SINGLE
        if (P == 1) alive[0] = 1;
        wb_atomic_max(&a.ctl->max_pairs, P);
        wb_atomic_add((unsigned long long*)&a.ctl->cand_pairs, (unsigned long long)P);
        a.nalive[s] = P;
        a.stage[s] = 0;   // no routine applied: `while len(df_bp.index) > 1`
        WB_END_SINGLE
        return true;
    }
    const int bxx = wb_bits_for_i(a.nlon), byy = wb_bits_for_i(a.nlat), pb = wb_bits_for_i(P);
    // Usual case: the western pairs (a quarter of all on a full-span contour) fit a table in LDS whose 64-bit entries
    // hold the key (x1 mod nlon, y1, x2 mod nlon, y2) above the pair index -- no table in global memory to clear and probe.
    // Two sweeps over the pairs: the western ones are counted and entered (the whole table is cleared in front: 32 KiB, four
    // stores per thread), the eastern ones look themselves up.  (Until round 5 a first sweep counted the western pairs to size
    // the table: a third of the routine's time.)
    bool counted = false;
    if (2 * (bxx + byy) + pb <= 63) {
        // One sweep over the pairs (their end points gathered, their keys made) enters the western ones in the table while it is
        // at most half full and leaves TWO LISTS of (key, pair) words in the contour's scratch: the western and the eastern pairs.
        // What follows reads those lists only, a quarter of the pairs each and no gathers: the eastern pairs look themselves up;
        // and when the western pairs were more than half the table (a contour of the 0.25 degree bench batch with more than
        // 8 000 candidate pairs, 150 of 2 920 -- with the table in global memory, 100-190 us each, they were the kernel's duration)
        // the keys are split by bits of their hash into as many parts as keep the table a third full, and per part the western
        // list is entered and the eastern list looked up.
        WB_SHARED(int, cntp, 3);             // eastern pairs; western pairs of the part; a part that did not fit
        WB_SHARED(unsigned long long, kills, 1);        // pairs removed (a thread counts its own, the waves add them up at the end)
        unsigned long long* const wlist = hkey;          // [P] (the table in global memory below, if it comes to that, starts over)
        unsigned long long* const elist = hval;          // [P]
        const int lmask = ltab_cap - 1;
        const unsigned long long pmask = (1ull << pb) - 1ull;
        WB_PFOR(q, ltab_cap)
            if (q == 0) { nwest[0] = 0; cnt[0] = 0; cntp[0] = 0; cntp[2] = 0; kills[0] = 0ull; }
            ltab[q] = WB_HEMPTY;
        WB_END_PFOR
        WB_DSTAMP(a.ctl, 1);
        auto enter = [&](unsigned long long mine, unsigned long long hk) WB_LAMBDA_INLINE {
            const unsigned long long key = mine >> pb;
            uint32_t sl = (uint32_t)hk & (uint32_t)lmask;
            for (int probe = 0; probe <= lmask; ++probe) {
                unsigned long long cur = ltab[sl];
                if (cur == WB_HEMPTY) {
                    cur = wb_atomic_cas(&ltab[sl], (unsigned long long)WB_HEMPTY, mine);
                    if (cur == WB_HEMPTY) break;
                }
                if ((cur >> pb) == key) { wb_atomic_min(&ltab[sl], mine); break; }
                sl = (sl + 1) & (uint32_t)lmask;
            }
        };
        unsigned long long mykills = 0ull;                   // (a thread's share; the host simulation's one thread: all of them)
        auto look_up = [&](unsigned long long mine, unsigned long long hk) WB_LAMBDA_INLINE {
            const unsigned long long key = mine >> pb;
            uint32_t sl = (uint32_t)hk & (uint32_t)lmask;
            for (int probe = 0; probe <= lmask; ++probe) {
                const unsigned long long cur = ltab[sl];
                if (cur == WB_HEMPTY) break;
                if ((cur >> pb) == key) {
                    const int p = (int)(mine & pmask), pw = (int)(cur & pmask);
                    alive[pairs[pw] < pairs[p] ? p : pw] = 0;             // the first in (ind1, ind2) order stays
                    ++mykills;
                    break;
                }
                sl = (sl + 1) & (uint32_t)lmask;
            }
        };
        WB_PAIRS_FOR4(P, pairs, pts)
            alive[p] = 1;
            const bool west = wb_pt_x(pi) < padd && wb_pt_x(pj) < padd;
            const bool east = !west && wb_pt_x(pi) >= a.nlon && wb_pt_x(pj) >= a.nlon;
            // (list slots by one atomic per wave: thousands of lanes adding to one LDS word take their turns -- a contour with 16 000
            //  pairs spent 50 us of its 250 waiting for them)
            const int widx = wb_append_slot(&nwest[0], west), eidx = wb_append_slot(&cntp[0], east);
            if (west || east) {
                const unsigned long long key = ((((unsigned long long)(wb_pt_x(pi) % a.nlon) << byy | (unsigned long long)wb_pt_y(pi)) << bxx |
                                                 (unsigned long long)(wb_pt_x(pj) % a.nlon)) << byy) | (unsigned long long)wb_pt_y(pj);
                const unsigned long long mine = (key << pb) | (unsigned long long)p;
                if (west) {
                    wlist[widx] = mine;
                    if (2 * widx < ltab_cap) enter(mine, wb_mix64(key));
                } else {
                    elist[eidx] = mine;
                }
            }
        WB_PAIRS_END4
        counted = true;
        WB_DSTAMP(a.ctl, 2);
        const int nW = nwest[0], nE = cntp[0];
        bool fits = 2 * nW <= ltab_cap;
        if (fits) {
            WB_PFOR(e, nE)
                const unsigned long long mine = elist[e];
                look_up(mine, wb_mix64(mine >> pb));
            WB_END_PFOR
            WB_DSTAMP(a.ctl, 3);
        } else if (ltab_cap >= 256) {
            int npart = 2;                       // parts that are 0.35 full on average (a part may hold half the table; the spread of a part's share is a few per cent)
            while (7LL * npart * ltab_cap < 20LL * nW && npart < 256) npart <<= 1;
            fits = true;
            for (int part = 0; part < npart && fits; ++part) {
                WB_PFOR(q, ltab_cap)
                    if (q == 0) cntp[1] = 0;
                    ltab[q] = WB_HEMPTY;
                WB_END_PFOR
                WB_PFOR(e, nW)
                    const unsigned long long mine = wlist[e], hk = wb_mix64(mine >> pb);
                    const bool mine_now = (int)((hk >> 40) & (unsigned long long)(npart - 1)) == part;
                    const int slot = wb_append_slot(&cntp[1], mine_now);
                    if (mine_now) {
                        if (2 * slot < ltab_cap) enter(mine, hk);
                        else cntp[2] = 1;
                    }
                WB_END_PFOR
                if (cntp[2]) { fits = false; break; }               // (a part with more keys than half the table: the table in global memory after all)
                WB_PFOR(e, nE)
                    const unsigned long long mine = elist[e], hk = wb_mix64(mine >> pb);
                    if ((int)((hk >> 40) & (unsigned long long)(npart - 1)) == part) look_up(mine, hk);
                WB_END_PFOR
            }
            if (fits) WB_HOSTSIM_PATH(0);
            WB_DSTAMP(a.ctl, 5);
        }
        if (fits) {
            wb_block_add_u64(&kills[0], mykills);
            WB_SINGLE
            wb_atomic_max(&a.ctl->max_pairs, P);
            wb_atomic_add((unsigned long long*)&a.ctl->cand_pairs, (unsigned long long)P);
            a.nalive[s] = P - (int)kills[0];
            a.stage[s] = 1;
            WB_END_SINGLE
            return true;
        }
        if (!ltab_only) {
            WB_PFOR(p, P)                            // (pairs the parts before have removed are found again below)
                alive[p] = 1;
            WB_END_PFOR
        }
    }
    if (ltab_only) return false;
    // (keys wider than the LDS table's entries, a tiny table, or a part that did not fit: the table in global memory)
    if (!counted) {
        WB_SINGLE
        nwest[0] = 0;
        WB_END_SINGLE
        WB_PAIRS_FOR4(P, pairs, pts)
            alive[p] = 1;
            if (wb_pt_x(pi) < padd && wb_pt_x(pj) < padd) wb_atomic_add(&nwest[0], 1);
        WB_PAIRS_END4
    }
    WB_SINGLE
    cnt[0] = 0;
    wb_atomic_max(&a.ctl->max_pairs, P);
    wb_atomic_add((unsigned long long*)&a.ctl->cand_pairs, (unsigned long long)P);
    WB_END_SINGLE
    // Two pairs coincide after x mod nlon only if one is the copy of the other one period further east: both end points
    // of the western pair lie in x < W - nlon, both of the eastern one in x >= nlon (a mixed pair would need its copy in
    // the opposite index order).  Only western pairs are entered into the table and only eastern pairs look themselves
    // up -- a quarter of the pairs each on a full-span contour; everything else survives without touching it.
    WB_HOSTSIM_PATH(1);
    int hsize = 256;                       // table sized to the pairs that go in, the western ones (clearing it is real traffic)
    while (hsize < 2 * nwest[0] && hsize < a.hash_size) hsize <<= 1;
    const int hmask = hsize - 1;
    WB_PFOR(q, hsize)
        hkey[q] = WB_HEMPTY;
        hval[q] = WB_HEMPTY;
    WB_END_PFOR
    WB_PAIRS_FOR4(P, pairs, pts)
        if (wb_pt_x(pi) < padd && wb_pt_x(pj) < padd) {
            const unsigned long long key = (unsigned long long)(wb_pt_x(pi) % a.nlon) | ((unsigned long long)wb_pt_y(pi) << 16) |
                                           ((unsigned long long)(wb_pt_x(pj) % a.nlon) << 32) |
                                           ((unsigned long long)wb_pt_y(pj) << 48);
            int fresh;
            const int slot = wb_hash_insert(hkey, hmask, key, &fresh);
            if (slot >= 0) wb_atomic_min(&hval[slot], (unsigned long long)p);
        }
    WB_PAIRS_END4
    WB_PAIRS_FOR4(P, pairs, pts)
        if (wb_pt_x(pi) >= a.nlon && wb_pt_x(pj) >= a.nlon && !(wb_pt_x(pi) < padd && wb_pt_x(pj) < padd)) {
            const unsigned long long key = (unsigned long long)(wb_pt_x(pi) % a.nlon) | ((unsigned long long)wb_pt_y(pi) << 16) |
                                           ((unsigned long long)(wb_pt_x(pj) % a.nlon) << 32) |
                                           ((unsigned long long)wb_pt_y(pj) << 48);
            const int slot = wb_hash_find(hkey, hmask, key);
            if (slot >= 0) {
                // of the two the first in frame order = (ind1, ind2) order stays
                const int pw = (int)hval[slot];
                alive[pairs[pw] < pr ? p : pw] = 0;
                wb_atomic_add(&cnt[0], 1);
            }
        }
    WB_PAIRS_END4
    WB_SINGLE
    a.nalive[s] = P - cnt[0];
    a.stage[s] = 1;
    WB_END_SINGLE
    WB_DSTAMP(a.ctl, 4);                    // (the table in global memory)
    return true;
}

// The per-contour filters take the contours in descending order of their candidate pairs (by power of two): a contour's
// cost grows with them, and one of the 2 920 contours of the 0.25 degree bench batch holds 50 000 pairs against a median of a
// few hundred -- started last, it alone ran for as long again as all the others together.
WB_KERNEL(wb_k_pairs_order, WbStreamerArgs a) {
    WB_SHARED(int, hist, 33);
    const int nsel = a.ctl->nsel;
    WB_PFOR(q, 33)
        hist[q] = 0;
    WB_END_PFOR
    WB_PFOR(s, nsel)
        wb_atomic_add(&hist[32 - wb_clz32((uint32_t)a.npairs[s])], 1);
    WB_END_PFOR
    WB_SINGLE
    int run = 0;
    for (int q = 32; q >= 0; --q) { const int c = hist[q]; hist[q] = run; run += c; }
    WB_END_SINGLE
    WB_PFOR(s, nsel)
        a.order[wb_atomic_add(&hist[32 - wb_clz32((uint32_t)a.npairs[s])], 1)] = (int)s;
    WB_END_PFOR
}
// persistent (two workgroups per CU): a contour with a few hundred pairs is ten microseconds of work for 1024 threads, and
// launched one workgroup per contour the kernel's duration was the launching of 2 920 fat workgroups
WB_KERNEL(wb_k_streamer_f1, WbStreamerArgs a) {
    WB_SHARED(unsigned long long, ltab, WB_F1_LTAB);
    WB_SHARED(int, item, 1);
    const int nsel = a.ctl->nsel;
    WB_DSTAMP_BEGIN
    for (;;) {
        WB_SINGLE                                    // (the previous contour ended behind a barrier: its table and counters are done with)
        item[0] = wb_atomic_add(&a.ctl->f1_cursor, 1);
        WB_END_SINGLE
        const int it = item[0];
        if (it >= nsel) break;
        const int s = a.order[it];
        const int job = a.sel_job[s], k = a.sel_cont[s];
        const size_t jb = (size_t)job * a.cap_seg;
        const uint32_t* pts = a.cpts + 2 * jb + a.c_off[jb + k];
#ifdef WB_WHATIF_BIG
        if (a.npairs[s] > WB_WHATIF_BIG) { if (WB_TID == 0) { a.nalive[s] = 0; a.stage[s] = 4; } continue; }
#endif
        WB_DSTAMP(a.ctl, 0);
        (void)wb_streamer_f1_body(a, s, job, pts, ltab, WB_F1_LTAB, false WB_DSTAMP_PASS);
    }
    WB_DSTAMP_FLUSH(a.ctl);
}

// routine 2: check_intersections (:185-200): one lane per candidate pair; contour blocks whose bounding box
// misses the chord's box are skipped, the rest is tested segment by segment with the exact predicate.
#define WB_F2_THREADS 256
#define WB_F2_SUP 8
#define WB_F2_SUPMAX 64
WB_KERNEL(wb_k_streamer_f2, WbStreamerArgs a) {
    // work list on the device: block w of the list = pairs [256 (w - f2_off[s]), +256) of the contour s with
    // f2_off[s] <= w < f2_off[s+1] (wb_k_f2_items; only contours that can still reach routine 2, streamer_index.py:261-264);
    // persistent workgroups stride over the list
    WB_SHARED(int, sbb, 4 * WB_F2_SUPMAX);
    const int nitems = a.ctl->n_f2_items, nsel = a.ctl->nsel;
    for (int w = WB_BLOCK_ID; w < nitems; w += WB_GRID_DIM) {
    int lo = 0, hi = nsel;                                // largest s with f2_off[s] <= w
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.f2_off[mid] <= w) lo = mid; else hi = mid;
    }
    const int s = lo;
    if (a.stage[s] != 1 || a.nalive[s] <= 1) continue;
    const int job = a.sel_job[s], k = a.sel_cont[s];
    const size_t jb = (size_t)job * a.cap_seg;
    const uint32_t* pts = a.cpts + 2 * jb + a.c_off[jb + k];
    const int N = a.c_len[jb + k];
    const uint32_t pe0 = pts[0], pe1 = pts[N - 1];            // the contour's end vertices: its boundary, as a point set (wb_chord_hits_segment)
    const int* bbx = a.bbx + 4 * ((a.poff[s] >> 5) + s);
    const int nblk = (N + WB_BLK - 1) / WB_BLK;
    const unsigned long long* pairs = a.pairs + (size_t)s * a.cap_pairs;
    int* alive = a.alive + (size_t)s * a.cap_pairs;
    const int P = a.npairs[s];
    const int p0 = (w - a.f2_off[s]) * WB_F2_THREADS;
    if (p0 >= P) continue;
    // second level of boxes: one per WB_F2_SUP consecutive blocks (a chord's box meets a handful of them)
    const int nsup = (nblk + WB_F2_SUP - 1) / WB_F2_SUP;
    const bool two = nsup <= WB_F2_SUPMAX;
    if (two) {
        WB_PFOR(sb, nsup)
            int v0 = 0x7FFFFFFF, v1 = -1, v2 = 0x7FFFFFFF, v3 = -1;
            for (int b = (int)sb * WB_F2_SUP; b < ((int)sb + 1) * WB_F2_SUP && b < nblk; ++b) {
                const int* bb = bbx + 4 * b;
                v0 = bb[0] < v0 ? bb[0] : v0; v1 = bb[1] > v1 ? bb[1] : v1;
                v2 = bb[2] < v2 ? bb[2] : v2; v3 = bb[3] > v3 ? bb[3] : v3;
            }
            sbb[4 * sb] = v0; sbb[4 * sb + 1] = v1; sbb[4 * sb + 2] = v2; sbb[4 * sb + 3] = v3;
        WB_END_PFOR
    }
    WB_PFOR(pp, WB_F2_THREADS)
        const int p = p0 + (int)pp;
        if (p >= P || !alive[p]) continue;
        const int i = (int)(pairs[p] >> 32), j = (int)(pairs[p] & 0xFFFFFFFFull);
        const int px = wb_pt_x(pts[i]), py = wb_pt_y(pts[i]), qx = wb_pt_x(pts[j]), qy = wb_pt_y(pts[j]);
        const int cx0 = px < qx ? px : qx, cx1 = px < qx ? qx : px, cy0 = py < qy ? py : qy, cy1 = py < qy ? qy : py;
        bool hit = false;
        for (int b = 0; b < nblk && !hit; ++b) {
            if (two && (b % WB_F2_SUP) == 0) {
                const int* sq = sbb + 4 * (b / WB_F2_SUP);
                if (sq[0] > cx1 || sq[1] < cx0 || sq[2] > cy1 || sq[3] < cy0) { b += WB_F2_SUP - 1; continue; }
            }
            const int* bb = bbx + 4 * b;
            if (bb[0] > cx1 || bb[1] < cx0 || bb[2] > cy1 || bb[3] < cy0) continue;
            const int mlo = b * WB_BLK, mhi = mlo + WB_BLK < N - 1 ? mlo + WB_BLK : N - 1;   // segments mlo .. mhi-1
            uint32_t pa = pts[mlo];
            for (int m = mlo; m < mhi && !hit; ++m) {
                const uint32_t pb = pts[m + 1];
                const int ax = wb_pt_x(pa), ay = wb_pt_y(pa), bx = wb_pt_x(pb), by = wb_pt_y(pb);
                pa = pb;
                // a segment whose box misses the chord's box cannot meet the chord: skip the exact predicate
                if ((ax < cx0 && bx < cx0) || (ax > cx1 && bx > cx1) || (ay < cy0 && by < cy0) || (ay > cy1 && by > cy1)) continue;
                hit = wb_chord_hits_segment(px, py, qx, qy, ax, ay, bx, by, pe0, pe1);
            }
        }
        if (hit) alive[p] = 2;   // 2 = rejected by routine 2 (turned into 0 by f3)
    WB_END_PFOR
    }   // work list
}

// check_groups, exact ties: which of the members of group g whose length equals the group's maximum comes first in the
// reference's ``list(first)`` (wb_core.h: CPython's set).  Members are the survivors m with lab[m] == g, ascending = the
// rows of the reference's frame; their crossing pairs in lexicographic order are what combine_shared saw.  ws: words of
// scratch.  Returns -1 when the group is too large for the scratch (the lowest member stays).
#define WB_TIE_MAX 64
WB_HD inline int wb_group_tie_winner(const int* lab, int M, int g, const uint32_t* pts, const int* mi, const int* mj,
                                     const unsigned long long* gval, unsigned long long bestbits, unsigned long long* ws, long long nws) {
    int nm = 0;
    for (int m = 0; m < M; ++m) if (lab[m] == g) ++nm;
    if (nm > WB_TIE_MAX) return -1;
    int cap = 8;
    while (cap < 8 * nm) cap <<= 1;
    const int maxp = nm * (nm + 1) / 2;                            // pairs: at most nm (nm + 1) / 2
    // 64-bit words: two tables, members, two pair columns, the order, the done flags
    const long long need = 2LL * cap + (nm + 2) / 2 + 2LL * ((maxp + 2) / 2) + (nm + 2) / 2 + (maxp + 8) / 8 + 4;
    if (need > nws) return -1;
    long long* tab = (long long*)ws;                               // 2 * cap
    int* mem = (int*)(ws + 2 * cap);                               // nm
    int* pp = mem + ((nm + 2) & ~1);
    int* pq = pp + ((maxp + 2) & ~1);
    int* out = pq + ((maxp + 2) & ~1);
    unsigned char* done = (unsigned char*)(out + ((nm + 2) & ~1));
    nm = 0;
    for (int m = 0; m < M; ++m) if (lab[m] == g) mem[nm++] = m;
    int np = 0;
    for (int x = 0; x < nm; ++x)
        for (int y = x; y < nm; ++y) {
            const int m1 = mem[x], m2 = mem[y];
            const uint32_t a1 = pts[mi[m1]], b1 = pts[mj[m1]], a2 = pts[mi[m2]], b2 = pts[mj[m2]];
            if (x == y || wb_segments_intersect(wb_pt_x(a1), wb_pt_y(a1), wb_pt_x(b1), wb_pt_y(b1), wb_pt_x(a2), wb_pt_y(a2),
                                                wb_pt_x(b2), wb_pt_y(b2))) { pp[np] = m1; pq[np] = m2; ++np; }
        }
    const int n = wb_pyset_group_order(pp, pq, np, done, tab, cap, out);
    for (int k = 0; k < n; ++k) if (gval[out[k]] == bestbits) return out[k];
    return -1;
}

// routines 3 and 4: check_overlapping (:202-222), check_groups (:224-251), then the polygons' end points
WB_KERNEL(wb_k_streamer_f34, WbStreamerArgs a) {
    const int s = WB_BLOCK_ID;
    if (s >= a.ctl->nsel) return;
    WB_SINGLE
    a.tie_m[s] = 0;
    WB_END_SINGLE
    const int job = a.sel_job[s], k = a.sel_cont[s];
    const size_t jb = (size_t)job * a.cap_seg;
    const uint32_t* pts = a.cpts + 2 * jb + a.c_off[jb + k];
    const int N = a.c_len[jb + k];
    const double* pre = a.pre + a.poff[s];
    const unsigned long long* pairs = a.pairs + (size_t)s * a.cap_pairs;
    int* alive = a.alive + (size_t)s * a.cap_pairs;
    int* rowmax = a.rowmax + a.poff[s];   // per ind1: largest surviving ind2
    int* aux = a.rowaux + a.poff[s];
    int* ev = a.ev_ind + (size_t)s * a.cap_ev * 2;
    unsigned long long* best = a.hval + (size_t)s * a.hash_size;  // reused: per group best value / index
    WB_SHARED(int, cnt, 4);
    const int P = a.npairs[s] > a.cap_pairs ? 0 : a.npairs[s];
    int stage = a.stage[s];
    int nal = a.nalive[s];

    bool rows_ready = false;
    if (stage == 1 && nal > 1) {   // finish routine 2's bookkeeping; the same pass already collects routine 3's row maxima
        WB_PFOR(i, N)
            rowmax[i] = -1;
            if (i == 0) cnt[0] = 0;
        WB_END_PFOR
        WB_PFOR(p, P)
            if (alive[p] == 2) alive[p] = 0;
            if (alive[p]) {
                wb_atomic_add(&cnt[0], 1);
                wb_atomic_max(&rowmax[(int)(pairs[p] >> 32)], (int)(pairs[p] & 0xFFFFFFFFull));
            }
        WB_END_PFOR
        nal = cnt[0];
        stage = 2;
        rows_ready = true;
    }
    if (stage == 2 && nal > 1) {   // routine 3: a pair inside another pair's [ind1, ind2] range goes
        if (!rows_ready) {
            WB_PFOR(i, N)
                rowmax[i] = -1;
            WB_END_PFOR
            WB_PFOR(p, P)
                if (alive[p]) wb_atomic_max(&rowmax[(int)(pairs[p] >> 32)], (int)(pairs[p] & 0xFFFFFFFFull));
            WB_END_PFOR
        }
        WB_PFOR(i, N)
            aux[i] = rowmax[i];
        WB_END_PFOR
        wb_block_scan_excl_max(aux, N, -1);   // aux[i] = max ind2 over rows < i
        WB_SINGLE
        cnt[0] = 0;
        WB_END_SINGLE
        WB_PFOR(p, P)
            if (alive[p]) {
                const int i = (int)(pairs[p] >> 32), j = (int)(pairs[p] & 0xFFFFFFFFull);
                if (j != rowmax[i] || aux[i] >= j) alive[p] = 0;
                else wb_atomic_add(&cnt[0], 1);
            }
        WB_END_PFOR
        nal = cnt[0];
        stage = 3;
    }

    // survivors in (ind1, ind2) order.  After routine 3 there is at most one per row; before it
    // (early exits) there is at most one survivor in total.
    WB_PFOR(i, N)
        rowmax[i] = -1;
    WB_END_PFOR
    WB_PFOR(p, P)
        if (alive[p] == 1) rowmax[(int)(pairs[p] >> 32)] = (int)(pairs[p] & 0xFFFFFFFFull);
    WB_END_PFOR
    WB_PFOR(i, N)
        aux[i] = rowmax[i] >= 0 ? 1 : 0;
    WB_END_PFOR
    const int M = wb_block_scan_excl(aux, N);   // aux[i] = rank of row i among survivors
    int* mi = (int*)a.hkey + (size_t)s * a.hash_size * 2;          // [M] ind1 of survivor m  (reuses the table)
    int* mj = mi + M;                                                // [M] ind2
    int* lab = mj + M;                                               // [M] group label
    WB_PFOR(i, N)
        if (rowmax[i] >= 0) {
            mi[aux[i]] = (int)i;
            mj[aux[i]] = rowmax[i];
        }
    WB_END_PFOR

    if (stage == 3 && nal > 1) {   // routine 4: group chords that intersect; keep the longest of each group
        WB_PFOR(m, M)
            lab[m] = (int)m;
        WB_END_PFOR
        for (int iter = 0; iter < M; ++iter) {
            const int cf = 1 + (iter & 1);       // alternating slots: a slot is rewritten two barriers after its last read
            WB_SINGLE
            cnt[cf] = 0;
            WB_END_SINGLE
            WB_PFOR(pair, (int64_t)M * M)
                const int m1 = (int)(pair / M), m2 = (int)(pair - (int64_t)m1 * M);
                if (m1 < m2 && lab[m1] != lab[m2]) {
                    const uint32_t a1 = pts[mi[m1]], b1 = pts[mj[m1]], a2 = pts[mi[m2]], b2 = pts[mj[m2]];
                    if (wb_segments_intersect(wb_pt_x(a1), wb_pt_y(a1), wb_pt_x(b1), wb_pt_y(b1), wb_pt_x(a2), wb_pt_y(a2),
                                              wb_pt_x(b2), wb_pt_y(b2))) {
                        // hook the larger root under the smaller one (labels are roots after the shortcut pass)
                        const int l1 = lab[m1], l2 = lab[m2];
                        if (l1 != l2) {
                            wb_atomic_min(&lab[l1 > l2 ? l1 : l2], l1 < l2 ? l1 : l2);
                            cnt[cf] = 1;
                        }
                    }
                }
            WB_END_PFOR
            WB_PFOR(m, M)   // shortcut labels to their root
                int l = lab[m];
                while (lab[l] != l) l = lab[l];
                lab[m] = l;
            WB_END_PFOR
            if (!cnt[cf]) break;
        }
        // per group: max of on[ind1 : ind2+1].sum() -- numpy's pairwise sum of the slice, not a prefix difference: two chords
        // shifted by one point along a lattice contour differ by on[ind1] - on[ind2 + 1], two steps of the same kind whose
        // lengths agree to the last bit or the one before, and the reference's argmax (streamer_index.py:241-246) then
        // turns on how the slices were added; ties -> lowest member
        WB_PFOR(m, M)
            best[m] = 0ull;
            best[M + m] = WB_HEMPTY;
            aux[m] = 0;
        WB_END_PFOR
        WB_PFOR(m, M)
            wb_atomic_add(&aux[lab[m]], 1);                        // members per group: a group of one needs no length
        WB_END_PFOR
        const double* onp = a.on + a.poff[s];
        unsigned long long* gval = best + 2 * (size_t)M;          // (the table holds 4 * cap_seg >= 4 M words)
        WB_PFOR(m, M)
            const double v = aux[lab[m]] > 1 ? wb_np_pairwise_sum(onp + mi[m], mj[m] - mi[m] + 1) : 0.0;
            unsigned long long bits;
            memcpy(&bits, &v, 8);
            gval[m] = bits;
            wb_atomic_max(&best[lab[m]], bits);
        WB_END_PFOR
        WB_PFOR(m, M)
            if (gval[m] == best[lab[m]]) wb_atomic_min(&best[M + lab[m]], (unsigned long long)m);
        WB_END_PFOR
        // several members with the SAME length (to the last bit): the reference keeps the one that comes first in a CPython
        // set's iteration order -- resolved by wb_k_streamer_ties after this kernel (rare; kept out of this kernel's
        // registers); here the lowest such member stands in and the contour is marked
        WB_PFOR(m, M)
            if (gval[m] == best[lab[m]] && (int)best[M + lab[m]] != (int)m) { a.tie_m[s] = M; a.ctl->tie_marks = 1; }
        WB_END_PFOR
        WB_PFOR(m, M)
            aux[m] = (lab[m] == (int)m) ? 1 : 0;   // group roots, in order of their lowest member
        WB_END_PFOR
        const int Gn = wb_block_scan_excl(aux, M);
        WB_PFOR(m, M)
            if (lab[m] == (int)m && aux[m] < a.cap_ev) {
                const int w = (int)best[M + m];
                ev[2 * aux[m] + 0] = mi[w];
                ev[2 * aux[m] + 1] = mj[w];
            }
        WB_END_PFOR
        WB_SINGLE
        a.ev_n[s] = Gn < a.cap_ev ? Gn : a.cap_ev;
        if (Gn > a.cap_ev) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_OVERFLOW_PAIRS);
        a.stage[s] = 4;
        a.nalive[s] = Gn;
        WB_END_SINGLE
    } else {
        WB_PFOR(m, M)
            if (m < a.cap_ev) {
                ev[2 * m + 0] = mi[m];
                ev[2 * m + 1] = mj[m];
            }
        WB_END_PFOR
        WB_SINGLE
        a.ev_n[s] = M < a.cap_ev ? M : a.cap_ev;
        if (M > a.cap_ev) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_OVERFLOW_PAIRS);
        a.stage[s] = stage;
        a.nalive[s] = nal;
        WB_END_SINGLE
    }
}

// routines 2, 3 and 4 in one kernel, routine 2 evaluated LAZILY.
//   check_overlapping (:202-222) keeps, of the pairs that passed check_intersections (:185-200), the maximal [ind1, ind2]
//   intervals: sweeping the rows ind1 in ascending order with M = the largest ind2 kept so far, a row contributes its
//   largest ind2 > M whose chord passes routine 2 -- and nothing else of that row, and no pair with ind2 <= M, matters,
//   whatever routine 2 would say about them.  So the chord test (the expensive part: chord against every contour segment)
//   runs only for the candidates the sweep actually asks about, in descending ind2 per row until one passes: a few dozen
//   tests per contour instead of thousands, each done by the whole workgroup over the contour held in LDS.
//   The reference's loop control (`while len(df_bp.index) > 1`, :261-264) is kept: routine 2 only runs when more than one
//   pair survived routine 1 (wb_k_streamer_f1's bookkeeping); routines 3 and 4 are the identity on fewer than two pairs,
//   so composing them unconditionally gives the same rows.
//   check_groups (:224-251): label propagation over the intersecting chords, the longest member of a group stays.
#ifndef WB_F234_LPTS
#define WB_F234_LPTS 4096          // contour points (and per-row maxima) held in LDS; longer contours work in global memory
#endif
#ifndef WB_F234_THREADS
#define WB_F234_THREADS 512
#endif
#ifndef WB_F1_THREADS
#define WB_F1_THREADS 1024         // routine 1: its duration is that of the few contours with tens of thousands of pairs (0.39 -> 0.27 ms against 256)
#endif
#ifndef WB_F234_OCC
#define WB_F234_OCC 6               // waves per SIMD the register allocation leaves room for (6: 80 registers, three workgroups per CU)
#endif
#define WB_F234_R4 256             // survivors whose routine-4 arrays are held in LDS
#ifndef WB_F234_CH
#define WB_F234_CH WB_F234_R4       // records listed per round (at most R4: the list lies where routine 4's arrays will); the tests build a 4-record variant
#endif
#define WB_F234_CC (2 * WB_F234_R4)   // chords tested per round
#define WB_F234_K 8                // candidates of a failing row tested per round
#define WB_F234_GROUP 128          // chords whose (chord, block) items are listed together
#define WB_F234_ITEMS 1024         // ... and how many the list holds
#define WB_F234_FMAX 8             // failing records whose candidates a round looks at
#define WB_F234_FSTAGE 128         // candidates in range of one of them held in LDS (more: only the largest is tested)
#define WB_F234_SCR (10 * WB_F234_R4)   // ints of the sweep's lists / routine 4's arrays (4 R4 ints + 3 R4 64-bit words)
      // per-contour work is a chain of short phases: many lanes = few dependent round trips per phase
WB_KERNEL_LB(wb_k_streamer_f234, WB_F234_THREADS, WB_F234_OCC, WbStreamerArgs a) {
    if (WB_BLOCK_ID >= a.ctl->nsel) return;
    const int s = a.order[WB_BLOCK_ID];
    if (WB_TID == 0) a.tie_m[s] = 0;
    const int job = a.sel_job[s], k = a.sel_cont[s];
    const size_t jb = (size_t)job * a.cap_seg;
    const uint32_t* gpts = a.cpts + 2 * jb + a.c_off[jb + k];
    const int N = a.c_len[jb + k];
    const double* pre = a.pre + a.poff[s];
    const unsigned long long* pairs = a.pairs + (size_t)s * a.cap_pairs;
    const int* alive = a.alive + (size_t)s * a.cap_pairs;
    int* ev = a.ev_ind + (size_t)s * a.cap_ev * 2;
    unsigned long long* best = a.hval + (size_t)s * a.hash_size;  // routine 4: per group best value / index
    const int P = a.npairs[s] > a.cap_pairs ? 0 : a.npairs[s];
#ifdef WB_WHATIF_BIG                   // development: what the launch costs without its few largest contours (their events are dropped)
    if (P > WB_WHATIF_BIG) { if (WB_TID == 0) a.ev_n[s] = 0; return; }
#endif
    int* mi = (int*)(a.hkey + (size_t)s * a.hash_size);             // [M] ind1 of survivor m (M <= min(P, N))
    int* mj = mi + (P < N ? P : N) + 1;                              // [M] ind2
    int* lab = mj + (P < N ? P : N) + 1;                             // [M] group label
    int* aux = lab + (P < N ? P : N) + 1;                            // [M] scratch of the group scan
    int* bk = aux + (P < N ? P : N) + 1;                             // [P] ind2 of the alive candidates, bucketed by row (the table's 8 * hash_size bytes hold 16 (min(P, N) + 1) + 4 P: hash_size >= 2 cap_pairs and >= 4 N)
    // the contour's points and per-row maxima in LDS, sized by the launch for the batch's longest contour (a.f234_cap points,
    // at most WB_F234_LPTS: with a fixed 4096 the kernel held three workgroups per CU, sized to a 0.25 degree contour four)
    WB_DYN_SHARED_BEGIN
    WB_DYN_SHARED(uint32_t, lpts, a.f234_cap);
    // lwork: per row the largest ind2 among its alive candidates (lrmax), then per row its alive candidates / the end of its bucket in
    // bk (rend) -- once the buckets stand (their ends go to global memory) the sweep's lists, then routine 4's arrays; before all
    // that, as one piece, routine 1's table of the western pairs
    const int capE = (a.f234_cap + 1) & ~1, scrE = a.f234_cap > WB_F234_SCR ? capE : WB_F234_SCR;
    WB_DYN_SHARED(int, lwork, capE + scrE);
    int* const lrmax = lwork;
    int* const rend = lwork + capE;
    WB_DYN_SHARED(int, lbb, 4 * (a.f234_cap / WB_BLK + 1));   // boxes of the blocks of WB_BLK segments (wb_k_streamer_prep: points 32 b .. 32 b + 32)
    WB_SHARED(uint32_t, tbits, WB_F234_LPTS / 16);   // two bits per row: the chord to its present bound rmax[r] has been tested / meets the contour
    WB_SHARED(int, items, WB_F234_ITEMS);        // (chord, block) pairs whose boxes meet
    WB_SHARED(uint32_t, hitw, WB_F234_CC / 32);  // a bit per chord of the round: it meets the contour
    WB_SHARED(int, fl, 4 * WB_F234_FMAX);        // the failing records of a round: place in the list, floor, candidates in range, the largest of them
    int* const lr4 = rend;                                                        // [4 R4] the sweep's record list and its bookkeeping, then routine 4's arrays
    unsigned long long* const lbest = (unsigned long long*)(rend + 4 * WB_F234_R4);   // [3 R4] the sweep's chords, then routine 4's best length / member per group
    int* const grend = a.rowaux + a.poff[s];     // the buckets' ends, for the sweep
    WB_SHARED(int, sh, 15);              // [2] survivors, [3..4] row found, [5..6] hit flag, [7..8] change flag, [9] next candidate, [10..11] blocks met by a chord, [12] by a batch of chords
                                         // (words alternate between two slots: a slot is rewritten two barriers after its last read)
    const bool in_lds = N <= a.f234_cap;
    const uint32_t* pts = in_lds ? lpts : gpts;
    const uint32_t pe0 = gpts[0], pe1 = gpts[N - 1];          // the contour's end vertices: its boundary, as a point set (wb_chord_hits_segment)
    int* rmax = in_lds ? lrmax : a.rowmx + a.poff[s];
    WB_STAMP_BEGIN
    const int nblk = (N + WB_BLK - 1) / WB_BLK;
    if (in_lds) {
        WB_PFOR(i, N)
            lpts[i] = gpts[i];
            if (i < 4 * nblk) lbb[i] = a.bbx[4 * ((a.poff[s] >> 5) + s) + i];
        WB_END_PFOR
    }
    if (a.f1_fused) {
        // routine 1 (check_duplicates) here, on the contour just loaded: its table of the western pairs lies where the row maxima
        // and buckets will (they are built from its result), a power of two of 64-bit entries; a contour with more western pairs
        // than half of that takes the table in global memory, as in wb_k_streamer_f1
        int lcap = 256;
        while (2 * lcap <= (capE + scrE) / 2) lcap <<= 1;
        {
#if !defined(WB_HOSTSIM) && defined(WB_STAMPS) && WB_STAMPS == 9
        long long wb_acc_[12] = {0}; long long wb_st_ = 0;
#endif
        (void)wb_streamer_f1_body(a, s, job, pts, (unsigned long long*)lwork, lcap, false WB_DSTAMP_PASS);
        }
        WB_SYNC();
    }
    const int stage = a.stage[s];
    const int nal = a.nalive[s];
    WB_PFOR(i, N)
        if (in_lds) rend[i] = 0;
        if (in_lds && i < (N + 15) / 16) tbits[i] = 0u;
        rmax[i] = -1;
    WB_END_PFOR
    WB_SINGLE
    sh[2] = 0; sh[9] = 0;
    sh[3] = in_lds ? 0 : 0x7FFFFFFF; sh[4] = sh[3];   // (global-memory sweep) search results of even / odd iterations
    sh[5] = 0; sh[6] = 0;                       // hit flags of even / odd chord tests
    sh[7] = 0; sh[8] = 0;                       // routine 4: change flags of even / odd passes
    sh[10] = 0; sh[11] = 0; sh[12] = 0;         // blocks met by even / odd chord tests, by a batch
    sh[13] = 0; sh[14] = 0;
    WB_END_SINGLE
    WB_STAMP(a.ctl, 0);
    if (stage == 1 && nal > 1) {
        // per row: the largest ind2 among the alive candidates; rows that hold one
        // (four pairs per lane and step, their flags and values requested together: the contour with 50 000 pairs)
        WB_PFOR(g4, (P + 3) / 4)
            int al[4];
            unsigned long long pr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pp = 4 * (int)g4 + u, pc = pp < P ? pp : P - 1;
                const int av = alive[pc];                      // (unconditional: a load behind a branch is waited for on the spot)
                al[u] = pp < P ? av : 0;
                pr[u] = pairs[pc];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (al[u]) {
                    const int i = (int)(pr[u] >> 32), j = (int)(pr[u] & 0xFFFFFFFFull);
                    wb_atomic_max(&rmax[i], j);
                    if (in_lds) wb_atomic_add(&rend[i], 1);
                }
            }
        WB_END_PFOR
        // the candidates bucketed by row: a row whose largest candidate fails routine 2 looks its other candidates up here
        // instead of scanning all P pairs (one contour of the 0.25 degree bench batch holds 50 000 pairs and 60 such rows:
        // it alone set the kernel's duration)
        if (in_lds) {
            wb_block_scan_excl(rend, N);
            WB_PFOR(g4, (P + 3) / 4)
                int al[4];
                unsigned long long pr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pp = 4 * (int)g4 + u, pc = pp < P ? pp : P - 1;
                    const int av = alive[pc];                      // (unconditional: a load behind a branch is waited for on the spot)
                al[u] = pp < P ? av : 0;
                    pr[u] = pairs[pc];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (al[u]) bk[wb_atomic_add(&rend[(int)(pr[u] >> 32)], 1)] = (int)(pr[u] & 0xFFFFFFFFull);
            WB_END_PFOR
            WB_PFOR(i, N)
                grend[i] = rend[i];
            WB_END_PFOR
        }
        WB_STAMP(a.ctl, 1);
        // The sweep (check_overlapping over the pairs that pass check_intersections).  Going through the rows in ascending order
        // with M = the largest ind2 kept so far, a row contributes the largest of its candidates above M whose chord passes
        // routine 2.  Whether a chord passes does not depend on M: with F[r] = the largest PASSING candidate of row r, the rows
        // that contribute are exactly the records of F (F[r] > F[r'] for every r' < r).  rmax[r] starts as the row's largest
        // candidate, an upper bound of F[r], and is only ever lowered towards it; one round
        //   * lists the records of rmax (wb_block_records) -- the only rows that can matter under the present bounds,
        //   * tests the chords of the records not tested yet, and for a record whose chord is known to meet the contour its next
        //     WB_F234_K candidates (below its bound, above the last passing record in front of it), all in ONE sweep,
        //   * lowers the failing records' bounds to their first passing candidate (or to the last one tested, or to "none"),
        // and the rounds end when every record's chord has passed: they are the survivors, in row order.  A chord is tested only
        // if the row-after-row sweep would test it too (bounds in front of it that are too high can only hide a row), but the
        // rows' tests are no longer one after the other: a contour takes a handful of rounds of nine barriers, whatever the
        // number of its rows -- the row-after-row sweep took 18 rows x 6 barriers on the average contour of the 0.25 degree bench
        // batch and 61 rows, 31 of them failing with a dozen candidates each, on the one that set the kernel's duration.
        int nsurv = 0;
        int M = -1, cur = 0;
        if (in_lds) {
            int* const chain = lr4;                              // [CH] the records of rmax[] above M from row cur on
            int* const kslot = lr4 + WB_F234_CH;                 // [CH] the record's own chord in this round's list / its place among the failing records
            int* const kcnt = lr4 + 2 * WB_F234_CH;              // [CH] -1: the record's own chord is tested (not tested before); 1: a failing record that has its candidates tested
            uint32_t* const cp = (uint32_t*)lbest;               // [CC] the chords of the round: near end,
            uint32_t* const cq = cp + WB_F234_CC;                // far end,
            int* const cj = (int*)(cq + WB_F234_CC);             // ind2 (-1: no chord)
            int* const stage = items;                            // [FMAX][FSTAGE] a failing record's candidates in range (before the items are listed)
            const uint32_t TB_TESTED = 1u, TB_FAILED = 2u;
            auto emit = [&](int nc) WB_LAMBDA_INLINE {           // every record's chord has passed: the survivors of rows cur ..
                WB_PFOR32(c, nc)
                    mi[nsurv + c] = chain[c];
                    mj[nsurv + c] = rmax[chain[c]];
                WB_END_PFOR
                nsurv += nc;
            };
            // counters of a round: sh[9] own chords, sh[10] records not yet passed, sh[12] / sh[13] items of even / odd groups,
            // sh[3 + par] failing records, sh[5 + par] "a record's own chord met the contour" (par: the round's parity); all zero
            // at the start of a round -- cleared in the verdict pass of the round before, two barriers after their last use
            for (int round = 0;; ++round) {
                const int nfw = 3 + (round & 1), hfw = 5 + (round & 1);
                const int nc = wb_block_records(rmax, cur, N, M, chain, WB_F234_CH);
                WB_STAMP(a.ctl, 4);
                if (nc == 0) break;
                WB_PFOR32(c, nc)
                    const int r = chain[c], u = rmax[r];
                    const uint32_t st = (tbits[r >> 4] >> (2 * (r & 15))) & 3u;
                    kcnt[c] = 0;
                    if (c < WB_F234_CC / 32) hitw[c] = 0u;           // (a round of nc records tests at most 8 nc chords)
                    if (!(st & TB_TESTED)) {
                        wb_atomic_add(&sh[10], 1);
                        const int slot = wb_atomic_add(&sh[9], 1);      // (at most CH of them: they always find room)
                        cp[slot] = pts[r]; cq[slot] = pts[u]; cj[slot] = u; kslot[c] = slot; kcnt[c] = -1;
                    } else if (st & TB_FAILED) {
                        wb_atomic_add(&sh[10], 1);
                        const int f = wb_atomic_add(&sh[nfw], 1);
                        if (f < WB_F234_FMAX) {
                            // the last record in front of this one whose chord has passed: its ind2 is final, candidates up to it are moot
                            int floor = M;
                            for (int b = (int)c - 1; b >= 0; --b) {
                                const int rb = chain[b];
                                const uint32_t sb = (tbits[rb >> 4] >> (2 * (rb & 15))) & 3u;
                                if (sb == TB_TESTED) { floor = rmax[rb]; break; }
                            }
                            kslot[c] = f; kcnt[c] = 1;
                            fl[f] = (int)c; fl[WB_F234_FMAX + f] = floor; fl[2 * WB_F234_FMAX + f] = 0; fl[3 * WB_F234_FMAX + f] = -1;
                        }                                            // (more failing records than a round takes: the next round)
                    }
                WB_END_PFOR
                WB_STAMP(a.ctl, 5);
                if (sh[10] == 0) {
                    emit(nc);
                    if (nc < WB_F234_CH) break;
                    cur = chain[nc - 1] + 1;                       // (more records than the list holds: on behind its last one)
                    M = rmax[chain[nc - 1]];
                    if (WB_TID == 0) { sh[7 - nfw] = 0; sh[11 - hfw] = 0; }   // (no verdict pass in this round)
                    WB_SYNC();
                    continue;
                }
                const int nu = sh[9], nf = sh[nfw] < WB_F234_FMAX ? sh[nfw] : WB_F234_FMAX;
                if (nf > 0) {
                    // a failing record's candidates in (floor, bound): 64 lanes go through its bucket, the candidates in range
                    // meet in LDS; the WB_F234_K largest of them are this round's chords of the row (rank by counting)
                    WB_PFOR32(t, nf * 64)
                        const int f = t >> 6, c = fl[f], r = chain[c], u = rmax[r], floor = fl[WB_F234_FMAX + f];
                        const int lo = r ? grend[r - 1] : 0, hi = grend[r];
                        for (int e = lo + (t & 63); e < hi; e += 64) {
                            const int j = bk[e];
                            if (j > floor && j < u) {
                                const int slot = wb_atomic_add(&fl[2 * WB_F234_FMAX + f], 1);
                                if (slot < WB_F234_FSTAGE) stage[f * WB_F234_FSTAGE + slot] = j;
                                wb_atomic_max(&fl[3 * WB_F234_FMAX + f], j);
                            }
                        }
                    WB_END_PFOR
                    WB_PFOR32(t, nf * WB_F234_FSTAGE)
                        const int f = t / WB_F234_FSTAGE, i = t % WB_F234_FSTAGE, all = fl[2 * WB_F234_FMAX + f];
                        const int r = chain[fl[f]], base = nu + f * WB_F234_K;
                        if (all > WB_F234_FSTAGE) {
                            // (more candidates in range than the staging holds: the largest one alone)
                            if (i == 0) { cp[base] = pts[r]; cq[base] = pts[fl[3 * WB_F234_FMAX + f]]; cj[base] = fl[3 * WB_F234_FMAX + f]; }
                            else if (i < WB_F234_K) cj[base + i] = -1;
                        } else {
                            if (i < all) {
                                const int j = stage[f * WB_F234_FSTAGE + i];
                                int rk = 0;
                                for (int q = 0; q < all; ++q) rk += stage[f * WB_F234_FSTAGE + q] > j ? 1 : 0;
                                if (rk < WB_F234_K) { cp[base + rk] = pts[r]; cq[base + rk] = pts[j]; cj[base + rk] = j; }
                            } else if (i < WB_F234_K) cj[base + i] = -1;     // (fewer than K candidates: ranks all .. K - 1 stay empty)
                        }
                    WB_END_PFOR
                }
                const int nch = nu + nf * WB_F234_K;
                WB_STAMP_COUNT(9, nch);                            // (development counters of the longest workgroup: chord slots, rounds)
                WB_STAMP_COUNT(11, 1);
                // the chords against the contour, WB_F234_GROUP at a time: the (chord, block of 32 segments) items whose boxes meet --
                // a lane per chord and block --, then a lane per item and segment: box, then the exact predicate.  hitw: a bit per
                // chord that meets the contour.
                auto seg_test = [&](int c, int m) WB_LAMBDA_INLINE {
                    const uint32_t pa = pts[m], pb = pts[m + 1], pp = cp[c], pq = cq[c];
                    const int ax = wb_pt_x(pa), ay = wb_pt_y(pa), bx = wb_pt_x(pb), by = wb_pt_y(pb);
                    const int px = wb_pt_x(pp), py = wb_pt_y(pp), qx = wb_pt_x(pq), qy = wb_pt_y(pq);
                    const int cx0 = px < qx ? px : qx, cx1 = px < qx ? qx : px, cy0 = py < qy ? py : qy, cy1 = py < qy ? qy : py;
                    // a segment whose box misses the chord's box cannot meet the chord
                    if (!((ax < cx0 && bx < cx0) || (ax > cx1 && bx > cx1) || (ay < cy0 && by < cy0) || (ay > cy1 && by > cy1)))
                        if (wb_chord_hits_segment(px, py, qx, qy, ax, ay, bx, by, pe0, pe1)) wb_atomic_or(&hitw[c >> 5], 1u << (c & 31));
                };
                for (int g0 = 0; g0 < nch; g0 += WB_F234_GROUP) {
                    const int ng = nch - g0 < WB_F234_GROUP ? nch - g0 : WB_F234_GROUP;
                    const int is = 12 + ((g0 / WB_F234_GROUP) & 1);   // (item counters of even / odd groups)
                    WB_PFOR32(t, ng << 7)                             // (nblk <= 128: a chord's blocks are 128 consecutive t, no division)
                        if (t == 0) sh[25 - is] = 0;                  // the next group's counter
                        const int c = g0 + (t >> 7), b = t & 127;
                        if (b >= nblk || cj[c] < 0) continue;
                        const int px = wb_pt_x(cp[c]), py = wb_pt_y(cp[c]), qx = wb_pt_x(cq[c]), qy = wb_pt_y(cq[c]);
                        const int cx0 = px < qx ? px : qx, cx1 = px < qx ? qx : px, cy0 = py < qy ? py : qy, cy1 = py < qy ? qy : py;
                        const int* bb = lbb + 4 * b;
                        if (!(bb[0] > cx1 || bb[1] < cx0 || bb[2] > cy1 || bb[3] < cy0)) {
                            const int slot = wb_atomic_add(&sh[is], 1);
                            if (slot < WB_F234_ITEMS) items[slot] = (c << 12) | b;
                        }
                    WB_END_PFOR
                    const int nit = sh[is];
                    if (nit <= WB_F234_ITEMS) {
                        WB_PFOR32(t, nit * WB_BLK)
                            const int it = items[t >> 5], m = (it & 0xFFF) * WB_BLK + (t & (WB_BLK - 1));
                            if (m < N - 1) seg_test(it >> 12, m);
                        WB_END_PFOR
                    } else {
                        // (more items than the list holds: every chord of the group against every segment)
                        WB_PFOR32(t, ng * (N - 1))
                            const int c = g0 + t / (N - 1);
                            if (cj[c] >= 0) seg_test(c, t % (N - 1));
                        WB_END_PFOR
                    }
                }
                WB_STAMP(a.ctl, 6);
                // the verdicts: a record's own chord is marked; a failing record takes its first candidate that passes as its new
                // bound (final: every larger candidate of the row meets the contour), or the last one tested, or none
                WB_PFOR32(c, nc)
                    if (c == 0) { sh[9] = 0; sh[10] = 0; sh[12] = 0; sh[7 - nfw] = 0; sh[11 - hfw] = 0; }   // the next round's counters
                    const int r = chain[c], kind = kcnt[c];
                    if (kind == -1) {
                        const int s0 = kslot[c];
                        const bool hit = (hitw[s0 >> 5] >> (s0 & 31)) & 1u;
                        if (hit) sh[hfw] = 1;
                        wb_atomic_or(&tbits[r >> 4], (hit ? (TB_TESTED | TB_FAILED) : TB_TESTED) << (2 * (r & 15)));
                    } else if (kind == 1) {
                        const int f = kslot[c], s0 = nu + f * WB_F234_K, all = fl[2 * WB_F234_FMAX + f];
                        const int n = all > WB_F234_FSTAGE ? 1 : (all < WB_F234_K ? all : WB_F234_K);
                        int found = -1;
                        for (int q = 0; q < n && found < 0; ++q)
                            if (!((hitw[(s0 + q) >> 5] >> ((s0 + q) & 31)) & 1u)) found = cj[s0 + q];
                        if (found >= 0) {
                            rmax[r] = found;                        // passes: tested, not failed
                            wb_atomic_and(&tbits[r >> 4], ~(TB_FAILED << (2 * (r & 15))));
                        } else if (n == all) {
                            rmax[r] = -1;                           // every candidate in range meets the contour (or there was none)
                        } else {
                            rmax[r] = cj[s0 + n - 1];               // (tested, failed: on below it in the next round)
                        }
                    }
                WB_END_PFOR
                if (sh[nfw] == 0 && sh[hfw] == 0) {
                    // no record failed and none had failed before: the list stands as it is
                    emit(nc);
                    if (nc < WB_F234_CH) break;
                    cur = chain[nc - 1] + 1;
                    M = rmax[chain[nc - 1]];
                    WB_SYNC();
                }
            }
        } else {
        // (a contour longer than the LDS holds: row after row in global memory, a search for the next row and a chord test each)
        int ntest = 0;
        for (int step = 0;; ++step) {
            const int fs = 3 + (step & 1);
            const int win = WB_NT > 64 ? WB_NT : 64;
            {
                WB_PFOR(t, win)
                    const int i = cur + (int)t;
                    wb_atomic_min_first(&sh[fs], i < N && rmax[i] > M, i);
                WB_END_PFOR
            }
            if (sh[fs] == 0x7FFFFFFF) {
                for (int base = cur + win; base < N; base += win) {
                    WB_PFOR(t, win)
                        const int i = base + (int)t;
                        wb_atomic_min_first(&sh[fs], i < N && rmax[i] > M, i);
                    WB_END_PFOR
                    if (sh[fs] != 0x7FFFFFFF) break;
                }
            }
            const int row = sh[fs];
            if (row == 0x7FFFFFFF) break;
            int jn = rmax[row];                 // the row's largest candidate first
            int bound = 0x7FFFFFFF;
            for (;;) {
                // routine 2 for this one chord: does the open chord meet the contour's interior? (all lanes, all segments)
                const int hf = 5 + (ntest & 1);
                ++ntest;
                const int px = wb_pt_x(pts[row]), py = wb_pt_y(pts[row]), qx = wb_pt_x(pts[jn]), qy = wb_pt_y(pts[jn]);
                const int cx0 = px < qx ? px : qx, cx1 = px < qx ? qx : px, cy0 = py < qy ? py : qy, cy1 = py < qy ? qy : py;
                WB_PFOR(m, N - 1)
                    if (m == 0) { sh[3 + ((step + 1) & 1)] = 0x7FFFFFFF; sh[5 + (ntest & 1)] = 0; }   // the next step's / test's slots
                    const uint32_t pa = pts[m], pb = pts[m + 1];
                    const int ax = wb_pt_x(pa), ay = wb_pt_y(pa), bx = wb_pt_x(pb), by = wb_pt_y(pb);
                    // a segment whose box misses the chord's box cannot meet the chord
                    if (!((ax < cx0 && bx < cx0) || (ax > cx1 && bx > cx1) || (ay < cy0 && by < cy0) || (ay > cy1 && by > cy1)))
                        if (wb_chord_hits_segment(px, py, qx, qy, ax, ay, bx, by, pe0, pe1)) sh[hf] = 1;
                WB_END_PFOR
                if (!sh[hf]) {                             // passes routine 2: the row's survivor
                    if (WB_TID == 0) { mi[nsurv] = row; mj[nsurv] = jn; }
                    ++nsurv;
                    M = jn;
                    break;
                }
                // the chord meets the contour: the row's largest candidate in (M, jn), one per round
                bound = jn;
                WB_SINGLE
                sh[9] = -1;
                WB_END_SINGLE
                WB_PFOR(p, P)
                    if (alive[p] && (int)(pairs[p] >> 32) == row) {
                        const int j = (int)(pairs[p] & 0xFFFFFFFFull);
                        if (j > M && j < bound) wb_atomic_max(&sh[9], j);
                    }
                WB_END_PFOR
                jn = sh[9];
                WB_SYNC();
                if (jn < 0) break;                         // every candidate of the row failed
            }
            cur = row + 1;
        }
        }
        WB_SINGLE
        sh[2] = nsurv;
        WB_END_SINGLE
    } else {
        // routine 2 never ran (`while len(df_bp.index) > 1`): at most one pair is left
        WB_PFOR(p, P)
            if (alive[p] == 1) {
                mi[0] = (int)(pairs[p] >> 32);
                mj[0] = (int)(pairs[p] & 0xFFFFFFFFull);
                sh[2] = 1;
            }
        WB_END_PFOR
    }
    const int M = sh[2];
    WB_STAMP(a.ctl, 7);

    // routine 4: group chords that intersect; keep the longest of each group.  Its arrays -- the survivors' end points, labels,
    // group scan, best length / member per group, every member's length -- are a dozen barrier-separated phases over M entries
    // each: in the contour's tables in global memory every phase was a round trip to L2 (22 us per contour, 15 % of the
    // kernel); up to WB_F234_R4 survivors they live in LDS and go back to the tables once, for wb_k_streamer_ties.
    auto groups = [&](const int* mi, const int* mj, int* lab, int* aux, unsigned long long* best) WB_LAMBDA_INLINE -> bool {
        WB_PFOR(m, M)
            lab[m] = (int)m;
            best[m] = 0ull;
            best[M + m] = WB_HEMPTY;
            aux[m] = 0;
        WB_END_PFOR
        bool apart = false;                                        // no two survivors' chords intersect
        for (int iter = 0; iter < M; ++iter) {
            const int cf = 7 + (iter & 1);                        // (change flags of even / odd passes, both zero at the start)
            WB_PFOR(pair, (int64_t)M * M)
                if (pair == 0) sh[15 - cf] = 0;                     // the next pass's flag
                const int m1 = (int)(pair / M), m2 = (int)(pair - (int64_t)m1 * M);
                if (m1 < m2 && lab[m1] != lab[m2]) {
                    const uint32_t a1 = pts[mi[m1]], b1 = pts[mj[m1]], a2 = pts[mi[m2]], b2 = pts[mj[m2]];
                    if (wb_segments_intersect(wb_pt_x(a1), wb_pt_y(a1), wb_pt_x(b1), wb_pt_y(b1), wb_pt_x(a2), wb_pt_y(a2),
                                              wb_pt_x(b2), wb_pt_y(b2))) {
                        // hook the larger root under the smaller one (labels are roots after the shortcut pass)
                        const int l1 = lab[m1], l2 = lab[m2];
                        if (l1 != l2) {
                            wb_atomic_min(&lab[l1 > l2 ? l1 : l2], l1 < l2 ? l1 : l2);
                            sh[cf] = 1;
                        }
                    }
                }
            WB_END_PFOR
            if (!sh[cf]) { apart = iter == 0; break; }            // (nothing hooked in this pass: the labels are roots already)
            WB_PFOR(m, M)   // shortcut labels to their root
                int l = lab[m];
                while (lab[l] != l) l = lab[l];
                lab[m] = l;
            WB_END_PFOR
        }
        if (apart) {
            // the usual case: every survivor is a group of its own -- it stays, no length is compared (the passes below would find
            // just that, in a dozen more barriers)
            WB_PFOR(m, M)
                if (m < a.cap_ev) { ev[2 * m + 0] = mi[m]; ev[2 * m + 1] = mj[m]; }
            WB_END_PFOR
            WB_SINGLE
            a.ev_n[s] = M < a.cap_ev ? M : a.cap_ev;
            if (M > a.cap_ev) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_OVERFLOW_PAIRS);
            WB_END_SINGLE
            return true;
        }
        // per group: max of on[ind1 : ind2+1].sum() -- numpy's pairwise sum of the slice, not a prefix difference: two chords
        // shifted by one point along a lattice contour differ by on[ind1] - on[ind2 + 1], two steps of the same kind whose
        // lengths agree to the last bit or the one before, and the reference's argmax (streamer_index.py:241-246) then
        // turns on how the slices were added; ties -> lowest member
        WB_PFOR(m, M)
            wb_atomic_add(&aux[lab[m]], 1);                        // members per group: a group of one needs no length
        WB_END_PFOR
        const double* onp = a.on + a.poff[s];
        unsigned long long* gval = best + 2 * (size_t)M;          // (the table holds 4 * cap_seg >= 4 M words)
        WB_PFOR(m, M)
            const double v = aux[lab[m]] > 1 ? wb_np_pairwise_sum(onp + mi[m], mj[m] - mi[m] + 1) : 0.0;
            unsigned long long bits;
            memcpy(&bits, &v, 8);
            gval[m] = bits;
            wb_atomic_max(&best[lab[m]], bits);
        WB_END_PFOR
        WB_PFOR(m, M)
            if (gval[m] == best[lab[m]]) wb_atomic_min(&best[M + lab[m]], (unsigned long long)m);
        WB_END_PFOR
        // several members with the SAME length (to the last bit): the reference keeps the one that comes first in a CPython
        // set's iteration order -- resolved by wb_k_streamer_ties after this kernel (rare; kept out of this kernel's
        // registers); here the lowest such member stands in and the contour is marked.  Then the group roots, in order of
        // their lowest member
        WB_PFOR(m, M)
            if (gval[m] == best[lab[m]] && (int)best[M + lab[m]] != (int)m) { a.tie_m[s] = M; a.ctl->tie_marks = 1; }
            aux[m] = (lab[m] == (int)m) ? 1 : 0;
        WB_END_PFOR
        int Gn;
        if (M <= WB_F234_R4) {
            // (a few dozen survivors in LDS: a root counts the roots in front of it -- a block scan is four barriers)
            WB_PFOR(m, M)
                int pos = 0;
                if (lab[m] == (int)m) {
                    for (int q = 0; q < (int)m; ++q) pos += aux[q];
                    if (pos < a.cap_ev) {
                        const int w = (int)best[M + m];
                        ev[2 * pos + 0] = mi[w];
                        ev[2 * pos + 1] = mj[w];
                    }
                }
                items[m] = pos;                                    // (the group's place among the events: wb_k_streamer_ties reads it from the table)
            WB_END_PFOR
            Gn = 0;
            for (int q = 0; q < M; ++q) Gn += aux[q];
        } else {
            Gn = wb_block_scan_excl(aux, M);
            WB_PFOR(m, M)
                if (lab[m] == (int)m && aux[m] < a.cap_ev) {
                    const int w = (int)best[M + m];
                    ev[2 * aux[m] + 0] = mi[w];
                    ev[2 * aux[m] + 1] = mj[w];
                }
            WB_END_PFOR
        }
        WB_SINGLE
        a.ev_n[s] = Gn < a.cap_ev ? Gn : a.cap_ev;
        if (Gn > a.cap_ev) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_OVERFLOW_PAIRS);
        WB_END_SINGLE
        return false;
    };
    if (M > 1 && M <= WB_F234_R4) {
        int *lmi = lr4, *lmj = lr4 + WB_F234_R4, *llab = lr4 + 2 * WB_F234_R4, *laux = lr4 + 3 * WB_F234_R4;
        WB_PFOR(m, M)
            lmi[m] = mi[m];
            lmj[m] = mj[m];
        WB_END_PFOR
        if (!groups(lmi, lmj, llab, laux, lbest)) {             // (the tables are read again by wb_k_streamer_ties, for groups only)
        WB_PFOR(m, M)
            lab[m] = llab[m];
            aux[m] = items[m];
            best[m] = lbest[m];
            best[M + m] = lbest[M + m];
            best[2 * (size_t)M + m] = lbest[2 * (size_t)M + m];
        WB_END_PFOR
        }
    } else if (M > 1) {
        (void)groups(mi, mj, lab, aux, best);
    } else {
        WB_SINGLE
        if (M == 1) { ev[0] = mi[0]; ev[1] = mj[0]; }
        a.ev_n[s] = M;
        WB_END_SINGLE
    }
    WB_STAMP(a.ctl, 8);
    WB_STAMP_FLUSH(a.ctl);
}

// ------------------------------------------------------------------------------------------------
// work lists and offsets, built on the device (single workgroups: a few thousand entries each)

// lengths of the selected contours -> packed point offsets, work items of the all-pairs kernel
// check_groups, exact ties (after wb_k_streamer_f234 / f34): for the contours marked there, the survivor of every group whose
// largest length is shared by several members becomes the member that comes first in the reference's CPython set
// (wb_group_tie_winner), and the event's end points are rewritten.  ONE wave looks through the marks (nearly always none;
// as a grid over the contours the launch queued behind the other batch's kernels for its 46 workgroups).
WB_DEV inline void wb_streamer_resolve_ties(const WbStreamerArgs& a, int s) {
    const int M = a.tie_m[s];
    if (M < 2) return;
    const int job = a.sel_job[s], k = a.sel_cont[s];
    const size_t jb = (size_t)job * a.cap_seg;
    const uint32_t* pts = a.cpts + 2 * jb + a.c_off[jb + k];
    const int N = a.c_len[jb + k];
    const int P = a.npairs[s] > a.cap_pairs ? 0 : a.npairs[s];
    unsigned long long* best = a.hval + (size_t)s * a.hash_size;
    const int* mi; const int* mj; const int* lab; const int* aux;
    if (a.eager) {
        mi = (int*)a.hkey + (size_t)s * a.hash_size * 2; mj = mi + M; lab = mj + M; aux = a.rowaux + a.poff[s];
    } else {
        const int w = (P < N ? P : N) + 1;
        mi = (int*)(a.hkey + (size_t)s * a.hash_size); mj = mi + w; lab = mj + w; aux = lab + w;
    }
    const unsigned long long* gval = best + 2 * (size_t)M;
    int* ev = a.ev_ind + (size_t)s * a.cap_ev * 2;
    for (int g = 0; g < M; ++g) {
        if (lab[g] != g) continue;
        int ntie = 0;
        for (int m = 0; m < M; ++m) if (lab[m] == g && gval[m] == best[g]) ++ntie;
        if (ntie < 2) continue;
        wb_atomic_add(&a.ctl->group_ties, 1);
        const int w = wb_group_tie_winner(lab, M, g, pts, mi, mj, gval, best[g], best + 3 * (size_t)M, (long long)a.hash_size - 3LL * M);
        if (w >= 0 && aux[g] < a.cap_ev) { ev[2 * aux[g] + 0] = mi[w]; ev[2 * aux[g] + 1] = mj[w]; }
    }
}

WB_KERNEL(wb_k_streamer_ties, WbStreamerArgs a) {
    if (!a.ctl->tie_marks) return;                  // (nearly always: no contour of the pass was marked -- going through their marks took 17 us)
    const int nsel = a.ctl->nsel;
    for (int s = WB_BLOCK_ID * WB_NT + WB_TID; s < nsel; s += WB_GRID_DIM * WB_NT) wb_streamer_resolve_ties(a, s);
}

WB_KERNEL(wb_k_sel_scan, WbStreamerArgs a, int* tmp) {
    const int nsel = a.ctl->nsel;
    WB_SHARED(unsigned long long, pp, 1);
    WB_SINGLE
    pp[0] = 0ull;
    WB_END_SINGLE
    unsigned long long mine = 0ull;                      // (a thread's share; hostsim: the one thread's total)
    WB_PFOR(s, nsel)
        const int len = a.c_len[(size_t)a.sel_job[s] * a.cap_seg + a.sel_cont[s]];
        tmp[s] = len;
        mine += (unsigned long long)((long long)len * (len - 1) / 2);                   // base-point pairs the all-pairs test covers
        const int tiles = (len - 1 + WB_PAIR_ROWS - 1) / WB_PAIR_ROWS;
        a.item_off[s] = len <= a.p2_cap ? 0 : (tiles + WB_PAIR_TPB - 1) / WB_PAIR_TPB;   // (short enough for wb_k_streamer_pairs2: no items)
    WB_END_PFOR
    wb_block_add_u64(&pp[0], mine);                      // (one atomic per wave: 2 920 lanes adding to one LDS word took 0.2 ms)
    const int npt = wb_block_scan_excl(tmp, nsel);
    const int nit = wb_block_scan_excl(a.item_off, nsel);
    WB_PFOR(s, nsel + 1)
        a.poff[s] = s < nsel ? (long long)tmp[s] : (long long)npt;
        if (s == nsel) a.item_off[s] = nit;
    WB_END_PFOR
    WB_SINGLE
    a.ctl->npt_sel = npt;
    a.ctl->n_pair_items = nit;
    a.ctl->point_pairs = (long long)pp[0];
    WB_END_SINGLE
}
WB_KERNEL(wb_k_pair_items, WbStreamerArgs a) {
    const int s = WB_BLOCK_ID;
    if (s >= a.ctl->nsel) return;
    const int first = a.item_off[s], n = a.item_off[s + 1] - first;
    WB_PFOR(k, n)
        a.item_sel[first + k] = s;
        a.item_tile[first + k] = (int)k * WB_PAIR_TPB;
    WB_END_PFOR
}
// routine 2's work list: 256-pair blocks of the contours that still hold more than one candidate (streamer_index.py:261-264)
WB_KERNEL(wb_k_f2_items, WbStreamerArgs a) {
    const int nsel = a.ctl->nsel;
    WB_PFOR(s, nsel)
        const int P = a.npairs[s] < a.cap_pairs ? a.npairs[s] : a.cap_pairs;
        a.f2_off[s] = P > 1 ? (P + WB_F2_THREADS - 1) / WB_F2_THREADS : 0;
    WB_END_PFOR
    const int n = wb_block_scan_excl(a.f2_off, nsel);
    WB_SINGLE
    a.f2_off[nsel] = n;
    a.ctl->n_f2_items = n;
    WB_END_SINGLE
}

#include "wb_pairs.h"
