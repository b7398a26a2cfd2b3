// wb_geom.h -- transform_polygons (wavebreaking/utils/index_utils.py:129-184) on the device: the event polygons of a pass in
// the data's own coordinates, written by the device into pinned host memory beside the event tables (round 5; until then
// wb_geometry.h did this on the host, one thread, 17 ms per 1 460-step pass -- four times the whole GPU pass).
//
// An event's polygon is a ring of index-lattice vertices in the periodically extended grid.  The reference maps it to
// (lon[x % nlon], lat[y]) (:139-146) and, when it reaches x >= nlon, first removes the strip nlon - 1 < x < nlon between the last
// meridian and its periodic copy (:148-173; GEOS linemerge / unary_union / polygonize / make_valid there -- unpinned here, see
// oracle/properties.py:split_rings_exact for the independent exact-arithmetic statement the tests compare ring sets with),
// keeping what lies on either side as separate rings, vertices truncated to the lattice (:135).
//
// Same statement as wb_geometry.h:wb_event_rings, ring for ring and bit for bit (tests/test_seam_split.py,
// tests/test_gpu_parity.py::test_device_geometry_equals_host_geometry): every vertex lies on exactly one side of the strip
// (x <= nlon - 1 or x >= nlon), the edges that change sides are cut on the side's meridian, a side's boundary chains are joined
// along the meridian by pairing the cut points sorted by y, pieces without area are dropped.  One wave per event: the cut edges
// (a handful) are sorted and paired by rank counting, a piece's vertices are written by all lanes.  Three launches per pass:
// count (rings and vertices per event) -> offsets (one workgroup) -> fill; an event with more than WB_GEOM_XMAX cut edges is
// flagged and done by the host function.
#pragma once
#include "wb_props.h"

#define WB_GEOM_XMAX 64
#define WB_GEOM_THREADS 64

struct WbGeomSrc {
    const int* d;            // event descriptor + contour points (events of a batch) ...
    const uint32_t* pts;
    const int32_t* px;       // ... or explicit vertex arrays (wb_transform_polygons_device)
    const int32_t* py;
};
WB_DEV inline void wb_geom_vertex(const WbGeomSrc& s, int k, double* x, double* y) {
    if (s.px) { *x = (double)s.px[k]; *y = (double)s.py[k]; return; }
    const uint32_t v = wb_event_vertex(s.d, s.pts, k);
    *x = (double)wb_pt_x(v); *y = (double)wb_pt_y(v);
}

struct WbGeomArgs {
    long long nev;                      // events, or -1: ctl->NE
    const WbCtl* ctl;
    WbCtl* ctl_out;                     // totals -> geom_rings / geom_verts, capacity flags -> overflow (bits 2, 3)
    int cap_seg, nlon, index_space;
    const int* ev; const long long* poff; const int* c_off; const uint32_t* cpts;          // batch events
    const long long* poly_offset; const int32_t* px; const int32_t* py;                       // or explicit polygons
    const double* lon; const double* lat;
    int* cnt_r; int* cnt_v;             // [nev + 1] rings / vertices per event -> exclusive scans
    int* gflag;                         // [nev] 1: more cut edges than the kernel holds (host fallback)
    long long cap_r, cap_v;
    long long* ring_offset;             // [nev + 1]   pinned host (or device, wb_transform_polygons_device)
    long long* vert_offset;             // [rings + 1]
    double* x; double* y;               // [vertices]
    long long* tot_out;                 // NULL, or [4] in pinned host memory <- rings, vertices, 1 if they do not fit cap_r / cap_v, flagged events: a
                                        // pass whose geometry kernels wb_indices does not wait for (option geometry_async) is judged from these
};

// one event by one single-wave workgroup; fill = false: counts only
WB_DEV inline void wb_geom_event(const WbGeomSrc& src, const int n, const int nlon, const double* lon, const double* lat, const int index_space,
                                 const bool fill, const long long ring0, const long long vert0, long long* vert_offset, double* ox, double* oy,
                                 int* out_nr, int* out_nv, int* out_flag) {
    WB_SHARED(int, gc, 8);              // 0 vertices with x >= nlon, 1 first of them, 2 first with x < nlon, 3 cut edges, 7 scratch
    WB_SHARED(int, xk, WB_GEOM_XMAX);   // cut edges (edge k runs from vertex k to k + 1), as found / ascending
    WB_SHARED(int, xs, WB_GEOM_XMAX);
    WB_SHARED(double, ly, WB_GEOM_XMAX);   // a side's cut points: y of chain i's entry (2 i) and exit (2 i + 1)
    WB_SHARED(int, lr, WB_GEOM_XMAX);
    WB_SHARED(int, linv, WB_GEOM_XMAX);
    WB_SHARED(int, cke, WB_GEOM_XMAX / 2);
    WB_SHARED(int, ckl, WB_GEOM_XMAX / 2);
    WB_SHARED(int, cnext, WB_GEOM_XMAX / 2);
    WB_SHARED(int, pchain, WB_GEOM_XMAX / 2);        // chains in piece order
    WB_SHARED(int, pstart, WB_GEOM_XMAX / 2 + 2);    // first entry of a piece in pchain
    WB_SHARED(int, cofs, WB_GEOM_XMAX / 2 + 2);      // first point of a chain inside its piece
    WB_SHARED(double, lw, 16);
    WB_SHARED(double, seq, 2);
    auto emit = [&](const long long at, const double px_, const double py_) {
        if (index_space) { ox[at] = px_; oy[at] = py_; return; }
        const long long xi = (long long)px_, yi = (long long)py_;                  // :135 truncation
        ox[at] = lon[((xi % nlon) + nlon) % nlon];                                    // :139-146
        oy[at] = lat[yi];
    };
    WB_SINGLE
    gc[0] = 0; gc[1] = n; gc[2] = n; gc[3] = 0;
    WB_END_SINGLE
    WB_PFOR(k, n)
        double x, y;
        wb_geom_vertex(src, (int)k, &x, &y);
        if (x >= (double)nlon) { wb_atomic_add(&gc[0], 1); wb_atomic_min(&gc[1], (int)k); }
        else wb_atomic_min(&gc[2], (int)k);
    WB_END_PFOR
    const int nB = gc[0];
    int nr = 0, nv = 0;
    if (nB == 0 || nB == n) {
        // one ring, vertex for vertex (index_utils.py:139-146; all of it beyond the last meridian: the cut leaves it whole)
        if (fill) {
            WB_SINGLE
            vert_offset[ring0] = vert0;
            WB_END_SINGLE
            WB_PFOR(k, n)
                double x, y;
                wb_geom_vertex(src, (int)k, &x, &y);
                emit(vert0 + k, x, y);
            WB_END_PFOR
        }
        *out_nr = 1; *out_nv = n; *out_flag = 0;
        return;
    }
    // the edges that change sides
    WB_PFOR(k, n)
        double xa, ya, xb, yb;
        wb_geom_vertex(src, (int)k, &xa, &ya);
        wb_geom_vertex(src, (int)((k + 1) % n), &xb, &yb);
        const bool cut = (xa >= (double)nlon) != (xb >= (double)nlon);
        const int slot = wb_append_slot(&gc[3], cut);
        if (cut && slot < WB_GEOM_XMAX) xk[slot] = (int)k;
    WB_END_PFOR
    const int m = gc[3];
    if (m > WB_GEOM_XMAX) { *out_nr = 0; *out_nv = 0; *out_flag = 1; return; }
    WB_PFOR(i, m)
        int r = 0;
        for (int j = 0; j < m; ++j) r += xk[j] < xk[i] ? 1 : 0;
        xs[r] = xk[i];
    WB_END_PFOR
    const int nc = m / 2;
    for (int side = 0; side < 2; ++side) {
        const double L = side == 0 ? (double)(nlon - 1) : (double)nlon;            // keep x <= nlon - 1, then x >= nlon
        const int s = side == 0 ? gc[1] : gc[2];                                      // the walk starts on the first OUTSIDE vertex
        int j0 = 0;
        for (int j = 0; j < m; ++j) j0 += xs[j] < s ? 1 : 0;                          // cut edges in front of it: the first one behind it enters
        WB_PFOR(i, nc)
            const int ke = xs[(j0 + 2 * (int)i) % m], kl = xs[(j0 + 2 * (int)i + 1) % m];
            cke[i] = ke; ckl[i] = kl;
            double ax, ay, bx, by;
            wb_geom_vertex(src, ke, &ax, &ay);
            wb_geom_vertex(src, (ke + 1) % n, &bx, &by);
            double t = (L - ax) / (bx - ax);
            ly[2 * i] = ay + t * (by - ay);
            wb_geom_vertex(src, kl, &ax, &ay);
            wb_geom_vertex(src, (kl + 1) % n, &bx, &by);
            t = (L - ax) / (bx - ax);
            ly[2 * i + 1] = ay + t * (by - ay);
        WB_END_PFOR
        // the cut points sorted by (y, entry before exit), stable: consecutive pairs are joined along the meridian
        WB_PFOR(q, 2 * nc)
            int r = 0;
            const double yq = ly[q];
            const int kq = (int)q & 1;
            for (int q2 = 0; q2 < 2 * nc; ++q2) {
                const double y2 = ly[q2];
                const int k2 = q2 & 1;
                r += (y2 < yq || (y2 == yq && (k2 < kq || (k2 == kq && q2 < (int)q)))) ? 1 : 0;
            }
            lr[q] = r;
            linv[r] = (int)q;
        WB_END_PFOR
        WB_PFOR(i, nc)
            cnext[i] = linv[lr[2 * i + 1] ^ 1] >> 1;                                  // the chain whose cut point is paired with this chain's exit
        WB_END_PFOR
        WB_SINGLE
        unsigned long long used = 0ull;
        int np = 0, pos = 0;
        for (int c0 = 0; c0 < nc; ++c0) {
            if ((used >> c0) & 1ull) continue;
            pstart[np++] = pos;
            int c = c0, guard = 0;
            while (!((used >> c) & 1ull) && guard <= nc) {
                used |= 1ull << c;
                pchain[pos++] = c;
                c = cnext[c];
                ++guard;
            }
        }
        pstart[np] = pos;
        gc[7] = np;
        WB_END_SINGLE
        const int np = gc[7];
        for (int pi = 0; pi < np; ++pi) {
            const int j0p = pstart[pi], j1p = pstart[pi + 1];
            WB_SINGLE
            int run = 0;
            for (int j = j0p; j < j1p; ++j) {
                const int c = pchain[j];
                cofs[j - j0p] = run;
                run += ((ckl[c] - cke[c] + n) % n) + 2;
            }
            cofs[j1p - j0p] = run;
            WB_END_SINGLE
            const int T = cofs[j1p - j0p];
            auto point = [&](const int t, double* x, double* y) {
                int j = 0;
                while (cofs[j + 1] <= t) ++j;
                const int c = pchain[j0p + j], u = t - cofs[j], len = (ckl[c] - cke[c] + n) % n;
                if (u == 0) { *x = L; *y = ly[2 * c]; }
                else if (u == len + 1) { *x = L; *y = ly[2 * c + 1]; }
                else wb_geom_vertex(src, (cke[c] + u) % n, x, y);
            };
            // twice the signed area, as wb_geometry.h adds it; a sum too close to zero to trust a parallel order is added again
            // by one lane in the host function's order
            double part = 0.0, pabs = 0.0;
            for (int t = WB_TID; t < T; t += WB_NT) {
                double px_, py_, qx_, qy_;
                point(t, &px_, &py_);
                point((t + 1) % T, &qx_, &qy_);
                const double term = px_ * qy_ - qx_ * py_;
                part += term;
                pabs += fabs(term);
            }
            double a2 = wb_block_sum_f64(part, lw);
            const double aabs = wb_block_sum_f64(pabs, lw);
            if (!(fabs(a2) > 1e-9 * aabs)) {
                WB_SINGLE
                double sacc = 0.0;
                for (int t = 0; t < T; ++t) {
                    double px_, py_, qx_, qy_;
                    point(t, &px_, &py_);
                    point((t + 1) % T, &qx_, &qy_);
                    sacc += px_ * qy_ - qx_ * py_;
                }
                seq[0] = sacc;
                WB_END_SINGLE
                a2 = seq[0];
            }
            if (T >= 3 && fabs(a2) > 0.0) {
                if (fill) {
                    WB_SINGLE
                    vert_offset[ring0 + nr] = vert0 + nv;
                    WB_END_SINGLE
                    WB_PFOR(t, T)
                        double px_, py_;
                        point((int)t, &px_, &py_);
                        emit(vert0 + nv + t, px_, py_);
                    WB_END_PFOR
                }
                nr += 1;
                nv += T;
            }
            WB_SYNC();                                     // (cofs / seq are rewritten by the next piece)
        }
    }
    *out_nr = nr; *out_nv = nv; *out_flag = 0;
}

WB_DEV inline int wb_geom_source(const WbGeomArgs& a, const long long e, WbGeomSrc* src) {
    if (a.px) {
        src->d = nullptr; src->pts = nullptr;
        src->px = a.px + a.poly_offset[e]; src->py = a.py + a.poly_offset[e];
        return (int)(a.poly_offset[e + 1] - a.poly_offset[e]);
    }
    const int* d = a.ev + (size_t)e * 8;
    const size_t jb = (size_t)d[1] * a.cap_seg;
    src->d = d; src->pts = a.cpts + 2 * jb + a.c_off[jb + d[2]];
    src->px = nullptr; src->py = nullptr;
    return (int)(a.poff[e + 1] - a.poff[e]);
}
WB_KERNEL(wb_k_geom_count, WbGeomArgs a) {
    const long long nev = a.nev >= 0 ? a.nev : (long long)a.ctl->NE;
    for (long long e = WB_BLOCK_ID; e < nev; e += WB_GRID_DIM) {
        WbGeomSrc src;
        const int n = wb_geom_source(a, e, &src);
        int nr = 0, nv = 0, fl = 0;
        wb_geom_event(src, n, a.nlon, a.lon, a.lat, a.index_space, false, 0, 0, nullptr, nullptr, nullptr, &nr, &nv, &fl);
        WB_SINGLE
        a.cnt_r[e] = nr; a.cnt_v[e] = nv; a.gflag[e] = fl;
        if (fl && a.ctl_out) wb_atomic_add(&a.ctl_out->geom_flagged, 1);
        WB_END_SINGLE
    }
}
// rings / vertices per event -> first ring / first vertex (one workgroup); the events' ring offsets go out here
WB_KERNEL(wb_k_geom_offsets, WbGeomArgs a) {
    const long long nev = a.nev >= 0 ? a.nev : (long long)a.ctl->NE;
    const int NR = wb_block_scan_excl(a.cnt_r, (int)nev);
    const int NV = wb_block_scan_excl(a.cnt_v, (int)nev);
    const bool fits = (long long)NR <= a.cap_r && (long long)NV <= a.cap_v;
    WB_PFOR(e, nev + 1)
        if (fits) a.ring_offset[e] = e < nev ? (long long)a.cnt_r[e] : (long long)NR;
    WB_END_PFOR
    WB_SINGLE
    a.cnt_r[nev] = NR; a.cnt_v[nev] = NV;
    if (fits) a.vert_offset[NR] = NV;
    if (a.ctl_out) {
        a.ctl_out->geom_rings = NR; a.ctl_out->geom_verts = NV;
        if (!fits) wb_atomic_or(&a.ctl_out->overflow, 4);
    }
    if (a.tot_out) {
        a.tot_out[1] = NV; a.tot_out[2] = fits ? 0 : 1; a.tot_out[3] = a.ctl_out ? (long long)a.ctl_out->geom_flagged : 0;
        a.tot_out[0] = NR;
    }
    WB_END_SINGLE
}
WB_KERNEL(wb_k_geom_fill, WbGeomArgs a) {
    const long long nev = a.nev >= 0 ? a.nev : (long long)a.ctl->NE;
    if ((long long)a.cnt_r[nev] > a.cap_r || (long long)a.cnt_v[nev] > a.cap_v) return;
    for (long long e = WB_BLOCK_ID; e < nev; e += WB_GRID_DIM) {
        WbGeomSrc src;
        const int n = wb_geom_source(a, e, &src);
        int nr = 0, nv = 0, fl = 0;
        wb_geom_event(src, n, a.nlon, a.lon, a.lat, a.index_space, true, (long long)a.cnt_r[e], (long long)a.cnt_v[e], a.vert_offset, a.x, a.y, &nr, &nv, &fl);
    }
}
