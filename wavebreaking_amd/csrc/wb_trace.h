// wb_trace.h -- kernel 2: segments -> ordered, rounded, de-duplicated contours (one workgroup per
// (time step, level) job).
//
// Restates, for one job, wavebreaking/indices/contour_index.py:103-156 of the reference:
//   * skimage's _assemble_contours (ordering rules of SURVEY.md 8a row C2) as successor linking
//     over integer edge ids + two rounds of pointer jumping (list ranking) instead of a sequential
//     dictionary walk;
//   * :109-117  closed flag, round (done by the scan kernel on the f64 crossing), order-preserving
//     first-occurrence dedupe (open-addressing table keyed by (contour, point), atomicMin of the
//     position), keep contours with >= 4 points;
//   * :119-147  periodic duplicate rule (set inclusion after x mod nlon);
//   * :154-155  number of distinct x (-> exp_lon) and sum of y (-> mean_lat).
#pragma once
#include "wb_core.h"
#include "wb_scan.h"

struct WbTraceArgs {
    int njobs, nlat, nlon, W, cap_seg, hash_size;  // hash_size: power of two >= 8*cap_seg
    int nstrips, cap_strip;
    // from the scan kernel, per (job, strip): raster-ordered segment lists
    const uint32_t* stg_k;   // [njobs][nstrips][cap_strip] key
    const void* stg_v;       // [njobs][nstrips][cap_strip][4] corner values (float or double)
    int dtype, nlev;
    double levels[WB_MAX_LEVELS];
    const int* stg_n;        // [njobs][nstrips]
    uint32_t* mseg[3];       // [njobs][cap_seg] merged key / from / to (scratch)
    // scratch, per job
    uint32_t* w[10];         // 10 arrays [njobs][cap_seg]
    uint32_t* rawpt;         // [njobs][2*cap_seg]
    uint32_t* rawcid;        // [njobs][2*cap_seg]
    int* keep;               // [njobs][2*cap_seg + 1]
    unsigned long long* hkey;  // [njobs][hash_size]
    uint32_t* hval;          // [njobs][hash_size]
    // outputs, per job
    uint32_t* cpts;          // [njobs][2*cap_seg]   points (x | y<<16) of all surviving contours, back to back
    uint32_t* cown;          // [njobs][2*cap_seg]   owner contour (pre-drop numbering) of each point
    int* c_off;              // [njobs][cap_seg]     final contour table
    int* c_len;
    int* c_closed;
    int* c_nux;
    int* c_sumy;
    int* ncont;              // [njobs]
    uint32_t* status;        // [njobs]
    int* max_nux;            // [1] batch-wide maximum of c_nux
    long long* dbg;          // optional [njobs][16] phase time stamps (wall_clock64), NULL in production
};

#define WB_HEMPTY 0xFFFFFFFFFFFFFFFFull
#if defined(WB_HOSTSIM)
#define WB_STAMP(k) ((void)0)
#else
#define WB_STAMP(k) do { if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)job * 16 + (k)] = (long long)wall_clock64(); } while (0)
#endif

// insert key; returns slot (or -1 when the table is full); *fresh = 1 if this call created the entry
WB_DEV inline int wb_hash_insert(unsigned long long* hkey, int hmask, unsigned long long key, int* fresh) {
    uint32_t s = (uint32_t)wb_mix64(key) & (uint32_t)hmask;
    for (int probe = 0; probe <= hmask; ++probe) {
        unsigned long long cur = hkey[s];
        if (cur == key) { *fresh = 0; return (int)s; }
        if (cur == WB_HEMPTY) {
            unsigned long long old = wb_atomic_cas(&hkey[s], (unsigned long long)WB_HEMPTY, key);
            if (old == WB_HEMPTY) { *fresh = 1; return (int)s; }
            if (old == key) { *fresh = 0; return (int)s; }
        }
        s = (s + 1) & (uint32_t)hmask;
    }
    *fresh = 0;
    return -1;
}
WB_DEV inline int wb_hash_find(const unsigned long long* hkey, int hmask, unsigned long long key) {
    uint32_t s = (uint32_t)wb_mix64(key) & (uint32_t)hmask;
    for (int probe = 0; probe <= hmask; ++probe) {
        unsigned long long cur = hkey[s];
        if (cur == key) return (int)s;
        if (cur == WB_HEMPTY) return -1;
        s = (s + 1) & (uint32_t)hmask;
    }
    return -1;
}

WB_KERNEL_LB(wb_k_trace, 256, 8, WbTraceArgs a) {
    const int job = WB_BLOCK_ID;
    if (job >= a.njobs) return;
    const int cap = a.cap_seg;
    const int W = a.W;
    const size_t jb = (size_t)job * cap;
    uint32_t* segk = a.mseg[0] + jb;
    uint32_t* segf = a.mseg[1] + jb;
    uint32_t* segt = a.mseg[2] + jb;
    uint32_t* succ = a.w[0] + jb;
    uint32_t* pred = a.w[1] + jb;
    uint32_t* jA = a.w[2] + jb;
    uint32_t* jB = a.w[3] + jb;
    uint32_t* mA = a.w[4] + jb;   // pass 1: window max ; pass 2: window start
    uint32_t* mB = a.w[5] + jb;
    uint32_t* dA = a.w[6] + jb;   // pass 2: window length
    uint32_t* dB = a.w[7] + jb;
    uint32_t* t0 = a.w[8] + jb;   // misc per-segment / per-contour
    uint32_t* t1 = a.w[9] + jb;
    uint32_t* rawpt = a.rawpt + 2 * jb;
    uint32_t* rawcid = a.rawcid + 2 * jb;
    int* keep = a.keep + (size_t)job * (2 * cap + 1);
    unsigned long long* hkey = a.hkey + (size_t)job * a.hash_size;
    uint32_t* hval = a.hval + (size_t)job * a.hash_size;
    int hsize = a.hash_size;            // shrunk below to what this job needs (clearing the table is real traffic)
    uint32_t* cpts = a.cpts + 2 * jb;
    uint32_t* cown = a.cown + 2 * jb;
    int* c_off = a.c_off + jb;
    int* c_len = a.c_len + jb;
    int* c_closed = a.c_closed + jb;
    int* c_nux = a.c_nux + jb;
    int* c_sumy = a.c_sumy + jb;

    WB_STAMP(0);
    // ---- P0: merge the sorted strip lists into one raster-ordered list: the rank of a segment is its
    //          position in its own list plus the number of smaller keys in every other list
    int n = 0;
    bool over = false;
    for (int s = 0; s < a.nstrips; ++s) {
        const int c = a.stg_n[job * a.nstrips + s];
        if (c > a.cap_strip) over = true;
        n += c;
    }
    if (over || n > cap) {  // overflow: the scan kernel kept counting but stopped writing
        WB_SINGLE
        wb_atomic_or(&a.status[job], (uint32_t)WB_ST_OVERFLOW_SEG);
        a.ncont[job] = 0;
        WB_END_SINGLE
        return;
    }
    if (n == 0) {
        WB_SINGLE
        a.ncont[job] = 0;
        WB_END_SINGLE
        return;
    }
    {   // one pass over all segments of all strips (strip found from the running counts)
        WB_PFOR(g, n)
            int s = 0, i = (int)g;
            while (i >= a.stg_n[job * a.nstrips + s]) { i -= a.stg_n[job * a.nstrips + s]; ++s; }
            const size_t sb = ((size_t)job * a.nstrips + s) * a.cap_strip;
            const uint32_t key = a.stg_k[sb + i];
            const uint32_t kk = key & 0x07FFFFFFu;
            int dest = i;
            for (int s2 = 0; s2 < a.nstrips; ++s2) {
                if (s2 == s) continue;
                const uint32_t* other = a.stg_k + ((size_t)job * a.nstrips + s2) * a.cap_strip;
                int lo = 0, hi = a.stg_n[job * a.nstrips + s2];
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if ((other[mid] & 0x07FFFFFFu) < kk) lo = mid + 1; else hi = mid;
                }
                dest += lo;
            }
            // crossing coordinates of the segment's two end points, rounded half-to-even (contour_index.py:111)
            double ul, ur, ll, lr;
            if (a.dtype == 0) {
                const float* v = (const float*)a.stg_v + (sb + i) * 4;
                ul = v[0]; ur = v[1]; ll = v[2]; lr = v[3];
            } else {
                const double* v = (const double*)a.stg_v + (sb + i) * 4;
                ul = v[0]; ur = v[1]; ll = v[2]; lr = v[3];
            }
            const int sq = wb_key_sq(key), r = sq / W, c = sq - r * W, cs = wb_key_case(key), sub = wb_key_sub(key);
            const double level = a.levels[job % a.nlev];
            // a corner that equals the level puts the crossing ON a grid vertex: skimage's coordinate joins and
            // the edge linking used here may then differ -> flag the job
            if (ul == level || ur == level || ll == level || lr == level) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_LEVEL_HIT);
            segk[dest] = key;
            segf[dest] = wb_crossing_point(wb_seg_from(cs, sub), r, c, ul, ur, ll, lr, level);
            segt[dest] = wb_crossing_point(wb_seg_to(cs, sub), r, c, ul, ur, ll, lr, level);
        WB_END_PFOR
    }

    WB_STAMP(1);
    {   // raw points <= 2n; keys per point <= 2 (point key + x key): load factor <= 0.5
        int want = 1024;
        while (want < 8 * n && want < a.hash_size) want <<= 1;
        hsize = want;
    }
    const int hmask = hsize - 1;

    // ---- P0: init
    WB_PFOR(i, n)
        pred[i] = WB_NIL;
        t0[i] = 0;  // "is first segment of a cut loop"
    WB_END_PFOR

    WB_STAMP(2);
    // ---- P1: successor of every segment = the segment of the neighbouring square that starts on
    //          the edge this one ends on (binary search by square key in the merged list)
    WB_PFOR(i, n)
        const uint32_t k = segk[i];
        const int sq = wb_key_sq(k), r = sq / W, c = sq - r * W;
        const int ts = wb_seg_to(wb_key_case(k), wb_key_sub(k));
        int rr = r, cc = c;
        if (ts == WB_TOP) rr = r - 1;
        else if (ts == WB_RIGHT) cc = c + 1;
        else if (ts == WB_BOTTOM) rr = r + 1;
        else cc = c - 1;
        const int want = ts ^ 2;  // opposite side
        uint32_t s = WB_NIL;
        if (rr >= 0 && rr < a.nlat - 1 && cc >= 0 && cc < W - 1) {
            int lo = 0, hi = n;
            const int target = rr * W + cc;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (wb_key_sq(segk[mid]) < target) lo = mid + 1; else hi = mid;
            }
            if (lo < n && wb_key_sq(segk[lo]) == target) {
                const int cs2 = wb_key_case(segk[lo]);
                if (wb_seg_from(cs2, 0) == want) s = (uint32_t)lo;
                else if (wb_case_nseg(cs2) == 2 && wb_seg_from(cs2, 1) == want) s = (uint32_t)lo + 1;
            }
        }
        succ[i] = s;
        if (s != WB_NIL) pred[s] = (uint32_t)i;
    WB_END_PFOR

    WB_STAMP(3);
    int rounds = 0;
    while ((1 << rounds) < n) ++rounds;

    // ---- P2: pointer jumping over pred with a running max: finds closed loops and their
    //          raster-last member (the contour starts right after it)
    WB_PFOR(i, n)
        jA[i] = pred[i];
        mA[i] = (uint32_t)i;
    WB_END_PFOR
    {
        uint32_t *ji = jA, *jo = jB, *mi = mA, *mo = mB;
        for (int it = 0; it < rounds; ++it) {
            WB_PFOR(i, n)
                const uint32_t j = ji[i];
                if (j != WB_NIL) {
                    const uint32_t m1 = mi[i], m2 = mi[j];
                    mo[i] = m1 > m2 ? m1 : m2;
                    jo[i] = ji[j];
                } else {
                    mo[i] = mi[i];
                    jo[i] = WB_NIL;
                }
            WB_END_PFOR
            uint32_t* t = ji; ji = jo; jo = t;
            t = mi; mi = mo; mo = t;
        }
        // cut every loop after its raster-last segment
        WB_PFOR(i, n)
            if (ji[i] != WB_NIL && mi[i] == (uint32_t)i) {
                const uint32_t s = succ[i];
                t0[s] = 1;
                pred[s] = WB_NIL;
                succ[i] = WB_NIL;
            }
        WB_END_PFOR
    }

    WB_STAMP(4);
    // ---- P3: list ranking: start segment and position of every segment
    WB_PFOR(i, n)
        jA[i] = pred[i];
        mA[i] = (uint32_t)i;
        dA[i] = 1;
    WB_END_PFOR
    uint32_t *start = mA, *dist = dA;
    {
        uint32_t *ji = jA, *jo = jB, *si = mA, *so = mB, *di = dA, *dob = dB;
        for (int it = 0; it < rounds; ++it) {
            WB_PFOR(i, n)
                const uint32_t j = ji[i];
                if (j != WB_NIL) {
                    so[i] = si[j];
                    dob[i] = di[i] + di[j];
                    jo[i] = ji[j];
                } else {
                    so[i] = si[i];
                    dob[i] = di[i];
                    jo[i] = WB_NIL;
                }
            WB_END_PFOR
            uint32_t* t = ji; ji = jo; jo = t;
            t = si; si = so; so = t;
            t = di; di = dob; dob = t;
        }
        start = si;
        dist = di;
    }
    WB_STAMP(5);
    // free now: succ, pred, the two j buffers and the two "other" halves.  Name what we reuse:
    uint32_t* minidx = succ;       // per start segment: smallest member index
    uint32_t* cid_of = pred;       // per start segment: contour number (raster order of first segment)
    uint32_t* spare0 = (start == mA) ? mB : mA;
    uint32_t* spare1 = (dist == dA) ? dB : dA;
    int* flag = (int*)jA;
    int* cnt = (int*)jB;           // per contour: number of segments, then raw base

    // ---- P4: contour numbering (sorted by raster-first segment) and raw point layout
    WB_PFOR(i, n)
        minidx[i] = WB_NIL;
        cnt[i] = 0;
    WB_END_PFOR
    WB_PFOR(i, n)
        wb_atomic_min(&minidx[start[i]], (uint32_t)i);
    WB_END_PFOR
    WB_PFOR(i, n)
        flag[i] = (minidx[start[i]] == (uint32_t)i) ? 1 : 0;
    WB_END_PFOR
    const int C = wb_block_scan_excl(flag, n);
    WB_PFOR(i, n)
        if (minidx[start[i]] == (uint32_t)i) cid_of[start[i]] = (uint32_t)flag[i];
    WB_END_PFOR
    WB_PFOR(i, n)
        wb_atomic_add(&cnt[cid_of[start[i]]], 1);
    WB_END_PFOR
    int* rawbase = (int*)spare0;   // [C+1]
    WB_PFOR(c, C)
        rawbase[c] = cnt[c] + 1;
    WB_END_PFOR
    const int R = wb_block_scan_excl(rawbase, C);
    int* closed_raw = (int*)spare1;  // [C]
    WB_PFOR(i, n)
        const uint32_t c = cid_of[start[i]];
        const int pos = (int)dist[i] - 1;
        const int p = rawbase[c] + pos + 1;
        rawpt[p] = segt[i];
        rawcid[p] = c;
        if (pos == 0) {
            rawpt[rawbase[c]] = segf[i];
            rawcid[rawbase[c]] = c;
            closed_raw[c] = (int)t0[i];
        }
    WB_END_PFOR

    WB_STAMP(6);
    // ---- P5: order-preserving first-occurrence dedupe inside each contour
    WB_PFOR(s, hsize)
        hkey[s] = WB_HEMPTY;
        hval[s] = WB_NIL;
    WB_END_PFOR
    WB_PFOR(p, R)
        int fresh;
        const unsigned long long key = ((unsigned long long)rawcid[p] << 32) | rawpt[p];
        const int s = wb_hash_insert(hkey, hmask, key, &fresh);
        if (s >= 0) wb_atomic_min(&hval[s], (uint32_t)p);
        else wb_atomic_or(&a.status[job], (uint32_t)WB_ST_HASH_FULL);
    WB_END_PFOR
    WB_PFOR(p, R + 1)
        int kp = 0;
        if (p < R) {
            const unsigned long long key = ((unsigned long long)rawcid[p] << 32) | rawpt[p];
            const int s = wb_hash_find(hkey, hmask, key);
            kp = (s >= 0 && hval[s] == (uint32_t)p) ? 1 : 0;
        }
        keep[p] = kp;
    WB_END_PFOR
    // keep[] becomes the exclusive scan = new position; remember the flags in t1 (bit per raw point
    // would do, but raw points can be 2*cap: use the rawcid high bit instead)
    WB_PFOR(p, R)
        if (keep[p]) rawcid[p] |= 0x80000000u;
    WB_END_PFOR
    const int Rk = wb_block_scan_excl(keep, R + 1);
    (void)Rk;
    // per contour: new length, kept flag (>= 4 points, contour_index.py:113-117)
    int* kc = (int*)t1;        // [C] kept flag -> scan -> id among kept
    WB_PFOR(c, C)
        const int b0 = rawbase[c], b1 = rawbase[c] + cnt[c] + 1;
        kc[c] = (keep[b1] - keep[b0]) >= 4 ? 1 : 0;
    WB_END_PFOR
    int* kcflag = (int*)t0;    // copy of the flag (t0 free now)
    WB_PFOR(c, C)
        kcflag[c] = kc[c];
    WB_END_PFOR
    const int Ck = wb_block_scan_excl(kc, C);
    // compact points; owner = id among kept contours (NIL for contours that are too short)
    WB_PFOR(p, R)
        if (rawcid[p] & 0x80000000u) {
            const uint32_t c = rawcid[p] & 0x7FFFFFFFu;
            const int q = keep[p];
            cpts[q] = rawpt[p];
            cown[q] = kcflag[c] ? (uint32_t)kc[c] : WB_NIL;
        }
    WB_END_PFOR
    // pre-drop contour table in the c_* arrays (indexed by id among kept)
    WB_PFOR(c, C)
        if (kcflag[c]) {
            const int f = kc[c];
            const int b0 = rawbase[c], b1 = rawbase[c] + cnt[c] + 1;
            c_off[f] = keep[b0];
            c_len[f] = keep[b1] - keep[b0];
            c_closed[f] = closed_raw[c];
            c_nux[f] = 0;
            c_sumy[f] = 0;
        }
    WB_END_PFOR

    WB_STAMP(7);
    // ---- P6: sets of (x mod nlon, y) and of x per contour, in one table (tag bit 63 = x-key)
    int* setsize = (int*)jA;   // [Ck]
    int* dropped = (int*)jB;   // [Ck]
    WB_PFOR(s, hsize)
        hkey[s] = WB_HEMPTY;
    WB_END_PFOR
    WB_PFOR(f, Ck)
        setsize[f] = 0;
        dropped[f] = 0;
    WB_END_PFOR
    const int npts = (Ck > 0) ? keep[R] : 0;
    WB_PFOR(q, npts)
        const uint32_t f = cown[q];
        if (f != WB_NIL) {
            const uint32_t pt = cpts[q];
            const int x = wb_pt_x(pt), y = wb_pt_y(pt);
            int fresh;
            const unsigned long long k1 = ((unsigned long long)f << 32) | (uint32_t)((x % a.nlon) | (y << 16));
            if (wb_hash_insert(hkey, hmask, k1, &fresh) < 0) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_HASH_FULL);
            if (fresh) wb_atomic_add(&setsize[f], 1);
            const unsigned long long k2 = (1ull << 63) | ((unsigned long long)f << 32) | (uint32_t)x;
            if (wb_hash_insert(hkey, hmask, k2, &fresh) < 0) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_HASH_FULL);
            if (fresh) wb_atomic_add(&c_nux[f], 1);
            wb_atomic_add(&c_sumy[f], y);
        }
    WB_END_PFOR

    WB_STAMP(8);
    // ---- P7: periodic duplicate rule (contour_index.py:119-147): for every ordered pair (i1,i2)
    //          with set(i1) a subset of set(i2): equal array lengths -> drop max(i1,i2), else drop the shorter
    WB_PFOR(pair, (int64_t)Ck * Ck)
        const int i1 = (int)(pair / Ck), i2 = (int)(pair - (int64_t)i1 * Ck);
        if (i1 == i2 || setsize[i1] > setsize[i2]) continue;
        const int o1 = c_off[i1], l1 = c_len[i1];
        bool sub = true;
        for (int q = 0; q < l1 && sub; ++q) {
            const uint32_t pt = cpts[o1 + q];
            const unsigned long long k1 =
                ((unsigned long long)i2 << 32) | (uint32_t)((wb_pt_x(pt) % a.nlon) | (wb_pt_y(pt) << 16));
            if (wb_hash_find(hkey, hmask, k1) < 0) sub = false;
        }
        if (sub) {
            const int l2 = c_len[i2];
            const int d = (l1 == l2) ? (i1 > i2 ? i1 : i2) : (l1 < l2 ? i1 : i2);
            dropped[d] = 1;
        }
    WB_END_PFOR

    WB_STAMP(9);
    // ---- P8: final table = kept and not dropped, order preserved
    int* fin = (int*)t0;  // [Ck] -> exclusive scan
    WB_PFOR(f, Ck)
        fin[f] = dropped[f] ? 0 : 1;
    WB_END_PFOR
    const int Cf = wb_block_scan_excl(fin, Ck);
    // move table entries down (read everything first, then write: entries only move to lower slots,
    // but different threads handle different slots, so stage through scratch)
    int* s_off = (int*)spare0;
    int* s_len = (int*)spare1;
    int* s_cl = (int*)t1;
    int* s_nux = (int*)succ;
    int* s_sy = (int*)pred;
    WB_PFOR(f, Ck)
        s_off[f] = c_off[f];
        s_len[f] = c_len[f];
        s_cl[f] = c_closed[f];
        s_nux[f] = c_nux[f];
        s_sy[f] = c_sumy[f];
    WB_END_PFOR
    WB_PFOR(f, Ck)
        if (!dropped[f]) {
            const int g = fin[f];
            c_off[g] = s_off[f];
            c_len[g] = s_len[f];
            c_closed[g] = s_cl[f];
            c_nux[g] = s_nux[f];
            c_sumy[g] = s_sy[f];
            wb_atomic_max(a.max_nux, s_nux[f]);
        }
    WB_END_PFOR
    WB_STAMP(10);
    WB_SINGLE
    a.ncont[job] = Cf;
    WB_END_SINGLE
}
