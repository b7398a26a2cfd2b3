// wb_trace.h -- kernels 2a-2c: segments -> ordered, rounded, de-duplicated contours (one workgroup per
// (time step, level) job in each of the three kernels).
//   wb_k_trace_link   : merge the strip lists, crossing points, successor / predecessor links (keys staged in LDS)
//   wb_k_trace_rank   : closed-loop detection + list ranking by pointer jumping, entirely in LDS (16-bit links)
//   wb_k_trace_finish : contour numbering, dedupe, periodic duplicates, table
// They are separate launches because they want different shapes: link 512 threads + 64 KiB of keys in LDS, rank 1024
// threads + 112 KiB of 16-bit links, finish 1024 threads + a 64 KiB LDS table (two workgroups per CU).
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
    uint32_t* w[11];         // 11 arrays [njobs][cap_seg]
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
    int* nseg;               // [njobs] merged segment count (-1: job skipped)
    int* p4;                 // [njobs][2] contours and raw points of a job the rank kernel has numbered itself (-1: finish does it)
    int link_rows;           // wb_k_trace_link: 1 = its dynamic LDS holds the row tables ((nstrips + 1) * nlat 16-bit entries), 0 = merge by binary searches
    int slow_pass;           // 0: the ordinary launch (slow jobs are left out and counted in ctl->n_slow); 1: the slow jobs only
    WbCtl* ctl;              // batch counters (capacity checks, maxima)
    const unsigned char* fdone;  // [njobs] or NULL: jobs wb_k_trace_fused (wb_trace_fused.h) has finished -- the kernels below leave them alone
};
#define WB_TRACE_SKIP_DONE(a, job) if ((a).fdone && (a).fdone[job]) return

#define WB_HEMPTY 0xFFFFFFFFFFFFFFFFull

// ---- cells EQUAL to the level (contour_index.py:103 on packed / rounded PV).  Such a cell counts as "not above", every
// crossing of its incident edges sits ON the grid vertex (fraction exactly 0 or 1), and skimage's _assemble_contours joins
// through that one coordinate: a segment with both ends on the vertex is skipped, the other segment ends on the vertex --
// at most two per square, four squares -- are matched in raster order: an arrival takes the pending departure or becomes
// the pending arrival, a departure takes the pending arrival or becomes the pending departure (oracle/marching.py:
// vertex_links, pinned by tests/golden/contours_levelhit.npz).  A pending end that gets overwritten leaves its piece open
// there; skimage's sequential walk may then even join differently or raise KeyError, so such a job is flagged
// WB_ST_LEVEL_ORDER and not returned as a result.
#define WB_PT_VERTEX 0x8000u       // in a packed crossing point (x | y << 16): the crossing sits exactly on that grid vertex
#define WB_PT_MASK 0xFFFF7FFFu
#define WB_ST_LEVEL_ORDER 32u
#define WB_ST_LEVEL_WALKED 128u      // ... and the job's pieces were joined by the sequential restatement of skimage's walk (wb_k_trace_walk)

// does the crossing on `side` of a square sit on a grid vertex (one end of that edge equals the level)?
WB_HD inline bool wb_side_on_vertex(int side, double ul, double ur, double ll, double lr, double level) {
    if (side == WB_TOP) return ul == level || ur == level;
    if (side == WB_BOTTOM) return ll == level || lr == level;
    if (side == WB_LEFT) return ul == level || ll == level;
    return ur == level || lr == level;
}
// is `side` of a square incident to its corner q (0 ul, 1 ur, 2 ll, 3 lr)?
WB_HD inline bool wb_side_at_corner(int side, int q) {
    if (q == 0) return side == WB_TOP || side == WB_LEFT;
    if (q == 1) return side == WB_TOP || side == WB_RIGHT;
    if (q == 2) return side == WB_BOTTOM || side == WB_LEFT;
    return side == WB_BOTTOM || side == WB_RIGHT;
}
// The state machine of one hit vertex (pr, pc) from the point of view of segment `me`, an arrival there: index of the
// departure it is joined to, or WB_NIL.  *overwritten is set when some pending end at this vertex gets overwritten.
WB_DEV inline uint32_t wb_vertex_match(const uint32_t* keys, int n, int W, int nlat, int pr, int pc, uint32_t me, bool* overwritten) {
    uint32_t arrival = WB_NIL, departure = WB_NIL, mine = WB_NIL;
    for (int q = 3; q >= 0; --q) {             // squares in raster order: the vertex is their corner lr, ll, ur, ul
        const int r = pr - (q >> 1), c = pc - (q & 1);
        if (r < 0 || c < 0 || r >= nlat - 1 || c >= W - 1) continue;
        const int target = r * W + c;
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (wb_key_sq(keys[mid]) < target) lo = mid + 1; else hi = mid;
        }
        if (lo >= n || wb_key_sq(keys[lo]) != target) continue;
        const int cs = wb_key_case(keys[lo]);
        for (int sub = 0; sub < wb_case_nseg(cs); ++sub) {
            const bool arr = wb_side_at_corner(wb_seg_to(cs, sub), q), dep = wb_side_at_corner(wb_seg_from(cs, sub), q);
            if (arr == dep) continue;          // neither end on this vertex, or both: a zero-length segment, skipped
            const uint32_t idx = (uint32_t)(lo + sub);
            if (arr) {
                if (departure != WB_NIL) { if (idx == me) mine = departure; departure = WB_NIL; }
                else { if (arrival != WB_NIL) { *overwritten = true; } arrival = idx; }
            } else {
                if (arrival != WB_NIL) { if (arrival == me) mine = idx; arrival = WB_NIL; }
                else { if (departure != WB_NIL) { *overwritten = true; } departure = idx; }
            }
        }
    }
    return mine;
}


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


// ---- compact tables.  When (contour, x, y) fits 32 bits the dedupe table holds ONE 64-bit word per entry,
// key << 32 | position: an insert is a CAS into an empty slot or an atomicMin on a slot that already carries the key
// (equal high halves, so the minimum is the smaller position); a probe touches one 8-byte word instead of a key and a
// value array.  The set table of the periodic-duplicate rule shrinks to 32-bit keys the same way.
#define WB_HEMPTY32 0xFFFFFFFFu
#ifndef WB_TRACE_PACK
#define WB_TRACE_PACK 1           // 0: always the two-array 64-bit-key tables (tests)
#endif
WB_DEV inline bool wb_hash_put_min(unsigned long long* tab, int hmask, uint32_t key, uint32_t pos) {
    const unsigned long long mine = ((unsigned long long)key << 32) | pos;
    uint32_t s = (uint32_t)wb_mix64(key) & (uint32_t)hmask;
    for (int probe = 0; probe <= hmask; ++probe) {
        unsigned long long cur = tab[s];
        if (cur == WB_HEMPTY) {
            cur = wb_atomic_cas(&tab[s], (unsigned long long)WB_HEMPTY, mine);
            if (cur == WB_HEMPTY) return true;
        }
        if ((uint32_t)(cur >> 32) == key) { wb_atomic_min(&tab[s], mine); return true; }
        s = (s + 1) & (uint32_t)hmask;
    }
    return false;
}
WB_DEV inline uint32_t wb_hash_get_min(const unsigned long long* tab, int hmask, uint32_t key) {
    uint32_t s = (uint32_t)wb_mix64(key) & (uint32_t)hmask;
    for (int probe = 0; probe <= hmask; ++probe) {
        const unsigned long long cur = tab[s];
        if (cur == WB_HEMPTY) return WB_NIL;
        if ((uint32_t)(cur >> 32) == key) return (uint32_t)cur;
        s = (s + 1) & (uint32_t)hmask;
    }
    return WB_NIL;
}
// 32-bit key set: returns 1 if this call created the entry, 0 if it was there, -1 if the table is full
WB_DEV inline int wb_set32_insert(uint32_t* tab, int hmask, uint32_t key) {
    uint32_t s = (uint32_t)wb_mix64(key) & (uint32_t)hmask;
    for (int probe = 0; probe <= hmask; ++probe) {
        uint32_t cur = tab[s];
        if (cur == key) return 0;
        if (cur == WB_HEMPTY32) {
            cur = wb_atomic_cas(&tab[s], (uint32_t)WB_HEMPTY32, key);
            if (cur == WB_HEMPTY32) return 1;
            if (cur == key) return 0;
        }
        s = (s + 1) & (uint32_t)hmask;
    }
    return -1;
}
WB_DEV inline bool wb_set32_has(const uint32_t* tab, int hmask, uint32_t key) {
    uint32_t s = (uint32_t)wb_mix64(key) & (uint32_t)hmask;
    for (int probe = 0; probe <= hmask; ++probe) {
        const uint32_t cur = tab[s];
        if (cur == key) return true;
        if (cur == WB_HEMPTY32) return false;
        s = (s + 1) & (uint32_t)hmask;
    }
    return false;
}
WB_DEV inline int wb_bits_for(int n) {   // bits that hold every value in [0, n)
    int b = 0;
    while ((1 << b) < n) ++b;
    return b;
}

#ifndef WB_TRACE_BIGLEN
#define WB_TRACE_BIGLEN 128       // contours at least this long count their distinct x in an LDS bitmap
#endif
#define WB_TRACE_NBIG 16
#define WB_TRACE_XWORDS 128        // bitmap words per long contour: W <= 4096
#define WB_TRACE_NCAND 128

#ifndef WB_TRACE_LDS_N
#define WB_TRACE_LDS_N 8192      // jobs with at most this many segments take the LDS paths
#endif
// threads per (time step, level) job of the three trace kernels
#ifndef WB_LINK_THREADS
#define WB_LINK_THREADS 512
#endif
#ifndef WB_LINK_ROWTAB_BYTES
#define WB_LINK_ROWTAB_BYTES (28 << 10)   // dynamic LDS of the link kernel's row tables at most (beside its 32 KiB of keys)
#endif
#ifndef WB_RANK_THREADS
#define WB_RANK_THREADS 1024
#endif
#ifndef WB_FINISH_THREADS
#define WB_FINISH_THREADS 1024
#endif

// -------------------------------------------------------------------------------------------- 2a: link
WB_KERNEL(wb_k_trace_link, WbTraceArgs a) {
    const int job = WB_BLOCK_ID;
    if (job >= a.njobs) return;
    WB_TRACE_SKIP_DONE(a, job);
    const int cap = a.cap_seg;
    const int W = a.W;
    const size_t jb = (size_t)job * cap;
    uint32_t* segk = a.mseg[0] + jb;
    uint32_t* segf = a.mseg[1] + jb;
    uint32_t* segt = a.mseg[2] + jb;
    uint32_t* succ = a.w[0] + jb;
    uint32_t* pred = a.w[1] + jb;
    uint32_t* t0 = a.w[8] + jb;
    // keys in LDS.  Row-table path (below): the merged list only.  Search path: first those of all strips, strip after strip
    // (the merge's binary searches), then -- the same 32 KiB -- the merged list (the successor searches)
    WB_SHARED(uint32_t, ks, WB_TRACE_LDS_N);
    uint32_t* mk = ks;
    WB_SHARED(int, sbase, WB_MAX_LISTS + 1);
    // row tables (dynamic LDS, a.link_rows > 0): rs[s * nrow1 + r] = first entry of strip list s whose square row is >= r, and
    // behind the strips' tables the same for the merged list.  The lists are in raster order and strip s' lies left of strip s
    // for s' < s (the periodic copies, columns >= nlon, come last), so the merged position of entry i of list s in row r is
    // i + sum_{s' < s} rs[s'][r + 1] + sum_{s' > s} rs[s'][r] -- one independent LDS read per other list, where the search path
    // makes a binary search per other list (seven searches of nine dependent reads each on the 0.25 degree grid: 70 % of this
    // kernel's time until round 4); and a segment's successor is searched among the few entries of ONE row.
    WB_DYN_SHARED_BEGIN
    const int nrow1 = a.nlat;                      // rows of squares + 1
    WB_DYN_SHARED(uint16_t, rs, a.link_rows > 0 ? (size_t)(a.nstrips + 1) * nrow1 : 1);

    // ---- P0: merge the sorted strip lists into one raster-ordered list: the rank of a segment is its
    //          position in its own list plus the number of smaller keys in every other list
    int n = 0, nmax = 0;
    bool over = false;
    for (int s = 0; s < a.nstrips; ++s) {
        const int c = a.stg_n[job * a.nstrips + s];
        if (c > a.cap_strip) over = true;
        nmax = c > nmax ? c : nmax;
        n += c;
    }
    WB_SINGLE
    if (n > a.cap_seg || over) {          // the host grows the capacities and re-runs the phase
        wb_atomic_max(&a.ctl->max_seg_job, n);
        wb_atomic_max(&a.ctl->max_seg_strip, nmax);
    }
    WB_END_SINGLE
    if (over || n > cap || n == 0) {  // overflow: the scan kernel kept counting but stopped writing
        WB_SINGLE
        if (n != 0) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_OVERFLOW_SEG);
        a.ncont[job] = 0;
        a.nseg[job] = n == 0 ? 0 : -1;
        WB_END_SINGLE
        return;
    }
    const bool lds = n <= WB_TRACE_LDS_N;
    const bool rows = lds && a.link_rows > 0;
    WB_LSTAMP_BEGIN                              // development stamps: 0 strip bases, 1 keys to LDS / row tables, 2 merge + crossing points, 3 merged keys to LDS, 4 successors
    WB_SINGLE
    int run = 0;
    for (int s = 0; s < a.nstrips; ++s) { sbase[s] = run; run += a.stg_n[job * a.nstrips + s]; }
    sbase[a.nstrips] = run;
    WB_END_SINGLE
    WB_LSTAMP(a.ctl, 0);
    if (rows) {
        // every table entry is written exactly once: entry i of a list opens the rows (row of entry i - 1, row of entry i]
        WB_PFOR(g, n + a.nstrips)
            if ((int)g < n) {
                int s = 0;
                while ((int)g >= sbase[s + 1]) ++s;
                const int i = (int)g - sbase[s];
                const uint32_t* lk = a.stg_k + ((size_t)job * a.nstrips + s) * a.cap_strip;
                const int r = wb_key_sq(lk[i]) / W;
                const int rp = i > 0 ? wb_key_sq(lk[i - 1]) / W : -1;
                for (int q = rp + 1; q <= r; ++q) rs[s * nrow1 + q] = (uint16_t)i;
            } else {                                   // ... and the rows behind a list's last entry (all of them for an empty list)
                const int s = (int)g - n, c = sbase[s + 1] - sbase[s];
                const uint32_t* lk = a.stg_k + ((size_t)job * a.nstrips + s) * a.cap_strip;
                const int rl = c > 0 ? wb_key_sq(lk[c - 1]) / W : -1;
                for (int q = rl + 1; q < nrow1; ++q) rs[s * nrow1 + q] = (uint16_t)c;
            }
        WB_END_PFOR
        WB_PFOR(q, nrow1)                              // the merged list's table
            int t = 0;
            for (int s = 0; s < a.nstrips; ++s) t += rs[s * nrow1 + q];
            rs[a.nstrips * nrow1 + q] = (uint16_t)t;
        WB_END_PFOR
    } else if (lds) {
        WB_PFOR(g, n)
            int s = 0;
            while ((int)g >= sbase[s + 1]) ++s;
            ks[g] = a.stg_k[((size_t)job * a.nstrips + s) * a.cap_strip + ((int)g - sbase[s])] & 0x07FFFFFFu;
        WB_END_PFOR
    }
    WB_LSTAMP(a.ctl, 1);
    // one segment: its place in the merged list, its two crossing points
    auto place = [&](const int g, const int s, const int i, const uint32_t key, const double ul, const double ur, const double ll, const double lr) {
        const uint32_t kk = key & 0x07FFFFFFu;
        const int sq = wb_key_sq(key), r = sq / W, c = sq - r * W, cs = wb_key_case(key), sub = wb_key_sub(key);
        int dest = i;
        if (rows) {
            for (int s2 = 0; s2 < a.nstrips; ++s2) dest += s2 < s ? rs[s2 * nrow1 + r + 1] : (s2 > s ? rs[s2 * nrow1 + r] : 0);
        } else {
        for (int s2 = 0; s2 < a.nstrips; ++s2) {       // (these searches in lockstep, eight independent LDS reads per step, changed nothing: the phase waits for its global loads and scattered stores)
            if (s2 == s) continue;
            int lo = 0, hi = sbase[s2 + 1] - sbase[s2];
            if (lds) {
                const uint32_t* other = ks + sbase[s2];
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (other[mid] < kk) lo = mid + 1; else hi = mid;
                }
            } else {
                const uint32_t* other = a.stg_k + ((size_t)job * a.nstrips + s2) * a.cap_strip;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if ((other[mid] & 0x07FFFFFFu) < kk) lo = mid + 1; else hi = mid;
                }
            }
            dest += lo;
        }
        }
        // crossing coordinates of the segment's two end points, rounded half-to-even (contour_index.py:111)
        const double level = a.levels[job % a.nlev];
        // a corner that equals the level puts the crossing ON a grid vertex: joined by the vertex rule below; the
        // job is flagged for the caller's information
        if (ul == level || ur == level || ll == level || lr == level) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_LEVEL_HIT);
        segk[dest] = key;
        if (rows) mk[dest] = key;                      // (ks holds no strip keys on this path: the merged list goes straight to LDS)
        const int sfrom = wb_seg_from(cs, sub), sto = wb_seg_to(cs, sub);
        segf[dest] = wb_crossing_point(sfrom, r, c, ul, ur, ll, lr, level) | (wb_side_on_vertex(sfrom, ul, ur, ll, lr, level) ? WB_PT_VERTEX : 0u);
        segt[dest] = wb_crossing_point(sto, r, c, ul, ur, ll, lr, level) | (wb_side_on_vertex(sto, ul, ur, ll, lr, level) ? WB_PT_VERTEX : 0u);
        pred[g] = WB_NIL;
        t0[g] = 0;  // "is first segment of a cut loop"
    };
    // (the square's four corner values requested together with its key: they are not needed before the table reads / searches
    //  are done, and loaded after them they were a second round trip to memory per segment; float32 fields: four segments'
    //  keys and values requested before the first is placed, WB_PFOR_B)
    if (a.dtype == 0) {
        uint32_t keya[WB_PFOR_B_WIDTH];
        int sa[WB_PFOR_B_WIDTH];
        float va[WB_PFOR_B_WIDTH][4];
        WB_PFOR_B(g, gc, u, n)
            int s = 0;
            while ((int)gc >= sbase[s + 1]) ++s;
            const size_t e = ((size_t)job * a.nstrips + s) * a.cap_strip + (size_t)((int)gc - sbase[s]);
            const float* v = (const float*)a.stg_v + e * 4;
            sa[u] = s;
            keya[u] = a.stg_k[e];
            va[u][0] = v[0]; va[u][1] = v[1]; va[u][2] = v[2]; va[u][3] = v[3];
        WB_PFOR_B_USE(g, u)
            place((int)g, sa[u], (int)g - sbase[sa[u]], keya[u], (double)va[u][0], (double)va[u][1], (double)va[u][2], (double)va[u][3]);
        WB_END_PFOR_B
    } else {
        WB_PFOR(g, n)
            int s = 0;
            while ((int)g >= sbase[s + 1]) ++s;
            const int i = (int)g - sbase[s];
            const size_t sb = ((size_t)job * a.nstrips + s) * a.cap_strip;
            const double* v = (const double*)a.stg_v + (sb + i) * 4;
            place((int)g, s, i, a.stg_k[sb + i], v[0], v[1], v[2], v[3]);
        WB_END_PFOR
    }

    WB_LSTAMP(a.ctl, 2);
    // ---- P1: successor of every segment = the segment of the neighbouring square that starts on
    //          the edge this one ends on (binary search by square key in the merged list)
    if (lds && !rows) {
        WB_PFOR(i, n)
            mk[i] = segk[i];                     // (the strips' keys are done with)
        WB_END_PFOR
    }
    const uint32_t* keys = lds ? mk : segk;
    const uint16_t* mrow = rs + (size_t)a.nstrips * nrow1;
    WB_LSTAMP(a.ctl, 3);
    uint32_t pta[WB_PFOR_B_WIDTH];
    WB_PFOR_B(i, ic, u, n)
        pta[u] = segt[ic];
    WB_PFOR_B_USE(i, u)
        const uint32_t k = keys[i];
        const int sq = wb_key_sq(k), r = sq / W, c = sq - r * W;
        const int ts = wb_seg_to(wb_key_case(k), wb_key_sub(k));
        int rr = r, cc = c;
        if (ts == WB_TOP) rr = r - 1;
        else if (ts == WB_RIGHT) cc = c + 1;
        else if (ts == WB_BOTTOM) rr = r + 1;
        else cc = c - 1;
        const int want = ts ^ 2;  // opposite side
        uint32_t s = WB_NIL;
        const uint32_t pt = pta[u];
        if (pt & WB_PT_VERTEX) {
            // the crossing sits on a grid vertex (a cell equal to the level): joined by skimage's coordinate rule
            if (!((segf[i] & WB_PT_VERTEX) && segf[i] == pt)) {              // not a zero-length segment
                bool overwritten = false;
                s = wb_vertex_match(keys, n, W, a.nlat, wb_pt_y(pt & WB_PT_MASK), wb_pt_x(pt & WB_PT_MASK), (uint32_t)i, &overwritten);
                if (overwritten) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_LEVEL_ORDER);
            }
        } else if (rr >= 0 && rr < a.nlat - 1 && cc >= 0 && cc < W - 1) {
            int lo = rows ? (int)mrow[rr] : 0, hi = rows ? (int)mrow[rr + 1] : n;     // (the entries of ONE row, where the table is there)
            const int rhi = hi;
            const int target = rr * W + cc;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (wb_key_sq(keys[mid]) < target) lo = mid + 1; else hi = mid;
            }
            if (lo < rhi && wb_key_sq(keys[lo]) == target) {
                const int cs2 = wb_key_case(keys[lo]);
                if (wb_seg_from(cs2, 0) == want) s = (uint32_t)lo;
                else if (wb_case_nseg(cs2) == 2 && wb_seg_from(cs2, 1) == want) s = (uint32_t)lo + 1;
            }
        }
        succ[i] = s;
        if (s != WB_NIL) pred[s] = (uint32_t)i;
    WB_END_PFOR_B
    WB_LSTAMP(a.ctl, 4);
    WB_LSTAMP_FLUSH(a.ctl);
    WB_SINGLE
    a.nseg[job] = n;
    // jobs for the second trace pass (wb_trace_is_slow): counted, so that the host launches that pass only when there is one
    // (the flag was set with atomics by this workgroup's other threads: read it where they wrote it, not through this CU's L1)
    if (n > WB_TRACE_LDS_N || (wb_atomic_or(&a.status[job], 0u) & WB_ST_LEVEL_ORDER)) wb_atomic_add(&a.ctl->n_slow, 1);
    WB_END_SINGLE
}

// -------------------------------------------------------------------------------------------- 2b: rank
// Pointer jumping, generic over the link type (uint16_t in LDS, uint32_t in global memory).
//   pass 1 (over pred, running max): closed loops = elements whose pointer never reaches NIL; the loop's
//          raster-last member is where the index equals the window max -> the contour starts right after it
//   pass 2 (Wyllie list ranking): start segment and 1-based position of every segment
template <class IT>
WB_DEV inline void wb_rank_lists(int n, IT nil, const uint32_t* succ_g, uint32_t* t0_g, IT* P, IT* J0, IT* J1, IT* A0, IT* A1,
                                 IT* D0, IT* D1, uint32_t* start_g, uint32_t* dist_g) {
    int rounds = 0;
    while ((1 << rounds) < n) ++rounds;
    WB_PFOR(i, n)
        J0[i] = P[i];
        A0[i] = (IT)i;
    WB_END_PFOR
    IT *ji = J0, *jo = J1, *ai = A0, *ao = A1;
    for (int it = 0; it < rounds; ++it) {
        WB_PFOR(i, n)
            const IT j = ji[i];
            if (j != nil) {
                const IT m1 = ai[i], m2 = ai[j];
                ao[i] = m1 > m2 ? m1 : m2;
                jo[i] = ji[j];
            } else {
                ao[i] = ai[i];
                jo[i] = nil;
            }
        WB_END_PFOR
        IT* t = ji; ji = jo; jo = t;
        t = ai; ai = ao; ao = t;
    }
    WB_PFOR(i, n)   // cut every loop after its raster-last segment
        if (ji[i] != nil && ai[i] == (IT)i) {
            const uint32_t s = succ_g[i];
            t0_g[s] = 1;
            P[s] = nil;
        }
    WB_END_PFOR
    WB_PFOR(i, n)
        J0[i] = P[i];
        A0[i] = (IT)i;
        D0[i] = (IT)1;
    WB_END_PFOR
    ji = J0; jo = J1; ai = A0; ao = A1;
    IT *di = D0, *dob = D1;
    for (int it = 0; it < rounds; ++it) {
        WB_PFOR(i, n)
            const IT j = ji[i];
            if (j != nil) {
                ao[i] = ai[j];
                dob[i] = (IT)(di[i] + di[j]);
                jo[i] = ji[j];
            } else {
                ao[i] = ai[i];
                dob[i] = di[i];
                jo[i] = nil;
            }
        WB_END_PFOR
        IT* t = ji; ji = jo; jo = t;
        t = ai; ai = ao; ao = t;
        t = di; di = dob; dob = t;
    }
    WB_PFOR(i, n)
        start_g[i] = (uint32_t)ai[i];
        dist_g[i] = (uint32_t)di[i];
    WB_END_PFOR
}

// The same two passes for a job whose links fit 16 bits, with ONE copy of every array in LDS (three arrays, 48 KiB for 8192
// segments: two workgroups per CU; the double-buffered version above needs seven arrays, 112 KiB, one workgroup per CU).  A
// round reads its new values into registers, waits, writes them back: a thread owns WB_RANK_EPT strided elements.  D shares
// P's storage (P is dead once pass 2 has copied it).
#define WB_RANK_EPT_MAX ((WB_TRACE_LDS_N + WB_RANK_THREADS - 1) / WB_RANK_THREADS)
#if defined(WB_HOSTSIM)
#define WB_RANK_TMP(name) std::vector<uint32_t> name((size_t)n + 1)
#define WB_RANK_TMP64(name) std::vector<unsigned long long> name((size_t)n + 1)
#define WB_RANK_K_LOOP for (int k = 0; k < n; ++k)
#define WB_RANK_I(k) (k)
#else
#define WB_RANK_TMP(name) uint32_t name[EPT]
#define WB_RANK_TMP64(name) unsigned long long name[EPT]
#define WB_RANK_K_LOOP _Pragma("unroll") for (int k = 0; k < EPT; ++k)
#define WB_RANK_I(k) (WB_TID + (k) * WB_NT)
#endif
// With `nb` (not for jobs the sequential walk will redo) the kernel goes on while start and position are still in LDS:
// contours numbered by their raster-first segment, the raw point layout and the raw points themselves -- what used to be
// the first ten barrier-separated phases of the finish kernel, each a few round trips to L2 (40 % of its time).
#ifndef WB_TRACE_FUSE_P4
#define WB_TRACE_FUSE_P4 1
#endif
struct WbNumberOut {
    const uint32_t* segf; const uint32_t* segt;
    uint32_t* rawpt; uint32_t* rawcid;      // [2 * cap_seg]
    int* cnt; int* rawbase; int* closed;     // per contour: segments, first raw point, closed flag
    int* CR;                                 // [2]: contours, raw points
    WbCtl* ctl;                              // (development stamps)
};
// One element's (J, A, D) live in ONE 64-bit LDS word during the two passes -- bits 0..15 J (the jump pointer), 16..31 A (pass 1:
// running maximum, pass 2: the chain's first segment), 32..47 D (pass 1: the predecessor, kept for pass 2; pass 2: the
// position): a round reads its own word and the word it jumps to and writes one word, where three 16-bit arrays took five
// reads and three writes -- the kernel's 32 waves per CU were bound by the LDS pipe (2.9 us per round of pass 2, round 4's
// stamps).  After pass 2 the words are unpacked into the three 16-bit arrays the numbering below works on.
// A round is STRAIGHT-LINE code over the thread's EPT elements: the own words are requested together, then the words they
// jump to (an element without a pointer, or past the job's end, reads a word it is entitled to -- its own, or word 0 -- and
// the arithmetic selects), so that a round costs two LDS latencies, not two per element: written with `if (i < n)` /
// `if (j != nil)` around the reads the compiler emitted a branch per element and the sixteen reads of a round waited for one
// another (and the eight-element version spilled its new values to scratch memory inside the loop).  EPT is a template
// parameter -- the job's segments / workgroup size rounded up to 3, 5 or 8 -- so that a job of 2 500 segments does not pay for 8 192.
#define WB_RANK_PACK(j, a, d) ((unsigned long long)(uint16_t)(j) | (unsigned long long)(uint16_t)(a) << 16 | (unsigned long long)(uint16_t)(d) << 32)
template <int EPT>
WB_DEV inline void wb_rank_lists_lds(int n, const uint32_t* succ_g, const uint32_t* pred_g, uint32_t* t0_g, unsigned long long* Q,
                                     uint32_t* start_g, uint32_t* dist_g, const bool fuse, const WbNumberOut nb) {
    const uint16_t nil = (uint16_t)0xFFFF;
    uint16_t* P = (uint16_t*)Q;                          // the 16-bit arrays of the numbering part, over the same storage
    uint16_t* D = P;
    uint16_t* J = P + WB_TRACE_LDS_N;
    uint16_t* A = P + 2 * WB_TRACE_LDS_N;
    (void)J;
    int rounds = 0;
    while ((1 << rounds) < n) ++rounds;
    WB_RANK_TMP(t1);                                     // a round's new values, two 16-bit fields per register
    WB_RANK_TMP(t2);
    WB_RANK_TMP64(wq);                                   // pass 2: the element's word, old then new
    WB_KSTAMP_BEGIN                                      // development stamps: 0 set-up, 1 pass 1, 2 cut, 3 pass 2, 4 .. 7 numbering, 8 raw points
    WB_PFOR(i, n)
        const uint32_t v = pred_g[i];
        const uint16_t p16 = v == WB_NIL ? nil : (uint16_t)v;
        Q[i] = WB_RANK_PACK(p16, i, p16);
    WB_END_PFOR
    WB_KSTAMP(nb.ctl, 0);
    for (int it = 0; it < rounds; ++it) {               // pass 1: running maximum over the predecessors
        WB_RANK_K_LOOP {
            const int i = WB_RANK_I(k);
            wq[k] = Q[i < n ? i : 0];
        }
        WB_RANK_K_LOOP {
            const int i = WB_RANK_I(k);
            const uint16_t j = (uint16_t)wq[k], m1 = (uint16_t)(wq[k] >> 16);
            const unsigned long long wj = Q[j != nil ? (int)j : (i < n ? i : 0)];   // (no pointer: the own word -- max(m1, m1), J stays nil)
            const uint16_t m2 = (uint16_t)(wj >> 16);
            t1[k] = (uint32_t)(m1 > m2 ? m1 : m2) << 16 | (uint16_t)wj;
        }
        WB_SYNC();
        WB_RANK_K_LOOP {
            const int i = WB_RANK_I(k);
            if (i < n) ((uint32_t*)(Q + i))[0] = t1[k];          // J and A: the word's low half (D keeps the predecessor)
        }
        WB_SYNC();
    }
    WB_KSTAMP(nb.ctl, 1);
    WB_PFOR(i, n)   // cut every loop after its raster-last segment
        const unsigned long long wi = Q[i];
        if ((uint16_t)wi != nil && (uint16_t)(wi >> 16) == (uint16_t)i) {
            const uint32_t s = succ_g[i];
            t0_g[s] = 1;
            ((uint16_t*)(Q + s))[2] = nil;                       // the kept predecessor of the loop's first segment
        }
    WB_END_PFOR
    WB_RANK_K_LOOP {
        const int i = WB_RANK_I(k);
        const uint16_t p16 = ((const uint16_t*)(Q + (i < n ? i : 0)))[2];
        wq[k] = WB_RANK_PACK(p16, i, 1);                 // (a job of one segment makes no round: its start and position stand)
        if (i < n) Q[i] = wq[k];
    }
    WB_SYNC();
    WB_KSTAMP(nb.ctl, 2);
    for (int it = 0; it < rounds; ++it) {               // pass 2: list ranking
        WB_RANK_K_LOOP {
            const int i = WB_RANK_I(k);
            wq[k] = Q[i < n ? i : 0];
        }
        WB_RANK_K_LOOP {
            const int i = WB_RANK_I(k);
            const unsigned long long wi = wq[k];
            const uint16_t j = (uint16_t)wi;
            const bool has = j != nil;
            const unsigned long long wj = Q[has ? (int)j : (i < n ? i : 0)];       // (no pointer: the own word -- J stays nil, A stays)
            const uint16_t d = (uint16_t)((uint16_t)(wi >> 32) + (has ? (uint16_t)(wj >> 32) : (uint16_t)0));
            wq[k] = WB_RANK_PACK((uint16_t)wj, (uint16_t)(wj >> 16), d);
        }
        WB_SYNC();
        WB_RANK_K_LOOP {
            const int i = WB_RANK_I(k);
            if (i < n) Q[i] = wq[k];
        }
        WB_SYNC();
    }
    // unpack: A (start) and D (position) as 16-bit arrays over the same storage -- every word is in its thread's registers
    // before the first array element is written
    WB_SYNC();
    WB_RANK_K_LOOP {
        const int i = WB_RANK_I(k);
        t1[k] = (uint32_t)(uint16_t)(wq[k] >> 16) | (uint32_t)(uint16_t)(wq[k] >> 32) << 16;      // start | position << 16
        if (i < n) { A[i] = (uint16_t)t1[k]; D[i] = (uint16_t)(t1[k] >> 16); }
    }
    WB_SYNC();
    WB_KSTAMP(nb.ctl, 3);
    if (!fuse) {
        WB_PFOR(i, n)
            start_g[i] = (uint32_t)A[i];
            dist_g[i] = (uint32_t)D[i];
        WB_END_PFOR
        WB_KSTAMP_FLUSH(nb.ctl);
        return;
    }
    // ---- numbering and raw points (contour_index.py:103: skimage's contours come in the order of their raster-first segments)
    // t1 = start | position << 16 | TAIL (last segment of its chain: no successor, or a successor that heads a chain -- the
    // cut of a loop) | LEAD (raster-first segment of its contour); positions are <= 8192, bits 30 and 31 are free
    const uint32_t TAIL = 1u << 30, LEAD = 1u << 31;
    WB_RANK_K_LOOP {                                      // the successors together (global memory), then their positions (LDS)
        const int i = WB_RANK_I(k);
        t2[k] = succ_g[i < n ? i : 0];
    }
    WB_RANK_K_LOOP {
        const int i = WB_RANK_I(k);
        const uint32_t s = t2[k];
        const uint16_t ds = D[s != WB_NIL ? (int)s : (i < n ? i : 0)];
        t1[k] |= (s == WB_NIL || ds == 1) ? TAIL : 0u;   // (t1 = start | position << 16 from the unpacking)
    }
    WB_SYNC();
    // contour_index.py:109 calls a contour closed when its first and last float coordinates are equal: besides the loops
    // cut above that is an OPEN chain that leaves and reaches the same grid vertex (cells equal to the level)
    WB_RANK_K_LOOP {
        const int i = WB_RANK_I(k);
        if (i < n && t2[k] == WB_NIL && (nb.segt[i] & WB_PT_VERTEX)) {
            const uint32_t s = t1[k] & 0xFFFFu;
            if (nb.segf[s] == nb.segt[i]) t0_g[s] = 1;
        }
    }
    uint32_t* M = (uint32_t*)P;                           // P's and J's storage as n 32-bit words; A's as 16-bit counts
    uint16_t* cnt16 = A;
    WB_RANK_K_LOOP {
        const int i = WB_RANK_I(k);
        if (i < n) M[i] = WB_NIL;
    }
    WB_SYNC();
    WB_RANK_K_LOOP {                                      // ascending i per thread: after the first sweep most compares fail without an atomic
        const int i = WB_RANK_I(k);
        if (i < n && (uint32_t)i < M[t1[k] & 0xFFFFu]) wb_atomic_min(&M[t1[k] & 0xFFFFu], (uint32_t)i);
    }
    WB_SYNC();
    WB_RANK_K_LOOP {
        const int i = WB_RANK_I(k);
        if (i < n && M[t1[k] & 0xFFFFu] == (uint32_t)i) t1[k] |= LEAD;
    }
    WB_SYNC();
    WB_RANK_K_LOOP {
        const int i = WB_RANK_I(k);
        if (i < n) M[i] = (t1[k] & LEAD) ? 1u : 0u;
    }
    WB_SYNC();
    WB_KSTAMP(nb.ctl, 4);
    const int C = wb_block_scan_excl((int*)M, n);
    WB_KSTAMP(nb.ctl, 5);
    WB_RANK_K_LOOP {
        const int i = WB_RANK_I(k);
        if (i < n && (t1[k] & LEAD)) t2[k] = M[i];
    }
    WB_SYNC();
    WB_RANK_K_LOOP {                                      // M becomes: contour number, by start segment
        const int i = WB_RANK_I(k);
        if (i < n && (t1[k] & LEAD)) M[t1[k] & 0xFFFFu] = t2[k];
    }
    WB_SYNC();
    WB_RANK_K_LOOP {
        const int i = WB_RANK_I(k);
        if (i < n) {
            t2[k] = M[t1[k] & 0xFFFFu];
            if (t1[k] & TAIL) cnt16[t2[k]] = (uint16_t)((t1[k] >> 16) & 0x3FFFu);   // the last segment's position = segments of the contour
        }
    }
    WB_SYNC();
    WB_PFOR(c, C)
        M[c] = (uint32_t)cnt16[c] + 1u;
        nb.cnt[c] = (int)cnt16[c];
    WB_END_PFOR
    WB_KSTAMP(nb.ctl, 6);
    const int R = wb_block_scan_excl((int*)M, C);
    WB_PFOR(c, C)
        nb.rawbase[c] = (int)M[c];
    WB_END_PFOR
    WB_KSTAMP(nb.ctl, 7);
    WB_RANK_TMP(sg);                                      // the segments' end points (global memory) and the contours' bases (LDS), requested together
    WB_RANK_TMP(bs);
    WB_RANK_K_LOOP {
        const int i = WB_RANK_I(k);
        sg[k] = nb.segt[i < n ? i : 0];
        bs[k] = M[i < n ? t2[k] : 0u];
    }
    WB_RANK_K_LOOP {
        const int i = WB_RANK_I(k);
        if (i < n) {
            const uint32_t c = t2[k];
            const int pos = (int)((t1[k] >> 16) & 0x3FFFu), base = (int)bs[k], p = base + pos;
            nb.rawpt[p] = sg[k] & WB_PT_MASK;
            nb.rawcid[p] = c;
            if (pos == 1) {
                nb.rawpt[base] = nb.segf[i] & WB_PT_MASK;
                nb.rawcid[base] = c;
                nb.closed[c] = (int)t0_g[i];
            }
        }
    }
    WB_SINGLE
    nb.CR[0] = C;
    nb.CR[1] = R;
    WB_END_SINGLE
    WB_KSTAMP(nb.ctl, 8);
    WB_KSTAMP_FLUSH(nb.ctl);
}

// contour_index.py:109 calls a contour closed when its first and last float coordinates are equal: besides the loops the
// ranking cuts that is an OPEN chain that leaves and reaches the same grid vertex (cells equal to the level)
WB_DEV inline void wb_rank_vertex_loops(int n, const uint32_t* succ, const uint32_t* segf, const uint32_t* segt, const uint32_t* start, uint32_t* t0) {
    WB_PFOR(i, n)
        if (succ[i] == WB_NIL && (segt[i] & WB_PT_VERTEX)) {
            const uint32_t s = start[i];
            if (segf[s] == segt[i]) t0[s] = 1;
        }
    WB_END_PFOR
}
WB_KERNEL_LB(wb_k_trace_rank, WB_RANK_THREADS, (WB_RANK_THREADS >= 1024 ? 8 : 4), WbTraceArgs a) {
    const int job = WB_BLOCK_ID;
    if (job >= a.njobs) return;
    WB_TRACE_SKIP_DONE(a, job);
    const int n = a.nseg[job];
    if (n <= 0 || n > WB_TRACE_LDS_N) return;             // (longer jobs: wb_k_trace_rank_big)
    const size_t jb = (size_t)job * a.cap_seg;
    const uint32_t* succ = a.w[0] + jb;
    uint32_t* pred = a.w[1] + jb;
    uint32_t* t0 = a.w[8] + jb;
    uint32_t* start = a.w[4] + jb;
    uint32_t* dist = a.w[6] + jb;
    const uint32_t* segf = a.mseg[1] + jb;
    const uint32_t* segt = a.mseg[2] + jb;
    WB_SHARED(unsigned long long, q64, WB_TRACE_LDS_N);   // 64 KiB: one word per segment during the passes (wb_rank_lists_lds), three 16-bit arrays after
    // the finish kernel's scratch roles: cnt = w[3], rawbase = w[5], closed_raw = w[7] (free here)
    WbNumberOut nb;
    nb.segf = segf; nb.segt = segt; nb.rawpt = a.rawpt + 2 * jb; nb.rawcid = a.rawcid + 2 * jb;
    nb.cnt = (int*)(a.w[3] + jb); nb.rawbase = (int*)(a.w[5] + jb); nb.closed = (int*)(a.w[7] + jb); nb.CR = a.p4 + 2 * (size_t)job; nb.ctl = a.ctl;
    const bool fuse = WB_TRACE_FUSE_P4 && !(a.status[job] & WB_ST_LEVEL_ORDER);   // (those the walk kernel ranks again)
    if (n <= 3 * WB_RANK_THREADS) wb_rank_lists_lds<3>(n, succ, pred, t0, q64, start, dist, fuse, nb);
    else if (n <= 5 * WB_RANK_THREADS) wb_rank_lists_lds<5>(n, succ, pred, t0, q64, start, dist, fuse, nb);
    else wb_rank_lists_lds<WB_RANK_EPT_MAX>(n, succ, pred, t0, q64, start, dist, fuse, nb);
    if (fuse) return;
    WB_SINGLE
    a.p4[2 * (size_t)job] = -1;
    WB_END_SINGLE
    wb_rank_vertex_loops(n, succ, segf, segt, start, t0);
}
// big job: the same algorithm on 32-bit links in global memory; buffers J: w[2], w[3]; A: w[5], w[7]; D: w[9], w[10].  A
// kernel of its own so that its registers are not the LDS kernel's (which has to stay within 64 for two workgroups per CU).
// Launched in the second trace pass only (run_phase1: when ctl->n_slow says there is such a job).
WB_KERNEL(wb_k_trace_rank_big, WbTraceArgs a) {
    const int job = WB_BLOCK_ID;
    if (job >= a.njobs) return;
    WB_TRACE_SKIP_DONE(a, job);
    const int n = a.nseg[job];
    if (n <= WB_TRACE_LDS_N) return;
    const size_t jb = (size_t)job * a.cap_seg;
    const uint32_t* succ = a.w[0] + jb;
    uint32_t* t0 = a.w[8] + jb;
    uint32_t* start = a.w[4] + jb;
    WB_SINGLE
    a.p4[2 * (size_t)job] = -1;
    WB_END_SINGLE
    wb_rank_lists<uint32_t>(n, (uint32_t)WB_NIL, succ, t0, a.w[1] + jb, a.w[2] + jb, a.w[3] + jb, a.w[5] + jb, a.w[7] + jb,
                            a.w[9] + jb, a.w[10] + jb, start, a.w[6] + jb);
    wb_rank_vertex_loops(n, succ, a.mseg[1] + jb, a.mseg[2] + jb, start, t0);
}

// -------------------------------------------------------------------------------------------- 2b': the sequential walk
// Jobs flagged WB_ST_LEVEL_ORDER by the link kernel: two contour ends competed for one dictionary key (cells equal to the
// level), and what scikit-image returns there depends on the order its _assemble_contours (site-packages/skimage/measure/
// _find_contours.py, 0.18.3; called at contour_index.py:103) takes the segments in -- or it raises KeyError.  For those
// jobs only, ONE lane restates the walk literally (oracle/marching.py:join_by_coordinates): segments in raster order, two
// dictionaries keyed by the exact coordinates of the pieces' first / last points -- a crossing's coordinate is identified
// by its grid edge, or by the grid vertex it sits on --, a zero-length segment is skipped, the piece created first survives
// a join.  A piece is a chain of segments, so the result is what the rank kernel writes: start segment and position of
// every segment, closed flag of every chain.  ``last_at.pop(after[-1])`` / ``first_at.pop(before[0])`` on a key that is not
// there is skimage's KeyError: the job gets WB_ST_LEVEL_KEYERROR and the Python layer raises it.
#define WB_ST_LEVEL_KEYERROR 64u
WB_DEV inline unsigned long long wb_walk_key(uint32_t segkey, int side, uint32_t pt, int W) {
    if (pt & WB_PT_VERTEX) return (1ull << 62) | (unsigned long long)(pt & WB_PT_MASK);       // the grid vertex itself
    const int sq = wb_key_sq(segkey), r = sq / W, c = sq - r * W;                              // the grid edge: H(r, c) even, V(r, c) odd
    if (side == WB_TOP) return 2ull * (unsigned long long)(r * W + c);
    if (side == WB_BOTTOM) return 2ull * (unsigned long long)((r + 1) * W + c);
    if (side == WB_LEFT) return 2ull * (unsigned long long)(r * W + c) + 1ull;
    return 2ull * (unsigned long long)(r * W + c + 1) + 1ull;
}
// dictionary: open addressing over (key << 1 | which); a popped entry keeps its key and holds WB_NIL
WB_DEV inline uint32_t wb_walk_pop(unsigned long long* hk, uint32_t* hv, int hmask, unsigned long long key) {
    uint32_t s = (uint32_t)wb_mix64(key) & (uint32_t)hmask;
    for (int probe = 0; probe <= hmask; ++probe) {
        if (hk[s] == key) { const uint32_t v = hv[s]; hv[s] = WB_NIL; return v; }
        if (hk[s] == WB_HEMPTY) return WB_NIL;
        s = (s + 1) & (uint32_t)hmask;
    }
    return WB_NIL;
}
WB_DEV inline void wb_walk_put(unsigned long long* hk, uint32_t* hv, int hmask, unsigned long long key, uint32_t val) {
    uint32_t s = (uint32_t)wb_mix64(key) & (uint32_t)hmask;
    for (int probe = 0; probe <= hmask; ++probe) {
        if (hk[s] == key || hk[s] == WB_HEMPTY) { hk[s] = key; hv[s] = val; return; }
        s = (s + 1) & (uint32_t)hmask;
    }
}
WB_KERNEL(wb_k_trace_walk, WbTraceArgs a) {
    const int job = WB_BLOCK_ID;
    if (job >= a.njobs) return;
    WB_TRACE_SKIP_DONE(a, job);
    const int n = a.nseg[job];
    if (n <= 0 || !(a.status[job] & WB_ST_LEVEL_ORDER)) return;
    const size_t jb = (size_t)job * a.cap_seg;
    const uint32_t* segk = a.mseg[0] + jb;
    const uint32_t* segf = a.mseg[1] + jb;
    const uint32_t* segt = a.mseg[2] + jb;
    uint32_t* nxt = a.w[0] + jb;          // successor inside a piece (the link kernel's succ / pred are replaced)
    uint32_t* prv = a.w[1] + jb;
    uint32_t* p_first = a.w[2] + jb;      // per piece, by creation number
    uint32_t* p_last = a.w[3] + jb;
    uint32_t* p_alive = a.w[5] + jb;
    uint32_t* start = a.w[4] + jb;
    uint32_t* dist = a.w[6] + jb;
    uint32_t* t0 = a.w[8] + jb;
    unsigned long long* hk = a.hkey + (size_t)job * a.hash_size;
    uint32_t* hv = a.hval + (size_t)job * a.hash_size;
    int hsize = 1024;
    while (hsize < 8 * n && hsize < a.hash_size) hsize <<= 1;
    const int hmask = hsize - 1;
    WB_PFOR(q, hsize)
        hk[q] = WB_HEMPTY;
        hv[q] = WB_NIL;
    WB_END_PFOR
    WB_PFOR(i, n)
        nxt[i] = WB_NIL; prv[i] = WB_NIL; t0[i] = 0;
    WB_END_PFOR
    WB_SINGLE
    uint32_t created = 0;
    bool keyerror = false;
    for (int i = 0; i < n && !keyerror; ++i) {
        const uint32_t k = segk[i];
        const int cs = wb_key_case(k), sub = wb_key_sub(k);
        const unsigned long long f = wb_walk_key(k, wb_seg_from(cs, sub), segf[i], a.W), t = wb_walk_key(k, wb_seg_to(cs, sub), segt[i], a.W);
        if (f == t) continue;                                               // from_point == to_point: skipped
        const uint32_t after = wb_walk_pop(hk, hv, hmask, t << 1), before = wb_walk_pop(hk, hv, hmask, (f << 1) | 1ull);
        if (before != WB_NIL && after != WB_NIL) {
            // the segment goes behind `before` either way
            const uint32_t bl = p_last[before];
            nxt[bl] = (uint32_t)i; prv[i] = bl;
            if (before == after) {                                          // the piece closes
                p_last[before] = (uint32_t)i;
            } else {
                const uint32_t af = p_first[after], al = p_last[after], bf = p_first[before];
                nxt[i] = af; prv[af] = (uint32_t)i;
                const uint32_t alk = segk[al], bfk = segk[bf];
                const unsigned long long a_end = wb_walk_key(alk, wb_seg_to(wb_key_case(alk), wb_key_sub(alk)), segt[al], a.W);
                const unsigned long long b_beg = wb_walk_key(bfk, wb_seg_from(wb_key_case(bfk), wb_key_sub(bfk)), segf[bf], a.W);
                if (after > before) {                                       // `after` was created second: before.extend(after)
                    if (wb_walk_pop(hk, hv, hmask, (a_end << 1) | 1ull) == WB_NIL) { keyerror = true; break; }    // last_at.pop(after[-1])
                    p_alive[after] = 0;
                    p_last[before] = al;
                    wb_walk_put(hk, hv, hmask, b_beg << 1, before);
                    wb_walk_put(hk, hv, hmask, (a_end << 1) | 1ull, before);
                } else {                                                    // after.extendleft(reversed(before))
                    if (wb_walk_pop(hk, hv, hmask, b_beg << 1) == WB_NIL) { keyerror = true; break; }             // first_at.pop(before[0])
                    p_alive[before] = 0;
                    p_first[after] = bf;
                    wb_walk_put(hk, hv, hmask, b_beg << 1, after);
                    wb_walk_put(hk, hv, hmask, (a_end << 1) | 1ull, after);
                }
            }
        } else if (before == WB_NIL && after == WB_NIL) {                   // a new piece
            p_first[created] = (uint32_t)i; p_last[created] = (uint32_t)i; p_alive[created] = 1;
            wb_walk_put(hk, hv, hmask, f << 1, created);
            wb_walk_put(hk, hv, hmask, (t << 1) | 1ull, created);
            ++created;
        } else if (before == WB_NIL) {                                      // after.appendleft(from_point)
            const uint32_t af = p_first[after];
            nxt[i] = af; prv[af] = (uint32_t)i;
            p_first[after] = (uint32_t)i;
            wb_walk_put(hk, hv, hmask, f << 1, after);
        } else {                                                            // before.append(to_point)
            const uint32_t bl = p_last[before];
            nxt[bl] = (uint32_t)i; prv[i] = bl;
            p_last[before] = (uint32_t)i;
            wb_walk_put(hk, hv, hmask, (t << 1) | 1ull, before);
        }
    }
    if (keyerror) {
        wb_atomic_or(&a.status[job], (uint32_t)WB_ST_LEVEL_KEYERROR);
        // (start / dist keep the rank kernel's values: the job's contours are not handed out)
    } else {
        // skipped (zero-length) segments: chains of their own, as the rank kernel leaves them
        for (int i = 0; i < n; ++i) { start[i] = (uint32_t)i; dist[i] = 1; }
        for (uint32_t p = 0; p < created; ++p) {
            if (!p_alive[p]) continue;
            const uint32_t first = p_first[p], last = p_last[p];
            uint32_t pos = 1;
            for (uint32_t i = first;; i = nxt[i]) {
                start[i] = first; dist[i] = pos++;
                if (i == last) break;
            }
            // contour_index.py:109: closed = first and last float coordinates equal
            const uint32_t fk = segk[first], lk = segk[last];
            const unsigned long long f = wb_walk_key(fk, wb_seg_from(wb_key_case(fk), wb_key_sub(fk)), segf[first], a.W);
            const unsigned long long t = wb_walk_key(lk, wb_seg_to(wb_key_case(lk), wb_key_sub(lk)), segt[last], a.W);
            t0[first] = f == t ? 1 : 0;
        }
        wb_atomic_or(&a.status[job], (uint32_t)WB_ST_LEVEL_WALKED);
    }
    WB_END_SINGLE
}

// -------------------------------------------------------------------------------------------- 2c: finish
// Workgroups of 1024 threads, two per CU: the same number of waves as eight workgroups of 256, but room for a 64 KiB
// table in LDS per workgroup -- the dedupe table and the set table of a typical job (a few thousand keys) live there
// instead of in L2 / HBM, where every probe costs a 64-byte sector.
// raw point p and its predecessor, all four words requested at once (p may be R: clamped, the caller tests p < R): written
// `p > 0 && rawcid[p] == rawcid[p - 1] && rawpt[p] == rawpt[p - 1]` the short-circuit made it three dependent round trips
#define WB_RAW_LOAD(p, R)                                                                           \
    const int wb_pc_ = (int)(p) < (R) ? (int)(p) : ((R) > 0 ? (R) - 1 : 0), wb_pm_ = wb_pc_ > 0 ? wb_pc_ - 1 : 0;   \
    const uint32_t rc1 = rawcid[wb_pc_], rc0 = rawcid[wb_pm_], rp1 = rawpt[wb_pc_], rp0 = rawpt[wb_pm_]; \
    const bool rdup = (wb_pc_ > 0) & (rc1 == rc0) & (rp1 == rp0)
// the same inside WB_PFOR_B: the request part for raw point `pc` (clamped by the caller), the use part for raw point p
#define WB_RAW_REQ(pc)                                                                              \
    { const int wb_pc_ = (int)(pc), wb_pm_ = wb_pc_ > 0 ? wb_pc_ - 1 : 0;                            \
      rc1a[u] = rawcid[wb_pc_]; rc0a[u] = rawcid[wb_pm_]; rp1a[u] = rawpt[wb_pc_]; rp0a[u] = rawpt[wb_pm_]; }
#define WB_RAW_USE(p)                                                                               \
    const uint32_t rc1 = rc1a[u], rp1 = rp1a[u];                                                    \
    const bool rdup = ((p) > 0) & (rc1 == rc0a[u]) & (rp1 == rp0a[u])
#ifndef WB_TRACE_LTAB
#define WB_TRACE_LTAB 8192        // 64-bit entries of the LDS table (= 16384 32-bit set entries)
#endif
WB_KERNEL_LB(wb_k_trace_finish, WB_FINISH_THREADS, (WB_FINISH_THREADS >= 1024 ? 8 : 4), WbTraceArgs a) {
    WB_SHARED(unsigned long long, ltab, WB_TRACE_LTAB);
    const int job = WB_BLOCK_ID;
    if (job >= a.njobs) return;
    WB_TRACE_SKIP_DONE(a, job);
    const int n = a.nseg[job];
    if (n <= 0) return;            // ncont already written by the link kernel
    {   // the slow jobs (big, or to be walked sequentially) wait for the second pass, which takes nothing else
        const bool slow = n > WB_TRACE_LDS_N || (a.status[job] & WB_ST_LEVEL_ORDER);
        if (slow != (a.slow_pass != 0)) {
            if (slow) {
                WB_SINGLE
                a.ncont[job] = 0;
                WB_END_SINGLE
            }
            return;
        }
    }
    const int cap = a.cap_seg;
    const size_t jb = (size_t)job * cap;
    const uint32_t* segf = a.mseg[1] + jb;
    const uint32_t* segt = a.mseg[2] + jb;
    uint32_t* succ = a.w[0] + jb;
    uint32_t* pred = a.w[1] + jb;
    uint32_t* jA = a.w[2] + jb;
    uint32_t* jB = a.w[3] + jb;
    uint32_t* start = a.w[4] + jb;
    uint32_t* dist = a.w[6] + jb;
    uint32_t* t0 = a.w[8] + jb;   // in: "is first segment of a cut loop"; later scratch
    uint32_t* t1 = a.w[9] + jb;
    uint32_t* rawpt = a.rawpt + 2 * jb;
    uint32_t* rawcid = a.rawcid + 2 * jb;
    int* keep = a.keep + (size_t)job * (2 * cap + 1);
    unsigned long long* hkey = a.hkey + (size_t)job * a.hash_size;
    uint32_t* hval = a.hval + (size_t)job * a.hash_size;
    uint32_t* cpts = a.cpts + 2 * jb;
    uint32_t* cown = a.cown + 2 * jb;
    int* c_off = a.c_off + jb;
    int* c_len = a.c_len + jb;
    int* c_closed = a.c_closed + jb;
    int* c_nux = a.c_nux + jb;
    int* c_sumy = a.c_sumy + jb;
    // the open-addressing table is sized per use to twice its number of keys (clearing it is real HBM traffic)
    int hsize = 1024, hmask = 1023;
    // scratch roles in this kernel
    uint32_t* minidx = succ;       // per start segment: smallest member index
    uint32_t* cid_of = pred;       // per start segment: contour number (raster order of first segment)
    uint32_t* spare0 = a.w[5] + jb;
    uint32_t* spare1 = a.w[7] + jb;
    int* flag = (int*)jA;
    int* cnt = (int*)jB;           // per contour: number of segments, then raw base

    WB_TSTAMP_BEGIN                        // development stamps: 0 numbering, 1 raw points, 2 key count, 3 dedupe insert, 4 dedupe lookup, 5 compaction + table, 6 set-up of the sets, 7 set inserts, 8 duplicate rule, 9 queued subset tests, 10 final table
    // ---- P4: contour numbering (sorted by raster-first segment) and raw point layout -- unless the rank kernel has done it
    int* rawbase = (int*)spare0;   // [C+1]
    int* closed_raw = (int*)spare1;  // [C]
    int C = a.p4[2 * (size_t)job], R = a.p4[2 * (size_t)job + 1];
    if (C < 0) {
    WB_PFOR(i, n)
        minidx[i] = WB_NIL;
        cnt[i] = 0;
    WB_END_PFOR
    WB_PFOR(i, n)
        // ascending i per thread: after the first sweep almost every compare fails without touching the atomic unit
        if ((uint32_t)i < minidx[start[i]]) wb_atomic_min(&minidx[start[i]], (uint32_t)i);
    WB_END_PFOR
    WB_PFOR(i, n)
        flag[i] = (minidx[start[i]] == (uint32_t)i) ? 1 : 0;
    WB_END_PFOR
    C = wb_block_scan_excl(flag, n);
    WB_PFOR(i, n)
        if (minidx[start[i]] == (uint32_t)i) cid_of[start[i]] = (uint32_t)flag[i];
    WB_END_PFOR
    WB_PFOR(i, n)
        wb_atomic_add_agg(cnt, (int)cid_of[start[i]], 1);
    WB_END_PFOR
    WB_TSTAMP(a.ctl, 0);
    WB_PFOR(c, C)
        rawbase[c] = cnt[c] + 1;
    WB_END_PFOR
    R = wb_block_scan_excl(rawbase, C);
    WB_PFOR(i, n)
        const uint32_t c = cid_of[start[i]];
        const int pos = (int)dist[i] - 1;
        const int p = rawbase[c] + pos + 1;
        rawpt[p] = segt[i] & WB_PT_MASK;
        rawcid[p] = c;
        if (pos == 0) {
            rawpt[rawbase[c]] = segf[i] & WB_PT_MASK;
            rawcid[rawbase[c]] = c;
            closed_raw[c] = (int)t0[i];
        }
    WB_END_PFOR
    }
    WB_TSTAMP(a.ctl, 1);
    // ---- P5: order-preserving first-occurrence dedupe inside each contour
    // A point equal to its predecessor in the same contour can never be a first occurrence: it is dropped without
    // touching the table (about half of all raw points: neighbouring crossings that round to the same lattice point).
    // The table is sized to twice the keys that do go in (clearing and probing it is real HBM traffic).
    WB_SHARED(int, nkeys, 1);
    WB_SINGLE
    nkeys[0] = 0;
    WB_END_SINGLE
    uint32_t rc1a[WB_PFOR_B_WIDTH], rc0a[WB_PFOR_B_WIDTH], rp1a[WB_PFOR_B_WIDTH], rp0a[WB_PFOR_B_WIDTH];   // (WB_PFOR_B: four raw points requested, then used)
    const int Rl = R > 0 ? R - 1 : 0;
    WB_PFOR_B(p, pc, u, R)
        WB_RAW_REQ(pc);
    WB_PFOR_B_USE(p, u)
        WB_RAW_USE(p);
        wb_atomic_add_agg(nkeys, 0, rdup ? 0 : 1);
    WB_END_PFOR_B
    WB_TSTAMP(a.ctl, 2);
    hsize = 1024;
    while (hsize < 2 * nkeys[0] && hsize < a.hash_size) hsize <<= 1;
    hmask = hsize - 1;
    const int bx = wb_bits_for(a.W), by = wb_bits_for(a.nlat);
    const bool pk5 = WB_TRACE_PACK && bx + by + wb_bits_for(C) <= 32;       // (contour, x, y) in 32 bits: one-word entries
    if (pk5 && hsize <= WB_TRACE_LTAB && 2 * nkeys[0] <= hsize) {           // ... and few enough for the LDS table
        WB_PFOR(s, hsize)
            ltab[s] = WB_HEMPTY;
        WB_END_PFOR
        WB_PFOR_B(p, pc, u, R)
            WB_RAW_REQ(pc);
        WB_PFOR_B_USE(p, u)
            WB_RAW_USE(p);
            if (rdup) continue;
            const uint32_t key = (rc1 << (bx + by)) | ((uint32_t)wb_pt_y(rp1) << bx) | (uint32_t)wb_pt_x(rp1);
            if (!wb_hash_put_min(ltab, hmask, key, (uint32_t)p)) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_HASH_FULL);
        WB_END_PFOR_B
        WB_TSTAMP(a.ctl, 3);
        WB_PFOR_B(p, pc, u, R + 1)
            WB_RAW_REQ(pc < R ? pc : Rl);
        WB_PFOR_B_USE(p, u)
            int kp = 0;
            WB_RAW_USE(p);
            if (p < R && !rdup) {
                const uint32_t key = (rc1 << (bx + by)) | ((uint32_t)wb_pt_y(rp1) << bx) | (uint32_t)wb_pt_x(rp1);
                kp = wb_hash_get_min(ltab, hmask, key) == (uint32_t)p ? 1 : 0;
            }
            keep[p] = kp;
        WB_END_PFOR_B
    } else if (pk5) {
        WB_PFOR(s, hsize)
            hkey[s] = WB_HEMPTY;
        WB_END_PFOR
        WB_PFOR_B(p, pc, u, R)
            WB_RAW_REQ(pc);
        WB_PFOR_B_USE(p, u)
            WB_RAW_USE(p);
            if (rdup) continue;
            const uint32_t key = (rc1 << (bx + by)) | ((uint32_t)wb_pt_y(rp1) << bx) | (uint32_t)wb_pt_x(rp1);
            if (!wb_hash_put_min(hkey, hmask, key, (uint32_t)p)) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_HASH_FULL);
        WB_END_PFOR_B
        WB_PFOR_B(p, pc, u, R + 1)
            WB_RAW_REQ(pc < R ? pc : Rl);
        WB_PFOR_B_USE(p, u)
            int kp = 0;
            WB_RAW_USE(p);
            if (p < R && !rdup) {
                const uint32_t key = (rc1 << (bx + by)) | ((uint32_t)wb_pt_y(rp1) << bx) | (uint32_t)wb_pt_x(rp1);
                kp = wb_hash_get_min(hkey, hmask, key) == (uint32_t)p ? 1 : 0;
            }
            keep[p] = kp;
        WB_END_PFOR_B
    } else {
        WB_PFOR(s, hsize)
            hkey[s] = WB_HEMPTY;
            hval[s] = WB_NIL;
        WB_END_PFOR
        WB_PFOR_B(p, pc, u, R)
            WB_RAW_REQ(pc);
        WB_PFOR_B_USE(p, u)
            WB_RAW_USE(p);
            if (rdup) continue;
            int fresh;
            const unsigned long long key = ((unsigned long long)rc1 << 32) | rp1;
            const int s = wb_hash_insert(hkey, hmask, key, &fresh);
            if (s >= 0) wb_atomic_min(&hval[s], (uint32_t)p);
            else wb_atomic_or(&a.status[job], (uint32_t)WB_ST_HASH_FULL);
        WB_END_PFOR_B
        WB_PFOR_B(p, pc, u, R + 1)
            WB_RAW_REQ(pc < R ? pc : Rl);
        WB_PFOR_B_USE(p, u)
            int kp = 0;
            WB_RAW_USE(p);
            if (p < R && !rdup) {
                const unsigned long long key = ((unsigned long long)rc1 << 32) | rp1;
                const int s = wb_hash_find(hkey, hmask, key);
                kp = (s >= 0 && hval[s] == (uint32_t)p) ? 1 : 0;
            }
            keep[p] = kp;
        WB_END_PFOR_B
    }
    WB_TSTAMP(a.ctl, 4);
    // keep[] becomes the exclusive scan = new position (R + 1 entries: a point was kept iff its successor's position is larger)
    const int Rk = wb_block_scan_excl(keep, R + 1);
    (void)Rk;
    // per contour: new length, kept flag (>= 4 points, contour_index.py:113-117)
    int* kc = (int*)t1;        // [C] kept flag -> scan -> id among kept
    int* kcflag = (int*)t0;    // copy of the flag (t0 free now)
    WB_PFOR(c, C)
        const int b0 = rawbase[c], b1 = rawbase[c] + cnt[c] + 1;
        const int f = (keep[b1] - keep[b0]) >= 4 ? 1 : 0;
        kc[c] = f;
        kcflag[c] = f;
    WB_END_PFOR
    const int Ck = wb_block_scan_excl(kc, C);
    // compact points; owner = id among kept contours (NIL for contours that are too short) -- and, the first C lanes, the
    // pre-drop contour table in the c_* arrays (indexed by id among kept)
    int k0a[WB_PFOR_B_WIDTH], k1a[WB_PFOR_B_WIDTH];
    WB_PFOR_B(p, pc, u, R)                                 // (a point's old and new position, its contour -- and, behind that, the contour's fate)
        k0a[u] = keep[pc];
        k1a[u] = keep[pc + 1];
        rp1a[u] = rawpt[pc];
        const uint32_t c = rawcid[pc];
        rc1a[u] = kcflag[c] ? (uint32_t)kc[c] : WB_NIL;
    WB_PFOR_B_USE(p, u)
        const int q = k0a[u];
        if (k1a[u] != q) {
            cpts[q] = rp1a[u];
            cown[q] = rc1a[u];
        }
        if (p < C && kcflag[p]) {
            const int c = (int)p, f = kc[c];
            const int b0 = rawbase[c], b1 = rawbase[c] + cnt[c] + 1;
            c_off[f] = keep[b0];
            c_len[f] = keep[b1] - keep[b0];
            c_closed[f] = closed_raw[c];
            c_nux[f] = 0;
            c_sumy[f] = 0;
        }
    WB_END_PFOR_B

    WB_TSTAMP(a.ctl, 5);
    // ---- P6: sets of (x mod nlon, y) and of x per contour, in one table (tag bit 63 = x-key)
    int* setsize = (int*)jA;   // [Ck]
    int* dropped = (int*)jB;   // [Ck]
    hsize = 1024;                  // sets: at most two keys per surviving point (x keys of long contours go to bitmaps)
    while (hsize < 4 * (keep[R] + 1) && hsize < a.hash_size) hsize <<= 1;
    const int bf = wb_bits_for(Ck);
    const bool pk6 = WB_TRACE_PACK && bf + bx + by <= 30;                    // 32-bit set keys, bit 31 tags the x-only keys
    // LDS table (as 32-bit entries) when three slots per surviving point fit
    int lsize = 2 * WB_TRACE_LTAB < 1024 ? 2 * WB_TRACE_LTAB : 1024;
    while (lsize < 3 * (keep[R] + 1) && lsize < 2 * WB_TRACE_LTAB) lsize <<= 1;
    const bool l6 = pk6 && 3 * (keep[R] + 1) <= lsize;
    if (l6) hsize = lsize;
    hmask = hsize - 1;
    uint32_t* set32 = l6 ? (uint32_t*)ltab : hval;
    if (pk6) {
        WB_PFOR(s, hsize)
            set32[s] = WB_HEMPTY32;
        WB_END_PFOR
    } else {
        WB_PFOR(s, hsize)
            hkey[s] = WB_HEMPTY;
        WB_END_PFOR
    }
    // the distinct-x count of a LONG contour (most points of a job belong to one or two of them) is taken from a bitmap
    // over x in LDS instead of one table insert per point
    WB_SHARED(uint32_t, xbits, WB_TRACE_NBIG * WB_TRACE_XWORDS);
    WB_SHARED(int, bigf, WB_TRACE_NBIG);
    WB_SHARED(int, nbig, 1);
    int* bigof = (int*)t0;         // [Ck] slot of a long contour, -1 otherwise (t0 is free until P8)
    const int xw = (a.W + 31) >> 5;
    WB_SINGLE
    nbig[0] = 0;
    WB_END_SINGLE
    WB_PFOR(f, Ck)
        setsize[f] = 0;
        dropped[f] = 0;
        int slot = -1;
        if (c_len[f] >= WB_TRACE_BIGLEN && xw <= WB_TRACE_XWORDS) {
            slot = wb_atomic_add(&nbig[0], 1);
            if (slot < WB_TRACE_NBIG) bigf[slot] = (int)f; else slot = -1;
        }
        bigof[f] = slot;
    WB_END_PFOR
    const int nb = nbig[0] < WB_TRACE_NBIG ? nbig[0] : WB_TRACE_NBIG;
    WB_PFOR(w, nb * xw)
        xbits[(w / xw) * WB_TRACE_XWORDS + (w % xw)] = 0u;
    WB_END_PFOR
    WB_TSTAMP(a.ctl, 6);
    const int npts = (Ck > 0) ? keep[R] : 0;
    WB_PFOR_B(q, qc, u, npts)                              // (a point, its owner -- and, behind that, whether the owner is a long contour)
        rp1a[u] = cpts[qc];
        rc1a[u] = cown[qc];
        k0a[u] = bigof[rc1a[u] != WB_NIL ? rc1a[u] : 0u];
    WB_PFOR_B_USE(q, u)
        const uint32_t f = rc1a[u];
        const uint32_t pt = rp1a[u];
        if (f != WB_NIL) {
            const int x = wb_pt_x(pt), y = wb_pt_y(pt);
            const int slot = k0a[u];
            int fresh1, fresh2 = 0;
            if (slot >= 0) wb_atomic_or(&xbits[slot * WB_TRACE_XWORDS + (x >> 5)], 1u << (x & 31));
            if (pk6) {
                fresh1 = wb_set32_insert(set32, hmask, (f << (bx + by)) | ((uint32_t)y << bx) | (uint32_t)(x % a.nlon));
                if (slot < 0) fresh2 = wb_set32_insert(set32, hmask, 0x80000000u | (f << bx) | (uint32_t)x);
                if (fresh1 < 0 || fresh2 < 0) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_HASH_FULL);
            } else {
                const unsigned long long k1 = ((unsigned long long)f << 32) | (uint32_t)((x % a.nlon) | (y << 16));
                if (wb_hash_insert(hkey, hmask, k1, &fresh1) < 0) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_HASH_FULL);
                if (slot < 0) {
                    const unsigned long long k2 = (1ull << 63) | ((unsigned long long)f << 32) | (uint32_t)x;
                    if (wb_hash_insert(hkey, hmask, k2, &fresh2) < 0) wb_atomic_or(&a.status[job], (uint32_t)WB_ST_HASH_FULL);
                }
            }
            if (fresh1 > 0) wb_atomic_add_agg(setsize, (int)f, 1);
            if (fresh2 > 0) wb_atomic_add(&c_nux[f], 1);
            wb_atomic_add_agg(c_sumy, (int)f, y);
        }
    WB_END_PFOR_B
    WB_PFOR(w, nb * xw)
        const uint32_t v = xbits[(w / xw) * WB_TRACE_XWORDS + (w % xw)];
        if (v) wb_atomic_add(&c_nux[bigf[w / xw]], wb_popc32(v));
    WB_END_PFOR

    WB_TSTAMP(a.ctl, 7);
    // ---- P7: periodic duplicate rule (contour_index.py:119-147): for every ordered pair (i1,i2)
    //          with set(i1) a subset of set(i2): equal array lengths -> drop max(i1,i2), else drop the shorter
    //          Almost every pair fails on the first point of i1; the few that pass it (a contour and its copy one period
    //          further east) are queued and their points are checked by the whole workgroup instead of one lane.
    WB_SHARED(int, cand, 2 * WB_TRACE_NCAND);
    WB_SHARED(int, ncand, 2);      // [0] queued pairs, [1] failure flag of the pair being checked
    WB_SINGLE
    ncand[0] = 0;
    WB_END_SINGLE
    WB_PFOR(pair, (int64_t)Ck * Ck)
        const int i1 = (int)(pair / Ck), i2 = (int)(pair - (int64_t)i1 * Ck);
        if (i1 == i2 || setsize[i1] > setsize[i2]) continue;
        const int o1 = c_off[i1], l1 = c_len[i1];
        bool sub = true;
        int checked = 0;
        for (int q = 0; q < l1 && sub; ++q) {
            if (q == 1) {                                      // first point is in: hand the rest to the workgroup
                const int slot = wb_atomic_add(&ncand[0], 1);
                if (slot < WB_TRACE_NCAND) { cand[2 * slot] = i1; cand[2 * slot + 1] = i2; checked = -1; break; }
            }
            const uint32_t pt = cpts[o1 + q];
            if (pk6) {
                sub = wb_set32_has(set32, hmask, ((uint32_t)i2 << (bx + by)) | ((uint32_t)wb_pt_y(pt) << bx) |
                                                     (uint32_t)(wb_pt_x(pt) % a.nlon));
            } else {
                const unsigned long long k1 =
                    ((unsigned long long)i2 << 32) | (uint32_t)((wb_pt_x(pt) % a.nlon) | (wb_pt_y(pt) << 16));
                if (wb_hash_find(hkey, hmask, k1) < 0) sub = false;
            }
        }
        if (sub && checked == 0) {
            const int l2 = c_len[i2];
            const int d = (l1 == l2) ? (i1 > i2 ? i1 : i2) : (l1 < l2 ? i1 : i2);
            dropped[d] = 1;
        }
    WB_END_PFOR
    WB_TSTAMP(a.ctl, 8);
    {
        const int nc = ncand[0] < WB_TRACE_NCAND ? ncand[0] : WB_TRACE_NCAND;
        for (int c = 0; c < nc; ++c) {
            const int i1 = cand[2 * c], i2 = cand[2 * c + 1];
            const int o1 = c_off[i1], l1 = c_len[i1];
            WB_SINGLE
            ncand[1] = 0;
            WB_END_SINGLE
            WB_PFOR(q, l1)
                const uint32_t pt = cpts[o1 + q];
                bool in;
                if (pk6) {
                    in = wb_set32_has(set32, hmask, ((uint32_t)i2 << (bx + by)) | ((uint32_t)wb_pt_y(pt) << bx) |
                                                        (uint32_t)(wb_pt_x(pt) % a.nlon));
                } else {
                    const unsigned long long k1 =
                        ((unsigned long long)i2 << 32) | (uint32_t)((wb_pt_x(pt) % a.nlon) | (wb_pt_y(pt) << 16));
                    in = wb_hash_find(hkey, hmask, k1) >= 0;
                }
                if (!in) ncand[1] = 1;
            WB_END_PFOR
            if (ncand[1] == 0) {
                WB_SINGLE
                const int l2 = c_len[i2];
                const int d = (l1 == l2) ? (i1 > i2 ? i1 : i2) : (l1 < l2 ? i1 : i2);
                dropped[d] = 1;
                WB_END_SINGLE
            } else {
                WB_SYNC();                                     // keep the barrier count of the two branches equal
            }
        }
    }

    WB_TSTAMP(a.ctl, 9);
    // ---- P8: final table = kept and not dropped, order preserved
    int* fin = (int*)t0;  // [Ck] -> exclusive scan
    // move table entries down (read everything first, then write: entries only move to lower slots,
    // but different threads handle different slots, so stage through scratch)
    int* s_off = (int*)spare0;
    int* s_len = (int*)spare1;
    int* s_cl = (int*)t1;
    int* s_nux = (int*)succ;
    int* s_sy = (int*)pred;
    WB_PFOR(f, Ck)
        fin[f] = dropped[f] ? 0 : 1;
        s_off[f] = c_off[f];
        s_len[f] = c_len[f];
        s_cl[f] = c_closed[f];
        s_nux[f] = c_nux[f];
        s_sy[f] = c_sumy[f];
    WB_END_PFOR
    const int Cf = wb_block_scan_excl(fin, Ck);
    WB_PFOR(f, Ck)
        if (!dropped[f]) {
            const int g = fin[f];
            c_off[g] = s_off[f];
            c_len[g] = s_len[f];
            c_closed[g] = s_cl[f];
            c_nux[g] = s_nux[f];
            c_sumy[g] = s_sy[f];
            wb_atomic_max(&a.ctl->max_nux, s_nux[f]);
        }
    WB_END_PFOR
    WB_TSTAMP(a.ctl, 10);
    WB_TSTAMP_FLUSH(a.ctl);
    WB_SINGLE
    a.ncont[job] = Cf;
    WB_END_SINGLE
}

// -------------------------------------------------------------------------------------------- phase-1 summary
// Bounds for phase 2, taken after the batch maximum of nux is final: how many contours reach it and how long they are
// (they are the selected ones unless another rank / batch holds a larger maximum, in which case nothing is selected),
// the closed contours (cutoff candidates), the work items of the all-pairs kernel.
#define WB_PAIR_ROWS 16
#ifndef WB_PAIR_TPB
#define WB_PAIR_TPB 8             // 16-row tiles per work item of the all-pairs kernel
#endif
struct WbSummaryArgs {
    int njobs, cap_seg;
    const int* ncont; const int* c_len; const int* c_nux; const int* c_closed;
    WbCtl* ctl;
};
WB_KERNEL(wb_k_phase1_summary, WbSummaryArgs a) {
    // a lane per job, sums in registers, one LDS atomic per lane and counter, one global atomic per workgroup and counter
    WB_SHARED(unsigned long long, acc, 8);
    WB_SINGLE
    for (int k = 0; k < 8; ++k) acc[k] = 0ull;
    WB_END_SINGLE
    const int mx = a.ctl->max_nux;
    WB_PFOR(t, WB_NT)
        unsigned long long v[7] = {0, 0, 0, 0, 0, 0, 0};
        int longest = 0;
        for (int job = WB_BLOCK_ID * WB_NT + (int)t; job < a.njobs; job += WB_GRID_DIM * WB_NT) {
            const size_t jb = (size_t)job * a.cap_seg;
            const int C = a.ncont[job];
            for (int k = 0; k < C; ++k) {
                const int len = a.c_len[jb + k];
                longest = len > longest ? len : longest;
                v[0] += 1; v[1] += (unsigned long long)len;
                if (a.c_nux[jb + k] == mx) {
                    const int tiles = (len - 1 + WB_PAIR_ROWS - 1) / WB_PAIR_ROWS;
                    v[2] += 1; v[3] += (unsigned long long)len; v[4] += (unsigned long long)((tiles + WB_PAIR_TPB - 1) / WB_PAIR_TPB);
                }
                if (a.c_closed[jb + k]) { v[5] += 1; v[6] += (unsigned long long)len; }
            }
        }
        for (int k = 0; k < 7; ++k) if (v[k]) wb_atomic_add(&acc[k], v[k]);
        if (longest) wb_atomic_max(&acc[7], (unsigned long long)longest);
    WB_END_PFOR
    WB_SINGLE
    if (acc[0]) {
        wb_atomic_max(&a.ctl->max_len, (int)acc[7]);
        wb_atomic_add(&a.ctl->ncont_total, (int)acc[0]);
        wb_atomic_add((unsigned long long*)&a.ctl->npts_total, acc[1]);
        if (acc[2]) {
            wb_atomic_add(&a.ctl->nsel_max, (int)acc[2]);
            wb_atomic_add((unsigned long long*)&a.ctl->npt_sel_max, acc[3]);
            wb_atomic_add(&a.ctl->pair_items_max, (int)acc[4]);
        }
        if (acc[5]) {
            wb_atomic_add(&a.ctl->nclosed, (int)acc[5]);
            wb_atomic_add((unsigned long long*)&a.ctl->npt_closed, acc[6]);
        }
    }
    WB_END_SINGLE
}
