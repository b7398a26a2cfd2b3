// wb_trace_fused.h -- kernel 2 (round 5): segments -> ordered, rounded, de-duplicated contours of one (time step, level) job in
// ONE launch, the job's working set resident in LDS from the first key to the final table.
//
// Restates, for one job, wavebreaking/indices/contour_index.py:103-156 of the reference exactly as the three kernels of
// wb_trace.h do (link -> rank -> finish; they stay: they are the host-simulation bodies' twins for the jobs this kernel
// declines -- cells equal to the level, more segments / chains than its LDS holds, capacity overflows -- and the CPU suite runs
// both).  What is different is the algorithm, chosen so that a job needs a few KB of LDS per thousand segments, a few
// barriers per phase and no global scratch at all:
//   * no merge of the strip lists.  A segment is known by its position g in the CONCATENATION of the job's raster-ordered lists;
//     raster order, where it matters (a closed contour starts behind its raster-last segment, contours are numbered by their
//     raster-first segment: SURVEY.md 8a row C2 (i)-(ii)), is the order of the 27-bit square keys themselves.  The successor of
//     a segment that leaves its square to the right / left is the next / previous entry of its own list, one that leaves
//     upwards / downwards is searched in the same list (the few entries between the two rows), one that crosses a strip's edge
//     in the neighbouring list;
//   * list ranking after Helman & JaJa instead of Wyllie's pointer jumping over every segment: every 16th segment and every
//     chain head is a splitter, a thread walks from its splitter to the next one (sub-lists of a dozen segments), and only the
//     few hundred splitters are ranked by pointer jumping -- O(n) work, ~20 barriers instead of ~50, 6 bytes of LDS per segment
//     instead of 8 + 8;
//   * closed loops: a loop that holds no splitter is found as the segments no walk visited and gets its key descents as
//     splitters; ONE ranking serves loops and open chains (round 6: the ranking word also carries the window's smallest ordinal
//     and the distance to it, which on a loop is the node's place in the loop cut at that node); the loop's raster-last segment is
//     the maximum key of its chain (a 64-bit atomic max of key | segment), and the positions are rotated so that the contour
//     starts behind it;
//   * first-occurrence dedupe (contour_index.py:110-112), distinct-x count (:154: a contour's points are a lattice path, so its
//     distinct x are xmin .. xmax) and the periodic-duplicate rule (:119-147) without hash tables: the raw points are bucketed by a
//     hash of (x mod nlon, y) (counting sort, 2 bytes per point); a bucket holds one or two points, a thread compares a bucket's
//     points in registers, and "does contour i2 hold this point of i1" is a scan of one bucket.
// Round 6: what sets this kernel's rate is the number of JOBS a CU holds -- the launch's LDS is sized by what the context's batches
// have asked for (segments, chains, splitters: wb_api.hip; 36-40 KB = four jobs per CU on the 0.25 degree bench batch, where a
// fixed 52 KB held three), and the body is written as data flow where the compiler would otherwise put an exec-mask branch around
// every LDS load ("Round 6"; profiles/r06_ab_sessions.txt has the steps).
// Development order: written in the WB_PFOR idiom so that the host simulation runs it (tests/test_hostsim_parity.py runs every
// parity case with the option trace_fused = 1 and = 0).
#pragma once
#include "wb_trace.h"

#ifndef WB_FUSED_THREADS
#define WB_FUSED_THREADS 256
#endif
#ifndef WB_FUSED_OCC
#define WB_FUSED_OCC 2              // waves per SIMD the register allocation leaves room for
#endif
#define WB_F_SPLIT 16                // every WB_F_SPLIT-th segment is a splitter
#define WB_F_NIL 0x1FFFu             // 13-bit links: a job of the fused path has at most 8190 segments
#define WB_F_HEAD 0x8000u            // flags in the upper bits of a link word
#define WB_F_SPLITTER 0x4000u
#define WB_F_NONE 0xFFFFFFFFu
#define WB_F_NCAND 128
#if defined(WB_HOSTSIM)
#define WB_F_QREG 1024               // splitters whose ranking word one thread carries across a barrier (the host simulation's single thread: all)
#else
#define WB_F_QREG ((1023 + WB_FUSED_THREADS) / WB_FUSED_THREADS)
#endif
#define WB_F_WALK 32                 // a walk hands over to a new splitter after this many segments (the longest walk sets the phase's duration)

// LDS layout (bytes) for a launch that admits jobs of up to nmax segments, smax splitters, cmax chains
struct WbFusedLayout {
    // region A: keys (link) -> owner / local position (ranking) -> raw points, packed contour | y | x
    // region B: links + flags (link, ranking) -> destination of a segment's point -> bucket order
    // region T: tables
    size_t oA, oB, oT, total;
    size_t t_rawbase, t_closed;      // per contour, from the numbering to the end
    // group 1 (link + ranking) and group 2 (finish) share what follows
    size_t g1_hp, g1_wpre, g1_ndseg, g1_nlen, g1_ncp, g1_nnext, g1_cix, g1_q0, g1_cmin, g1_cmaxk, g1_lpos, g1_ctot, g1_cloop, g1_cnum;
    size_t g2_cnt, g2_keep, g2_blk, g2_newstart, g2_kid, g2_nux, g2_sumy, g2_setsize, g2_nuxc, g2_sumyc, g2_setc, g2_dropped, g2_ckept, g2_fin, g2_tcount, g2_cand;
};
WB_HD inline size_t wb_f_al(size_t v) { return (v + 7) & ~(size_t)7; }
// buckets of the raw points: the largest power of two <= nlon (+ 1: see phase F)
WB_HD inline int wb_f_nbuckets(int nlon) {
    int nb = 1;
    while (2 * nb <= nlon) nb <<= 1;
    return nb;
}
WB_HD inline WbFusedLayout wb_fused_layout(int nmax, int smax, int cmax, int nlon, int nthreads) {
    WbFusedLayout L;
    const size_t R = (size_t)nmax + cmax + 2;
    size_t o = 0;
    L.oA = o; o += wb_f_al(4 * R);
    L.oB = o; o += wb_f_al(2 * R);
    L.oT = o;
    size_t t = 0;
    const size_t C1 = (size_t)cmax + 2;
    L.t_rawbase = t; t += wb_f_al(4 * C1);
    L.t_closed = t; t += wb_f_al(C1);
    size_t g1 = t;
    const size_t S1 = (size_t)smax + 2;
    L.g1_hp = g1; g1 += wb_f_al(4 * (((size_t)nmax + 31) / 32 + 1));
    L.g1_wpre = g1; g1 += wb_f_al(4 * (((size_t)nmax + 31) / 32 + 2));
    L.g1_ndseg = g1; g1 += wb_f_al(2 * S1);
    L.g1_nlen = g1; g1 += wb_f_al(2 * S1);
    L.g1_ncp = g1; g1 += wb_f_al(2 * S1);
    L.g1_nnext = g1; g1 += wb_f_al(2 * S1);
    L.g1_cix = g1; g1 += wb_f_al(4 * S1);
    L.g1_q0 = g1; g1 += wb_f_al(8 * S1);
    L.g1_cmin = g1; g1 += wb_f_al(4 * C1);
    L.g1_cmaxk = g1; g1 += wb_f_al(8 * C1);
    L.g1_lpos = g1; g1 += wb_f_al(2 * C1);
    L.g1_ctot = g1; g1 += wb_f_al(4 * C1);
    L.g1_cloop = g1; g1 += wb_f_al(C1);
    L.g1_cnum = g1; g1 += wb_f_al(2 * C1);
    size_t g2 = t;
    L.g2_cnt = g2; g2 += wb_f_al(4 * ((size_t)wb_f_nbuckets(nlon) + 3));
    L.g2_keep = g2; g2 += wb_f_al(4 * ((R + 31) / 32 + 1));
    L.g2_blk = g2; g2 += wb_f_al(2 * ((R + 31) / 32 + 2));
    L.g2_newstart = g2; g2 += wb_f_al(4 * C1);
    L.g2_kid = g2; g2 += wb_f_al(4 * C1);
    L.g2_nux = g2; g2 += wb_f_al(4 * C1);
    L.g2_sumy = g2; g2 += wb_f_al(4 * C1);
    L.g2_setsize = g2; g2 += wb_f_al(4 * C1);
    L.g2_nuxc = g2; g2 += wb_f_al(4 * C1);
    L.g2_sumyc = g2; g2 += wb_f_al(4 * C1);
    L.g2_setc = g2; g2 += wb_f_al(4 * C1);
    L.g2_dropped = g2; g2 += wb_f_al(C1);
    L.g2_ckept = g2; g2 += wb_f_al(2 * C1);
    L.g2_fin = g2; g2 += wb_f_al(4 * C1);
    {   // the threads' counters; also one entry per contour (its first segment), and one per word of the keep bits (the compaction's scan)
        size_t tc = (size_t)(nthreads > cmax ? nthreads : cmax);
        if ((R + 31) / 32 + 2 > tc) tc = (R + 31) / 32 + 2;
        L.g2_tcount = g2; g2 += wb_f_al(4 * (tc + 2));
    }
    L.g2_cand = g2; g2 += wb_f_al(4 * 2 * WB_F_NCAND);
    L.total = L.oT + (g1 > g2 ? g1 : g2);
    return L;
}

// the reference's arithmetic for a crossing's coordinate along its edge (wb_crossing_point): kept out of line, the fused kernel
// needs it for one crossing in a few hundred
WB_DEV WB_NOINLINE int wb_f_along_exact(int base, double va, double vb, double level) {
    return wb_rint((double)base + wb_fraction(va, vb, level));
}

// position g in the concatenation of a job's lists -> (list, index inside it)
WB_DEV inline void wb_f_locate(const int* sbase, int g, int* s, int* i) {
    int q = 0;
    while (g >= sbase[q + 1]) ++q;
    *s = q;
    *i = g - sbase[q];
}
// ... with the list of the thread's previous element as the starting guess (a thread's elements ascend: amortised O(1))
WB_DEV inline int wb_f_locate_run(const int* sbase, int g, int* srun) {
    int q = *srun;
    while (g < sbase[q]) --q;
    while (g >= sbase[q + 1]) ++q;
    *srun = q;
    return g - sbase[q];
}
// first entry of K[lo, hi) whose square is >= target
WB_DEV inline int wb_f_lower(const uint32_t* K, int lo, int hi, int target) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (wb_key_sq(K[mid]) < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct WbFusedArgs {
    WbTraceArgs t;
    int nmax, smax, cmax;            // what the launch's LDS admits
    int nwork;                       // base strips (lists nwork .. nstrips - 1 are the periodic copies)
    unsigned char* fdone;            // [njobs] 1: this kernel finished the job, 0: the three-kernel chain must
};

WB_KERNEL_LB(wb_k_trace_fused, WB_FUSED_THREADS, WB_FUSED_OCC, WbFusedArgs fa) {
    const WbTraceArgs& a = fa.t;
    const int job = WB_BLOCK_ID;
    if (job >= a.njobs) return;
    const int NS = a.nstrips, W = a.W, nlon = a.nlon, nlat = a.nlat;
    const size_t jb = (size_t)job * a.cap_seg;
    WB_SHARED(int, sbase, WB_MAX_LISTS + 2);
    WB_SHARED(int, ctr, 8);          // [0] bail flag, [1] splitters, [4] queued pairs, [5] pair failure flag, [6] queued successor searches, [7] contours' first segments
    WB_DYN_SHARED_BEGIN
    const WbFusedLayout L = wb_fused_layout(fa.nmax, fa.smax, fa.cmax, nlon, WB_FUSED_THREADS);
    WB_DYN_SHARED(unsigned char, lds, L.total);
    uint32_t* const K = (uint32_t*)(lds + L.oA);               // keys; later raw points
    // (ranking) a link word S[g]: 0x8000 | successor until a walk has visited the segment, then owner << 5 | (position in its
    // sub-list - 1) -- 10 + 5 bits: at most 1023 splitters, sub-lists of at most WB_F_WALK = 32 segments.  The keys stay in
    // region A until the raw points replace them: the chains' smallest / largest keys are read from LDS, not from the lists again.
#define WB_F_UNVISITED 0x8000u
#define WB_F_OWN(g) ((int)(S[g] >> 5))
#define WB_F_LP(g) ((int)(S[g] & 31u) + 1)
    uint32_t* const rawpt = K;                                 // raw points in contour order: contour << (xb + yb) | y << xb | x
    uint16_t* const S = (uint16_t*)(lds + L.oB);               // successor | flags; later: destination of the segment's point; later: bucket order
    uint16_t* const dest = S;
    uint16_t* const order = S;
    unsigned char* const T = lds + L.oT;
    int* const rawbase = (int*)(T + L.t_rawbase);
    unsigned char* const closedn = T + L.t_closed;
    uint32_t* const hp = (uint32_t*)(T + L.g1_hp);             // has-predecessor bits (link); then: splitter bits (ranking)
    uint32_t* const spb = hp;
    int* const wpre = (int*)(T + L.g1_wpre);                   // splitters in front of a word of spb
    uint16_t* const ndseg = (uint16_t*)(T + L.g1_ndseg);
    uint16_t* const nlen = (uint16_t*)(T + L.g1_nlen);
    uint16_t* const ncp = (uint16_t*)(T + L.g1_ncp);
    uint16_t* const nnext = (uint16_t*)(T + L.g1_nnext);
    uint16_t* const ndend = nnext;                             // (the sub-list's end, until it has become the next splitter's ordinal)
    int* const cix = (int*)(T + L.g1_cix);
    unsigned long long* const Q = (unsigned long long*)(T + L.g1_q0);
    uint32_t* const cmin = (uint32_t*)(T + L.g1_cmin);
    unsigned long long* const cmaxk = (unsigned long long*)(T + L.g1_cmaxk);      // a loop's largest key << 16 | the segment that has it
    uint16_t* const lpos = (uint16_t*)(T + L.g1_lpos);
    int* const ctot = (int*)(T + L.g1_ctot);
    unsigned char* const cloop = T + L.g1_cloop;
    uint16_t* const cnum = (uint16_t*)(T + L.g1_cnum);
    int* const cnt = (int*)(T + L.g2_cnt);
    uint32_t* const keepb = (uint32_t*)(T + L.g2_keep);
    uint16_t* const blk = (uint16_t*)(T + L.g2_blk);
    int* const newstart = (int*)(T + L.g2_newstart);
    int* const kid = (int*)(T + L.g2_kid);
    int* const nuxl = (int*)(T + L.g2_nux);
    int* const sumyl = (int*)(T + L.g2_sumy);
    int* const setsize = (int*)(T + L.g2_setsize);
    int* const nuxc = (int*)(T + L.g2_nuxc);                  // per contour BEFORE the >= 4 points filter: the largest x (nuxc) and the smallest (xminc) of its raw points,
    int* const xminc = kid;                                   // (kid[] is not needed before the extents have been taken: see there) the sum of y over the kept points and the distinct (x mod nlon, y) (one bucket scan serves the dedupe and them)
    int* const sumyc = (int*)(T + L.g2_sumyc);
    int* const setc = (int*)(T + L.g2_setc);
    unsigned char* const dropped = T + L.g2_dropped;
    uint16_t* const ckept = (uint16_t*)(T + L.g2_ckept);
    int* const fin = (int*)(T + L.g2_fin);
    int* const tcount = (int*)(T + L.g2_tcount);
    int* const cand = (int*)(T + L.g2_cand);
    // a raw point is one word: contour << (xb + yb) | y << xb | x (what does not fit -- a finer grid, more contours -- is declined)
    const int xb = wb_bits_for(W), yb = wb_bits_for(nlat), sh = xb + yb;
    const uint32_t xmask = (1u << xb) - 1u, ymask = (1u << yb) - 1u;
    const int cmax = (sh <= 28 && (1 << (32 - sh)) < fa.cmax) ? (1 << (32 - sh)) : fa.cmax;

    // ---- the job's lists
    int n = 0;
    bool over = false;
    for (int s = 0; s < NS; ++s) {
        const int c = a.stg_n[job * NS + s];
        if (c > a.cap_strip) over = true;
        n += c;
    }
    WB_SINGLE
    int run = 0;
    for (int s = 0; s < NS; ++s) { sbase[s] = run; run += a.stg_n[job * NS + s]; }
    sbase[NS] = run;
    sbase[NS + 1] = 0x7FFFFFFF;
    for (int k = 0; k < 8; ++k) ctr[k] = 0;
    if (n > 0 && !over && n <= a.cap_seg) wb_atomic_max(&a.ctl->max_nseg, n);
    WB_END_SINGLE
    // declined: the three-kernel chain takes the job (capacity overflows are reported by its link kernel)
#define WB_F_DECLINE()                                   \
    do {                                                 \
        WB_SINGLE                                        \
        fa.fdone[job] = 0;                               \
        a.ncont[job] = 0;                                \
        wb_atomic_add(&a.ctl->n_unfused, 1);             \
        WB_END_SINGLE                                    \
        return;                                          \
    } while (0)
    if (n == 0) {
        WB_SINGLE
        fa.fdone[job] = 1;
        a.ncont[job] = 0;
        a.nseg[job] = 0;
        WB_END_SINGLE
        return;
    }
    if (over || n > a.cap_seg || n > fa.nmax || n >= (int)WB_F_NIL || sh > 28) WB_F_DECLINE();
    const double level = a.levels[job % a.nlev];
    const int NSB = fa.nwork;

    WB_FSTAMP_BEGIN                              // development stamps: 0 keys, 1 link, 2 splitters + first walk, 3 loops' walk + splitter links, 4 splitter ranking + chains, 5 key pass + loops, 6 numbering + destinations, 7 crossing points, 8 buckets, 9 dedupe + compaction, 10 per-contour counts, 11 duplicate rule + table
    // ---- A: keys to LDS
    int srun = 0;                                               // (wb_f_locate_run: the list of this thread's previous element)
    {
        constexpr int BW = 16;                                  // (a thread's keys -- fifteen of a bench job's 3 800 -- requested together: one memory latency, not four)
        for (int g0 = 0; g0 < n; g0 += BW * WB_NT) {
            uint32_t ka[BW];
#pragma unroll
            for (int u = 0; u < BW; ++u) {
                const int g = g0 + u * WB_NT + WB_TID;
                const int i = wb_f_locate_run(sbase, g < n ? g : n - 1, &srun);
                ka[u] = a.stg_k[((size_t)job * NS + srun) * a.cap_strip + i];
            }
#pragma unroll
            for (int u = 0; u < BW; ++u) {
                const int g = g0 + u * WB_NT + WB_TID;
                if (g < n) K[g] = ka[u];
            }
        }
        for (int w = WB_TID; w < (n + 31) / 32 + 1; w += WB_NT) hp[w] = 0u;
        WB_SYNC();
    }

    WB_FSTAMP(a.ctl, 0);
    // ---- B: successor of every segment = the segment of the neighbouring square that starts on the edge this one ends on.
    // Every lane does the SAME thing (a wave runs the sum of its lanes' branches, and a data-dependent search loop per branch
    // cost 40 us per job): the four entries of its own list next to the segment -- behind it when it leaves its square to the
    // right or downwards, in front of it when to the left or upwards -- are requested together and hold the successor, or show
    // that there is none, for nearly every segment (the entries between two rows of a strip are few).  What they leave open --
    // the target square lies in another strip's list, or further away -- goes to a short queue that is searched afterwards.
    const float invW = 1.0f / (float)W;
    uint16_t* const queue = (uint16_t*)Q;                       // (the splitters' ranking words are not in use yet: four entries per splitter)
    const int QMAX = 4 * fa.smax;
    // square and side a segment leads to -> the list that holds the square, or -1 (off the grid: the contour ends there)
    auto aim = [&](const uint32_t k, int* target, int* want) -> int {
        const int sq = wb_key_sq(k), cs = wb_key_case(k), sub = wb_key_sub(k);
        int r = (int)(((float)sq + 0.5f) * invW), c = sq - r * W;     // sq / W without the integer division (sq < 2^24: exact up to the fix-up)
        if (c < 0) { --r; c += W; } else if (c >= W) { ++r; c -= W; }
        const int ts = wb_seg_to(cs, sub);
        int rr = r, cc = c;
        if (ts == WB_TOP) rr = r - 1;
        else if (ts == WB_RIGHT) cc = c + 1;
        else if (ts == WB_BOTTOM) rr = r + 1;
        else cc = c - 1;
        *want = ts ^ 2;
        *target = rr * W + cc;
        if (!(rr >= 0 && rr < nlat - 1 && cc >= 0 && cc < W - 1)) return -1;
        const int tl = cc < nlon ? cc / WB_STRIP : NSB + (cc - nlon) / WB_STRIP;
        return tl < NS ? tl : -1;
    };
    // the segment of square entry f (its first) that starts on side `want`
    auto pick = [&](const int f, const int want) -> uint32_t {
        const int cs2 = wb_key_case(K[f]);
        if (wb_seg_from(cs2, 0) == want) return (uint32_t)f;
        if (wb_case_nseg(cs2) == 2 && wb_seg_from(cs2, 1) == want) return (uint32_t)f + 1u;
        return WB_F_NIL;
    };
    // (The main loop knows nothing of the lists' ends: a square lives in ONE list, so an entry of the window that belongs to a
    // neighbouring list can never equal the target, and "at or beyond the target" read off such an entry is either right -- the
    // window has left the segment's own list without meeting the target: there is no such square -- or leaves the segment to the
    // queue.  The segment's own list and the target's are told apart by their columns.)
    // Written as data flow -- flags combined with & and |, values chosen by flags, every load unconditional from a clamped position:
    // the compiler turns `if`s and short-circuit conditions around LDS loads into exec-mask branches, a dozen of them per segment.
    static_assert(WB_STRIP == 256, "the strip of a column is taken by a shift");
    WB_PFOR32(g, n)
        const uint32_t k = K[g];
        const int sq = wb_key_sq(k), cs = wb_key_case(k), sub = wb_key_sub(k);
        int r = (int)(((float)sq + 0.5f) * invW), c = sq - r * W;     // sq / W without the integer division (sq < 2^24: exact up to the fix-up)
        const int adj = (c >= W ? 1 : 0) - (c < 0 ? 1 : 0);
        r += adj; c -= adj * W;
        const int ts = wb_seg_to(cs, sub), want = ts ^ 2;
        const int fwd = (ts == WB_RIGHT ? 1 : 0) | (ts == WB_BOTTOM ? 1 : 0), step = 2 * fwd - 1;
        const int rr = r + (ts == WB_BOTTOM ? 1 : 0) - (ts == WB_TOP ? 1 : 0), cc = c + (ts == WB_RIGHT ? 1 : 0) - (ts == WB_LEFT ? 1 : 0);
        const int target = rr * W + cc;
        const int tl = cc < nlon ? (cc >> 8) : NSB + ((cc - nlon) >> 8);
        const int ls = c < nlon ? (c >> 8) : NSB + ((c - nlon) >> 8);
        const int there = (rr >= 0 ? 1 : 0) & (rr < nlat - 1 ? 1 : 0) & (cc >= 0 ? 1 : 0) & (cc < W - 1 ? 1 : 0) & (tl < NS ? 1 : 0);      // (0: off the grid, the contour ends)
        const int b = fwd ? (int)g - sub + wb_case_nseg(cs) : (int)g - sub - 1;
        uint32_t kw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int at = b + q * step;
            at = at < 0 ? 0 : at;
            kw[q] = K[at > n - 1 ? n - 1 : at];
        }
        int undecided = there & (tl == ls ? 1 : 0), hit = -1, hcs = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int at = b + q * step, sq2 = wb_key_sq(kw[q]);
            const int inside = (at >= 0 ? 1 : 0) & (at < n ? 1 : 0);
            const int beyond = (inside ^ 1) | (fwd ? (sq2 >= target ? 1 : 0) : (sq2 <= target ? 1 : 0));       // at or beyond the target, seen from the segment
            const int found = undecided & inside & (sq2 == target ? 1 : 0);
            hit = found ? at - (fwd ? 0 : wb_key_sub(kw[q])) : hit;                        // (the square's first segment)
            hcs = found ? wb_key_case(kw[q]) : hcs;                                          // (both segments of a square carry its case)
            undecided &= beyond ^ 1;
        }
        bool open = (there & ((tl == ls ? 0 : 1) | undecided)) != 0;                       // another strip's list, or further away than the window
        const int f0 = (hit >= 0 ? 1 : 0) & (wb_seg_from(hcs, 0) == want ? 1 : 0);
        const int f1 = (hit >= 0 ? 1 : 0) & (f0 ^ 1) & (wb_case_nseg(hcs) == 2 ? 1 : 0) & (wb_seg_from(hcs, 1) == want ? 1 : 0);
        uint32_t sc = f0 ? (uint32_t)hit : (f1 ? (uint32_t)hit + 1u : (uint32_t)WB_F_NIL);
        const int slot = wb_append_slot(&ctr[6], open);
        if (open) {
            if (slot < QMAX) queue[slot] = (uint16_t)g;
            else {                                                 // (queue full: searched here)
                const int q = wb_f_lower(K, sbase[tl], sbase[tl + 1], target);
                if (q < sbase[tl + 1] && wb_key_sq(K[q]) == target) sc = pick(q, want);
                open = false;
            }
        }
        if (!open) {
            S[g] = (uint16_t)(sc | WB_F_UNVISITED);
            if (sc != WB_F_NIL) wb_atomic_or(&hp[sc >> 5], 1u << (sc & 31u));
        }
    WB_END_PFOR
    WB_FSTAMP_FINE(a.ctl, 12);                   // (fine stamps: 12 the successors' main loop, 1 the queue; 13 splitters, 14 first walk, 15 loops' splitters + walk; 16 raw points into the buckets' counters, 8 the rest of the buckets; 17 dedupe scan, 18 compaction counts, 19 compaction, 9 the rest)
    {
        const int nq = ctr[6] < QMAX ? ctr[6] : QMAX;
        WB_PFOR32(qi, nq)
            const int g = (int)queue[qi];
            int target, want;
            const int tl = aim(K[g], &target, &want);
            uint32_t sc = WB_F_NIL;
            const int q = wb_f_lower(K, sbase[tl], sbase[tl + 1], target);
            if (q < sbase[tl + 1] && wb_key_sq(K[q]) == target) sc = pick(q, want);
            S[g] = (uint16_t)(sc | WB_F_UNVISITED);
            if (sc != WB_F_NIL) wb_atomic_or(&hp[sc >> 5], 1u << (sc & 31u));
        WB_END_PFOR
    }

    WB_FSTAMP(a.ctl, 1);
    // ---- C: list ranking.  C1: splitters = chain heads (no predecessor) + every 16th segment, a word of 32 flags at a time:
    // the has-predecessor bits turn into the splitter bits in place; a splitter's ordinal is the number of flags in front of it
    const int nwords = (n + 31) / 32;
    WB_PFOR32(w, nwords)
        const uint32_t valid = ((int)w + 1) * 32 <= n ? 0xFFFFFFFFu : ((1u << (n & 31)) - 1u);
        const uint32_t f = (~hp[w] | 0x00010001u) & valid;           // (WB_F_SPLIT = 16: bits 0 and 16 of every word)
        spb[w] = f;
        wpre[w] = wb_popc32(f);
    WB_END_PFOR
    const int S1 = wb_block_scan_excl(wpre, nwords);
    // (what a job asked for sizes the context's next launch -- also when this one declines it)
#define WB_F_NOTE(field, v) do { if (WB_TID == 0) wb_atomic_max(&a.ctl->field, (int)(v)); } while (0)
    if (S1 > fa.smax || fa.smax > 1023) { WB_F_NOTE(max_splitters, S1 + S1 / 4 + 16); WB_F_DECLINE(); }
    WB_PFOR32(w, nwords)
        if (w == 0) ctr[1] = S1;                                  // (the walks append splitters of their own behind these)
        uint32_t f = spb[w];
        int o = wpre[w];
        while (f) {
            const int bit = wb_ctz32(f);
            f &= f - 1u;
            ndseg[o++] = (uint16_t)((int)w * 32 + bit);
        }
    WB_END_PFOR
    // C2: a thread walks from its splitter to the next one (no two walks meet: successors are one-to-one).  The longest walk
    // sets the phase's duration: after WB_F_WALK segments the walk makes the next segment a splitter of its own and goes on as
    // that one.  A visited segment's link word becomes owner | position (the walk has read the successor before).
    // (The splitter flag read off the next segment's own link word -- flagged while unvisited, position 0 once visited -- instead of
    // the separate bit array: one LDS read less per step, and the walks took 44 ms of workgroup time per launch instead of 36.)
    WB_FSTAMP_FINE(a.ctl, 13);
    auto walk = [&](int o) {
        int j = (int)ndseg[o], cntw = 0;
        uint32_t end = WB_F_NIL, w = S[j];
        for (;;) {
            S[j] = (uint16_t)(((uint32_t)o << 5) | (uint32_t)cntw);    // (position - 1 = cntw before the increment)
            ++cntw;
            const uint32_t nx = w & WB_F_NIL;
            if (nx == WB_F_NIL) break;
            const uint32_t fl = spb[nx >> 5], wn = S[nx];                // (independent reads: the next segment's flag word and its link)
            if ((fl >> (nx & 31u)) & 1u) { end = nx; break; }
            j = (int)nx;
            w = wn;
            if (cntw == WB_F_WALK) {
                const int o2 = wb_atomic_add(&ctr[1], 1);
                if (o2 >= fa.smax) { ctr[0] = 1; end = nx; break; }
                nlen[o] = (uint16_t)cntw;
                ndend[o] = (uint16_t)nx;
                o = o2;
                ndseg[o] = (uint16_t)nx;
                cntw = 0;
            }
        }
        nlen[o] = (uint16_t)cntw;
        ndend[o] = (uint16_t)end;
    };
    WB_PFOR32(o, S1)
        walk((int)o);
    WB_END_PFOR
    WB_FSTAMP_FINE(a.ctl, 14);
    // C3: segments no walk reached lie on closed loops without a splitter: their key descents become splitters
    const int S2 = ctr[1];
    if (ctr[0] || S2 > fa.smax) { WB_F_NOTE(max_splitters, 2 * fa.smax); WB_F_DECLINE(); }      // (how many the walks would have added is not known: twice as many next time)
    const uint16_t NN = (uint16_t)0xFFFF;                      // "no node"
    WB_PFOR32(g, n > fa.smax + 2 ? n : fa.smax + 2)
        if (g < fa.smax + 2) ncp[g] = NN;                         // (C4's previous-splitter links start empty: no loop of their own for that)
        if (g < n) {
            bool want = false;
            const uint32_t w = S[g];
            if (w & WB_F_UNVISITED) {
                const uint32_t nx = w & WB_F_NIL;
                want = nx == WB_F_NIL || (K[g] & 0x07FFFFFFu) > (K[nx] & 0x07FFFFFFu);
            }
            const int slot = wb_append_slot(&ctr[1], want);
            if (want && slot < fa.smax) { ndseg[slot] = (uint16_t)g; wb_atomic_or(&spb[g >> 5], 1u << (g & 31)); }
        }
    WB_END_PFOR
    const int S3 = ctr[1];
    if (S3 > fa.smax) { WB_F_NOTE(max_splitters, S3 + S3 / 8); WB_F_DECLINE(); }
    WB_PFOR32(o, S3 - S2)
        walk(S2 + (int)o);
    WB_END_PFOR
    const int NSP = ctr[1];
    if (ctr[0] || NSP > fa.smax) { WB_F_NOTE(max_splitters, 2 * fa.smax); WB_F_DECLINE(); }
    WB_FSTAMP_FINE(a.ctl, 15);
    // C4: the splitters' own list: next splitter, previous splitter
    WB_PFOR32(o, NSP)
        const uint16_t e = ndend[o];
        const uint16_t nx = e == (uint16_t)WB_F_NIL ? NN : (uint16_t)WB_F_OWN(e);
        nnext[o] = nx;                                         // (in place: ndend is nnext's storage)
        if (nx != NN) ncp[nx] = (uint16_t)o;
    WB_END_PFOR
    WB_FSTAMP(a.ctl, 3);
    // C5: pointer jumping over the splitters along the PREVIOUS links.  One 64-bit word per node:
    //   J (10 bits) jump pointer: the node 2^k nodes in front, or none;  D (13) segments between that node's start and this node's --
    //   once J is none: the segments in front of the node in its chain;  H (10) first node of the chain;
    //   M (10) smallest ordinal among the node and the 2^k - 1 in front of it, DM (13) segments between THAT node's start and this one's.
    // An open chain ends with J = none, named by H, positions from D.  On a loop J stays alive; once the window has gone around,
    // M is the loop's smallest ordinal -- its name -- and DM the node's place in the loop cut in front of that node (the nearest
    // occurrence wins a tie, and D, which then counts laps and overflows its field, is not read any more): ONE ranking serves
    // both, where the loops used to be cut at their raster-last segment and ranked again (another 2 x 9 barriers).
#define WB_FQ_NN 0x3FFu
#define WB_FQ(j, h, m, d, dm) ((unsigned long long)((j) & 0x3FFu) | (unsigned long long)((h) & 0x3FFu) << 10 | (unsigned long long)((m) & 0x3FFu) << 20 | (unsigned long long)((d) & 0x1FFFu) << 30 | (unsigned long long)((dm) & 0x1FFFu) << 43)
#define WB_FQ_J(w) ((uint32_t)(w) & 0x3FFu)
#define WB_FQ_H(w) ((uint32_t)((w) >> 10) & 0x3FFu)
#define WB_FQ_M(w) ((uint32_t)((w) >> 20) & 0x3FFu)
#define WB_FQ_D(w) ((uint32_t)((w) >> 30) & 0x1FFFu)
#define WB_FQ_DM(w) ((uint32_t)((w) >> 43) & 0x1FFFu)
    int rounds = 0;
    while ((1 << rounds) < NSP) ++rounds;
    // In place: a thread holds the new words of its nodes (at most WB_F_QREG: smax <= 1023) across a barrier -- a second array of
    // words was 3 KB of the job's LDS, and the LDS decides how many jobs a CU holds.  (ONE wave ranking all the nodes without any
    // workgroup barrier -- a wave's LDS operations are served in order -- took 31 ms of workgroup time per launch instead of 20:
    // five nodes per lane and round, two dependent reads each.)
    {
        WB_PFOR32(o, NSP)
            const uint16_t p = ncp[o];
            Q[o] = WB_FQ(p == NN ? WB_FQ_NN : (uint32_t)p, (uint32_t)o, (uint32_t)o, p != NN ? (uint32_t)nlen[p] : 0u, 0u);
        WB_END_PFOR
        for (int it = 0; it < rounds; ++it) {
            unsigned long long nwq[WB_F_QREG];
#pragma unroll
            for (int u = 0; u < WB_F_QREG; ++u) {
                const int o = WB_TID + u * WB_NT;
                if (o >= NSP) break;
                const unsigned long long w = Q[o];
                const uint32_t j = WB_FQ_J(w);
                unsigned long long nw = w;
                if (j != WB_FQ_NN) {
                    const unsigned long long wj = Q[j];
                    const uint32_t d = WB_FQ_D(w), m1 = WB_FQ_M(w), m2 = WB_FQ_M(wj);
                    const bool far = m2 < m1;
                    nw = WB_FQ(WB_FQ_J(wj), WB_FQ_H(wj), far ? m2 : m1, d + WB_FQ_D(wj), far ? WB_FQ_DM(wj) + d : WB_FQ_DM(w));
                }
                nwq[u] = nw;
            }
            WB_SYNC();
#pragma unroll
            for (int u = 0; u < WB_F_QREG; ++u) {
                const int o = WB_TID + u * WB_NT;
                if (o >= NSP) break;
                Q[o] = nwq[u];
            }
            WB_SYNC();
        }
    }
    // chains: an open chain is named by its first node, a loop (jump pointer still alive) by its smallest ordinal
    WB_PFOR32(o, NSP)
        const unsigned long long w = Q[o];
        const bool loop = WB_FQ_J(w) != WB_FQ_NN;
        const uint32_t id = loop ? WB_FQ_M(w) : WB_FQ_H(w);
        cix[o] = id == (uint32_t)o ? 1 : 0;
    WB_END_PFOR
    const int C = wb_block_scan_excl(cix, NSP);               // cix[o] (for a naming node): the chain's index
    if (C > cmax) { WB_F_NOTE(max_chains, C); WB_F_DECLINE(); }
    // every node learns its chain's index (the naming node's entry is read before anything overwrites it: two steps)
    WB_PFOR32(o, NSP)
        const unsigned long long w = Q[o];
        const bool loop = WB_FQ_J(w) != WB_FQ_NN;
        const uint32_t id = loop ? WB_FQ_M(w) : WB_FQ_H(w);
        nnext[o] = (uint16_t)cix[id];                          // (nnext is done with: chain index of the node, for now)
        if (id == (uint32_t)o) { cloop[cix[o]] = loop ? 1 : 0; cmin[cix[o]] = 0xFFFFFFFFu; cmaxk[cix[o]] = 0ull; ctot[cix[o]] = 0; }
    WB_END_PFOR
    // ... and its chain the number of its segments: the largest (segments in front of a node + its own)
    WB_PFOR32(o, NSP)
        const unsigned long long w = Q[o];
        const int ch = (int)nnext[o];
        cix[o] = ch;
        wb_atomic_max(&ctot[ch], (int)(WB_FQ_J(w) != WB_FQ_NN ? WB_FQ_DM(w) : WB_FQ_D(w)) + (int)nlen[o]);
    WB_END_PFOR
    WB_FSTAMP(a.ctl, 4);
    // the keys again: every chain's smallest key (it numbers the contours: contour_index.py:103, find_contours' list order) and every
    // loop's largest WITH its segment -- the loop's raster-last segment: the contour starts behind it
    WB_PFOR32(g, n)
        const uint32_t key = K[g] & 0x07FFFFFFu;
        const int ch = cix[WB_F_OWN(g)];
        wb_atomic_min(&cmin[ch], key);
        if (cloop[ch]) wb_atomic_max(&cmaxk[ch], (unsigned long long)key << 16 | (unsigned long long)g);
    WB_END_PFOR
    WB_PFOR32(ch, C)
        if (cloop[ch]) {                                       // the place of that segment in the loop as the ranking cut it
            const int g = (int)(cmaxk[ch] & 0xFFFFull);
            lpos[ch] = (uint16_t)((int)WB_FQ_DM(Q[WB_F_OWN(g)]) + WB_F_LP(g));
        }
    WB_END_PFOR
    WB_FSTAMP(a.ctl, 5);
    // ---- D: contours in the order of their raster-first segments (contour_index.py:103: find_contours' list order)
    WB_PFOR32(ch, C)
        const uint32_t mine = cmin[ch];
        int num = 0;
        for (int c2 = 0; c2 < C; ++c2) num += cmin[c2] < mine ? 1 : 0;
        cnum[ch] = (uint16_t)num;
        rawbase[num] = ctot[ch] + 1;                           // raw points of the contour: its segments' end points + the first start point
        closedn[num] = cloop[ch];
    WB_END_PFOR
    const int R = wb_block_scan_excl(rawbase, C);
    // where every segment's end point goes (bit 15: the segment is its contour's first -- its start point goes in front).
    // Reads region A (own / lp) and the ranking's tables, writes region B.
    WB_PFOR32(g, n)
        if (g == 0) rawbase[C] = R;
        const int o = WB_F_OWN(g), ch = cix[o];
        const unsigned long long w = Q[o];
        int q = WB_F_LP(g);                                      // position along the chain, from 1 (dest is S's storage: read, then written, by this thread)
        if (cloop[ch]) {                                         // (a loop: rotated so that it starts behind its raster-last segment)
            const int pl = (int)lpos[ch], tot = ctot[ch];
            q += (int)WB_FQ_DM(w);
            q = q > pl ? q - pl : q - pl + tot;
        } else q += (int)WB_FQ_D(w);
        dest[g] = (uint16_t)((rawbase[cnum[ch]] + q) | (q == 1 ? 0x8000 : 0));
    WB_END_PFOR
    if (R >= 0x8000) WB_F_DECLINE();                            // (cannot happen: n < 8191, C <= cmax)
    // the contour of raw point 32 b, for every b (the ranking's tables are done with: group 2 of the tables from here on)
    WB_PFOR32(b, (R + 31) / 32 + 1)
        const int p = (int)b * 32;
        int lo = 0, hi = C;                                      // first contour whose end lies behind p
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (rawbase[mid + 1] <= p) lo = mid + 1; else hi = mid;
        }
        blk[b] = (uint16_t)lo;
    WB_END_PFOR
    WB_FSTAMP(a.ctl, 6);
    // ---- E: the crossing points, rounded half-to-even (contour_index.py:111), straight to their places in contour order.
    // Region A becomes the raw points (own / lp are done with).
    auto pack = [&](const int p, const uint32_t pt) -> uint32_t {
        int c = (int)blk[p >> 5];
        while (rawbase[c + 1] <= p) ++c;
        return ((uint32_t)c << sh) | ((uint32_t)wb_pt_y(pt) << xb) | (uint32_t)wb_pt_x(pt);
    };
    // The lattice point of a crossing is base + (0 or 1) along its edge: the reference rounds base + f, f = (level - a) / (b - a) in
    // float64 (wb_crossing_point), and unless f is within 1e-3 of one half a float32 quotient says on which side of base + 1/2 that
    // sum lies (|error| < 1e-6: the two differences are exact or correctly rounded float64, f lies in [0, 1] on an edge the contour
    // crosses).  The exact arithmetic -- a float64 division, a sixth of this kernel's instructions -- is kept for the quotients near
    // one half, infinite or tiny differences and whatever else fails the test (NaN fails every comparison).
    auto crossing = [&](const int side, const int r, const int c, const double ul, const double ur, const double ll, const double lr) -> uint32_t {
        // (flags and selections, no branch but the rare exact evaluation: see the successors' loop)
        const int top = side == WB_TOP ? 1 : 0, bot = side == WB_BOTTOM ? 1 : 0, lef = side == WB_LEFT ? 1 : 0;
        const int horiz = top | bot;                                   // the crossing moves along a row: its column is the fraction
        const double va = (top | lef) ? ul : (bot ? ll : ur);
        const double vb = top ? ur : (lef ? ll : lr);
        const int base = horiz ? c : r, other = horiz ? r + bot : c + (lef ^ 1);
        const float numf = (float)(level - va), denf = (float)(vb - va);
        const float fa = numf * wb_rcp_f32(denf), ad = fabsf(denf);
        const int fast = (fabsf(fa - 0.5f) > 1e-3f ? 1 : 0) & (fa > -0.01f ? 1 : 0) & (fa < 1.01f ? 1 : 0) & (ad > 1e-30f ? 1 : 0) & (ad < 1e30f ? 1 : 0);
        int along = base + (fa > 0.5f ? 1 : 0);
        if (!fast) along = wb_f_along_exact(base, va, vb, level);
        const uint32_t lo = horiz ? (uint32_t)along : (uint32_t)other, hi = horiz ? (uint32_t)other : (uint32_t)along;
        return lo | (hi << 16);
    };
    int* const firsts = tcount;                                   // (at most one per contour: wb_fused_layout sizes the threads' counters for that too)
    auto crossings = [&](auto tag) {
        typedef decltype(tag) T;
        struct alignas(16) V4 { T v[4]; };
        const V4* const V = (const V4*)a.stg_v;
        constexpr int BW = 8;                                        // (a segment's key and the four corner values of its square, eight segments' worth requested together)
        for (int g0 = 0; g0 < n; g0 += BW * WB_NT) {
            uint32_t keya[BW];
            V4 va[BW];
#pragma unroll
            for (int u = 0; u < BW; ++u) {
                const int g = g0 + u * WB_NT + WB_TID, gc = g < n ? g : n - 1;
                const int i = wb_f_locate_run(sbase, gc, &srun);
                const size_t e = ((size_t)job * NS + srun) * a.cap_strip + (size_t)i;
                keya[u] = a.stg_k[e];
                va[u] = V[e];
            }
#pragma unroll
            for (int u = 0; u < BW; ++u) {
                const int g = g0 + u * WB_NT + WB_TID;
                if (g >= n) break;
                const uint32_t key = keya[u];
                const double ul = (double)va[u].v[0], ur = (double)va[u].v[1], ll = (double)va[u].v[2], lr = (double)va[u].v[3];
                // a corner equal to the level puts crossings ON grid vertices (skimage joins those by coordinate): the chain of wb_trace.h
                if ((ul == level ? 1 : 0) | (ur == level ? 1 : 0) | (ll == level ? 1 : 0) | (lr == level ? 1 : 0)) ctr[0] = 1;
                const int sq = wb_key_sq(key), cs = wb_key_case(key), sub = wb_key_sub(key);
                int r = (int)(((float)sq + 0.5f) * invW), c = sq - r * W;
                const int adj = (c >= W ? 1 : 0) - (c < 0 ? 1 : 0);
                r += adj; c -= adj * W;
                const uint16_t d = dest[g];
                const int p = (int)(d & 0x7FFFu);
                rawpt[p] = pack(p, crossing(wb_seg_to(cs, sub), r, c, ul, ur, ll, lr));
                if (d & 0x8000u) firsts[wb_atomic_add(&ctr[7], 1)] = g;      // (a contour's first segment: its start point goes in front, below)
            }
        }
        WB_SYNC();
        WB_PFOR32(f, ctr[7])
            const int g = firsts[f];
            int s1 = 0;
            const int i = wb_f_locate_run(sbase, g, &s1);
            const size_t e = ((size_t)job * NS + s1) * a.cap_strip + (size_t)i;
            const uint32_t key = a.stg_k[e];
            const V4 v = V[e];
            const int sq = wb_key_sq(key), cs = wb_key_case(key), sub = wb_key_sub(key);
            int r = (int)(((float)sq + 0.5f) * invW), c = sq - r * W;
            if (c < 0) { --r; c += W; } else if (c >= W) { ++r; c -= W; }
            const int p = (int)(dest[g] & 0x7FFFu) - 1;
            rawpt[p] = pack(p, crossing(wb_seg_from(cs, sub), r, c, (double)v.v[0], (double)v.v[1], (double)v.v[2], (double)v.v[3]));
        WB_END_PFOR
    };

    if (a.dtype == 0) crossings(0.0f); else crossings(0.0);
    if (ctr[0]) WB_F_DECLINE();

    WB_FSTAMP(a.ctl, 7);
    // ---- F: buckets of the raw points by (x mod nlon, y).  A point equal to its predecessor in
    // the same contour can never be a first occurrence (about half of all raw points: neighbouring crossings that round to the
    // same lattice point): it stays out.
    const int NBP = wb_f_nbuckets(nlon);                         // buckets: the largest power of two the counters hold
    const int NB = NBP + 1;
    {
        const int kw0 = (R + 31) / 32 + 1, n0 = NB + 1 > kw0 ? (NB + 1 > C + 1 ? NB + 1 : C + 1) : (kw0 > C + 1 ? kw0 : C + 1);
        WB_PFOR32(x, n0)                                          // (one loop, one barrier: the bucket counters, the keep bits, the contours' counters)
            if (x < NB + 1) cnt[x] = 0;
            if (x < kw0) keepb[x] = 0u;
            if (x < C + 1) { newstart[x] = 0; nuxl[x] = 0; sumyl[x] = 0; setsize[x] = 0; dropped[x] = 0; nuxc[x] = -1; xminc[x] = 0x7FFFFFFF; sumyc[x] = 0; setc[x] = 0; }
        WB_END_PFOR
    }
#define WB_F_XM(pw) (((pw) & xmask) >= (uint32_t)nlon ? ((pw) & xmask) - (uint32_t)nlon : ((pw) & xmask))   /* x mod nlon: x < W <= 2 nlon */
#define WB_F_BUCKET(pw) ((int)((WB_F_XM(pw) + (((pw) >> xb) & ymask) * 347u) & (uint32_t)(NBP - 1)))    /* by (x mod nlon, y) only: what the dedupe, the (x mod nlon, y) sets and in_set compare */
    // A thread owns a RUN of consecutive raw points (fifteen of a bench job's): the predecessor is the word it read last (a point equal to
    // its predecessor stays out of the buckets: no flag array), and the extent in x of the contour it is in is kept in registers and
    // goes to LDS when the run leaves the contour -- one LDS read per point where a point per thread and step read four.
    const int chunk = (R + WB_NT - 1) / WB_NT;
    WB_PFOR32(t, WB_NT)
        const int p0 = (int)t * chunk, p1 = p0 + chunk < R ? p0 + chunk : R;
        if (p0 < p1) {
            uint32_t prev = p0 > 0 ? rawpt[p0 - 1] : 0u;
            int curc = -1, xmax = 0, xmin = 0;
#pragma unroll 4
            for (int p = p0; p < p1; ++p) {
                const uint32_t pw = rawpt[p];
                const bool dup = p > 0 && pw == prev;                     // (equal words: the same contour and the same lattice point)
                prev = pw;
                if (!dup) {
                    wb_atomic_add(&cnt[WB_F_BUCKET(pw)], 1);
                    // the contour's extent in x: its points form a lattice path (consecutive crossings of the marching squares lie on
                    // the edges of one cell and round to corners of that cell: |dx| <= 1), so its distinct x are the integers xmin .. xmax
                    const int c = (int)(pw >> sh), x = (int)(pw & xmask);
                    if (c != curc) {
                        if (curc >= 0) { wb_atomic_max(&nuxc[curc], xmax); wb_atomic_min(&xminc[curc], xmin); }
                        curc = c; xmax = x; xmin = x;
                    } else { xmax = x > xmax ? x : xmax; xmin = x < xmin ? x : xmin; }
                }
            }
            if (curc >= 0) { wb_atomic_max(&nuxc[curc], xmax); wb_atomic_min(&xminc[curc], xmin); }
        }
    WB_END_PFOR
    WB_FSTAMP_FINE(a.ctl, 16);
    const int U = wb_block_scan_excl(cnt, NB);                  // cnt[b]: first entry of bucket b
    WB_PFOR32(t, WB_NT)
        const int p0 = (int)t * chunk, p1 = p0 + chunk < R ? p0 + chunk : R;
        if (p0 < p1) {
            uint32_t prev = p0 > 0 ? rawpt[p0 - 1] : 0u;
#pragma unroll 4
            for (int p = p0; p < p1; ++p) {
                const uint32_t pw = rawpt[p];
                const bool dup = p > 0 && pw == prev;
                prev = pw;
                if (!dup) {
                    const int e = wb_atomic_add(&cnt[WB_F_BUCKET(pw)], 1);          // (afterwards cnt[b] = END of bucket b = start of bucket b + 1)
                    order[e] = (uint16_t)p;
                }
            }
        }
    WB_END_PFOR
    WB_FSTAMP(a.ctl, 8);
    // first-occurrence dedupe (contour_index.py:110-112: dict.fromkeys over the contour's points) -- and, in the same scan of the
    // point's bucket, the contour's distinct (x mod nlon, y) and the sum of y over the kept points (-> mean_lat, :155): an entry
    // counts for (x mod nlon, y) when no entry IN FRONT of it in the bucket has them.  (The buckets are by a hash of (x mod nlon, y):
    // one or two entries each.  By x mod nlon alone -- needed while the distinct x were counted here too, -> exp_lon, :154 -- a
    // bucket held every crossing of a meridian by every contour of the job and its copy one period east, forty entries, and this
    // scan was a fifth of the kernel.)
    // A thread takes a BUCKET, not a point: its entries' points (up to eight; a longer bucket is scanned entry by entry) go to
    // registers once and are compared there -- two LDS reads per entry and three round trips per bucket, where every point used to
    // read its whole bucket (five round trips per point).  The sums of the thread's first contour -- with a full-span contour in the
    // job: nearly every point's -- are kept in registers and added once per WAVE (wb_atomic_add_wave): one address per contour is
    // what sixty-four lanes' atomics would queue on.
    (void)U;
    {
        int c0 = -1, accy = 0, accs = 0;
        auto tally = [&](const uint32_t pw, const bool first, const bool firsty, const int p) {
            const int c = (int)(pw >> sh), y = (int)((pw >> xb) & ymask);
            if (first) wb_atomic_or(&keepb[p >> 5], 1u << (p & 31));
            if (c0 < 0) c0 = c;
            if (c == c0) { accy += first ? y : 0; accs += firsty ? 1 : 0; }
            else {
                if (first) wb_atomic_add(&sumyc[c], y);
                if (firsty) wb_atomic_add(&setc[c], 1);
            }
        };
        WB_PFOR32(bk, NBP)
            const int b0 = bk > 0 ? cnt[bk - 1] : 0, b1 = cnt[bk], m = b1 - b0;
            if (m > 0 && m <= 8) {
                int pp[8];
                uint32_t ww[8], cy[8], xm[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) pp[i] = (int)order[b0 + (i < m ? i : m - 1)];
#pragma unroll
                for (int i = 0; i < 8; ++i) { ww[i] = rawpt[pp[i]]; cy[i] = ww[i] >> xb; xm[i] = WB_F_XM(ww[i]); }
                uint32_t nf = 0u, ny = 0u;                       // bit i: entry i is not the first occurrence of its point in its contour / of its (x mod nlon, y) in the bucket
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int j = i + 1; j < 8; ++j) {
                        const uint32_t vj = j < m ? 1u : 0u;
                        const uint32_t eqw = ww[i] == ww[j] ? vj : 0u;
                        const uint32_t eqy = (cy[i] == cy[j] ? vj : 0u) & (xm[i] == xm[j] ? 1u : 0u);
                        nf |= eqw << (pp[i] < pp[j] ? j : i);
                        ny |= eqy << j;
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (i < m) tally(ww[i], !((nf >> i) & 1u), !((ny >> i) & 1u), pp[i]);
            } else if (m > 8) {
                // (one bucket in a hundred of the bench batch, up to nineteen entries: left to the wave that finds it, its m x m
                // LDS reads were the whole phase -- queued, and scanned below an ENTRY per thread)
                const int slot = m <= 32 ? wb_atomic_add(&ctr[3], 1) : WB_F_NCAND;
                if (slot < WB_F_NCAND) { cand[2 * slot] = b0; cand[2 * slot + 1] = m; }
                else {
                    for (int e = b0; e < b1; ++e) {
                        const int p = (int)order[e];
                        const uint32_t pw = rawpt[p];
                        bool first = true, firsty = true;
                        for (int e2 = b0; e2 < b1; ++e2) {
                            const int p2 = (int)order[e2];
                            const uint32_t w2 = rawpt[p2];
                            if (p2 < p && w2 == pw) first = false;
                            if (e2 < e && (w2 >> xb) == (pw >> xb) && WB_F_XM(w2) == WB_F_XM(pw)) firsty = false;
                        }
                        tally(pw, first, firsty, p);
                    }
                }
            }
        WB_END_PFOR
        const int nbig = ctr[3] < WB_F_NCAND ? ctr[3] : WB_F_NCAND;
        WB_PFOR32(t, nbig * 32)
            const int b0 = cand[2 * ((int)t >> 5)], m = cand[2 * ((int)t >> 5) + 1], i = (int)t & 31;
            if (i < m) {
                const int e = b0 + i, b1 = b0 + m, p = (int)order[e];
                const uint32_t pw = rawpt[p], cyp = pw >> xb, xmp = WB_F_XM(pw);
                bool first = true, firsty = true;
                for (int e2 = b0; e2 < b1; e2 += 4) {           // (four entries' positions, then their points: two LDS latencies per four)
                    int p2[4];
                    uint32_t w2[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) p2[q] = (int)order[e2 + q < b1 ? e2 + q : b1 - 1];
#pragma unroll
                    for (int q = 0; q < 4; ++q) w2[q] = rawpt[p2[q]];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (p2[q] < p && w2[q] == pw) first = false;
                        if (e2 + q < e && (w2[q] >> xb) == cyp && WB_F_XM(w2[q]) == xmp) firsty = false;
                    }
                }
                tally(pw, first, firsty, p);
            }
        WB_END_PFOR
        wb_atomic_add_wave(sumyc, c0 < 0 ? 0 : c0, accy);     // (read behind the compaction's barriers)
        wb_atomic_add_wave(setc, c0 < 0 ? 0 : c0, accs);
    }
    WB_FSTAMP_FINE(a.ctl, 17);
    // compaction: the kept points go to the job's output array in their order (the points of contours that turn out too short
    // included, as the finish kernel of wb_trace.h lays them out); a point's place = the kept points in front of it, from the
    // keep bits: a scan over their words' counts + a count inside its word
    uint32_t* const cpts = a.cpts + 2 * jb;
    const int kwords = (R + 31) / 32 + 1;
    WB_PFOR32(w, kwords)
        tcount[w] = wb_popc32(keepb[w]);
    WB_END_PFOR
    const int NK = wb_block_scan_excl(tcount, kwords);
    WB_FSTAMP_FINE(a.ctl, 18);
    WB_PFOR32(p, R > C + 1 ? R : C + 1)
        if (p < R) {
            const uint32_t bits = keepb[p >> 5];
            if ((bits >> (p & 31)) & 1u) {
                const uint32_t pw = rawpt[p];
                cpts[tcount[p >> 5] + wb_popc32(bits & ((1u << (p & 31)) - 1u))] = (pw & xmask) | (((pw >> xb) & ymask) << 16);
            }
        }
        if (p < C + 1) {                                        // (the same loop: the kept points in front of a contour's first raw point)
            const int pc = rawbase[p];                          // (rawbase[C] = R: all the kept points)
            newstart[p] = tcount[pc >> 5] + wb_popc32(keepb[pc >> 5] & ((1u << (pc & 31)) - 1u));
        }
    WB_END_PFOR
    (void)NK;
    WB_FSTAMP_FINE(a.ctl, 19);
    // contours with at least four points stay (contour_index.py:113-117)
    WB_PFOR32(c, C)
        nuxc[c] = nuxc[c] - xminc[c] + 1;                          // distinct x (xminc lies in kid[])
        kid[c] = (newstart[c + 1] - newstart[c]) >= 4 ? 1 : 0;
        fin[c] = kid[c];
    WB_END_PFOR
    const int Ck = wb_block_scan_excl(kid, C);                  // kid[c]: number among the kept ones; fin[c]: kept flag
    // ---- G: the kept contours' counters, by their number among the kept ones
    WB_PFOR32(c, C)
        if (fin[c]) { const int f = kid[c]; ckept[f] = (uint16_t)c; nuxl[f] = nuxc[c]; setsize[f] = setc[c]; sumyl[f] = sumyc[c]; }
    WB_END_PFOR
    WB_FSTAMP(a.ctl, 9);
    // ---- periodic duplicate rule (contour_index.py:119-147): for every ordered pair (i1, i2) with set(i1) a subset of set(i2)
    // (points taken x mod nlon): equal array lengths -> drop max(i1, i2), else drop the shorter.  Almost every pair fails on the
    // first point of i1; the few that pass it are queued and checked by the whole workgroup.
    auto in_set = [&](const uint32_t pw, const int c2) -> bool {
        const uint32_t xm = WB_F_XM(pw), cy = ((uint32_t)c2 << yb) | ((pw >> xb) & ymask);
        const int bk = WB_F_BUCKET(pw), b0 = bk > 0 ? cnt[bk - 1] : 0, b1 = cnt[bk];
        for (int e2 = b0; e2 < b1; ++e2) {
            const uint32_t pw2 = rawpt[order[e2]];
            if ((pw2 >> xb) == cy && WB_F_XM(pw2) == xm) return true;
        }
        return false;
    };
    auto drop_of = [&](const int i1, const int i2) -> int {
        const int c1 = (int)ckept[i1], c2 = (int)ckept[i2];
        const int l1 = newstart[c1 + 1] - newstart[c1], l2 = newstart[c2 + 1] - newstart[c2];
        return (l1 == l2) ? (i1 > i2 ? i1 : i2) : (l1 < l2 ? i1 : i2);
    };
    WB_PFOR(pair, (int64_t)Ck * Ck)
        const int i1 = (int)(pair / Ck), i2 = (int)(pair - (int64_t)i1 * Ck);
        if (i1 == i2 || setsize[i1] > setsize[i2]) continue;
        const int c1 = (int)ckept[i1], c2 = (int)ckept[i2];
        const int pa = rawbase[c1], pb = rawbase[c1 + 1];
        if (!in_set(rawpt[pa], c2)) continue;
        const int slot = wb_atomic_add(&ctr[4], 1);
        if (slot < WB_F_NCAND) { cand[2 * slot] = i1; cand[2 * slot + 1] = i2; continue; }
        bool sub = true;                                        // (queue full: this lane checks the pair by itself)
        for (int p = pa + 1; p < pb && sub; ++p) sub = in_set(rawpt[p], c2);
        if (sub) dropped[drop_of(i1, i2)] = 1;
    WB_END_PFOR
    {
        // a candidate per WAVE (its lanes over the points of i1, a ballot for the verdict): no workgroup barrier per candidate -- there
        // were three, and a job of the bench batch has a few candidates (the loops of the first 480 columns and their copies)
        const int nc = ctr[4] < WB_F_NCAND ? ctr[4] : WB_F_NCAND;
        const int wv = WB_TID / WB_WAVE, nwv = WB_NT / WB_WAVE > 0 ? WB_NT / WB_WAVE : 1, ln = WB_TID % WB_WAVE;
        for (int q = wv; q < nc; q += nwv) {
            const int i1 = cand[2 * q], i2 = cand[2 * q + 1];
            const int c1 = (int)ckept[i1], c2 = (int)ckept[i2];
            const int pa = rawbase[c1], pb = rawbase[c1 + 1];
            bool miss = false;
            for (int p0 = pa; p0 < pb && !miss; p0 += WB_WAVE) {
                const int p = p0 + ln;
                miss = wb_wave_any(p < pb && !in_set(rawpt[p], c2));
            }
            if (!miss && ln == 0) dropped[drop_of(i1, i2)] = 1;
        }
        WB_SYNC();
    }
    // ---- H: final table = kept and not dropped, order preserved (fin: the kept flag by contour until here, by kept contour from here)
    WB_PFOR32(f, Ck)
        fin[f] = dropped[f] ? 0 : 1;
    WB_END_PFOR
    const int Cf = wb_block_scan_excl(fin, Ck);
    WB_PFOR32(f, Ck)
        if (!dropped[f]) {
            const int g = fin[f], c = (int)ckept[f];
            (a.c_off + jb)[g] = newstart[c];
            (a.c_len + jb)[g] = newstart[c + 1] - newstart[c];
            (a.c_closed + jb)[g] = (int)closedn[c];
            (a.c_nux + jb)[g] = nuxl[f];
            (a.c_sumy + jb)[g] = sumyl[f];
            wb_atomic_max(&a.ctl->max_nux, nuxl[f]);
        }
    WB_END_PFOR
    WB_FSTAMP(a.ctl, 11);
    WB_FSTAMP_FLUSH(a.ctl);
    if (WB_TID == 0) {
    a.ncont[job] = Cf;
    a.nseg[job] = n;
    fa.fdone[job] = 1;
    wb_atomic_max(&a.ctl->max_chains, C);                       // (size the next batch's launch: wb_fused_layout)
    wb_atomic_max(&a.ctl->max_splitters, NSP);
    }
#undef WB_F_DECLINE
#undef WB_F_NOTE
#undef WB_FQ
#undef WB_FQ_NN
#undef WB_FQ_J
#undef WB_FQ_H
#undef WB_FQ_M
#undef WB_FQ_D
#undef WB_FQ_DM
#undef WB_F_OWN
#undef WB_F_LP
#undef WB_F_BUCKET
#undef WB_F_XM
}
