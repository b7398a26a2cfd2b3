// wb_core.h -- arithmetic shared by every kernel: marching-squares case logic, crossing
// coordinates, exact integer lattice predicates.  Pure functions, host+device.
//
// Reference semantics restated (file:line under the reference tree):
//   * skimage.measure.find_contours as called at wavebreaking/indices/contour_index.py:103
//     (case table / crossing formula: SURVEY.md 8a row C2, pinned by tests/golden/contours_*.npz)
//   * np.round(...).astype(int) at contour_index.py:111  -> round-half-even of the f64 SUM c0+f
//   * LineString.touches / intersects at streamer_index.py:197,234 -> exact int64 orientation tests
#pragma once
#include "wb_spmd.h"

#define WB_NIL 0xFFFFFFFFu

enum { WB_TOP = 0, WB_RIGHT = 1, WB_BOTTOM = 2, WB_LEFT = 3 };

// segment table: case -> (from_side, to_side) for sub-segment 0 and 1; nibble-packed
// entry = from0 | to0<<2 | from1<<4 | to1<<6 ; number of segments in WB_NSEG.
//  1:(T->L) 2:(R->T) 3:(R->L) 4:(L->B) 5:(T->B) 6:(R->T),(L->B) 7:(R->B)
//  8:(B->R) 9:(T->L),(B->R) 10:(B->T) 11:(B->L) 12:(L->R) 13:(T->R) 14:(L->T)
// (tables packed into constants, two bits per case: no constant array lands in scratch, no branch on the case)
WB_HD inline int wb_case_nseg(int c) { return (int)((0x15596554u >> (2 * c)) & 3u); }     // 0 / 15: none, 6 / 9: two, else one
// sides T=0 R=1 B=2 L=3; the second segment of a saddle: 6:(L->B)  9:(B->R)
WB_HD inline int wb_seg_from(int c, int sub) {
    return (int)(((sub ? 0xAAAABAAAu : 0x33a25350u) >> (2 * c)) & 3u);
}
WB_HD inline int wb_seg_to(int c, int sub) {
    return (int)(((sub ? 0x55556555u : 0x05cd8accu) >> (2 * c)) & 3u);
}

// crossing fraction, exactly skimage's _get_fraction (float64)
WB_HD inline double wb_fraction(double a, double b, double level) {
    return (a == b) ? 0.0 : (level - a) / (b - a);
}

// round half to even of a non-negative-ish double -> int (np.round(x).astype(int))
WB_HD inline int wb_rint(double v) { return (int)rint(v); }

// lattice point (x | y<<16) of the crossing on `side` of the square with upper-left corner (r,c)
WB_HD inline uint32_t wb_crossing_point(int side, int r, int c, double ul, double ur, double ll, double lr,
                                        double level) {
    double pr, pc;
    if (side == WB_TOP) {
        pr = (double)r;
        pc = (double)c + wb_fraction(ul, ur, level);
    } else if (side == WB_BOTTOM) {
        pr = (double)(r + 1);
        pc = (double)c + wb_fraction(ll, lr, level);
    } else if (side == WB_LEFT) {
        pr = (double)r + wb_fraction(ul, ll, level);
        pc = (double)c;
    } else {
        pr = (double)r + wb_fraction(ur, lr, level);
        pc = (double)(c + 1);
    }
    return (uint32_t)wb_rint(pc) | ((uint32_t)wb_rint(pr) << 16);
}

WB_HD inline int wb_pt_x(uint32_t p) { return (int)(p & 0xFFFFu); }
WB_HD inline int wb_pt_y(uint32_t p) { return (int)(p >> 16); }

// segment key: ((r*W + c) << 1 | sub) | case << 27   (r*W+c < 2^26)
WB_HD inline uint32_t wb_seg_key(int r, int c, int W, int sub, int cs) {
    return ((uint32_t)(r * W + c) << 1) | (uint32_t)sub | ((uint32_t)cs << 27);
}
WB_HD inline int wb_key_sq(uint32_t k) { return (int)((k & 0x07FFFFFFu) >> 1); }
WB_HD inline int wb_key_sub(uint32_t k) { return (int)(k & 1u); }
WB_HD inline int wb_key_case(uint32_t k) { return (int)(k >> 27) & 15; }

// ------------------------------------------------------------------ control block
// Counters of a batch that live on the device and are read back by the host ONCE per ABI phase (a small pinned copy):
// launch sizes and allocations come from the phase-1 bounds, loop limits inside the kernels from the device copy, so no
// kernel of a phase waits for the host.
struct WbCtl {
    // phase 1 (wb_contours)
    int max_nux;                 // largest distinct-longitude count of a contour (exp_lon.max() / dlon of this batch)
    int max_seg_job;             // largest segment count of a (step, level) job / of a strip list: capacity checks
    int max_seg_strip;
    int ncont_total;             // contours of the batch
    long long npts_total;        // their points
    int nsel_max;                // contours with nux == max_nux (= the selected ones unless another rank holds a larger maximum)
    int nclosed;                 // closed contours (cutoff candidates whatever the selection maximum)
    long long npt_sel_max;       // points of the former / of the latter
    long long npt_closed;
    int pair_items_max;          // work items of the all-pairs kernel for the contours with nux == max_nux
    int scan_cursor;             // streaming scan: the next (time step, band) work item (a workgroup takes one when it is free)
    int max_len;                 // longest contour of the batch, in points
    int n_slow;                  // jobs the second trace pass redoes: more segments than the LDS paths hold, or an order-dependent join (cells equal to the level)
    int n_unfused;               // jobs wb_k_trace_fused declined (cells equal to the level, more segments / chains than its LDS holds): the three-kernel chain takes them
    int max_nseg;                // largest segment count of a job (sizes the fused kernel's LDS for the context's next batch)
    int max_chains, max_splitters;   // ... largest number of chains (raw contours) / of ranking splitters of a job the fused kernel finished
    // phase 2 (wb_indices)
    int nsel, ncut;              // selected contours / cutoff contours
    long long npt_sel;
    int n_pair_items, n_f2_items;
    int max_pairs;               // largest candidate count of a contour (capacity check)
    int NS, NO, NE;              // streamer / overturning events, all events
    int overflow;                // bit 0: more events than cap_ne, bit 1: more polygon vertices than cap_poly
    long long cand_pairs;        // candidate pairs that passed both distance tests
    long long point_pairs;       // sum of N (N - 1) / 2 over the selected contours
    long long evaluated_pairs;   // pairs whose great-circle term was actually evaluated
    long long poly_total;        // polygon vertices of all events
    int band[2];                 // guard-band re-evaluations (geo, cont)
    int com_rechecks[3];
    int ne_demand;               // events found when they exceeded cap_ne
    unsigned long long cell_count;
    int props_cursor;            // next event of the property kernel's work queue
    int group_ties;              // check_groups: groups whose maximum length was shared by several members (exact ties)
    int tie_marks;               // ... some contour of the pass holds such a group (wb_k_streamer_ties has something to do)
    int n_recheck;               // pairs wb_k_streamer_pairs2 left to wb_k_streamer_recheck (inside a guard band)
    int n_boxq, boxq_cursor;     // overturning boxes queued for wb_k_box_com (wb_k_overturn) / the next one its persistent workgroups take
    int p2_cursor;               // next (contour, part) item of wb_k_streamer_pairs2's persistent workgroups
    int f1_cursor;               // next contour (in wb_k_pairs_order's order) of wb_k_streamer_f1's persistent workgroups
    int geom_rings, geom_verts;  // transform_polygons on the device (wb_geom.h): rings and vertices of the pass's events in lon / lat
    int geom_flagged;            // events with more cut edges than its kernel holds (the host function does the pass's geometry then)
    int box_bound;               // queued boxes wb_k_box_com settled from the exact totals of their terms alone (no rounding boundary within the steps' error bound)
    long long dbg[24];           // development: per-phase clock sums of an instrumented kernel (WB_STAMP)
};
// development aid: clock ticks (100 MHz wall clock) spent between two stamps of thread 0, summed over the workgroups.
// -DWB_STAMPS=1 instruments wb_k_streamer_f234 (WB_STAMP), =2 wb_k_streamer_pairs (WB_PSTAMP), =3 wb_k_props (WB_RSTAMP), =4 wb_k_trace_finish (WB_TSTAMP), =5 wb_k_box_com (WB_BSTAMP), =6 wb_k_trace_link (WB_LSTAMP), =7 wb_k_trace_rank (WB_KSTAMP), =8 wb_k_trace_fused (WB_FSTAMP); WB_DEBUG_STAMPS=1 prints.
#if !defined(WB_HOSTSIM) && defined(WB_STAMPS)
// sums kept in registers of thread 0, one global atomic per phase and workgroup at the end (..._FLUSH)
#if defined(WB_STAMPS_FINE)
// -DWB_STAMPS_FINE: twenty sums (a kernel's WB_xSTAMP_FINE stamps cut its phases further), no statistics of the longest workgroup
#define WB_STAMPX_BEGIN long long wb_acc_[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long wb_st_ = (threadIdx.x == 0) ? (long long)wall_clock64() : 0;
#define WB_STAMPX(ctl, k) do { if (threadIdx.x == 0) { const long long wb_n_ = (long long)wall_clock64(); wb_acc_[k] += wb_n_ - wb_st_; wb_st_ = wb_n_; } } while (0)
#define WB_STAMPX_FLUSH(ctl) do { if (threadIdx.x == 0) { for (int wb_q_ = 0; wb_q_ < 20; ++wb_q_) if (wb_acc_[wb_q_]) atomicAdd((unsigned long long*)&(ctl)->dbg[wb_q_], (unsigned long long)wb_acc_[wb_q_]); } } while (0)
#else
#define WB_STAMPX_BEGIN long long wb_acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long wb_st_ = (threadIdx.x == 0) ? (long long)wall_clock64() : 0;
#define WB_STAMPX(ctl, k) do { if (threadIdx.x == 0) { const long long wb_n_ = (long long)wall_clock64(); wb_acc_[k] += wb_n_ - wb_st_; wb_st_ = wb_n_; } } while (0)
// (dbg[20]: the longest workgroup; dbg[21..23]: workgroups that took more than 200 / 400 / 800 us, printed as count * 1e-5)
#define WB_STAMPX_FLUSH(ctl) do { if (threadIdx.x == 0) { long long wb_t_ = 0; for (int wb_q_ = 0; wb_q_ < 12; ++wb_q_) if (wb_acc_[wb_q_]) { wb_t_ += wb_acc_[wb_q_]; atomicAdd((unsigned long long*)&(ctl)->dbg[wb_q_], (unsigned long long)wb_acc_[wb_q_]); } \
    if (atomicMax((unsigned long long*)&(ctl)->dbg[20], (unsigned long long)wb_t_) < (unsigned long long)wb_t_) for (int wb_q_ = 0; wb_q_ < 8; ++wb_q_) (ctl)->dbg[12 + wb_q_] = wb_acc_[wb_q_ < 3 ? 4 + wb_q_ : wb_q_ + 5];   /* the longest workgroup's own stamps 4, 5, 6 and 8 .. 11 (slots 12 ..) */ \
    for (int wb_q_ = 0; wb_q_ < 3; ++wb_q_) if (wb_t_ > (20000ll << wb_q_)) atomicAdd((unsigned long long*)&(ctl)->dbg[21 + wb_q_], 1ull); } } while (0)
#endif
#endif
#if !defined(WB_HOSTSIM) && defined(WB_STAMPS) && WB_STAMPS == 1
#define WB_STAMP_BEGIN WB_STAMPX_BEGIN
#define WB_STAMP(ctl, k) WB_STAMPX(ctl, k)
#define WB_STAMP_COUNT(k, v) do { if (threadIdx.x == 0) wb_acc_[k] += (long long)(v) * 100000; } while (0)   /* printed as the count itself */
#define WB_STAMP_FLUSH(ctl) WB_STAMPX_FLUSH(ctl)
#else
#define WB_STAMP_BEGIN
#define WB_STAMP(ctl, k) ((void)0)
#define WB_STAMP_COUNT(k, v) ((void)0)
#define WB_STAMP_FLUSH(ctl) ((void)0)
#endif
#if !defined(WB_HOSTSIM) && defined(WB_STAMPS) && WB_STAMPS == 3
#define WB_RSTAMP_BEGIN WB_STAMPX_BEGIN
#define WB_RSTAMP(ctl, k) WB_STAMPX(ctl, k)
#define WB_RSTAMP_FLUSH(ctl) WB_STAMPX_FLUSH(ctl)
#else
#define WB_RSTAMP_BEGIN
#define WB_RSTAMP(ctl, k) ((void)0)
#define WB_RSTAMP_FLUSH(ctl) ((void)0)
#endif
#if !defined(WB_HOSTSIM) && defined(WB_STAMPS) && WB_STAMPS == 4
#define WB_TSTAMP_BEGIN WB_STAMPX_BEGIN
#define WB_TSTAMP(ctl, k) WB_STAMPX(ctl, k)
#define WB_TSTAMP_FLUSH(ctl) WB_STAMPX_FLUSH(ctl)
#else
#define WB_TSTAMP_BEGIN
#define WB_TSTAMP(ctl, k) ((void)0)
#define WB_TSTAMP_FLUSH(ctl) ((void)0)
#endif
#if !defined(WB_HOSTSIM) && defined(WB_STAMPS) && WB_STAMPS == 7
#define WB_KSTAMP_BEGIN WB_STAMPX_BEGIN
#define WB_KSTAMP(ctl, k) WB_STAMPX(ctl, k)
#define WB_KSTAMP_FLUSH(ctl) WB_STAMPX_FLUSH(ctl)
#else
#define WB_KSTAMP_BEGIN
#define WB_KSTAMP(ctl, k) ((void)0)
#define WB_KSTAMP_FLUSH(ctl) ((void)0)
#endif
#if !defined(WB_HOSTSIM) && defined(WB_STAMPS) && WB_STAMPS == 6
#define WB_LSTAMP_BEGIN WB_STAMPX_BEGIN
#define WB_LSTAMP(ctl, k) WB_STAMPX(ctl, k)
#define WB_LSTAMP_FLUSH(ctl) WB_STAMPX_FLUSH(ctl)
#else
#define WB_LSTAMP_BEGIN
#define WB_LSTAMP(ctl, k) ((void)0)
#define WB_LSTAMP_FLUSH(ctl) ((void)0)
#endif
#if !defined(WB_HOSTSIM) && defined(WB_STAMPS) && WB_STAMPS == 5
#define WB_BSTAMP_BEGIN WB_STAMPX_BEGIN
#define WB_BSTAMP(ctl, k) WB_STAMPX(ctl, k)
#define WB_BSTAMP_FLUSH(ctl) WB_STAMPX_FLUSH(ctl)
#else
#define WB_BSTAMP_BEGIN
#define WB_BSTAMP(ctl, k) ((void)0)
#define WB_BSTAMP_FLUSH(ctl) ((void)0)
#endif
#if !defined(WB_HOSTSIM) && defined(WB_STAMPS) && WB_STAMPS == 8
#define WB_FSTAMP_BEGIN WB_STAMPX_BEGIN
#define WB_FSTAMP(ctl, k) WB_STAMPX(ctl, k)
#define WB_FSTAMP_FLUSH(ctl) WB_STAMPX_FLUSH(ctl)
#if defined(WB_STAMPS_FINE)
#define WB_FSTAMP_FINE(ctl, k) WB_STAMPX(ctl, k)
#else
#define WB_FSTAMP_FINE(ctl, k) ((void)0)
#endif
#else
#define WB_FSTAMP_BEGIN
#define WB_FSTAMP(ctl, k) ((void)0)
#define WB_FSTAMP_FINE(ctl, k) ((void)0)
#define WB_FSTAMP_FLUSH(ctl) ((void)0)
#endif
#if !defined(WB_HOSTSIM) && defined(WB_STAMPS) && WB_STAMPS == 9      // wb_k_streamer_f1 (routine 1)
#define WB_DSTAMP_BEGIN WB_STAMPX_BEGIN
#define WB_DSTAMP(ctl, k) WB_STAMPX(ctl, k)
#define WB_DSTAMP_FLUSH(ctl) WB_STAMPX_FLUSH(ctl)
#define WB_DSTAMP_ARGS , long long* wb_acc_, long long& wb_st_
#define WB_DSTAMP_PASS , wb_acc_, wb_st_
#else
#define WB_DSTAMP_BEGIN
#define WB_DSTAMP(ctl, k) ((void)0)
#define WB_DSTAMP_FLUSH(ctl) ((void)0)
#define WB_DSTAMP_ARGS
#define WB_DSTAMP_PASS
#endif
#if !defined(WB_HOSTSIM) && defined(WB_STAMPS) && WB_STAMPS == 2
#define WB_PSTAMP_BEGIN WB_STAMPX_BEGIN
#define WB_PSTAMP(ctl, k) WB_STAMPX(ctl, k)
#define WB_PSTAMP_FLUSH(ctl) WB_STAMPX_FLUSH(ctl)
#else
#define WB_PSTAMP_BEGIN
#define WB_PSTAMP(ctl, k) ((void)0)
#define WB_PSTAMP_FLUSH(ctl) ((void)0)
#endif

// ------------------------------------------------------------------ numpy's pairwise summation
// ``a[0:n].sum()`` of a contiguous float64 array exactly as numpy adds it (loops_utils.h.src, DOUBLE_pairwise_sum): fewer than
// 8 elements one after the other; up to 128 in eight interleaved accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
// plus the tail; longer arrays split at n/2 rounded down to a multiple of 8.  (check_groups compares such sums,
// streamer_index.py:241-246: shifted chords of a lattice contour tie to the last bit.)
WB_HD inline double wb_np_sum_leaf(const double* a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res = res + a[i];
        return res;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 = r0 + a[i]; r1 = r1 + a[i + 1]; r2 = r2 + a[i + 2]; r3 = r3 + a[i + 3];
        r4 = r4 + a[i + 4]; r5 = r5 + a[i + 5]; r6 = r6 + a[i + 6]; r7 = r7 + a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res = res + a[i];
    return res;
}
WB_HD inline double wb_np_pairwise_block(const double* a, int n) {
    // the recursion, unrolled onto a small stack (a block has at most 8192 elements: depth 7); a frame is (offset, length,
    // stage) + the left half's sum
    int fo[10], fn[10];
    signed char fs[10];
    double fl[10];
    int sp = 0;
    fo[0] = 0; fn[0] = n; fs[0] = 0; fl[0] = 0.0;
    double ret = 0.0;
    while (sp >= 0) {
        if (fs[sp] == 0) {
            if (fn[sp] <= 128) {
                ret = wb_np_sum_leaf(a + fo[sp], fn[sp]);
                --sp;
            } else {
                int n2 = fn[sp] / 2; n2 -= n2 % 8;
                fs[sp] = 1;
                fo[sp + 1] = fo[sp]; fn[sp + 1] = n2; fs[sp + 1] = 0; fl[sp + 1] = 0.0;
                ++sp;
            }
        } else if (fs[sp] == 1) {
            int n2 = fn[sp] / 2; n2 -= n2 % 8;
            fl[sp] = ret;
            fs[sp] = 2;
            fo[sp + 1] = fo[sp] + n2; fn[sp + 1] = fn[sp] - n2; fs[sp + 1] = 0; fl[sp + 1] = 0.0;
            ++sp;
        } else { ret = fl[sp] + ret; --sp; }
    }
    return ret;
}

// ``a[0:n].sum()``: numpy's reduction hands its inner loop at most 8192 elements at a time (the ufunc buffer size) and adds
// the blocks' pairwise sums one after the other (checked against numpy 2.2 for lengths up to 100 000)
WB_HD inline double wb_np_pairwise_sum(const double* a, int n) {
    if (n <= 8192) return wb_np_pairwise_block(a, n);
    double s = wb_np_pairwise_block(a, 8192);
    for (int i = 8192; i < n; i += 8192) s = s + wb_np_pairwise_block(a + i, n - i < 8192 ? n - i : 8192);
    return s;
}

// ------------------------------------------------------------------ CPython's set, for one tie
// check_groups (streamer_index.py:224-251) keeps ``item[np.argmax(lengths)]`` per group, ``item`` being ``list(first)`` of
// combine_shared (index_utils.py:187-214): when two members' lengths are EQUAL to the last bit -- chords shifted by one
// point along a lattice contour, every few thousand time steps -- the survivor is whichever comes first in the iteration
// order of a CPython set of small integers, i.e. in the slot order of its hash table after the very sequence of
// ``set(row)`` / ``first |= set(row)`` operations combine_shared performs.  Restated from Objects/setobject.c (CPython 3.7
// - 3.12: hash(int) = int, 8 initial slots, LINEAR_PROBES 9, PERTURB_SHIFT 5, growth when fill * 5 >= mask * 3 to the
// first power of two above 4 * used, ``|=`` growing up front to above 2 * (used + other's used)); pinned against the
// running interpreter's sets in tests/test_hostsim_parity.py.  Keys are non-negative ints; -1 marks an empty slot.
struct WbPySet { long long* tab; long long* spare; int mask, fill, used; };
WB_HD inline void wb_pyset_insert_clean(long long* tab, int mask, long long key) {
    unsigned long long perturb = (unsigned long long)key;
    unsigned long long i = (unsigned long long)key & (unsigned long long)mask;
    for (;;) {
        if (tab[i] < 0) { tab[i] = key; return; }
        if (i + 9 <= (unsigned long long)mask) {
            for (int j = 1; j <= 9; ++j) if (tab[i + j] < 0) { tab[i + j] = key; return; }
        }
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (unsigned long long)mask;
    }
}
WB_HD inline void wb_pyset_resize(WbPySet& s, int minused) {
    int newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    long long* nt = s.spare;
    for (int i = 0; i < newsize; ++i) nt[i] = -1;
    for (int i = 0; i <= s.mask; ++i) if (s.tab[i] >= 0) wb_pyset_insert_clean(nt, newsize - 1, s.tab[i]);
    s.spare = s.tab; s.tab = nt; s.mask = newsize - 1; s.fill = s.used;
}
WB_HD inline bool wb_pyset_contains(const WbPySet& s, long long key) {
    unsigned long long perturb = (unsigned long long)key, mask = (unsigned long long)s.mask;
    unsigned long long i = (unsigned long long)key & mask;
    for (;;) {
        const int probes = (i + 9 <= mask) ? 9 : 0;
        for (int j = 0; j <= probes; ++j) {
            if (s.tab[i + j] < 0) return false;
            if (s.tab[i + j] == key) return true;
        }
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & mask;
    }
}
WB_HD inline void wb_pyset_add(WbPySet& s, long long key) {
    unsigned long long perturb = (unsigned long long)key, mask = (unsigned long long)s.mask;
    unsigned long long i = (unsigned long long)key & mask;
    for (;;) {
        const int probes = (i + 9 <= mask) ? 9 : 0;
        for (int j = 0; j <= probes; ++j) {
            if (s.tab[i + j] == key) return;
            if (s.tab[i + j] < 0) {
                s.tab[i + j] = key;
                ++s.fill; ++s.used;
                if ((long long)s.fill * 5 < (long long)s.mask * 3) return;
                wb_pyset_resize(s, s.used > 50000 ? s.used * 2 : s.used * 4);
                return;
            }
        }
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & mask;
    }
}
// s |= set([a, b])
WB_HD inline void wb_pyset_merge_pair(WbPySet& s, long long a, long long b) {
    long long ot[8];
    for (int i = 0; i < 8; ++i) ot[i] = -1;
    wb_pyset_insert_clean(ot, 7, a);                       // set(iterable): a, then b (two keys never fill 8 slots)
    int oused = 1;
    if (b != a) { wb_pyset_insert_clean(ot, 7, b); oused = 2; }
    if ((long long)(s.fill + oused) * 5 >= (long long)s.mask * 3) wb_pyset_resize(s, (s.used + oused) * 2);
    for (int i = 0; i < 8; ++i) if (ot[i] >= 0) wb_pyset_add(s, ot[i]);
}
// combine_shared for ONE group: ``pp[k] <= pq[k]``, k < np, its crossing pairs in the reference's order (lexicographic; the
// first is the group's smallest member with itself); ws: 2 * cap slots, cap a power of two >= 8 * members.  Writes the
// members in ``list(first)`` order to out, returns their number.
WB_HD inline int wb_pyset_group_order(const int* pp, const int* pq, int np, unsigned char* done, long long* ws, int cap, int* out) {
    WbPySet s;
    s.tab = ws; s.spare = ws + cap; s.mask = 7; s.fill = 0; s.used = 0;
    for (int i = 0; i < 8; ++i) s.tab[i] = -1;
    wb_pyset_add(s, pp[0]);
    wb_pyset_add(s, pq[0]);                                 // first = set(first row)
    for (int k = 0; k < np; ++k) done[k] = k == 0;
    int lf = -1;
    while (s.used > lf) {
        lf = s.used;
        for (int k = 1; k < np; ++k) {
            if (done[k]) continue;
            if (wb_pyset_contains(s, pp[k]) || wb_pyset_contains(s, pq[k])) { wb_pyset_merge_pair(s, pp[k], pq[k]); done[k] = 1; }
        }
    }
    int n = 0;
    for (int i = 0; i <= s.mask; ++i) if (s.tab[i] >= 0) out[n++] = (int)s.tab[i];
    return n;
}

// ------------------------------------------------------------------ integer predicates
WB_HD inline int64_t wb_orient(int64_t ax, int64_t ay, int64_t bx, int64_t by, int64_t cx, int64_t cy) {
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
}
WB_HD inline bool wb_opposite(int64_t a, int64_t b) { return (a > 0 && b < 0) || (a < 0 && b > 0); }

// Does the OPEN chord (p,q) meet the closed contour segment [a,b] in a point that is interior to the contour?  This is the
// per-segment term of `not chord.touches(contour)`, streamer_index.py:192-198 (DE-9IM: the interiors must not meet).  The
// contour's boundary is the POINT SET of its two end vertices e0, e1 (packed x | y << 16; equal: a closed line, no
// boundary): a location that is an end vertex is boundary wherever else the line passes through it -- a later segment with a
// long step, (3,0) after the first-occurrence dedupe of contour_index.py:110-112, runs OVER earlier vertices, the first one
// included.  Until round 5 "end vertex" was decided by the segment's index, which differs exactly there: a chord through
// the first vertex that crosses such a segment AT that vertex (found by the randomized differential test against
// tests/ogc_reference.py, an independent exact-rational statement; never on a traced full-span contour, whose end
// vertices lie on the grid's edge columns).
WB_HD inline bool wb_chord_hits_segment(int px, int py, int qx, int qy, int ax, int ay, int bx, int by, uint32_t e0, uint32_t e1) {
    const int64_t d1 = wb_orient(px, py, qx, qy, ax, ay);
    const int64_t d2 = wb_orient(px, py, qx, qy, bx, by);
    const int64_t dx = qx - px, dy = qy - py;
    const int64_t L = dx * dx + dy * dy;
    const int64_t ta = (int64_t)(ax - px) * dx + (int64_t)(ay - py) * dy;
    const int64_t tb = (int64_t)(bx - px) * dx + (int64_t)(by - py) * dy;
    if (d1 == 0 && d2 == 0) {  // collinear: overlap of positive length (infinitely many points, at most two of them boundary)
        int64_t lo = ta < tb ? ta : tb, hi = ta < tb ? tb : ta;
        if (lo < 0) lo = 0;
        if (hi > L) hi = L;
        return hi > lo;
    }
    const bool open_ends = e0 != e1;
    const int e0x = (int)(e0 & 0xFFFFu), e0y = (int)(e0 >> 16), e1x = (int)(e1 & 0xFFFFu), e1y = (int)(e1 >> 16);
    const int64_t d3 = wb_orient(ax, ay, bx, by, px, py);
    const int64_t d4 = wb_orient(ax, ay, bx, by, qx, qy);
    if (wb_opposite(d1, d2) && wb_opposite(d3, d4)) {
        // proper crossing, in the ONE point the two lines share: interior unless that point is an end vertex of the contour
        if (open_ends && wb_orient(px, py, qx, qy, e0x, e0y) == 0 && wb_orient(ax, ay, bx, by, e0x, e0y) == 0) return false;
        if (open_ends && wb_orient(px, py, qx, qy, e1x, e1y) == 0 && wb_orient(ax, ay, bx, by, e1x, e1y) == 0) return false;
        return true;
    }
    // a contour vertex on the open chord
    if (d1 == 0 && ta > 0 && ta < L && !(open_ends && ((ax == e0x && ay == e0y) || (ax == e1x && ay == e1y)))) return true;
    if (d2 == 0 && tb > 0 && tb < L && !(open_ends && ((bx == e0x && by == e0y) || (bx == e1x && by == e1y)))) return true;
    return false;
}
// ... for callers that hold the end vertices as coordinates
WB_HD inline bool wb_chord_hits_segment(int px, int py, int qx, int qy, int ax, int ay, int bx, int by, int e0x, int e0y, int e1x, int e1y) {
    return wb_chord_hits_segment(px, py, qx, qy, ax, ay, bx, by, (uint32_t)e0x | ((uint32_t)e0y << 16), (uint32_t)e1x | ((uint32_t)e1y << 16));
}

WB_HD inline bool wb_on_closed(int ax, int ay, int bx, int by, int cx, int cy) {
    return (ax < bx ? ax : bx) <= cx && cx <= (ax < bx ? bx : ax) && (ay < by ? ay : by) <= cy &&
           cy <= (ay < by ? by : ay);
}
// closed segments [p1,q1] and [p2,q2] share a point (streamer_index.py:233-236, `intersects`)
WB_HD inline bool wb_segments_intersect(int p1x, int p1y, int q1x, int q1y, int p2x, int p2y, int q2x, int q2y) {
    const int64_t d1 = wb_orient(p1x, p1y, q1x, q1y, p2x, p2y);
    const int64_t d2 = wb_orient(p1x, p1y, q1x, q1y, q2x, q2y);
    const int64_t d3 = wb_orient(p2x, p2y, q2x, q2y, p1x, p1y);
    const int64_t d4 = wb_orient(p2x, p2y, q2x, q2y, q1x, q1y);
    if (wb_opposite(d1, d2) && wb_opposite(d3, d4)) return true;
    if (d1 == 0 && wb_on_closed(p1x, p1y, q1x, q1y, p2x, p2y)) return true;
    if (d2 == 0 && wb_on_closed(p1x, p1y, q1x, q1y, q2x, q2y)) return true;
    if (d3 == 0 && wb_on_closed(p2x, p2y, q2x, q2y, p1x, p1y)) return true;
    if (d4 == 0 && wb_on_closed(p2x, p2y, q2x, q2y, q1x, q1y)) return true;
    return false;
}

// 64-bit mix for the open-addressing tables (splitmix64 finaliser)
// ---- shared by wb_k_overturn (which decides whether a box needs the reference-order sums) and wb_props.h
// Is the truncation of num / den safe against an error of `ulps` units of 2^-53 (relative) in the quotient?
WB_HD inline bool wb_trunc_unsafe(double num, double den, double ulps) {
    const double q = num / den;
    const double tol = (fabs(q) + 1.0) * ulps * 1.1102230246251565e-16;
    return fabs(q - rint(q)) <= tol;
}
// relative error bound (in units of 2^-53) between the row-wise sums over `rows` rows and the reference's compensated
// cell-wise sums, for the QUOTIENT of two such sums; x4 safety
WB_HD inline double wb_com_ulps(int rows) { return 4.0 * (2.0 * (double)rows + 16.0); }

// v = m * 2^e, m < 2^53 (v >= 0, finite)
WB_HD inline void wb_dbl_split(double v, unsigned long long* m, int* e) {
    unsigned long long b;
    memcpy(&b, &v, 8);
    const int ef = (int)((b >> 52) & 0x7FF);
    const unsigned long long fr = b & 0xFFFFFFFFFFFFFull;
    if (ef == 0) { *m = fr; *e = -1074; } else { *m = fr | (1ull << 52); *e = ef - 1075; }
}


#ifndef WB_BOXCOM_BOUND
#define WB_BOXCOM_BOUND 1            // 1: wb_k_overturn settles a box whose three sums are decided by the exact totals of their terms and queues the others for
                                     // wb_k_box_com's chained passes; 0 (a test build): every such box is queued; 2 (a test build): every box is queued and
                                     // wb_k_box_com evaluates it both ways -- a box on which the two differ is counted in dbg[23] (must be 0)
#endif
// ---- the overturning boxes' centre-of-mass sums from the EXACT totals of their terms (wb_k_overturn, wb_k_box_com; the argument
// is in wb_props.h in front of wb_k_box_com): 128-bit fixed point in units of 2^ue
typedef unsigned __int128 wb_u128;
WB_HD inline wb_u128 wb_fix128(double v, int ue) {
    unsigned long long m; int e;
    wb_dbl_split(v, &m, &e);
    if (!m) return (wb_u128)0;
    if (e >= ue) return (wb_u128)m << (e - ue);
    // a small multiple of 2^ue (a compensation term): its 53-bit mantissa ends in zeros
    return ue - e > 63 ? (wb_u128)0 : (wb_u128)(m >> (ue - e));
}
// state (s, c) of the compensated sum whose exact running total is E * 2^ue, every step having been exact: s = RN(E),
// c = s - E.  false: c does not fit a double (such a state cannot have been reached by exact steps)
WB_HD inline bool wb_state_from_exact(wb_u128 E, int ue, double* s, double* c) {
    if (E == 0) { *s = 0.0; *c = 0.0; return true; }
    const unsigned long long hi = (unsigned long long)(E >> 64), lo = (unsigned long long)E;
    const int msb = hi ? 127 - __builtin_clzll(hi) : 63 - __builtin_clzll(lo);
    if (msb <= 52) { *s = ldexp((double)lo, ue); *c = 0.0; return true; }
    const int sh = msb - 52;
    if (sh > 53) return false;
    unsigned long long q = (unsigned long long)(E >> sh);
    const wb_u128 rem = E - ((wb_u128)q << sh), half = (wb_u128)1 << (sh - 1);
    const bool up = rem > half || (rem == half && (q & 1ull));
    if (up) q += 1;
    *s = ldexp((double)q, sh + ue);
    if (up) *c = ldexp((double)(unsigned long long)(((wb_u128)q << sh) - E), ue);
    else *c = rem ? -ldexp((double)(unsigned long long)rem, ue) : 0.0;
    return true;
}


// totals of a set of rows of a box: T[3] exact totals of the terms of the three sums (areas: w, x_com: x * w, y_com: y * w, each
// rounded once as the reference rounds it), B[3] the bound on what the compensated sums' inexact steps can add, edge the smallest
// distance of a term to the top or the bottom of its binade (in the term's own units: never more than in units of 2^ue)
struct WbBoxRows { wb_u128 T[3]; unsigned long long B[3]; unsigned long long edge; };
WB_HD inline void wb_box_rows_clear(WbBoxRows* o) { o->T[0] = o->T[1] = o->T[2] = 0; o->B[0] = o->B[1] = o->B[2] = 0; o->edge = ~0ull; }
// row y of the box [x0, x1] x [.., ..] added to o.  first: the box's first row (its sums start from s = c = 0: that step is exact)
WB_HD inline void wb_box_row_terms(const double* wlat, int y, bool first, int x0, int x1, int ue, WbBoxRows* o) {
    const int cols = x1 - x0 + 1;
    const double w = wlat[y], yw = (double)y * w;
    // the term in front of this row's first, in each of the three sequences
    const double wp = first ? 0.0 : wlat[y - 1];
    unsigned long long m, mp; int e, ep;
    // a step whose y = v - c is rounded costs at most ulp(v): y is rounded to half a unit of its own binade, which is v's or -- c < 0
    // -- the next one up.  It can be rounded only where v's unit is coarser than the unit c is a multiple of: the unit of the
    // PREVIOUS term's y -- that term's own unit, unless its y fell into the binade below (the term within |c| of the bottom of its
    // binade) -- or where v - c leaves v's binade upwards (v within |c| of the top): `edge`
    wb_dbl_split(w, &m, &e);
    wb_dbl_split(wp, &mp, &ep);
    if (m) {
        o->T[0] += ((wb_u128)m * (unsigned)cols) << (e - ue);
        if (!first && (!mp || e > ep)) o->B[0] += 1ull << (e - ue);
        const unsigned long long dtop = (1ull << 53) - m, dbot = m - (1ull << 52);
        const unsigned long long dd = dtop < dbot ? dtop : dbot;
        if (dd < o->edge) o->edge = dd;
    }
    wb_dbl_split(yw, &m, &e);
    wb_dbl_split((double)(y - 1) * wp, &mp, &ep);
    if (m) {
        o->T[2] += ((wb_u128)m * (unsigned)cols) << (e - ue);
        if (!first && (!mp || e > ep)) o->B[2] += 1ull << (e - ue);
        const unsigned long long dtop = (1ull << 53) - m, dbot = m - (1ull << 52);
        const unsigned long long dd = dtop < dbot ? dtop : dbot;
        if (dd < o->edge) o->edge = dd;
    }
    wb_dbl_split((double)x1 * wp, &mp, &ep);
    // the row's terms x * w: mantissas of a run of equal exponent are added in 64 bits (at most 2047 of them), the run goes into
    // the 128-bit total when the exponent changes
    int ecur = (!first && mp) ? ep : (!first ? -100000 : 100000);      // exponent of the term in front (the sequence's, across the rows)
    unsigned long long acc = 0ull;
    int cnt = 0;
    for (int x = x0; x <= x1; ++x) {
        const double p = (double)x * w;
        wb_dbl_split(p, &m, &e);
        if (!m) continue;
        if (e != ecur || cnt == 2047) {
            if (cnt) o->T[1] += (wb_u128)acc << (ecur - ue);
            if (e > ecur) o->B[1] += 1ull << (e - ue);
            acc = 0ull; cnt = 0; ecur = e;
        }
        acc += m; ++cnt;
        const unsigned long long dtop = (1ull << 53) - m, dbot = m - (1ull << 52);
        const unsigned long long dd = dtop < dbot ? dtop : dbot;
        if (dd < o->edge) o->edge = dd;
    }
    if (cnt) o->T[1] += (wb_u128)acc << (ecur - ue);
}
// the three sums, if the totals decide them: S = RN(T + d) with |d| <= B, and no term so close to an edge of its binade that a
// step outside the bound's accounting could have been rounded (|c| <= ulp(S) / 2 = 2^(msb - 53) units; four times that is asked for)
WB_HD inline bool wb_box_decide(const wb_u128* T, const wb_u128* Bin, unsigned long long edge, int ue, double* sv) {
    for (int q = 0; q < 3; ++q) {
        const wb_u128 B = Bin[q] + 1;                                 // (never an empty interval)
        if (T[q] <= B) return false;
        double s0, s1, c0, c1;
        if (!(wb_state_from_exact(T[q] - B, ue, &s0, &c0) && wb_state_from_exact(T[q] + B, ue, &s1, &c1) && s0 == s1)) return false;
        sv[q] = s0;
        const unsigned long long hi = (unsigned long long)(T[q] >> 64), lo = (unsigned long long)T[q];
        const int msb = hi ? 127 - __builtin_clzll(hi) : 63 - __builtin_clzll(lo | 1ull);
        if (msb > 52 && (msb - 52 >= 63 || edge <= (2ull << (msb - 52)))) return false;
    }
    return true;
}

WB_HD inline uint64_t wb_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
