// wb_stream.h -- the field scan as a pure streaming kernel (THE HBM-bound kernel of the path).
//
// wb_k_scan (wb_scan.h) classifies AND emits segments in one kernel: its emit path costs 116 VGPRs, which leaves 4 waves
// per SIMD with 4 row loads each in flight -- 16 KiB per SIMD, and the HBM pipe of this chip wants several times that
// (loaded latency is ~4 us: bandwidth grew in proportion to the bytes in flight in every variant measured).  This kernel
// does the least a pass over the field must do:
//   * a WORKGROUP works on a band of rows of one time step, a wave on one 256-column strip of it, streaming top to bottom
//     (bands: a few times more work items than resident workgroups, so the last round of the persistent grid is nearly
//     full; a band re-reads one row of its neighbour): the workgroup reads every row as one contiguous 4 * nlon byte run.
//     The one value a strip needs from its right neighbour -- the halo of lane 63 -- is handed over through LDS: every
//     strip publishes the first of its 256 columns for every row it receives (value, then a tag of work item and row)
//     and reads its neighbour's slot, sleeping until the tag is there; a strip publishes BEFORE it waits, so the waits
//     cannot form a cycle, a barrier per work item keeps a fast strip from overwriting slots, and the tags are cleared
//     at kernel start (LDS keeps the previous launch's contents -- tags of the same work items included).  The halo used
//     to be a scalar load per strip and row: the L2 does not merge it with the neighbour's vector miss of that line, and
//     a wait for a scalar load is a wait for all of them (its latency was exposed in two of ten rows).  With the exchange,
//     no wrap-around re-read of the row's first columns by the last strip and a spare row for the loads past a band's
//     end, the L2's request counter shows 47.5 M requests per launch for the field's 47.4 M lines (54.1 M before).
//     Fields that are not float, or whose rows cannot be loaded as vectors, keep the scalar load (HL = false);
//   * a row buffer is the lane's 16 bytes and nothing else; halo, minimum and maximum are formed when a row is consumed,
//     so an 8-deep ring (7 loads in flight) fits in 64 VGPRs -> 8 waves per SIMD;
//   * a lane whose 2 x 5 patch holds a crossing of ANY level (min(patch) <= level < max(patch), evaluated in the field's own
//     precision against the largest T <= level: exactly skimage's strict `>` on the float64 copy, contour_index.py:103)
//     appends the patch to the list of its (time step, band, strip) by ballot compaction, raster order -- through a small
//     per-wave LDS buffer that is flushed to HBM 32+ patches at a time with full-width stores (the patches of one row are
//     a handful of 44-byte items: stored directly they cost a dozen nearly empty store instructions per row, and those,
//     not the loads, set the kernel's duration: 1.6 ms against 1.1 ms with the stores switched off).
// wb_k_scan_build (wb_scan.h) turns the patches into the per-level segment lists the trace kernels read.
//
// Algorithmic HBM traffic: nlat * nlon * sizeof(T) per time step read once (+ 44 B per captured patch, ~6 %).
#pragma once
#include "wb_scan.h"

#if !defined(WB_HOSTSIM)
#ifndef WB_STREAM_FLUSH
#define WB_STREAM_FLUSH 64         // patches in the wave's LDS buffer that trigger a flush (of its whole 128-byte lines)
#endif
#ifndef WB_STREAM_LBUF
#define WB_STREAM_LBUF 160         // its capacity: a row adds at most 64
#endif
// D = ring depth (D - 1 row loads in flight per wave); WPS = waves per SIMD the register allocation must leave room for
// spin until the LDS word at byte address `lds_addr` equals `tag` (both wave-uniform).  One opaque instruction for the
// compiler: a loop in the source would sit inside every unrolled step of the kernel below and cost it its register
// allocation (the ring rows rotate by name; with back-edges in between they were copied and spilled).
// The spin is BOUNDED (round 5): `budget` (wave-uniform, an SGPR) counts the sleeps a wave may still spend waiting in this
// launch; at zero every wait falls through at once (the value is then whatever the slot holds) and the caller flags the time
// step (WB_ST_SCAN_TIMEOUT -> RuntimeError in the Python layer): a lost tag -- none has been seen since the tags are
// cleared at kernel start -- ends the launch with an error instead of hanging the device until the driver's watchdog.
__device__ inline float wb_lds_wait_eq(uint32_t addr_tag, uint32_t tag, uint32_t addr_val, uint32_t& budget) {
    uint32_t tmp;
    float val;
    asm volatile(
        "1:\n\t"
        "ds_read_b32 %0, %3\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u32 vcc, %0, %4\n\t"
        "s_cbranch_vccnz 2f\n\t"
        "s_cmp_eq_u32 %2, 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_sub_u32 %2, %2, 1\n\t"
        "s_sleep 1\n\t"
        "s_branch 1b\n\t"
        "2:\n\t"
        "ds_read_b32 %1, %5\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(tmp), "=&v"(val), "+s"(budget) : "v"(addr_tag), "s"(tag), "v"(addr_val) : "vcc", "scc", "memory");
    return val;
}
// ... the same for a 64-bit value (float64 fields, round 6)
__device__ inline double wb_lds_wait_eq64(uint32_t addr_tag, uint32_t tag, uint32_t addr_val, uint32_t& budget) {
    uint32_t tmp;
    double val;
    asm volatile(
        "1:\n\t"
        "ds_read_b32 %0, %3\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u32 vcc, %0, %4\n\t"
        "s_cbranch_vccnz 2f\n\t"
        "s_cmp_eq_u32 %2, 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_sub_u32 %2, %2, 1\n\t"
        "s_sleep 1\n\t"
        "s_branch 1b\n\t"
        "2:\n\t"
        "ds_read_b64 %1, %5\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(tmp), "=&v"(val), "+s"(budget) : "v"(addr_tag), "s"(tag), "v"(addr_val) : "vcc", "scc", "memory");
    return val;
}
template <typename T> __device__ inline T wb_lds_wait_val(uint32_t addr_tag, uint32_t tag, uint32_t addr_val, uint32_t& budget);
template <> __device__ inline float wb_lds_wait_val<float>(uint32_t addr_tag, uint32_t tag, uint32_t addr_val, uint32_t& budget) { return wb_lds_wait_eq(addr_tag, tag, addr_val, budget); }
template <> __device__ inline double wb_lds_wait_val<double>(uint32_t addr_tag, uint32_t tag, uint32_t addr_val, uint32_t& budget) { return wb_lds_wait_eq64(addr_tag, tag, addr_val, budget); }
#ifndef WB_STREAM_SPIN_BUDGET
#define WB_STREAM_SPIN_BUDGET (1u << 22)     // sleeps of ~64 clocks + an LDS read each: a few tenths of a second per wave and launch
#endif

// HL: the strips of a workgroup exchange their halo columns through LDS (a.halo_rows slots per strip); else scalar loads
// (deeper rings at three waves per SIMD -- 16 x 3, 18 x 3 with a 512-thread launch bound -- were slower: 1.33 ms against 1.17)
// KO ("keys only", round 5): a captured patch is stored as its key alone -- 4 bytes instead of 44 -- and wb_k_scan_build takes the
// ten values from the field itself (the rows were just streamed: a few hundred KB per time step, mostly still in the L2).  The
// patch stores were what made this kernel take 1.10 ms in one placement of its scratch and 1.28 ms in the next (0.2 GB of
// stores beside 6 GB of reads): with a tenth of the bytes the dependence goes with them.
template <typename T, bool VEC, int NLEV, int D, int WPS, bool HL, bool KO = false>
__global__ __launch_bounds__(1024, WPS) void wb_k_scan_stream(WbScanArgs a) {
    const int lane = threadIdx.x & 63;
    const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // strip = wave of the workgroup
    const int NSB = a.nwork;
    const int nlon = a.nlon, nlat = a.nlat;
    const unsigned long long ltmask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int cmax = nlon < a.W - 1 ? nlon : a.W - 1;
    T lvl[NLEV];
#pragma unroll
    for (int l = 0; l < NLEV; ++l) lvl[l] = wb_level_lo<T>(a.levels[l]);
    const int c0 = s * WB_STRIP + 4 * lane;
    const bool live = c0 < cmax;                              // the lane owns at least one square
    int cm = c0 >= nlon ? c0 - nlon : c0;
    if (cm > nlon - 4) cm = nlon - 4;
    // (with the halo exchange the lanes past the row's end own nothing and hand nothing on: they re-read the last lane's
    // 16 bytes instead of wrapping to the row's first columns -- three more lines per row, requested while strip 0's own
    // request for them is in flight, i.e. three more requests to the memory side: 3.2 M of 50 M per launch)
    if (HL && c0 >= nlon) cm = nlon - 4;
    int chs = (s + 1) * WB_STRIP;                             // first column of the next strip: halo of lane 63 (wave-uniform)
    chs = chs >= nlon ? chs - nlon : chs;
    if (chs > nlon - 1) chs = nlon - 1;
    typedef typename WbVec4<T>::type V4;
    // the halo of lane 63 is wave-uniform: read through the constant address space it becomes a scalar load (its own
    // counter; the field is not written by this kernel)
    typedef const __attribute__((address_space(4))) T* CT;
    // this wave's patch buffer in LDS: keys, then the ten values as ten arrays (plain 4 / 8 byte LDS stores, no tuples)
    extern __shared__ __attribute__((aligned(16))) uint32_t wb_stream_lds[];
    constexpr int LW = KO ? WB_STREAM_LBUF : WB_STREAM_LBUF * (1 + 10 * (int)(sizeof(T) / 4));      // dwords per wave
    uint32_t* lkey = wb_stream_lds + (size_t)s * LW;
    T* lval = reinterpret_cast<T*>(lkey + WB_STREAM_LBUF);
    // halo exchange between the strips of the workgroup (when a.halo_rows > 0): strip s publishes, for every row of the
    // band, the first of its 256 columns -- the value lane 63 of strip s - 1 needs --, tagged with (work item, row).  The
    // tag is written after the value and LDS operations execute in order, so a reader that sees the tag sees the value.
    const int HR = HL ? a.halo_rows : 0;
    typedef __attribute__((address_space(3))) volatile T* LdsT;             // explicit LDS pointers: ds_read / ds_write, never flat
    typedef __attribute__((address_space(3))) volatile uint32_t* LdsU;
    const LdsT hval = (LdsT)(wb_stream_lds + NSB * LW);
    const LdsU htag = (LdsU)(hval + NSB * HR);
    // whose first column is this strip's halo: the next strip, or strip 0 for the strip that holds the row's end (the right
    // column of its last square is column 0: the periodic wrap)
    const int hsrc = (s + 1) * WB_STRIP < nlon ? s + 1 : 0;
    // the lane that takes the halo from the exchange: 63, or the last lane inside the row
    const int hlane = (s + 1) * WB_STRIP <= nlon ? 63 : (nlon - s * WB_STRIP) / 4 - 1;
    if (HL) {
        // LDS keeps what the previous launch left there -- tags of the very same work items included: no tag is valid
        // until this workgroup has written it (0 is never a tag)
        for (int i = threadIdx.x; i < NSB * HR; i += blockDim.x) htag[i] = 0u;
        __syncthreads();
    }

    uint32_t spin_budget = WB_STREAM_SPIN_BUDGET;            // (wb_lds_wait_eq: the halo waits of this wave are bounded)
    const int nitems = a.nt * a.nbands;
    // persistent: a workgroup takes the next (time step, row band) item when it is free.  Dealing the items out by workgroup
    // number (it += gridDim.x) assumes every workgroup is resident from the start: with another batch's kernels on the
    // device some are not, start late, and their whole share sets the launch's duration.
    __shared__ int wb_item;
    int it = blockIdx.x;
    for (uint32_t mine = 1;; ++mine) {                        // mine: items this workgroup has taken (the halo tags are built from it)
        if (a.cursor) {
            if (s == 0 && lane == 0) wb_item = atomicAdd(a.cursor, 1);
            __syncthreads();
            it = __builtin_amdgcn_readfirstlane(wb_item);
        } else if (mine > 1) it += gridDim.x;
        if (it >= nitems) break;
        const int t = it / a.nbands, band = it - t * a.nbands;
        const int rb0 = band * a.band_rows;                   // squares of rows rb0 .. rb1 - 1: field rows rb0 .. rb1
        const int rb1 = rb0 + a.band_rows < nlat - 1 ? rb0 + a.band_rows : nlat - 1;
        const int g = it * NSB + s;                           // patch list of this (time step, band, strip)
        const T* f = (const T*)a.field + (size_t)t * nlat * nlon;
        bool nanflag = false;
        int count = 0;                                        // patches appended so far (wave-uniform)
        int fill = 0;                                         // patches waiting in the wave's LDS buffer (wave-uniform)
        uint32_t* ck = a.cap_k + (size_t)g * a.cap_vs;        // the list in HBM: keys, and value k of patch i at cv[k * cap_vs + i]
        T* cv = (T*)a.cap_v + (size_t)g * a.cap_vs * 10;
        auto flush = [&](const bool all) {
            // patches [count, count + n) of the list <- the LDS buffer, eleven stores per 64 patches.  Until the item's last
            // flush only WHOLE 128-byte lines go out (n a multiple of 32, and count stays one: the arrays start on a line) and
            // the few patches behind them move to the front of the buffer.  The scan WITHOUT its stores takes 1.02 ms wherever
            // its buffers lie; the stores -- 0.2 GB beside 6 GB of reads -- add 0.1 ms in one process and 0.3 ms in the next
            // (tools/stream_pattern_probe.py: that is the kernel's two modes).  Written as they came, 30 to 60 patches at a time
            // with every one of the eleven stores beginning and ending inside a line, they added 0.13 / 0.30; whole lines: 0.10 /
            // 0.25.  Padded strides (option scan_pad) and non-temporal stores moved another 1-2 %: the rest is not understood.
            const int n = all ? fill : (fill & ~31);
            int w = a.cap_rec - count;
            w = w < 0 ? 0 : (w < n ? w : n);
            if (a.debug & 1) w = 0;
            for (int j = lane; j < w; j += 64) {
                ck[count + j] = lkey[j];
                if (!KO) {
#pragma unroll
                    for (int k = 0; k < 10; ++k) cv[(size_t)k * a.cap_vs + count + j] = lval[k * WB_STREAM_LBUF + j];
                }
            }
            const int rem = fill - n;                             // < 32, and n >= 32 whenever rem > 0: sources lie behind every destination
            if (rem > 0) {
                uint32_t kk = 0;
                T vv[10];
                if (lane < rem) {
                    kk = lkey[n + lane];
                    if (!KO) {
#pragma unroll
                        for (int k = 0; k < 10; ++k) vv[k] = lval[k * WB_STREAM_LBUF + n + lane];
                    }
                }
                if (lane < rem) {
                    lkey[lane] = kk;
                    if (!KO) {
#pragma unroll
                        for (int k = 0; k < 10; ++k) lval[k * WB_STREAM_LBUF + lane] = vv[k];
                    }
                }
            }
            count += n;
            fill = rem;
        };
        V4 q[D];                                              // the ring: raw row data only (register-resident: every index
        T h[D];                                               // below is a compile-time constant after unrolling)
        auto load = [&](int r, V4& qq, T& hh) {
            // the ring runs D - 1 rows ahead, past the band's last row too: those loads (never consumed) go to one spare row
            // that every workgroup keeps hot in the CU's L1 -- repeating the band's last row, still in flight, cost a
            // second request to the memory side for half of them (2.2 M of 50 M per launch)
            const T* row = r <= rb1 ? f + (size_t)r * nlon : (const T*)a.spare_row;
            if (VEC) qq = *reinterpret_cast<const V4*>(row + cm);
            else {
                int cc = c0; cc = cc >= nlon ? cc - nlon : cc; qq.x = row[cc > nlon - 1 ? nlon - 1 : cc];
                cc = c0 + 1; cc = cc >= nlon ? cc - nlon : cc; qq.y = row[cc > nlon - 1 ? nlon - 1 : cc];
                cc = c0 + 2; cc = cc >= nlon ? cc - nlon : cc; qq.z = row[cc > nlon - 1 ? nlon - 1 : cc];
                cc = c0 + 3; cc = cc >= nlon ? cc - nlon : cc; qq.w = row[cc > nlon - 1 ? nlon - 1 : cc];
            }
            hh = T(0);
            if (!HL) hh = ((CT)row)[chs];
        };
        // slots of this item, indexed by the row number itself; the tag of a row is (work item, row)
        // (the item number of THIS workgroup, not the global one: two items 65536 apart would share their tags, and a
        // workgroup could take both back to back on a grid that is a multiple of 65536; a slot is rewritten by every item)
        const uint32_t tagbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((mine & 0xFFFFu) << 16) - (uint32_t)rb0 + 1u));
        const LdsT pubv = hval + (s * HR - rb0);
        const LdsU pubt = htag + (s * HR - rb0);
        const LdsT srcv = hval + ((hsrc < 0 ? 0 : hsrc) * HR - rb0);
        const LdsU srct = htag + ((hsrc < 0 ? 0 : hsrc) * HR - rb0);
        auto exchange = [&](int r, const V4& qq, T hh) -> T {
            // row r has arrived: publish its first column, then take the halo from the strip to the right (which publishes
            // BEFORE it waits itself: no cycle of waits).  A strip that runs ahead of its neighbour sleeps here.
            if (!HL) return hh;
            const uint32_t tag = tagbase + (uint32_t)r;
            // every lane stores lane 0's value to the SAME word (one LDS access, no divergence: an `if (lane == 0)` here
            // cost the kernel its register allocation), the tag after the value
            const T x0 = __shfl(qq.x, 0, 64);
            pubv[r] = x0;
            pubt[r] = tag;
            if (hsrc < 0) return hh;
            return wb_lds_wait_val<T>((uint32_t)(uintptr_t)(srct + r), tag, (uint32_t)(uintptr_t)(srcv + r), spin_budget);
        };
        T p4, pmn, pmx;                                       // previous row, completed: its halo, minimum and maximum
        auto finish = [&](const V4& qq, T hh, T& v4, T& mn, T& mx) {
            const T sh = wb_shl1(qq.x);                       // unconditional: every lane must be active for the DPP read
            v4 = (lane == (HL ? hlane : 63)) ? hh : sh;
            nanflag |= __builtin_isunordered(qq.x, qq.y) | __builtin_isunordered(qq.z, qq.w);
            mn = wb_min3(qq.x, qq.y, wb_min3(qq.z, qq.w, v4));
            mx = wb_max3(qq.x, qq.y, wb_max3(qq.z, qq.w, v4));
        };
        auto step = [&](int r, const V4& P, const V4& N, T hN) {   // rows (r, r + 1) = (P, N)
            T n4, nmn, nmx;
            finish(N, exchange(r + 1, N, hN), n4, nmn, nmx);
            const T lo = wb_min(pmn, nmn), hi = wb_max(pmx, nmx);
            bool hit = false;
#pragma unroll
            for (int l = 0; l < NLEV; ++l) hit |= (hi > lvl[l]) && !(lo > lvl[l]);
            hit = hit && live;
            const unsigned long long m = __ballot(hit);
            if (m) {
                if (hit) {
                    const int pos = fill + __popcll(m & ltmask);
                    lkey[pos] = ((uint32_t)r << 16) | (uint32_t)c0;
                    if (!KO) {
                        T* d = lval + pos;                    // value k of the buffered patch pos at lval[k * WB_STREAM_LBUF + pos]
                        d[0] = P.x; d[WB_STREAM_LBUF] = P.y; d[2 * WB_STREAM_LBUF] = P.z; d[3 * WB_STREAM_LBUF] = P.w; d[4 * WB_STREAM_LBUF] = p4;
                        d[5 * WB_STREAM_LBUF] = N.x; d[6 * WB_STREAM_LBUF] = N.y; d[7 * WB_STREAM_LBUF] = N.z; d[8 * WB_STREAM_LBUF] = N.w;
                        d[9 * WB_STREAM_LBUF] = n4;
                    }
                }
                fill += __popcll(m);
                if (fill > WB_STREAM_LBUF - 64) flush(false); // no room for another full row (rare: the regular flush is below)
            }
            p4 = n4; pmn = nmn; pmx = nmx;
        };
#pragma unroll
        for (int k = 0; k < D - 1; ++k) load(rb0 + k, q[k], h[k]);
        finish(q[0], exchange(rb0, q[0], h[0]), p4, pmn, pmx);
        int r = rb0;
        for (; r + D <= rb1; r += D) {                        // D - 1 row loads always in flight
#pragma unroll
            for (int k = 0; k < D; ++k) {
                load(r + k + D - 1, q[(k + D - 1) % D], h[(k + D - 1) % D]);
                __builtin_amdgcn_sched_barrier(0);            // a step's work stays between its load and the next one
                step(r + k, q[k], q[(k + 1) % D], h[(k + 1) % D]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (fill >= WB_STREAM_FLUSH) flush(false);        // once per D rows
        }
#pragma unroll
        for (int k = 0; k < D - 1; ++k) {
            if (r + k < rb1) {
                if (k == 0) load(r + D - 1, q[D - 1], h[D - 1]);
                step(r + k, q[k], q[k + 1], h[k + 1]);
            }
        }
        flush(true);
        if (lane == 0) a.cap_n[g] = count;
        if (__any(nanflag && (c0 < nlon)) && lane == 0) {
#pragma unroll
            for (int l = 0; l < NLEV; ++l) atomicOr(&a.status[t * NLEV + l], WB_ST_HAS_NAN);
        }
        if (HL && spin_budget == 0u && lane == 0) {             // a halo tag never came: this step's contours are not to be trusted
#pragma unroll
            for (int l = 0; l < NLEV; ++l) atomicOr(&a.status[t * NLEV + l], WB_ST_SCAN_TIMEOUT);
        }
        if (HL || a.cursor) __syncthreads();                  // nobody overwrites a halo slot a slower strip still waits for (nor wb_item)
    }
}
#endif
