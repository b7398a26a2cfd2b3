// wb_scan.h -- kernel 1: marching-squares scan of the field (THE HBM-bound kernel).
//
// One workgroup per time step streams the (nlat, nlon) field row by row, exactly once, for all
// contour levels at once.  Two consecutive rows of the periodically extended field
// (contour_index.py:94-100: columns nlon..W-1 repeat columns 0..padd-1) live in LDS; every lane
// owns one 2x2 square per 64-square group, classifies it (case = ul>lvl | ur>lvl<<1 | ll>lvl<<2 |
// lr>lvl<<3, squares with a NaN corner skipped), and the rare squares that carry a segment are
// compacted in raster order with wave ballots + one wave scan over the per-group counts.  Each
// emitted segment carries its key (square, sub-segment, case) and the two crossing points already
// rounded half-to-even on the f64 sum c0 + (lvl - v0)/(v1 - v0) (contour_index.py:111).
//
// Algorithmic HBM traffic: nlat*nlon*sizeof(T) read per time step (+ ~12 B per segment written).
#pragma once
#include "wb_core.h"
#include "wb_trace.h"

#define WB_MAX_LEVELS 4

struct WbScanArgs {
    int nt, nlat, nlon, W, padd, nlev, cap_seg;
    const void* field;      // [nt][nlat][nlon], T
    double levels[WB_MAX_LEVELS];
    uint32_t* segk;         // [nt*nlev][cap_seg]
    uint32_t* segf;
    uint32_t* segt;
    uint32_t* rowptr;       // [nt*nlev][nlat]
    int* nseg;              // [nt*nlev]
    uint32_t* status;       // [nt*nlev]
};

template <typename T>
WB_HD inline void wb_emit_square(const WbScanArgs& a, int job, int r, int c, int cs, int ns, double ul, double ur,
                                 double ll, double lr, double level, int idx) {
    const size_t jb = (size_t)job * a.cap_seg;
    for (int sub = 0; sub < ns; ++sub) {
        if (idx + sub < a.cap_seg) {
            const int fs = wb_seg_from(cs, sub), ts = wb_seg_to(cs, sub);
            a.segk[jb + idx + sub] = wb_seg_key(r, c, a.W, sub, cs);
            a.segf[jb + idx + sub] = wb_crossing_point(fs, r, c, ul, ur, ll, lr, level);
            a.segt[jb + idx + sub] = wb_crossing_point(ts, r, c, ul, ur, ll, lr, level);
        }
    }
}

WB_HD inline int wb_square_case(double ul, double ur, double ll, double lr, double level) {
    if (ul != ul || ur != ur || ll != ll || lr != lr) return 0;  // NaN corner: square skipped
    return (ul > level ? 1 : 0) | (ur > level ? 2 : 0) | (ll > level ? 4 : 0) | (lr > level ? 8 : 0);
}

#if defined(WB_HOSTSIM)
// serial statement of the same scan (host simulation only)
template <typename T>
void wb_scan_host(const WbScanArgs& a) {
    for (int t = 0; t < a.nt; ++t) {
        const T* f = (const T*)a.field + (size_t)t * a.nlat * a.nlon;
        for (int l = 0; l < a.nlev; ++l) {
            const int job = t * a.nlev + l;
            const double level = a.levels[l];
            int run = 0;
            uint32_t st = 0;
            for (int r = 0; r < a.nlat; ++r)
                for (int c = 0; c < a.nlon; ++c) {
                    const double v = (double)f[(size_t)r * a.nlon + c];
                    if (v != v) st |= WB_ST_HAS_NAN;
                    if (v == level) st |= WB_ST_LEVEL_HIT;
                }
            for (int r = 0; r < a.nlat - 1; ++r) {
                a.rowptr[(size_t)job * a.nlat + r] = (uint32_t)run;
                for (int c = 0; c < a.W - 1; ++c) {
                    const double ul = (double)f[(size_t)r * a.nlon + c % a.nlon];
                    const double ur = (double)f[(size_t)r * a.nlon + (c + 1) % a.nlon];
                    const double ll = (double)f[(size_t)(r + 1) * a.nlon + c % a.nlon];
                    const double lr = (double)f[(size_t)(r + 1) * a.nlon + (c + 1) % a.nlon];
                    const int cs = wb_square_case(ul, ur, ll, lr, level);
                    const int ns = wb_case_nseg(cs);
                    if (ns) wb_emit_square<T>(a, job, r, c, cs, ns, ul, ur, ll, lr, level, run);
                    run += ns;
                }
            }
            a.rowptr[(size_t)job * a.nlat + a.nlat - 1] = (uint32_t)run;
            a.nseg[job] = run;
            a.status[job] = st;
        }
    }
}
#else
// ------------------------------------------------------------------------------------------------
// gfx950 kernel.  blockDim.x = 256 (4 waves); dynamic LDS = 2*W*sizeof(T) + 2*nlev*64*sizeof(int).
// Limit: ceil((W-1)/64) <= 64 groups per row, i.e. W <= 4097.
#define WB_SCAN_PF 8   // row prefetch depth in registers: covers nlon <= 8*256

template <typename T>
__global__ __launch_bounds__(256) void wb_k_scan(WbScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wb_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x;
    const int W = a.W, nlon = a.nlon, nlat = a.nlat, nlev = a.nlev, padd = a.padd;
    T* rowbuf[2];
    rowbuf[0] = (T*)wb_smem;
    rowbuf[1] = rowbuf[0] + W;
    int* cnt = (int*)(rowbuf[1] + W);  // [2][nlev][64] double-buffered per-group segment counts
    const T* f = (const T*)a.field + (size_t)t * nlat * nlon;
    const int NG = (W - 1 + 63) >> 6;  // 64-square groups per row
    const unsigned long long ltmask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    int base[WB_MAX_LEVELS];
    uint32_t st[WB_MAX_LEVELS];
    double lev[WB_MAX_LEVELS];
#pragma unroll
    for (int l = 0; l < WB_MAX_LEVELS; ++l) {
        base[l] = 0;
        st[l] = 0;
        lev[l] = (l < nlev) ? a.levels[l] : 0.0;
    }

    // rows 0 and 1 -> LDS
    for (int rr = 0; rr < 2 && rr < nlat; ++rr) {
        for (int j = tid; j < nlon; j += 256) {
            const T v = f[(size_t)rr * nlon + j];
            rowbuf[rr][j] = v;
            if (j < padd) rowbuf[rr][nlon + j] = v;
            const double dv = (double)v;
#pragma unroll
            for (int l = 0; l < WB_MAX_LEVELS; ++l)
                if (l < nlev) st[l] |= (dv != dv ? WB_ST_HAS_NAN : 0u) | (dv == lev[l] ? WB_ST_LEVEL_HIT : 0u);
        }
    }
    for (int e = tid; e < 2 * nlev * 64; e += 256) cnt[e] = 0;
    __syncthreads();

    for (int r = 0; r < nlat - 1; ++r) {
        const T* top = rowbuf[r & 1];
        const T* bot = rowbuf[(r + 1) & 1];
        int* cn = cnt + (r & 1) * nlev * 64;

        // prefetch row r+2 into registers (consumed after the barrier)
        T pf[WB_SCAN_PF];
        const bool have_next = (r + 2 < nlat);
        if (have_next) {
#pragma unroll
            for (int m = 0; m < WB_SCAN_PF; ++m) {
                const int j = tid + 256 * m;
                if (j < nlon) pf[m] = f[(size_t)(r + 2) * nlon + j];
            }
        }

        // classify: one square per lane per group, ballots give the per-group counts
        for (int g = wave; g < NG; g += 4) {
            const int c = g * 64 + lane;
            const bool in = c < W - 1;
            const double ul = in ? (double)top[c] : 0.0, ur = in ? (double)top[c + 1] : 0.0;
            const double ll = in ? (double)bot[c] : 0.0, lr = in ? (double)bot[c + 1] : 0.0;
#pragma unroll
            for (int l = 0; l < WB_MAX_LEVELS; ++l) {
                if (l < nlev) {
                    const int ns = in ? wb_case_nseg(wb_square_case(ul, ur, ll, lr, lev[l])) : 0;
                    const unsigned long long m1 = __ballot(ns >= 1), m2 = __ballot(ns == 2);
                    if (lane == 0) cn[l * 64 + g] = __popcll(m1) + __popcll(m2);
                }
            }
        }
        __syncthreads();

        // every wave scans the <=64 group counts of each level (lane = group)
        int gpre[WB_MAX_LEVELS], rowtot[WB_MAX_LEVELS];
        int any = 0;
#pragma unroll
        for (int l = 0; l < WB_MAX_LEVELS; ++l) {
            gpre[l] = 0;
            rowtot[l] = 0;
            if (l < nlev) {
                const int v = (lane < NG) ? cn[l * 64 + lane] : 0;
                const int inc = wb_wave_scan_incl(v);
                gpre[l] = inc - v;
                rowtot[l] = __shfl(inc, 63, 64);
                any |= rowtot[l];
            }
        }
        if (tid == 0) {
#pragma unroll
            for (int l = 0; l < WB_MAX_LEVELS; ++l)
                if (l < nlev) a.rowptr[(size_t)(t * nlev + l) * nlat + r] = (uint32_t)base[l];
        }
        if (any) {  // rare path: this row carries segments
            for (int g = wave; g < NG; g += 4) {
                const int c = g * 64 + lane;
                const bool in = c < W - 1;
                const double ul = in ? (double)top[c] : 0.0, ur = in ? (double)top[c + 1] : 0.0;
                const double ll = in ? (double)bot[c] : 0.0, lr = in ? (double)bot[c + 1] : 0.0;
#pragma unroll
                for (int l = 0; l < WB_MAX_LEVELS; ++l) {
                    if (l < nlev) {
                        const int goff = __shfl(gpre[l], g, 64);
                        const int cs = in ? wb_square_case(ul, ur, ll, lr, lev[l]) : 0;
                        const int ns = wb_case_nseg(cs);
                        const unsigned long long m1 = __ballot(ns >= 1), m2 = __ballot(ns == 2);
                        if (ns) {
                            const int rank = __popcll(m1 & ltmask) + __popcll(m2 & ltmask);
                            wb_emit_square<T>(a, t * nlev + l, r, c, cs, ns, ul, ur, ll, lr, lev[l],
                                              base[l] + goff + rank);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int l = 0; l < WB_MAX_LEVELS; ++l) base[l] += rowtot[l];

        // row r+2 replaces row r in LDS (everyone is past the barrier, nobody reads `top` any more
        // except the emit path above, which this thread has finished; other threads' emit reads are
        // protected by the barrier below only for the NEXT iteration -> write after a second barrier)
        __syncthreads();
        if (have_next) {
            T* dst = rowbuf[r & 1];
#pragma unroll
            for (int m = 0; m < WB_SCAN_PF; ++m) {
                const int j = tid + 256 * m;
                if (j < nlon) {
                    const T v = pf[m];
                    dst[j] = v;
                    if (j < padd) dst[nlon + j] = v;
                    const double dv = (double)v;
#pragma unroll
                    for (int l = 0; l < WB_MAX_LEVELS; ++l)
                        if (l < nlev) st[l] |= (dv != dv ? WB_ST_HAS_NAN : 0u) | (dv == lev[l] ? WB_ST_LEVEL_HIT : 0u);
                }
            }
            for (int j = tid + 256 * WB_SCAN_PF; j < nlon; j += 256) {  // very wide grids: late loads
                const T v = f[(size_t)(r + 2) * nlon + j];
                dst[j] = v;
                if (j < padd) dst[nlon + j] = v;
                const double dv = (double)v;
#pragma unroll
                for (int l = 0; l < WB_MAX_LEVELS; ++l)
                    if (l < nlev) st[l] |= (dv != dv ? WB_ST_HAS_NAN : 0u) | (dv == lev[l] ? WB_ST_LEVEL_HIT : 0u);
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int l = 0; l < WB_MAX_LEVELS; ++l) {
        if (l < nlev) {
            const int job = t * nlev + l;
            if (st[l]) atomicOr(&a.status[job], st[l]);
            if (tid == 0) {
                a.rowptr[(size_t)job * nlat + nlat - 1] = (uint32_t)base[l];
                a.nseg[job] = base[l];
            }
        }
    }
}
#endif
