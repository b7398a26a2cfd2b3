// wb_spatial.h -- the two pre-processing steps next to the hot path.
//
//   calculate_smoothed_field (processing/spatial.py:60-128): per time step `passes` x
//       scipy.ndimage.convolve(temp, weights, mode) / np.sum(weights), then rows arange(-b, b) -> NaN.
//   calculate_momentum_flux  (processing/spatial.py:27-57):  (u - mean_lon u) * (v - mean_lon v).
//
// Bit-exact restatements of the third-party arithmetic (scipy NI_Correlate tap order and double accumulator,
// numpy's pairwise row sum, numpy >= 2 dtype promotion) -- see oracle/spatial.py for the pinned CPU version.
//
// Kernels
//   wb_k_smooth_plus<T, P>  the default case -- a 3x3 plus-shaped kernel with mode="wrap" -- with P passes FUSED
//       in one sweep: a wave owns a 128-column strip (2 columns per lane) and streams down the rows; each pass
//       is a pipeline stage that keeps a 3-row window of the previous stage in registers, horizontal neighbours
//       come from the adjacent lane by DPP wave shifts.  HBM traffic: input once (x 128/(128-2H) halo columns)
//       + float64 output once, whatever P is (unfused: 16 B per cell and pass).  No LDS.
//   wb_k_smooth_pass        any kernel up to 7x7 / any scipy boundary mode, one pass per launch.
//   wb_k_mflux<T>           one workgroup per (time, lat) row: numpy's pairwise sum tree evaluated in parallel
//       (8 strided partials per <=128-element leaf, leaves combined by a host-built postfix program).
#pragma once
#include "wb_spmd.h"

#define WB_SM_MAXTAP 49
#define WB_SM_REFLECT 0
#define WB_SM_CONSTANT 1
#define WB_SM_NEAREST 2
#define WB_SM_MIRROR 3
#define WB_SM_WRAP 4

struct WbSmoothArgs {
    long long nt;
    int nlat, nlon;
    const void* in;          // [nt][nlat][nlon] float (in_f32) or double
    double* out;             // [nt][nlat][nlon]
    int in_f32;              // input cube is float32
    int cast_f32;            // first pass of a float32 cube: the accumulator is rounded to float32 before the division
    int nan_rows;            // > 0 on the last pass: rows [0, b) and [nlat-b, nlat) are NaN (spatial.py:104-107)
    int mode;
    double wsum;             // np.sum(weights)
    double rsum;             // RN(1 / wsum)
    int fastdiv;             // wsum qualifies for wb_sm_div's short sequence
    int unit4;               // fused kernel: the four outer taps are present with weight exactly 1
    // taps in accumulation order (C order of the flipped kernel, zero weights dropped)
    int ntap;
    int tr[WB_SM_MAXTAP], tc[WB_SM_MAXTAP];
    double tw[WB_SM_MAXTAP];
    // fused plus-kernel: weights / presence of (up, left, centre, right, down) in accumulation order
    double pw[5];
    int puse[5];
    int nstrips, strip_w, halo;
    int vec;                 // rows can be moved as aligned 2-element vectors (nlon even, bases aligned)
};

// scipy ni_support.c boundary extension; -1 = constant (cval 0)
static WB_HD inline int wb_sm_extend(int i, int n, int mode) {
    if (i >= 0 && i < n) return i;
    switch (mode) {
        case WB_SM_CONSTANT: return -1;
        case WB_SM_NEAREST: return i < 0 ? 0 : n - 1;
        case WB_SM_WRAP: { int m = i % n; return m < 0 ? m + n : m; }
        case WB_SM_REFLECT: {
            const int p = 2 * n;
            int m = i % p; if (m < 0) m += p;
            return m < n ? m : p - 1 - m;
        }
        default: {   // mirror
            if (n == 1) return 0;
            const int p = 2 * n - 2;
            int m = i % p; if (m < 0) m += p;
            return m < n ? m : p - m;
        }
    }
}

// tmp / S.  With `fast` (host: S = k * 2^j, k an odd integer below 2^26, 2^-60 < |S| < 2^60) the quotient is formed
// as q = a*y, r = fma(-S, q, a), q' = fma(r, y, q) with y = RN(1/S) -- Markstein's sequence.  It is the correctly
// rounded a / S for such S: q is within one ulp of x = a/S, so r = S(x - q) is exact and q' = RN(x + (x - q) * d),
// |d| <= 2^-53, i.e. RN of x perturbed by less than 2^(e-105) (x in [2^e, 2^(e+1))); a rounding boundary m (an odd
// multiple of 2^(e-53), or of 2^(e-54) below 2^e) cannot be that close because a - S*m is a non-zero odd multiple of
// 2^j * 2^(e-54) at least, which puts |x - m| above 2^(e-54) / k.  Zeros keep their sign through q; infinities, NaN
// and magnitudes near the ends of the exponent range take the ordinary division.
static WB_HD inline double wb_sm_div(double a, double S, double y, bool fast) {
    if (!fast) return a / S;
    const double q = a * y;
    double q1 = __builtin_fma(__builtin_fma(-S, q, a), y, q);
    const double aa = __builtin_fabs(a);
    if (!(aa > 0x1p-900 && aa < 0x1p900)) {
        if (a == 0.0) q1 = q;
        else q1 = a / S;
    }
    return q1;
}

#define WB_SM_CHUNK 2048
WB_KERNEL(wb_k_smooth_pass, WbSmoothArgs a) {
    const long long ncell = a.nt * a.nlat * a.nlon;
    const long long nchunk = (ncell + WB_SM_CHUNK - 1) / WB_SM_CHUNK;
    const double qnan = __builtin_nan("");
    for (long long ch = WB_BLOCK_ID; ch < nchunk; ch += WB_GRID_DIM) {
        const long long base = ch * WB_SM_CHUNK;
        const long long n = ncell - base < WB_SM_CHUNK ? ncell - base : WB_SM_CHUNK;
        WB_PFOR(i, n)
            const long long cell = base + i;
            const int c = (int)(cell % a.nlon);
            const long long tr = cell / a.nlon;
            const int r = (int)(tr % a.nlat);
            const long long plane = (tr / a.nlat) * a.nlat * a.nlon;
            double tmp = 0.0;
            for (int k = 0; k < a.ntap; ++k) {
                const int rr = wb_sm_extend(r + a.tr[k], a.nlat, a.mode), cc = wb_sm_extend(c + a.tc[k], a.nlon, a.mode);
                double x = 0.0;
                if (rr >= 0 && cc >= 0) {
                    const long long idx = plane + (long long)rr * a.nlon + cc;
                    x = a.in_f32 ? (double)((const float*)a.in)[idx] : ((const double*)a.in)[idx];
                }
                tmp += a.tw[k] * x;
            }
            if (a.cast_f32) tmp = (double)(float)tmp;
            double y = wb_sm_div(tmp, a.wsum, a.rsum, a.fastdiv != 0);
            if (a.nan_rows > 0 && (r < a.nan_rows || r >= a.nlat - a.nan_rows)) y = qnan;
            a.out[cell] = y;
        WB_END_PFOR
    }
}

// ------------------------------------------------------------------------------------------------ momentum flux
#define WB_MF_MAXLEAF 64
#define WB_MF_MAXLVL 16
struct WbMfluxArgs {
    long long nrows;         // nt * nlat
    int nlon;
    const void* u;
    const void* v;
    void* out;
    // numpy's pairwise summation tree over one row: leaves of <= 128 elements; inner nodes as "slot[dst] += slot[src]"
    // (a node's value lives in the slot of its leftmost leaf), grouped by height so one level is one parallel step
    int nleaf, nlvl;
    int leaf_off[WB_MF_MAXLEAF];
    short leaf_len[WB_MF_MAXLEAF];
    short lvl_start[WB_MF_MAXLVL + 1];
    short op_dst[WB_MF_MAXLEAF], op_src[WB_MF_MAXLEAF];
};

// numpy pairwise_sum recursion on [off, off+n) (loops_utils.h.src): returns the height of the node or -1 when a table
// is full; ops are collected per height in tmp_* and flattened by wb_mf_plan
struct WbMfPlanTmp { int n = 0; short h[WB_MF_MAXLEAF], dst[WB_MF_MAXLEAF], src[WB_MF_MAXLEAF]; };
static inline int wb_mf_plan_rec(WbMfluxArgs& a, WbMfPlanTmp& t, int off, int n, int* slot) {
    if (n <= 128) {
        if (a.nleaf >= WB_MF_MAXLEAF) return -1;
        a.leaf_off[a.nleaf] = off; a.leaf_len[a.nleaf] = (short)n;
        *slot = a.nleaf++;
        return 0;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    int sl = 0, sr = 0;
    const int hl = wb_mf_plan_rec(a, t, off, n2, &sl);
    if (hl < 0) return -1;
    const int hr = wb_mf_plan_rec(a, t, off + n2, n - n2, &sr);
    if (hr < 0 || t.n >= WB_MF_MAXLEAF) return -1;
    const int h = (hl > hr ? hl : hr) + 1;
    if (h > WB_MF_MAXLVL) return -1;
    t.h[t.n] = (short)h; t.dst[t.n] = (short)sl; t.src[t.n] = (short)sr; ++t.n;
    *slot = sl;
    return h;
}
static inline int wb_mf_plan(WbMfluxArgs& a, int n) {
    WbMfPlanTmp t;
    int slot = 0;
    const int H = wb_mf_plan_rec(a, t, 0, n, &slot);
    if (H < 0) return 1;
    a.nlvl = H;
    int k = 0;
    for (int h = 1; h <= H; ++h) {
        a.lvl_start[h - 1] = (short)k;
        for (int i = 0; i < t.n; ++i) if (t.h[i] == h) { a.op_dst[k] = t.dst[i]; a.op_src[k] = t.src[i]; ++k; }
    }
    a.lvl_start[H] = (short)k;
    return 0;
}

template <typename T>
WB_KERNEL(wb_k_mflux, WbMfluxArgs a) {
    WB_SHARED(T, part, 2 * WB_MF_MAXLEAF * 8);
    WB_SHARED(T, leafsum, 2 * WB_MF_MAXLEAF);
    WB_SHARED(int, nanc, 2);
    const int L = a.nleaf, nlon = a.nlon;
    for (long long row = WB_BLOCK_ID; row < a.nrows; row += WB_GRID_DIM) {
        const T* ur = (const T*)a.u + row * nlon;
        const T* vr = (const T*)a.v + row * nlon;
        WB_PFOR(f, 2)
            nanc[f] = 0;
        WB_END_PFOR
        // 8 strided partial sums per leaf (r[k] = a[k]; r[k] += a[i + k], i = 8, 16, ..), NaN counted and taken as 0.
        // A task owns chain k of one leaf for BOTH fields; its <= 16 + 16 elements are fetched by independent loads
        // first (one memory latency per task), then added in order.
        WB_PFOR(task, L * 8)
            const int k = (int)(task & 7), leaf = (int)(task >> 3);
            const int n = a.leaf_len[leaf], body = n - (n % 8), cnt = n >= 8 ? body / 8 : 0;
            const T* xu = ur + a.leaf_off[leaf] + k;
            const T* xv = vr + a.leaf_off[leaf] + k;
            T eu[16], ev[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                eu[i] = i < cnt ? xu[8 * i] : (T)0;
                ev[i] = i < cnt ? xv[8 * i] : (T)0;
            }
            int nu = 0, nv = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (eu[i] != eu[i]) { eu[i] = (T)0; ++nu; }
                if (ev[i] != ev[i]) { ev[i] = (T)0; ++nv; }
            }
            T ru = eu[0], rv = ev[0];
#pragma unroll
            for (int i = 1; i < 16; ++i)
                if (i < cnt) { ru = ru + eu[i]; rv = rv + ev[i]; }
            part[task] = ru;
            part[L * 8 + task] = rv;
            if (nu) wb_atomic_add(&nanc[0], nu);
            if (nv) wb_atomic_add(&nanc[1], nv);
        WB_END_PFOR
        WB_PFOR(task, 2 * L)
            const int leaf = (int)(task % L), f = (int)(task / L);
            const T* x = (f ? vr : ur) + a.leaf_off[leaf];
            const int n = a.leaf_len[leaf];
            const T* r = part + task * 8;
            T res;
            int i, nn = 0;
            if (n >= 8) {
                res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                i = n - (n % 8);
            } else {
                res = (T)0;
                i = 0;
            }
            for (; i < n; ++i) { const T e = x[i]; const bool isn = e != e; nn += isn; res = res + (isn ? (T)0 : e); }
            leafsum[task] = res;
            if (nn) wb_atomic_add(&nanc[f], nn);
        WB_END_PFOR
        for (int h = 0; h < a.nlvl; ++h) {
            const int o0 = a.lvl_start[h], no = a.lvl_start[h + 1] - o0;
            WB_PFOR(q, 2 * no)
                const int f = (int)(q / no), op = o0 + (int)(q % no);
                leafsum[f * L + a.op_dst[op]] = leafsum[f * L + a.op_dst[op]] + leafsum[f * L + a.op_src[op]];
            WB_END_PFOR
        }
        // np.nanmean: true_divide(total, count) evaluated in double, stored in the array dtype
        const T um = (T)((double)leafsum[0] / (double)(nlon - nanc[0]));
        const T vm = (T)((double)leafsum[L] / (double)(nlon - nanc[1]));
        T* o = (T*)a.out + row * nlon;
        WB_PFOR(c, nlon)
            o[c] = (ur[c] - um) * (vr[c] - vm);
        WB_END_PFOR
    }
}

// ------------------------------------------------------------------------------------------------ fused passes
#if !defined(WB_HOSTSIM)
// value of the previous lane (lane 0: 0) / of the next lane (lane 63: 0); DPP wave shifts, all lanes must be active
__device__ inline double wb_sp_prev(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ inline double wb_sp_next(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int V> struct WbInt { static constexpr int value = V; };
template <typename T> struct WbVec2;
template <> struct WbVec2<float> { typedef float2 type; };
template <> struct WbVec2<double> { typedef double2 type; };

// One strip of one time step.  UNIT4 = the four outer taps have weight 1 (no multiply).  FAST = divisions by wb_sm_div's
// short sequence WITHOUT its per-call range test: the host admits it only for 2^-20 <= |S|, |centre weight| <= 2^20, and the
// strip watches its inputs instead -- if every input is NaN, zero or of magnitude 2^-100 .. 2^100 then every intermediate
// of up to 8 passes is NaN, +0 or of magnitude 2^-850 .. 2^430 (a non-zero sum of such terms is at least one ulp of the
// smallest, a pass shrinks by at most 2^-93 and grows by at most 2^41), where the sequence is exact.  Returns true if an
// input fell outside; the caller then redoes the strip with the IEEE division.
template <typename T, int P, bool CAST, bool UNIT4, bool FAST>
__device__ __forceinline__ bool wb_sp_strip(const WbSmoothArgs& a) {
    const int lane = threadIdx.x & 63;
    const long long g = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // work item = (time step, strip)
    if (g >= a.nt * a.nstrips) return false;                                    // whole waves leave together
    const long long t = g / a.nstrips;
    const int s = (int)(g - t * a.nstrips);
    const int nlat = a.nlat, nlon = a.nlon, H = a.halo;
    const int cs = s * a.strip_w;                                               // first output column of the strip
    const int v0 = cs - H + 2 * lane;                                           // virtual column of the lane's first cell
    int c0 = v0 % nlon; if (c0 < 0) c0 += nlon;
    int c1 = (v0 + 1) % nlon; if (c1 < 0) c1 += nlon;
    const bool vec = a.vec != 0;                                                // then c0 is even and c1 = c0 + 1
    const bool st0 = v0 >= cs && v0 < cs + a.strip_w && v0 < nlon;
    const bool st1 = v0 + 1 >= cs && v0 + 1 < cs + a.strip_w && v0 + 1 < nlon;
    const T* f = (const T*)a.in + t * nlat * nlon;
    double* o = a.out + t * nlat * nlon;
    const double wU = a.pw[0], wL = a.pw[1], wC = a.pw[2], wR = a.pw[3], wD = a.pw[4];
    const bool uU = a.puse[0], uL = a.puse[1], uC = a.puse[2], uR = a.puse[3], uD = a.puse[4];
    const double S = a.wsum, Y = a.rsum, qnan = __builtin_nan("");
    bool bad = false;

    double win[P][3][2];                                                        // stage s-1 output: rows (newest-2 .. newest)
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
        for (int w = 0; w < 3; ++w) { win[q][w][0] = 0.0; win[q][w][1] = 0.0; }

    T ld[3][2];                                                                 // prefetch ring: input rows tau+1 .. tau+3
    auto load = [&](int tau, T* dst) {
        int r = (tau - P) % nlat; if (r < 0) r += nlat;
        const T* row = f + (long long)r * nlon;
        if (vec) {
            const typename WbVec2<T>::type q = *reinterpret_cast<const typename WbVec2<T>::type*>(row + c0);
            dst[0] = q.x; dst[1] = q.y;
        } else {
            dst[0] = row[c0]; dst[1] = row[c1];
        }
    };
    auto div = [&](double v) {
        if (!FAST) return v / S;
        const double q = v * Y;
        return __builtin_fma(__builtin_fma(-S, q, v), Y, q);
    };
    const int nticks = nlat + 2 * P;
    auto tick = [&](int tau, auto ph) {
        constexpr int N = decltype(ph)::value, M = (N + 2) % 3, O = (N + 1) % 3;   // newest / middle / oldest slot
        double x0 = (double)ld[N][0], x1 = (double)ld[N][1];
        if (tau + 3 < nticks) load(tau + 3, ld[N]);
        if (FAST) {
            const double a0 = __builtin_fabs(x0), a1 = __builtin_fabs(x1);     // comparisons are false for NaN
            bad |= (a0 > 0x1p100) | (a1 > 0x1p100) | ((a0 < 0x1p-100) & (a0 > 0.0)) | ((a1 < 0x1p-100) & (a1 > 0.0));
        }
#pragma unroll
        for (int q = 0; q < P; ++q) {
            win[q][N][0] = x0; win[q][N][1] = x1;
            const double b0 = win[q][M][0], b1 = win[q][M][1];
            const double bl = wb_sp_prev(b1), br = wb_sp_next(b0);
            double t0 = 0.0, t1 = 0.0;
            if (UNIT4) {
                t0 += win[q][O][0]; t1 += win[q][O][1];
                t0 += bl; t1 += b0;
                if (uC) { t0 += wC * b0; t1 += wC * b1; }
                t0 += b1; t1 += br;
                t0 += x0; t1 += x1;
            } else {
                if (uU) { t0 += wU * win[q][O][0]; t1 += wU * win[q][O][1]; }
                if (uL) { t0 += wL * bl; t1 += wL * b0; }
                if (uC) { t0 += wC * b0; t1 += wC * b1; }
                if (uR) { t0 += wR * b1; t1 += wR * br; }
                if (uD) { t0 += wD * x0; t1 += wD * x1; }
            }
            if (CAST && q == 0) { t0 = (double)(float)t0; t1 = (double)(float)t1; }
            x0 = div(t0); x1 = div(t1);
        }
        const int r = tau - 2 * P;                                              // row the last stage just finished
        if (r >= 0) {
            if (a.nan_rows > 0 && (r < a.nan_rows || r >= nlat - a.nan_rows)) { x0 = qnan; x1 = qnan; }
            double* dst = o + (long long)r * nlon;
            if (vec) {
                if (st0) *reinterpret_cast<double2*>(dst + c0) = make_double2(x0, x1);   // st0 == st1: strip_w, nlon even
            } else {
                if (st0) dst[c0] = x0;
                if (st1) dst[c1] = x1;
            }
        }
    };
    load(0, ld[0]);
    if (1 < nticks) load(1, ld[1]);
    if (2 < nticks) load(2, ld[2]);
    int tau = 0;
    for (; tau + 3 <= nticks; tau += 3) {
        tick(tau, WbInt<0>());
        tick(tau + 1, WbInt<1>());
        tick(tau + 2, WbInt<2>());
    }
    if (tau < nticks) { tick(tau, WbInt<0>()); ++tau; }
    if (tau < nticks) { tick(tau, WbInt<1>()); ++tau; }
    return bad;
}

// T = element type of the input cube, P = passes fused in this launch, CAST = stage 1 rounds its accumulator to float
template <typename T, int P, bool CAST, bool UNIT4, bool FAST>
__global__ __launch_bounds__(256) void wb_k_smooth_plus(WbSmoothArgs a) {
    if (FAST) {
        if (__any(wb_sp_strip<T, P, CAST, UNIT4, true>(a))) (void)wb_sp_strip<T, P, CAST, UNIT4, false>(a);
    } else {
        (void)wb_sp_strip<T, P, CAST, UNIT4, false>(a);
    }
}
#endif
