// wb_scan.h -- kernel 1: marching-squares scan of the field (THE HBM-bound kernel).
//
// Work decomposition: the nlon base columns are cut into strips of 256 squares; ONE WAVE streams one strip
// of one time step top to bottom, for all contour levels at once, with no LDS and no barrier (persistent
// waves loop over (time step, strip) work items).  A lane owns 4 consecutive squares per row: one float4 /
// double4 load (1 KiB per wave-load), right halo from the next lane by a DPP wave shift, six row buffers
// that rotate by name so that four row loads are always in flight.  Every field value is loaded from HBM
// once; the periodic columns nlon..W-1 of contour_index.py:94-100 are never read: a square of the first
// padd-1 base columns is emitted twice, at c and at c + nlon (second copy -> the strip's "periodic copy" list).
//
// Classification per lane and row: the lane's 2 x 5 patch of values is connected, so it contains a crossing
// of lvl iff  min(patch) <= lvl < max(patch), i.e.
//
//     max(patch) > lvl_lo  and not  min(patch) > lvl_lo ,     lvl_lo = largest T <= level,
//
// evaluated in the field's own precision.  That is EXACT, not a conservative filter: for a T-valued v,
// (double)v > level  <=>  v > lvl_lo, which is skimage's strict `>` on the float64 copy
// (contour_index.py:103).  Only when some lane of the wave sees a crossing does the wave take the emit path:
// per-square case, ballots for raster-order offsets inside the strip; the segment's key and the four corner
// values are stored (the f64 crossing coordinate c0 + (lvl - v0)/(v1 - v0) and its round-half-even,
// contour_index.py:111, are evaluated when the trace kernel merges the strips).  Squares with a NaN corner
// are skipped like skimage does.
//
// Output: per (job, list) segments in raster order (key + corner values); the trace kernel merges the
// <= 32 sorted lists of a job by rank (wb_trace.h, link kernel).
//
// Algorithmic HBM traffic: nlat*nlon*sizeof(T) per time step (+ 4+4*sizeof(T) B per segment, < 1 %).
#pragma once
#include "wb_core.h"

#define WB_MAX_LEVELS 4
#define WB_STRIP 256          // squares per strip = 64 lanes x 4 squares
#define WB_MAX_STRIPS 16      // nlon <= 4096
#define WB_MAX_LISTS 32       // base strips + their periodic copies

#define WB_ST_OVERFLOW_SEG 1u   // more segments than capacity: job skipped, host retries with a larger cap
#define WB_ST_LEVEL_HIT 2u      // some cell on a contour equals the level exactly (joined by the vertex rule, wb_trace.h)
#define WB_ST_HAS_NAN 4u
#define WB_ST_HASH_FULL 8u
#define WB_ST_OVERFLOW_PAIRS 16u
#define WB_ST_SCAN_TIMEOUT 256u  // the streaming scan's halo exchange ran out of its spin budget (wb_stream.h): the step's contours are not to be trusted

struct WbScanArgs {
    int nt, nlat, nlon, W, nlev, cap_strip;
    int nwork;              // strips of 256 base columns that are actually streamed: ceil(nlon / 256)
    int nstrips;            // segment lists per job: nwork base lists + one "periodic copy" list for every base strip
                            // that reaches into the first padd-1 columns (list nwork + s holds the copies of strip s)
    int padd;               // W - nlon
    const void* field;      // [nt][nlat][nlon], T
    double levels[WB_MAX_LEVELS];
    uint32_t* stg_k;        // [njobs][nstrips][cap_strip] key
    void* stg_v;            // [njobs][nstrips][cap_strip][4] corner values ul, ur, ll, lr (T)
    int* stg_n;             // [njobs][nstrips] segments found (may exceed cap_strip: overflow)
    uint32_t* status;       // [njobs]
    // two-kernel scan: patches (2 x 5 values of one lane and row) that hold a crossing of some level, per (time step, strip)
    uint32_t* cap_k;        // [lists][cap_rec]      r << 16 | c0
    void* cap_v;            // [lists][10][cap_rec]  upper row v0..v4, lower row v0..v4 (T): value k of patch i at [k * cap_rec + i]
    int* cap_n;             // [nt][nwork] patches found (may exceed cap_rec: overflow)
    int cap_rec;
    int cap_vs;             // stride (elements) between a list's key array and the next list's, and between the ten value planes of a
                            // list: cap_rec + a pad -- with the power of two itself every plane of every list started on the same
                            // 4 KB phase, and the scan's patch stores took 0.1 or 0.3 ms by where the scratch lay
    int debug;              // development: bit 0 = the streaming scan stores no patches (streaming ceiling of the traversal)
    int keys_only;          // two-kernel scan: the patch lists hold keys only (cap_v unused); wb_k_scan_build reads the values from the field
    int nbands, band_rows;  // streaming scan: a time step is cut into nbands bands of band_rows rows of squares; the lists above
                            // are then per (time step, band, strip): index (t * nbands + band) * nwork + strip
    const void* spare_row;  // streaming scan: nlon values that every workgroup reads instead of rows past its band's end (always cached)
    int halo_rows;          // streaming scan: > 0 = the strips of a workgroup hand each other the halo column through LDS
                            // (band_rows + 1 slots per strip); 0 = every strip loads its halo itself (a scalar load)
    int* cursor;            // streaming scan: next (time step, band) work item, zero at launch (WbCtl::scan_cursor); NULL = items
                            // dealt out by workgroup number
};

// a segment is stored as its key + the four corner values of its square; the crossing coordinates
// (f64 divisions) are evaluated by the trace kernel's merge phase, which keeps this kernel's registers low
template <typename T>
WB_HD inline void wb_store_square(uint32_t* stg_k, T* stg_v, int cap_strip, int W, int r, int c, int cs, int ns, T ul, T ur,
                                  T ll, T lr, int idx) {
    for (int sub = 0; sub < ns; ++sub) {
        if (idx + sub < cap_strip) {
            stg_k[idx + sub] = wb_seg_key(r, c, W, sub, cs);
            T* v = stg_v + (size_t)(idx + sub) * 4;
            v[0] = ul; v[1] = ur; v[2] = ll; v[3] = lr;
        }
    }
}

WB_HD inline int wb_square_case(double ul, double ur, double ll, double lr, double level) {
    if (ul != ul || ur != ur || ll != ll || lr != lr) return 0;  // NaN corner: square skipped
    return (ul > level ? 1 : 0) | (ur > level ? 2 : 0) | (ll > level ? 4 : 0) | (lr > level ? 8 : 0);
}

#if defined(WB_HOSTSIM)
// serial statement of the same scan with the same output layout (host simulation only)
template <typename T>
void wb_scan_host(const WbScanArgs& a) {
    for (int t = 0; t < a.nt; ++t) {
        const T* f = (const T*)a.field + (size_t)t * a.nlat * a.nlon;
        for (int l = 0; l < a.nlev; ++l) {
            const int job = t * a.nlev + l;
            const double level = a.levels[l];
            uint32_t st = 0;
            for (int r = 0; r < a.nlat; ++r)
                for (int c = 0; c < a.nlon; ++c) {
                    const double v = (double)f[(size_t)r * a.nlon + c];
                    if (v != v) st |= WB_ST_HAS_NAN;
                }
            const int cmax = a.nlon < a.W - 1 ? a.nlon : a.W - 1;     // base squares: c < min(nlon, W-1)
            for (int s = 0; s < a.nwork; ++s) {
                const size_t base = ((size_t)job * a.nstrips + s) * a.cap_strip;
                const size_t cbase = ((size_t)job * a.nstrips + a.nwork + s) * a.cap_strip;
                const bool has_copy = a.nwork + s < a.nstrips;
                int run = 0, crun = 0;
                for (int r = 0; r < a.nlat - 1; ++r) {
                    for (int c = s * WB_STRIP; c < (s + 1) * WB_STRIP && c < cmax; ++c) {
                        const double ul = (double)f[(size_t)r * a.nlon + c % a.nlon];
                        const double ur = (double)f[(size_t)r * a.nlon + (c + 1) % a.nlon];
                        const double ll = (double)f[(size_t)(r + 1) * a.nlon + c % a.nlon];
                        const double lr = (double)f[(size_t)(r + 1) * a.nlon + (c + 1) % a.nlon];
                        const int cs = wb_square_case(ul, ur, ll, lr, level);
                        const int ns = wb_case_nseg(cs);
                        if (ns) wb_store_square<T>(a.stg_k + base, (T*)a.stg_v + base * 4, a.cap_strip, a.W, r, c, cs, ns, (T)ul, (T)ur, (T)ll, (T)lr, run);
                        run += ns;
                        if (has_copy && c < a.padd - 1) {                 // the same square in the periodic columns
                            if (ns) wb_store_square<T>(a.stg_k + cbase, (T*)a.stg_v + cbase * 4, a.cap_strip, a.W, r, c + a.nlon, cs, ns, (T)ul, (T)ur, (T)ll, (T)lr, crun);
                            crun += ns;
                        }
                    }
                }
                a.stg_n[job * a.nstrips + s] = run;
                if (has_copy) a.stg_n[job * a.nstrips + a.nwork + s] = crun;
            }
            a.status[job] = st;
        }
    }
}

// Host simulation of the TWO-kernel scan (wb_k_scan_stream + wb_k_scan_build), statement by statement what the device
// kernels decide: which 2 x 5 patches a (time step, row band, strip) list captures -- min(patch) <= lvl_lo < max(patch) in
// the field's own precision for ANY level, NaN operands ignored as v_min / v_max ignore them, the halo column wrapping to
// column 0 at the row's end, a band re-reading its neighbour's first row -- and how the build expands the patches into the
// per-level segment lists (square cases against lvl_lo, raster positions, periodic copies, overflow counts).
template <typename T> inline T wb_level_lo_host(double level);
template <> inline float wb_level_lo_host<float>(double level) {
    float f = (float)level;                                  // round to nearest, then down to the largest float <= level
    if ((double)f > level) f = nextafterf(f, -INFINITY);
    return f;
}
template <> inline double wb_level_lo_host<double>(double level) { return level; }
template <typename T> inline T wb_minnum(T a, T b) { return a != a ? b : (b != b ? a : (a < b ? a : b)); }
template <typename T> inline T wb_maxnum(T a, T b) { return a != a ? b : (b != b ? a : (a > b ? a : b)); }

template <typename T>
void wb_scan_capture_host(const WbScanArgs& a) {
    const int nlon = a.nlon, nlat = a.nlat, NSB = a.nwork;
    const int cmax = nlon < a.W - 1 ? nlon : a.W - 1;
    T lvl[WB_MAX_LEVELS];
    for (int l = 0; l < a.nlev; ++l) lvl[l] = wb_level_lo_host<T>(a.levels[l]);
    for (int t = 0; t < a.nt; ++t) {
        const T* f = (const T*)a.field + (size_t)t * nlat * nlon;
        bool nan = false;
        for (int band = 0; band < a.nbands; ++band) {
            const int rb0 = band * a.band_rows, rb1 = rb0 + a.band_rows < nlat - 1 ? rb0 + a.band_rows : nlat - 1;
            for (int s = 0; s < NSB; ++s) {
                const size_t g = ((size_t)t * a.nbands + band) * NSB + s;
                uint32_t* ck = a.cap_k + g * a.cap_vs;
                T* cv = (T*)a.cap_v + g * a.cap_vs * 10;
                int count = 0;
                for (int r = rb0; r < rb1; ++r) {
                    for (int lane = 0; lane < 64; ++lane) {
                        const int c0 = s * WB_STRIP + 4 * lane;
                        if (c0 >= cmax) continue;                              // the lane owns no square
                        T v[10];
                        for (int row = 0; row < 2; ++row)
                            for (int k = 0; k < 5; ++k) {
                                int c = c0 + k;
                                if (c >= nlon) c -= nlon;                      // the halo past the row's end is column 0 (periodic wrap)
                                if (c > nlon - 1) c = nlon - 1;
                                v[5 * row + k] = f[(size_t)(r + row) * nlon + c];
                            }
                        for (int k = 0; k < 4; ++k) nan |= (v[k] != v[k]) || (v[5 + k] != v[5 + k]);
                        T lo = v[0], hi = v[0];
                        for (int k = 1; k < 10; ++k) { lo = wb_minnum(lo, v[k]); hi = wb_maxnum(hi, v[k]); }
                        bool hit = false;
                        for (int l = 0; l < a.nlev; ++l) hit |= (hi > lvl[l]) && !(lo > lvl[l]);
                        if (!hit) continue;
                        if (count < a.cap_rec) {
                            ck[count] = ((uint32_t)r << 16) | (uint32_t)c0;
                            for (int k = 0; k < 10; ++k) cv[(size_t)k * a.cap_vs + count] = v[k];
                        }
                        ++count;
                    }
                }
                a.cap_n[g] = count;
            }
        }
        if (nan) for (int l = 0; l < a.nlev; ++l) a.status[t * a.nlev + l] |= WB_ST_HAS_NAN;
    }
}

template <typename T>
void wb_scan_build_host(const WbScanArgs& a) {
    const int NS = a.nstrips, NSB = a.nwork, W = a.W, nlon = a.nlon;
    const int cmax = nlon < W - 1 ? nlon : W - 1;
    const size_t cs = (size_t)a.cap_vs;                       // (strides; the capacity is cap_rec)
    for (int t = 0; t < a.nt; ++t)
        for (int s = 0; s < NSB; ++s) {
            const bool has_copy = NSB + s < NS;
            int excess = 0;
            for (int band = 0; band < a.nbands; ++band) {
                const int n = a.cap_n[((size_t)t * a.nbands + band) * NSB + s];
                if (n > a.cap_rec) excess += n - a.cap_rec;
            }
            for (int l = 0; l < a.nlev; ++l) {
                const T lv = wb_level_lo_host<T>(a.levels[l]);
                const size_t base = ((size_t)(t * a.nlev + l) * NS + s) * a.cap_strip;
                const size_t cbase = ((size_t)(t * a.nlev + l) * NS + NSB + s) * a.cap_strip;
                int mine = 0, cmine = 0;
                for (int band = 0; band < a.nbands; ++band) {
                    const size_t gl = ((size_t)t * a.nbands + band) * NSB + s;
                    const int n = a.cap_n[gl] < a.cap_rec ? a.cap_n[gl] : a.cap_rec;
                    for (int li = 0; li < n; ++li) {
                        const uint32_t key = a.cap_k[gl * cs + li];
                        const int r = (int)(key >> 16), c0 = (int)(key & 0xFFFFu);
                        const T* d = (const T*)a.cap_v + gl * cs * 10 + li;
                        for (int k = 0; k < 4; ++k) {
                            const T ul = d[k * cs], ur = d[(k + 1) * cs], ll = d[(5 + k) * cs], lr = d[(6 + k) * cs];
                            const bool ok = (c0 + k < cmax) && ul == ul && ur == ur && ll == ll && lr == lr;
                            const int cs2 = ok ? ((ul > lv ? 1 : 0) | (ur > lv ? 2 : 0) | (ll > lv ? 4 : 0) | (lr > lv ? 8 : 0)) : 0;
                            const int ns = wb_case_nseg(cs2);
                            if (ns) {
                                wb_store_square<T>(a.stg_k + base, (T*)a.stg_v + base * 4, a.cap_strip, W, r, c0 + k, cs2, ns, ul, ur, ll, lr, mine);
                                if (has_copy && c0 + k < a.padd - 1)
                                    wb_store_square<T>(a.stg_k + cbase, (T*)a.stg_v + cbase * 4, a.cap_strip, W, r, c0 + k + nlon, cs2, ns, ul, ur, ll, lr, cmine);
                            }
                            mine += ns;
                            if (c0 + k < a.padd - 1) cmine += ns;
                        }
                    }
                }
                a.stg_n[(t * a.nlev + l) * NS + s] = excess ? a.cap_strip + 1 + excess : mine;
                if (has_copy) a.stg_n[(t * a.nlev + l) * NS + NSB + s] = cmine;
            }
        }
}
#else
// ------------------------------------------------------------------------------------------------
// gfx950 kernel: blockDim.x = 256 = 4 independent persistent waves.
//   VEC  : rows are read with one 16 / 32 byte load per lane (needs nlon % 4 == 0 and an aligned base)
//   NLEV : number of contour levels (compile time, so the level loops vanish)
template <typename T> struct WbVec4;
template <> struct WbVec4<float> { typedef float4 type; };
template <> struct WbVec4<double> { typedef double4 type; };

// largest T <= level
template <typename T> __device__ inline T wb_level_lo(double level);
template <> __device__ inline float wb_level_lo<float>(double level) {
    float f = (float)level;                     // round to nearest
    if ((double)f > level) f = nextafterf(f, -INFINITY);
    return f;
}
template <> __device__ inline double wb_level_lo<double>(double level) { return level; }

// plain v_min / v_max (IEEE minNum/maxNum: a NaN operand is ignored); written as asm so that hipcc does not
// wrap every call in canonicalising v_max x,x (cdna_hip_programming.md, Appendix B GEMM note)
__device__ inline float wb_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline float wb_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline double wb_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline double wb_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// value of the next lane (lane 63 gets 0): DPP wave shift, one VALU op per dword, no LDS
__device__ inline float wb_shl1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xF, 0xF, false));
}
__device__ inline double wb_shl1(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), 0x130, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ inline float wb_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ inline float wb_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ inline double wb_min3(double a, double b, double c) { return wb_min(wb_min(a, b), c); }
__device__ inline double wb_max3(double a, double b, double c) { return wb_max(wb_max(a, b), c); }

// Emit path of one strip row (rare).  Returns segments written to the base list | segments written to the copy list << 16.
template <typename T>
__device__ __forceinline__ int wb_scan_emit(uint32_t* stg_k, T* stg_v, int off, uint32_t* cpy_k, T* cpy_v, int coff, int nlon,
                                            int cap_strip, int W, int r, int c0, unsigned inmask, unsigned eligmask, T lv,
                                            unsigned long long ltmask, T p0, T p1, T p2, T p3, T p4, T n0, T n1, T n2, T n3,
                                            T n4) {
    const T p[5] = {p0, p1, p2, p3, p4}, n[5] = {n0, n1, n2, n3, n4};
    int cs[4], ns[4], below = 0, total = 0, cbelow = 0, ctotal = 0;
    const bool any_elig = __any(eligmask != 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const T ul = p[k], ur = p[k + 1], ll = n[k], lr = n[k + 1];
        const bool ok = ((inmask >> k) & 1u) && ul == ul && ur == ur && ll == ll && lr == lr;
        // v > lvl_lo in T  <=>  (double)v > level : skimage's strict compare on the float64 copy
        cs[k] = ok ? ((ul > lv ? 1 : 0) | (ur > lv ? 2 : 0) | (ll > lv ? 4 : 0) | (lr > lv ? 8 : 0)) : 0;
        ns[k] = wb_case_nseg(cs[k]);
        const unsigned long long m1 = __ballot(ns[k] >= 1), m2 = __ballot(ns[k] == 2);
        below += __popcll(m1 & ltmask) + __popcll(m2 & ltmask);
        total += __popcll(m1) + __popcll(m2);
        if (any_elig) {                                    // wave-uniform: only the first strips have periodic copies
            const bool el = (eligmask >> k) & 1u;
            const unsigned long long c1 = __ballot(ns[k] >= 1 && el), c2 = __ballot(ns[k] == 2 && el);
            cbelow += __popcll(c1 & ltmask) + __popcll(c2 & ltmask);
            ctotal += __popcll(c1) + __popcll(c2);
        }
    }
    if (total == 0) return 0;
    int mine = off + below, cmine = coff + cbelow;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (ns[k]) {
            wb_store_square<T>(stg_k, stg_v, cap_strip, W, r, c0 + k, cs[k], ns[k], p[k], p[k + 1], n[k], n[k + 1], mine);
            if ((eligmask >> k) & 1u)
                wb_store_square<T>(cpy_k, cpy_v, cap_strip, W, r, c0 + k + nlon, cs[k], ns[k], p[k], p[k + 1], n[k], n[k + 1], cmine);
        }
        mine += ns[k];
        if ((eligmask >> k) & 1u) cmine += ns[k];
    }
    return total | (ctotal << 16);
}

// One row of a lane's patch: its 4 columns, the right halo, min / max over the 5 values (NaNs ignored).
template <typename T> struct WbRow { T v[5]; T h63; T mn, mx; };

template <typename T, bool VEC, int NLEV>
__global__ __launch_bounds__(256) void wb_k_scan(WbScanArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> strip / row addresses are scalar
    const int NS = a.nstrips, NSB = a.nwork;
    const int W = a.W, nlon = a.nlon, nlat = a.nlat;
    const unsigned long long ltmask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int nwork = a.nt * NSB;
    const int cmax = nlon < W - 1 ? nlon : W - 1;             // base squares: c < min(nlon, W-1)

    T lvl[NLEV];
#pragma unroll
    for (int l = 0; l < NLEV; ++l) lvl[l] = wb_level_lo<T>(a.levels[l]);

    for (int g = blockIdx.x * 4 + wave; g < nwork; g += gridDim.x * 4) {     // work item = (time step, strip)
        const int t = g / NSB, s = g - t * NSB;
        const T* f = (const T*)a.field + (size_t)t * nlat * nlon;
        const int c0 = s * WB_STRIP + 4 * lane;
        unsigned inmask = 0, eligmask = 0;                    // bit k: square c0+k exists / also exists at c0+k+nlon
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            inmask |= (c0 + k < cmax) ? (1u << k) : 0u;
            eligmask |= (c0 + k < a.padd - 1) ? (1u << k) : 0u;
        }
        // column offsets inside a row; out-of-range lanes read a clamped (valid, ignored) address
        int cm = c0 >= nlon ? c0 - nlon : c0;
        if (cm > nlon - 4) cm = nlon - 4;
        int chs = (s + 1) * WB_STRIP;                         // first column of the next strip (wave-uniform)
        chs = chs >= nlon ? chs - nlon : chs;
        if (chs > nlon - 1) chs = nlon - 1;
        bool nanflag = false;
        int cursor[NLEV], ccursor[NLEV];                      // segments written so far (base list / copy list), wave-uniform
#pragma unroll
        for (int l = 0; l < NLEV; ++l) { cursor[l] = 0; ccursor[l] = 0; }

        // A row load = the lane's 4 columns (one vector load) + the first column of the next strip, which is the
        // halo of lane 63 only and wave-uniform, hence a SCALAR load (its own counter: the vector loads' vmcnt
        // bookkeeping stays exact).  The halo of the other lanes is the next lane's first column, fetched with a
        // DPP wave shift only when the row is consumed, so the loads stay in flight for four iterations (depth-4
        // prefetch; six row buffers that rotate by name: the loop is unrolled 6x and no register is copied).
        auto load_row = [&](int r, WbRow<T>& R) {
            const T* row = f + (size_t)(r < nlat ? r : nlat - 1) * nlon;
            if (VEC) {
                const typename WbVec4<T>::type q = *reinterpret_cast<const typename WbVec4<T>::type*>(row + cm);
                R.v[0] = q.x; R.v[1] = q.y; R.v[2] = q.z; R.v[3] = q.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int c = c0 + k;
                    c = c >= nlon ? c - nlon : c;
                    R.v[k] = row[c > nlon - 1 ? nlon - 1 : c];
                }
            }
            R.h63 = row[chs];
        };
        auto finish_row = [&](WbRow<T>& R) {                  // halo, NaN flag, patch-row min / max
            const T sh = wb_shl1(R.v[0]);                     // unconditional: every lane must be active for the DPP read
            R.v[4] = (lane == 63) ? R.h63 : sh;
            nanflag |= __builtin_isunordered(R.v[0], R.v[1]) | __builtin_isunordered(R.v[2], R.v[3]);
            R.mn = wb_min3(R.v[0], R.v[1], wb_min3(R.v[2], R.v[3], R.v[4]));
            R.mx = wb_max3(R.v[0], R.v[1], wb_max3(R.v[2], R.v[3], R.v[4]));
        };
        // classify the lane's 2 x 5 patch of rows (P, N): the patch is connected, so it holds a crossing of lvl
        // iff min(patch) <= lvl < max(patch); squares with a NaN corner are sorted out by the emit path
        auto step = [&](const WbRow<T>& P, WbRow<T>& N, int r) {
            finish_row(N);
            const T lo = wb_min(P.mn, N.mn), hi = wb_max(P.mx, N.mx);
#pragma unroll
            for (int l = 0; l < NLEV; ++l) {
                if (__any((hi > lvl[l]) && !(lo > lvl[l]))) {   // rare: this strip row carries a segment of level l
                    const size_t base = ((size_t)(t * NLEV + l) * NS + s) * a.cap_strip;
                    const size_t cbase = ((size_t)(t * NLEV + l) * NS + NSB + s) * a.cap_strip;   // touched only if eligible
                    const int w = wb_scan_emit<T>(a.stg_k + base, (T*)a.stg_v + base * 4, cursor[l], a.stg_k + cbase,
                                                  (T*)a.stg_v + cbase * 4, ccursor[l], nlon, a.cap_strip, W, r, c0, inmask, eligmask,
                                                  lvl[l], ltmask, P.v[0], P.v[1], P.v[2], P.v[3], P.v[4], N.v[0], N.v[1], N.v[2],
                                                  N.v[3], N.v[4]);
                    cursor[l] += w & 0xFFFF;
                    ccursor[l] += w >> 16;
                }
            }
        };

        WbRow<T> R0, R1, R2, R3, R4, R5;
        load_row(0, R0);
        load_row(1, R1);
        load_row(2, R2);
        load_row(3, R3);
        load_row(4, R4);
        finish_row(R0);
        int r = 0;
        for (; r + 6 <= nlat - 1; r += 6) {                   // rows (r, r+1) .. (r+5, r+6); 4 row loads always in flight
            load_row(r + 5, R5); step(R0, R1, r);
            load_row(r + 6, R0); step(R1, R2, r + 1);
            load_row(r + 7, R1); step(R2, R3, r + 2);
            load_row(r + 8, R2); step(R3, R4, r + 3);
            load_row(r + 9, R3); step(R4, R5, r + 4);
            load_row(r + 10, R4); step(R5, R0, r + 5);
        }
        if (r < nlat - 1) { load_row(r + 5, R5); step(R0, R1, r); ++r; }
        if (r < nlat - 1) { step(R1, R2, r); ++r; }
        if (r < nlat - 1) { step(R2, R3, r); ++r; }
        if (r < nlat - 1) { step(R3, R4, r); ++r; }
        if (r < nlat - 1) { step(R4, R5, r); ++r; }

        if (lane == 0) {
#pragma unroll
            for (int l = 0; l < NLEV; ++l) {
                a.stg_n[(t * NLEV + l) * NS + s] = cursor[l];
                if (NSB + s < NS) a.stg_n[(t * NLEV + l) * NS + NSB + s] = ccursor[l];
            }
        }
        if (__any(nanflag && (c0 < nlon)) && lane == 0) {
#pragma unroll
            for (int l = 0; l < NLEV; ++l) atomicOr(&a.status[t * NLEV + l], WB_ST_HAS_NAN);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The scan as two kernels: wb_k_scan_stream (wb_stream.h) streams the field and captures the 2 x 5 patches that hold a
// crossing; wb_k_scan_build below, one workgroup per (time step, strip), turns the patches into the segment lists of every
// level (square cases, block scan for the raster-ordered positions, key + four corner values per segment, periodic
// copies) -- the staging format the trace kernels read.  It touches the captured patches only (a few % of the field).
// Patch layout: value k (0..4 upper row, 5..9 lower row) of patch i of a list at cap_v[list][k * cap_rec + i].
#define WB_BUILD_CHUNK 1024
#ifndef WB_BUILD_KO_CHUNK
#define WB_BUILD_KO_CHUNK 128
#endif
// in-place exclusive scan of 64-bit values in LDS (two 32-bit counters per word, no carry between them as long as each
// stays below 2^32); every thread of the block calls it; returns the total
__device__ inline unsigned long long wb_block_scan_excl_u64(unsigned long long* a, int n) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
    if (tid == 0) carry_s = 0ull;
    __syncthreads();
    for (int base = 0; base < n; base += 8 * blockDim.x) {
        const int i0 = base + 8 * tid;
        unsigned long long v[8], sum = 0ull;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = (i0 + k < n) ? a[i0 + k] : 0ull; sum += v[k]; }
        unsigned long long inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned long long woff = 0ull;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const unsigned long long carry = carry_s;
        unsigned long long run = carry + woff + inc - sum;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (i0 + k < n) a[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 0) {
            unsigned long long tot = 0ull;
            for (int w = 0; w < nw; ++w) tot += wsum[w];
            carry_s = carry + tot;
        }
        __syncthreads();
    }
    const unsigned long long total = carry_s;                // (as in wb_block_scan_excl: the total is read before the next scan may clear it)
    __syncthreads();
    return total;
}

// KO (a.keys_only, round 6 a template parameter): the pool is then cb[KOCH] + lcase[KOCH] + the patches' ten values -- 7 KB for KOCH =
// WB_BUILD_KO_CHUNK patches (float; float64: half as many) -- and the launch has WB_BUILD_KO_THREADS threads: the kernel is a chain
// of latency-bound phases per workgroup (patch keys, ten gathers from the field, two block scans and the segment stores per
// level), and what hides them is workgroups per CU -- 16 of two waves instead of 8 of four (16 KB of LDS each)
#ifndef WB_BUILD_KO_THREADS
#define WB_BUILD_KO_THREADS 128
#endif
template <typename T, int NLEV, bool KO>
__global__ __launch_bounds__(256) void wb_k_scan_build(WbScanArgs a) {
    // key + values: one 16 KB pool, cb[1024] + lcase[1024]; keys only: see above
    constexpr int KOCH = sizeof(T) == 4 ? WB_BUILD_KO_CHUNK : WB_BUILD_KO_CHUNK / 2;    // keys only: patches per chunk
    __shared__ unsigned long long wb_build_pool[KO ? (2 * KOCH + (10 * KOCH * sizeof(T) + 7) / 8) : 2 * WB_BUILD_CHUNK];
    const int CH = KO ? KOCH : WB_BUILD_CHUNK;
    unsigned long long* const cb = wb_build_pool;              // per patch: segments for the base list | periodic-copy list << 32 -> scan
    unsigned long long* const lcase = wb_build_pool + CH;
    T* const lvf = (T*)(wb_build_pool + 2 * CH);              // (keys only: 2 * KOCH words in front, 10 * KOCH values = 10 KB behind them: inside the pool)
    __shared__ int boff[17];                                   // first patch of a band in the concatenation of the bands' lists
    const int g = blockIdx.x;
    const int NS = a.nstrips, NSB = a.nwork;
    const int t = g / NSB, s = g - t * NSB;
    const int W = a.W, nlon = a.nlon;
    const int cmax = nlon < W - 1 ? nlon : W - 1;
    const int nbands = a.nbands > 0 ? a.nbands : 1;
    const size_t cs = (size_t)a.cap_vs;                       // (strides; the capacity is cap_rec)
    const bool has_copy = NSB + s < NS;
    if (threadIdx.x == 0) {
        int run = 0;
        for (int band = 0; band < nbands; ++band) {
            boff[band] = run;
            const int n = a.cap_n[((size_t)t * nbands + band) * NSB + s];
            run += n < a.cap_rec ? n : a.cap_rec;
        }
        boff[nbands] = run;
    }
    __syncthreads();
    const int total = boff[nbands];
    int excess = 0;
    for (int band = 0; band < nbands; ++band) {
        const int n = a.cap_n[((size_t)t * nbands + band) * NSB + s];
        if (n > a.cap_rec) excess += n - a.cap_rec;
    }
    // patch i of the concatenation: its list and its index there
    auto locate = [&](int i, size_t& gl, int& li) {
        int band = 0;
        while (band + 1 < nbands && i >= boff[band + 1]) ++band;
        gl = ((size_t)t * nbands + band) * NSB + s;
        li = i - boff[band];
    };
    // A chunk of patches at a time: (1) every patch's ten values are read ONCE and the square cases of all levels go to LDS
    // (four bits per square, sixteen per level); (2) per level: segment counts from the cached cases, block scan, and the
    // values are read again only by the patches that hold a segment of THAT level (the two PV levels live in different
    // hemispheres: half the patches).  Reading the values per level and pass cost four reads of every patch.
    T lvl[NLEV];
    unsigned long long cur[NLEV];                            // segments written so far: base list | copy list << 32
#pragma unroll
    for (int l = 0; l < NLEV; ++l) { lvl[l] = wb_level_lo<T>(a.levels[l]); cur[l] = 0ull; }
    // keys only: a patch's ten values are read from the FIELD (two short runs of a row each: not coalesced) -- once: they stay in
    // LDS for the levels' emit passes below (chunks of 512 patches then: 20 KB)
    for (int q0 = 0; q0 < total; q0 += CH) {
        const int nq = total - q0 < CH ? total - q0 : CH;
        for (int i = threadIdx.x; i < nq; i += blockDim.x) {
            size_t gl; int li;
            locate(q0 + i, gl, li);
            const uint32_t key = a.cap_k[gl * cs + li];
            const int c0 = (int)(key & 0xFFFFu);
            T v[10];
            if (KO) {
                // the patch's 2 x 5 values from the field: columns c0 .. c0 + 4 (mod nlon) of rows r and r + 1, as the streaming
                // kernel saw them (a lane's four columns + its right neighbour's first / column 0 behind the row's end)
                const T* f0 = (const T*)a.field + ((size_t)t * a.nlat + (key >> 16)) * nlon;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int cc = c0 + k >= nlon ? c0 + k - nlon : c0 + k;
                    v[k] = f0[cc];
                    v[5 + k] = f0[nlon + cc];
                }
#pragma unroll
                for (int k = 0; k < 10; ++k) lvf[k * KOCH + i] = v[k];
            } else {
                const T* d = (const T*)a.cap_v + gl * cs * 10 + li;
#pragma unroll
                for (int k = 0; k < 10; ++k) v[k] = d[k * cs];
            }
            unsigned long long word = 0ull;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const T ul = v[k], ur = v[k + 1], ll = v[5 + k], lr = v[6 + k];
                const bool ok = (c0 + k < cmax) && ul == ul && ur == ur && ll == ll && lr == lr;
#pragma unroll
                for (int l = 0; l < NLEV; ++l) {
                    const T lv = lvl[l];
                    const int cs2 = ok ? ((ul > lv ? 1 : 0) | (ur > lv ? 2 : 0) | (ll > lv ? 4 : 0) | (lr > lv ? 8 : 0)) : 0;
                    word |= (unsigned long long)cs2 << (16 * l + 4 * k);
                }
            }
            lcase[i] = word;
        }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < NLEV; ++l) {
            const size_t base = ((size_t)(t * NLEV + l) * NS + s) * a.cap_strip;
            const size_t cbase = ((size_t)(t * NLEV + l) * NS + NSB + s) * a.cap_strip;
            for (int i = threadIdx.x; i < nq; i += blockDim.x) {
                const unsigned cw = (unsigned)(lcase[i] >> (16 * l)) & 0xFFFFu;
                unsigned nb = 0, nc = 0;
                if (cw) {
                    size_t gl; int li;
                    locate(q0 + i, gl, li);
                    const int c0 = (int)(a.cap_k[gl * cs + li] & 0xFFFFu);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int ns = wb_case_nseg((int)(cw >> (4 * k)) & 15);
                        nb += ns;
                        if (c0 + k < a.padd - 1) nc += ns;
                    }
                }
                cb[i] = (unsigned long long)nb | ((unsigned long long)nc << 32);
            }
            __syncthreads();
            const unsigned long long tot = wb_block_scan_excl_u64(cb, nq);
            for (int i = threadIdx.x; i < nq; i += blockDim.x) {
                const unsigned cw = (unsigned)(lcase[i] >> (16 * l)) & 0xFFFFu;
                if (!cw) continue;                                   // no segment of this level in the patch: nothing to read
                size_t gl; int li;
                locate(q0 + i, gl, li);
                const uint32_t key = a.cap_k[gl * cs + li];
                const int r = (int)(key >> 16), c0 = (int)(key & 0xFFFFu);
                const T* d = KO ? nullptr : (const T*)a.cap_v + gl * cs * 10 + li;
                const unsigned long long at = cur[l] + cb[i];
                int mine = (int)(at & 0xFFFFFFFFull), cmine = (int)(at >> 32);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cs2 = (int)(cw >> (4 * k)) & 15;
                    const int ns = wb_case_nseg(cs2);
                    if (ns) {
                        T ul, ur, ll, lr;
                        if (KO) {
                            ul = lvf[k * KOCH + i]; ur = lvf[(k + 1) * KOCH + i];
                            ll = lvf[(5 + k) * KOCH + i]; lr = lvf[(6 + k) * KOCH + i];
                        } else { ul = d[k * cs]; ur = d[(k + 1) * cs]; ll = d[(5 + k) * cs]; lr = d[(6 + k) * cs]; }
                        wb_store_square<T>(a.stg_k + base, (T*)a.stg_v + base * 4, a.cap_strip, W, r, c0 + k, cs2, ns, ul, ur, ll, lr, mine);
                        if (has_copy && c0 + k < a.padd - 1)
                            wb_store_square<T>(a.stg_k + cbase, (T*)a.stg_v + cbase * 4, a.cap_strip, W, r, c0 + k + nlon, cs2, ns, ul, ur, ll, lr, cmine);
                    }
                    mine += ns;
                    if (c0 + k < a.padd - 1) cmine += ns;
                }
            }
            __syncthreads();
            cur[l] += tot;
        }
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int l = 0; l < NLEV; ++l) {
            // an overflowed patch list makes the segment count overflow too: the host grows the capacities and re-runs
            a.stg_n[(t * NLEV + l) * NS + s] = excess ? a.cap_strip + 1 + excess : (int)(cur[l] & 0xFFFFFFFFull);
            if (has_copy) a.stg_n[(t * NLEV + l) * NS + NSB + s] = (int)(cur[l] >> 32);
        }
    }
}
#endif
