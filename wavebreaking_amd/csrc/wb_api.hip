// wb_api.hip -- the C ABI (include/wavebreaking_hip.h): buffer management, kernel sequencing, result
// marshalling.  Compiled by hipcc for gfx950 into libwb_hip.so (the product) and, with -DWB_HOSTSIM
// by g++, into tests/_hostsim/libwb_hostsim.so (test infrastructure: same sources, kernels executed
// block by block on the host under ASan/UBSan; never loaded by the package).
#include "../../include/wavebreaking_hip.h"
#include "../../include/wavebreaking_hip_dev.h"

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>

#include "wb_comm.h"
#include "wb_core.h"
#include "wb_geometry.h"
#include "wb_indices.h"
#include "wb_props.h"
#include "wb_geom.h"
#include "wb_scan.h"
#include "wb_stream.h"
#include "wb_trace.h"
#include "wb_trace_fused.h"
#include "wb_track.h"

#include "wb_platform.h"

// ------------------------------------------------------------------------------------------------ batch
struct wb_batch {
    wb_ctx* ctx = nullptr;
    wb_params p;
    std::vector<double> lat, lon, levels;
    int W = 0, padd = 0, njobs = 0, cap_seg = 0, hash_size = 0;
    std::vector<void*> owned;       // device allocations to free
    const void* d_field = nullptr;
    const void* d_intensity = nullptr;
    // phase 1
    uint32_t *stg_k = nullptr, *status = nullptr;
    void* stg_v = nullptr;
    int* stg_n = nullptr;
    int nstrips = 0, nwork_strips = 0, cap_strip = 0;
    uint32_t *cpts = nullptr, *cown = nullptr;
    int *c_off = nullptr, *c_len = nullptr, *c_closed = nullptr, *c_nux = nullptr, *c_sumy = nullptr, *ncont = nullptr;
    std::vector<void*> phase1_scratch;
    int max_nux = 0;
    std::vector<int> h_ncont;
    std::vector<uint32_t> h_status;
    // tables on the device
    double *d_lat = nullptr, *d_latrad = nullptr, *d_lonrad = nullptr, *d_sphi2 = nullptr, *d_cphi2 = nullptr,
           *d_cphi = nullptr, *d_slam2 = nullptr, *d_clam2 = nullptr, *d_wlat = nullptr;
    // phase 2
    int nsel = 0, ncut = 0;
    int *sel_job = nullptr, *sel_cont = nullptr, *cut_job = nullptr, *cut_cont = nullptr;
    long long band[2] = {0, 0};
    long long pair_stats[3] = {0, 0, 0};    // base-point pairs covered by the all-pairs test / candidates found / pairs evaluated
    WbCtl* d_ctl = nullptr;                 // device control block of the batch
    WbCtl h_ctl;                            // its phase-1 values (host copy)
    int fused_declined = 0, fused_per_cu = -1;      // wb_k_trace_fused: jobs it declined, jobs a CU holds at the launch's LDS (wb_get_counter)
    long long fused_lds = 0;
    std::vector<std::pair<void*, size_t>> hres;   // pinned host blocks of the result copies
    // device-resident result of the last wb_indices (kept when the context option "keep_device" is set): packed rows and
    // polygon vertices for the device-to-device gather of a time-sharded run
    double* dev_rows = nullptr;
    int *dev_px = nullptr, *dev_py = nullptr, *d_job_level = nullptr;
    long long dev_n[3] = {0, 0, 0}, dev_nv = 0, step_offset = 0;
    long long com_rechecks[3] = {0, 0, 0};     // events / boxes whose centre-of-mass sums were re-evaluated in the reference's order
    long long box_bound = 0, box_bound_mismatch = 0;   // queued boxes settled from the exact totals of their terms (wb_k_box_com) / test build WB_BOXCOM_BOUND = 2: boxes the two paths disagree on
    long long group_ties = 0;                  // check_groups: groups whose largest length was shared by several members
    // batches built from a user-supplied contour frame: one job per contour row
    std::vector<int> job_step, job_level;
    int* d_job_step = nullptr;
    // host result tables
    bool have_contours = false;
    std::vector<int32_t> ct_step, ct_level, ct_closed, ct_nux, ct_npts, ct_x, ct_y;
    std::vector<int64_t> ct_sumy, ct_off, ct_base;   // ct_base[job] = first contour row of the job
    struct Ev {
        std::vector<int32_t> step, level, ind1, ind2, box, orient, ncells, flags, cx, cy;
        std::vector<int64_t> contour, celloff, polyoff;
        size_t pbase = 0;              // first polygon vertex of this table in wb_batch::poly_x / poly_y
        std::vector<double> sums, props;
    } ev[3];
    WbRingSink geo[3];                              // event geometry in lon / lat (wb_get_event_geometry), built on demand
    // ... or cut and mapped by the device during wb_indices (context option "geometry"): pinned host arrays over ALL events
    std::vector<std::pair<void*, size_t>> gres;
    long long* g_ring_offset = nullptr; long long* g_vert_offset = nullptr; double* g_x = nullptr; double* g_y = nullptr;
    long long g_rings = -1, g_verts = 0;            // -1: not there (option off, or an event the kernel flagged: host function)
    // the last wb_indices' event descriptors and polygon offsets stay on the device: index polygons are packed when wb_get_events
    // first asks for them (context option index_polygons = 0), the geometry kernels may still be reading them (geometry_async)
    int* d_ev = nullptr; long long* d_poff = nullptr;
    int d_ev_cap_seg = 0;
    // geometry kernels wb_indices did not wait for: their event, their totals in pinned memory, their scratch
    bool g_pending = false;
    long long* g_tot = nullptr;                     // [4] rings, vertices, overflow, flagged (inside the ring_offset block)
    long long g_cap_r = 0, g_cap_v = 0;
    int *g_cnt_r = nullptr, *g_cnt_v = nullptr, *g_flag = nullptr;
    // context option geometry_dma: the kernels' device arrays (x, y: what the copy behind the kernels left out is fetched by
    // wb_geom_settle) and how many vertices that copy covered
    double *g_dx = nullptr, *g_dy = nullptr;
    long long* g_dro = nullptr; long long* g_dvo = nullptr;
    long long g_copied = 0;
#if !defined(WB_HOSTSIM)
    hipEvent_t g_event = nullptr;
#endif
    int32_t *poly_x = nullptr, *poly_y = nullptr;   // polygon vertices of all events (streamers, overturnings, cutoffs)
    size_t poly_n = 0, poly_bytes = 0;
    WbTimer timer;
    size_t timer_phase1 = 0;                // stages recorded by wb_contours
    std::string timer_names;
};

// widest periodically extended grid: a lattice point is 16 + 16 bits (wb_core.h), a segment key holds nlat * W < 2^26 squares;
// beyond the widths the fast kernels hold (16 strips of 256 columns in the streaming scan, 8190 segments per job in the fused
// trace) the single-kernel scan and the three-kernel trace chain take over
#define WB_MAX_W 16384
static int wb_geom_settle(wb_batch* b);        // (defined in front of wb_indices)
static int wb_pack_polys_now(wb_batch* b);

#ifndef WB_LIST_THREADS
#define WB_LIST_THREADS 256          // single-workgroup list / offset kernels
#endif
static int wb_num_cus(wb_ctx* ctx) {
    if (ctx->ncu <= 0) {
        ctx->ncu = 256;
#if !defined(WB_HOSTSIM)
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) ctx->ncu = prop.multiProcessorCount;
        if (ctx->cu_count > 0 && ctx->cu_count < ctx->ncu) ctx->ncu = ctx->cu_count;      // (option "cu_mask": persistent grids fill the context's share)
#endif
    }
    return ctx->ncu;
}
template <class T>
static T* balloc(wb_batch* b, size_t n) {
    void* p = dmalloc(b->ctx, n * sizeof(T));
    if (p) b->owned.push_back(p);
    return (T*)p;
}
static void bfree_one(wb_batch* b, void* p) {
    for (size_t i = 0; i < b->owned.size(); ++i)
        if (b->owned[i] == p) { b->owned.erase(b->owned.begin() + i); break; }
    dfree(b->ctx, p);
}
template <class T>
static T* upload(wb_batch* b, const std::vector<T>& v) {
    T* d = balloc<T>(b, v.size());
    if (d && !v.empty()) h2d(b->ctx, d, v.data(), v.size() * sizeof(T));
    return d;
}

static int next_pow2(long long v) {
    long long p = 1;
    while (p < v) p <<= 1;
    return (int)p;
}

extern "C" {

const char* wb_backend(void) { return wb_backend_name(); }

int wb_create(int device_id, wb_ctx** out) {
    wb_ctx* c = new wb_ctx();
    c->device = device_id;
    if (!wb_platform_create(c)) {
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

void wb_destroy(wb_ctx* c) {
    if (!c) return;
    wb_comm_destroy(c);
    for (auto& kv : c->hpool) hrelease(kv.second);
    for (wb_ctx::Tables* t : c->tables) delete t;
    wb_platform_destroy(c);
    delete c;
}

const char* wb_last_error(wb_ctx* c) { return c ? c->err.c_str() : "no context"; }

void* wb_dev_alloc(wb_ctx* c, uint64_t bytes) { wb_use(c); return dmalloc(c, (size_t)bytes); }
// Milliseconds one streaming read of a device buffer takes (best of `reps`).  On this platform the same kernel reads one
// 6 GB allocation at 5.4 TB/s and the next at 4.6 -- it follows the BUFFER (tools/scan_placement_probe.py), presumably how
// contiguous the physical pages behind it are --, and nothing but a measurement tells which kind an allocation is.
#if !defined(WB_HOSTSIM)
__global__ __launch_bounds__(256) void wb_k_probe_read(const uint4* p, size_t n16, unsigned* sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0u;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {                // four independent 16-byte loads per lane in flight
        const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { const uint4 a = p[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x9E3779B9u) sink[0] = acc;                         // (keeps the loads)
}
// the same bytes read as the scan reads them: `grid` workgroups of 512 threads, each streaming chunks of chunk16 16-byte words
// from start to end (chunk b, b + grid, ...), eight 8 KB steps in flight
__global__ __launch_bounds__(512) void wb_k_probe_streams(const uint4* p, size_t n16, size_t chunk16, unsigned* sink) {
    unsigned acc = 0u;
    const size_t nchunks = (n16 + chunk16 - 1) / chunk16;
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t lo = c * chunk16, hi = lo + chunk16 < n16 ? lo + chunk16 : n16;
        size_t i = lo + threadIdx.x;
        for (; i + 7 * 512 < hi; i += 8 * 512) {
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p[i + (size_t)k * 512];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
        for (; i < hi; i += 512) { const uint4 a = p[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}
#endif
extern "C" double wb_probe_streams_ms(wb_ctx* c, const void* dev, uint64_t bytes, uint64_t chunk_bytes, int grid, int reps) {
    wb_use(c);
#if defined(WB_HOSTSIM)
    (void)dev; (void)bytes; (void)chunk_bytes; (void)grid; (void)reps;
    return 0.0;
#else
    unsigned* sink = (unsigned*)dmalloc(c, 256);
    if (!sink) return -1.0;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { dfree(c, sink); return -1.0; }
    double best = -1.0;
    if (grid < 1) grid = wb_num_cus(c);
    for (int r = 0; r < (reps > 0 ? reps : 1); ++r) {
        (void)hipEventRecord(e0, c->stream);
        hipLaunchKernelGGL(wb_k_probe_streams, dim3((unsigned)grid), dim3(512), 0, c->stream, (const uint4*)dev, (size_t)(bytes / 16), (size_t)(chunk_bytes / 16 > 0 ? chunk_bytes / 16 : 512), sink);
        (void)hipEventRecord(e1, c->stream);
        if (hipEventSynchronize(e1) != hipSuccess) { best = -1.0; break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (best < 0.0 || ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    dfree(c, sink);
    return best;
#endif
}
double wb_probe_read_ms(wb_ctx* c, const void* dev, uint64_t bytes, int reps) {
    wb_use(c);
#if defined(WB_HOSTSIM)
    (void)dev; (void)bytes; (void)reps;
    return 0.0;
#else
    unsigned* sink = (unsigned*)dmalloc(c, 256);
    if (!sink) return -1.0;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { dfree(c, sink); return -1.0; }
    double best = -1.0;
    const int grid = wb_num_cus(c) * 8;
    for (int r = 0; r < (reps > 0 ? reps : 1); ++r) {
        (void)hipEventRecord(e0, c->stream);
        hipLaunchKernelGGL(wb_k_probe_read, dim3((unsigned)grid), dim3(256), 0, c->stream, (const uint4*)dev, (size_t)(bytes / 16), sink);
        (void)hipEventRecord(e1, c->stream);
        if (hipEventSynchronize(e1) != hipSuccess) { best = -1.0; break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (best < 0.0 || ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    dfree(c, sink);
    return best;
#endif
}
void wb_dev_free(wb_ctx* c, void* p) { wb_use(c); dfree(c, p); }
int wb_memcpy_h2d(wb_ctx* c, void* d, const void* s, uint64_t b) { wb_use(c); return h2d(c, d, s, (size_t)b); }
int wb_memcpy_d2h(wb_ctx* c, void* d, const void* s, uint64_t b) { wb_use(c); return d2h(c, d, s, (size_t)b); }

void wb_free_batch(wb_batch* b) {
    if (!b) return;
    wb_use(b->ctx);
    double wb_t0_ = wb_now_ms();
#if !defined(WB_HOSTSIM)
    // (may be called by another thread than the one that drives the context: it takes the allocators' lock only.  Geometry
    // kernels that nobody has waited for yet still read the batch's contours and write its pinned arrays)
    if (b->g_event) { if (b->g_pending) (void)hipEventSynchronize(b->g_event); (void)hipEventDestroy(b->g_event); b->g_event = nullptr; }
#endif
    b->g_pending = false;
    b->timer.collect();                         // (the events of a stage nobody collected: the geometry kernels' when no timings were asked for)
    for (void* p : b->owned) dfree(b->ctx, p);
    hfree(b->ctx, b->poly_x, b->poly_bytes);
    hfree(b->ctx, b->poly_y, b->poly_bytes);
    for (auto& hb : b->gres) hfree(b->ctx, hb.first, hb.second);
    WB_TICK("free_batch device");
    delete b;
    WB_TICK("free_batch host");
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ phase 1
// bounds for phase 2 + the ONE read-back of phase 1: control block and contour counts per job
static int phase1_summary(wb_batch* b) {
    wb_ctx* ctx = b->ctx;
    WbCtl* hc = host_ctl(ctx);
    if (!hc) return wb_fail(ctx, "out of pinned host memory");
    WbSummaryArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.njobs = b->njobs; sa.cap_seg = b->cap_seg; sa.ncont = b->ncont; sa.c_len = b->c_len; sa.c_nux = b->c_nux;
    sa.c_closed = b->c_closed; sa.ctl = b->d_ctl;
    b->timer.begin(ctx, "summary");
    WB_LAUNCH(wb_k_phase1_summary, (b->njobs + 63) / 64 < 256 ? (b->njobs + 63) / 64 : 256, 64, sa);
    b->timer.end(ctx);
    size_t got = 0;
    int* h_nc = (int*)hmalloc(ctx, (size_t)(b->njobs + 1) * sizeof(int), &got);
    if (!h_nc) return wb_fail(ctx, "out of pinned host memory");
    d2h_async(ctx, &hc[0], b->d_ctl, sizeof(WbCtl));
    d2h_async(ctx, h_nc, b->ncont, (size_t)b->njobs * sizeof(int));
    const int rc = dsync(ctx);
    b->h_ctl = hc[0];
    b->max_nux = b->h_ctl.max_nux;
    b->h_ncont.assign(h_nc, h_nc + b->njobs);
    hfree(ctx, h_nc, got);
    return rc;
}

static int run_phase1(wb_batch* b) {
    wb_ctx* ctx = b->ctx;
    const wb_params& p = b->p;
    const int J = b->njobs, cap = b->cap_seg, nlat = p.nlat;
    const int NS = b->nstrips, caps = b->cap_strip;
    double wb_t0_ = wb_now_ms();
    b->hash_size = next_pow2(8LL * cap);
    const size_t per = (size_t)J * cap;
    const size_t pers = (size_t)J * NS * caps;
    b->stg_k = balloc<uint32_t>(b, pers);
    b->stg_v = balloc<unsigned char>(b, pers * 4 * (p.dtype == 0 ? 4 : 8));
    b->stg_n = balloc<int>(b, (size_t)J * NS);
    b->status = balloc<uint32_t>(b, J);
    b->cpts = balloc<uint32_t>(b, 2 * per);
    b->cown = balloc<uint32_t>(b, 2 * per);
    b->c_off = balloc<int>(b, per);
    b->c_len = balloc<int>(b, per);
    b->c_closed = balloc<int>(b, per);
    b->c_nux = balloc<int>(b, per);
    b->c_sumy = balloc<int>(b, per);
    b->ncont = balloc<int>(b, J);
    b->d_ctl = balloc<WbCtl>(b, 1);
    if (!b->d_ctl) return 1;
    WbTraceArgs ta;
    memset(&ta, 0, sizeof(ta));
    std::vector<void*>& sc = b->phase1_scratch;
    ta.nseg = balloc<int>(b, J); sc.push_back(ta.nseg);
    ta.p4 = balloc<int>(b, 2 * (size_t)J); sc.push_back(ta.p4);
    if (!ta.p4) return 1;
    // the global scratch of the three-kernel chain (merged segments, links, raw points, hash tables: 20 GB for a year of 0.25
    // degree) is only made when that chain runs -- a job the fused kernel declined, or the option trace_fused = 0
    auto chain_scratch = [&]() -> bool {
        if (ta.hval) return true;
        for (int i = 0; i < 11; ++i) { ta.w[i] = balloc<uint32_t>(b, per); sc.push_back(ta.w[i]); }
        for (int i = 0; i < 3; ++i) { ta.mseg[i] = balloc<uint32_t>(b, per); sc.push_back(ta.mseg[i]); }
        ta.rawpt = balloc<uint32_t>(b, 2 * per); sc.push_back(ta.rawpt);
        ta.rawcid = balloc<uint32_t>(b, 2 * per); sc.push_back(ta.rawcid);
        ta.keep = balloc<int>(b, (size_t)J * (2 * cap + 1)); sc.push_back(ta.keep);
        ta.hkey = balloc<unsigned long long>(b, (size_t)J * b->hash_size); sc.push_back(ta.hkey);
        ta.hval = balloc<uint32_t>(b, (size_t)J * b->hash_size); sc.push_back(ta.hval);
        return ta.hval != nullptr;
    };
    WB_TICK("phase1 allocations");
    if (wb_debug()) { dsync(ctx); WB_TICK("phase1 pre-sync"); }
    // the phase's zero fills: queued, ONE launch (dzero_flush, below: the streaming scan adds its spare row)
    dzero_q(ctx, b->status, (size_t)J * sizeof(uint32_t));
    dzero_q(ctx, b->d_ctl, sizeof(WbCtl));
    dzero_q(ctx, b->stg_n, (size_t)J * NS * sizeof(int));

    WbScanArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.nt = p.nt; sa.nlat = nlat; sa.nlon = p.nlon; sa.W = b->W; sa.nlev = p.nlevels; sa.nstrips = NS; sa.cap_strip = caps;
    sa.nwork = b->nwork_strips; sa.padd = b->padd;
    sa.field = b->d_field;
    for (int l = 0; l < p.nlevels; ++l) sa.levels[l] = b->levels[l];
    sa.stg_k = b->stg_k; sa.stg_v = b->stg_v; sa.stg_n = b->stg_n; sa.status = b->status;
    wb_scan_gate_enter(ctx);                       // (option scan_exclusive: behind the other contexts' scans, before the stage's first event)
    b->timer.begin(ctx, "scan");
#if defined(WB_HOSTSIM)
    if (ctx->scan_split && nlat < 65536 && b->W <= 65536) {
        // the two-kernel scan as the product runs it: patch capture per (time step, row band, strip), then the list build
        const size_t esz = p.dtype == 0 ? 4 : 8;
        int nbands = ctx->scan_grid_mult > 1 ? ctx->scan_grid_mult : 3;          // (any band count gives the same lists)
        if (nbands > nlat - 1) nbands = nlat - 1;
        sa.band_rows = (nlat - 1 + nbands - 1) / nbands;
        sa.nbands = (nlat - 1 + sa.band_rows - 1) / sa.band_rows;
        sa.cap_rec = caps;
        sa.cap_vs = caps + ctx->scan_pad;
        const long long lists = (long long)p.nt * b->nwork_strips * sa.nbands;
        sa.cap_k = balloc<uint32_t>(b, (size_t)lists * sa.cap_vs);
        sa.cap_v = balloc<unsigned char>(b, (size_t)lists * sa.cap_vs * 10 * esz);
        sa.cap_n = balloc<int>(b, (size_t)lists);
        if (!sa.cap_n || !sa.cap_v || !sa.cap_k) return 1;
        b->phase1_scratch.push_back(sa.cap_k); b->phase1_scratch.push_back(sa.cap_v); b->phase1_scratch.push_back(sa.cap_n);
        if (p.dtype == 0) { wb_scan_capture_host<float>(sa); wb_scan_build_host<float>(sa); }
        else { wb_scan_capture_host<double>(sa); wb_scan_build_host<double>(sa); }
    } else {
        if (p.dtype == 0) wb_scan_host<float>(sa); else wb_scan_host<double>(sa);
    }
#else
    {
        const size_t esz = p.dtype == 0 ? 4 : 8;
        // float4 / double4 row loads need 4-column granularity and a 16 / 32 byte aligned base
        const bool vec = (p.nlon % 4 == 0) && (((uintptr_t)b->d_field) % (4 * esz) == 0);
        long long nwg = ((long long)p.nt * b->nwork_strips + 3) / 4;
        const dim3 block(256);
        int per_cu = 4;
#define WB_SCAN_OCC(T, V, N) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wb_k_scan<T, V, N>, 256, 0)
#define WB_SCAN_OCC_N(T, V) \
        switch (p.nlevels) { case 1: WB_SCAN_OCC(T, V, 1); break; case 2: WB_SCAN_OCC(T, V, 2); break; \
                             case 3: WB_SCAN_OCC(T, V, 3); break; default: WB_SCAN_OCC(T, V, 4); break; }
        if (p.dtype == 0) { if (vec) { WB_SCAN_OCC_N(float, true) } else { WB_SCAN_OCC_N(float, false) } }
        else { if (vec) { WB_SCAN_OCC_N(double, true) } else { WB_SCAN_OCC_N(double, false) } }
#undef WB_SCAN_OCC_N
#undef WB_SCAN_OCC
        if (per_cu < 1) per_cu = 1;
        const int ncu = ctx->scan_cu_count > 0 ? ctx->scan_cu_count : wb_num_cus(ctx);   // (option "scan_cu_mask": the scan's own share)
        if (nwg > (long long)ncu * per_cu) nwg = (long long)ncu * per_cu;   // persistent: exactly the resident workgroups
        // the streaming kernel keeps a patch buffer per strip (and the halo slots) in dynamic LDS and runs one wave per strip:
        // fields whose strips do not fit the device's limits (float64 beyond 12 strips, any dtype beyond 16) take the
        // single-kernel scan below
        size_t lds_limit = 64 << 10;
        int threads_limit = 1024;
        { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) { lds_limit = prop.sharedMemPerBlock; threads_limit = prop.maxThreadsPerBlock; } }
        // (float fields and, with the option scan_f64_fast, float64 fields that load as vectors store their patches as keys only)
        const bool ko_likely = ctx->scan_keys_only && vec && ctx->scan_halo_lds && (p.dtype == 0 ? ctx->scan_variant == 2 : ctx->scan_f64_fast != 0);
        const size_t stream_lds = (size_t)b->nwork_strips * WB_STREAM_LBUF * (ko_likely ? 4 : 4 + 10 * esz) + (size_t)b->nwork_strips * 257 * (esz + 4) + 64;
        const bool stream_fits = stream_lds <= lds_limit && 64 * b->nwork_strips <= threads_limit;
        bool streamed = false;
        if (ctx->scan_split && nlat < 65536 && b->W <= 65536 && stream_fits) {
            // two kernels (wb_scan.h): the streaming kernel captures the patches that hold a crossing, the build
            // kernel turns them into the per-level segment lists
            const long long items = (long long)p.nt * b->nwork_strips;
            // one workgroup per (time step, row band) item (a wave per strip), persistent; ring depth / waves per SIMD by option
            const int variant = ctx->scan_variant;
            const int wps = !(p.dtype == 0 && vec) ? 4 : ((variant == 1 || variant == 4 || variant == 5) ? 5 : (variant == 2 ? 4 : (variant == 3 ? 8 : 6)));
            int wg_per_cu = (wps * 4) / b->nwork_strips;
            if (wg_per_cu < 1) wg_per_cu = 1;
            if (ctx->scan_wg_per_cu > 0 && ctx->scan_wg_per_cu < wg_per_cu) wg_per_cu = ctx->scan_wg_per_cu;   // (option "scan_wg_per_cu": fewer resident workgroups than fit -- room beside them for the other batch's kernels)
            const long long slots = (long long)ncu * wg_per_cu;
            // bands: about six work items per resident workgroup, so that the last round of the persistent grid is nearly full
            int nbands = (int)((6 * slots + p.nt - 1) / p.nt);
            if (nbands < 1) nbands = 1;
            if (nbands > 8) nbands = 8;
            // a coarse grid (1 degree: 181 rows of 1.4 KB): ONE band -- a work item of a few dozen short rows is all ring
            // fill and drain (0.18 ms for a year of 360 x 181 with eight bands, 0.12 ms with one)
            if (nlat - 1 < 256) nbands = 1;
            if (ctx->scan_bands > 0) nbands = ctx->scan_bands;             // (option "scan_bands": development override)
            {   // bands short enough for the LDS halo exchange of the streaming kernel (wb_k_scan_build handles up to 16)
                const int minb = (nlat - 1 + 254) / 255;
                if (vec && ctx->scan_halo_lds && nbands < minb && minb <= 16) nbands = minb;
            }
            if (nbands > nlat - 1) nbands = nlat - 1;
            sa.band_rows = (nlat - 1 + nbands - 1) / nbands;
            if (ctx->scan_band_rows > 0 && ctx->scan_band_rows <= 255 && ctx->scan_band_rows >= sa.band_rows) sa.band_rows = ctx->scan_band_rows;   // (option "scan_band_rows": longer bands, a shorter last one -- other start addresses for the same number of lists)
            sa.nbands = (nlat - 1 + sa.band_rows - 1) / sa.band_rows;
            sa.cap_rec = caps;
            sa.cap_vs = caps + ctx->scan_pad;
            sa.debug = ctx->scan_debug;
            const long long lists = items * sa.nbands;
            // the strips of a workgroup hand each other the halo column through LDS when a band's rows fit (vector loads only)
            // (float64 since round 6: option "scan_f64_fast", default 1 -- until then float64 fields, what calculate_smoothed_field
            //  returns, took the scalar-halo kernel with a 6-deep ring and 84-byte patches)
            const bool f64_fast = p.dtype != 0 && ctx->scan_f64_fast && ctx->scan_keys_only;
            sa.halo_rows = (vec && (p.dtype == 0 || f64_fast) && ctx->scan_halo_lds && sa.band_rows + 1 <= 256) ? sa.band_rows + 1 : 0;
            // keys only (option "scan_keys_only", the default where the 13 x 4 streaming kernel runs): a patch is stored as its
            // key, wb_k_scan_build reads the values from the field
            sa.keys_only = (ctx->scan_keys_only && vec && sa.halo_rows && (p.dtype == 0 ? variant == 2 : f64_fast)) ? 1 : 0;
            if (p.dtype != 0 && !sa.keys_only) sa.halo_rows = 0;
            sa.cap_k = balloc<uint32_t>(b, (size_t)lists * sa.cap_vs);
            sa.cap_v = balloc<unsigned char>(b, sa.keys_only ? 256 : (size_t)lists * sa.cap_vs * 10 * esz);
            sa.cap_n = balloc<int>(b, (size_t)lists);
            unsigned char* spare = balloc<unsigned char>(b, (size_t)p.nlon * esz + 64);
            if (!sa.cap_n || !sa.cap_v || !sa.cap_k || !spare) return 1;
            dzero_q(ctx, spare, (size_t)p.nlon * esz + 64);
            dzero_flush(ctx);
            sa.spare_row = spare;
            b->phase1_scratch.push_back(sa.cap_k); b->phase1_scratch.push_back(sa.cap_v); b->phase1_scratch.push_back(sa.cap_n);
            b->phase1_scratch.push_back(spare);
            const size_t slds = (size_t)b->nwork_strips * WB_STREAM_LBUF * (sa.keys_only ? 4 : 4 + 10 * esz)    // a patch buffer per wave
                                + (size_t)b->nwork_strips * sa.halo_rows * (esz + 4);
            if (slds <= lds_limit) {                       // (else -- a field whose bands do not admit the mode the estimate above assumed -- the single-kernel scan)
            streamed = true;
            sa.cursor = ctx->scan_dynamic ? &b->d_ctl->scan_cursor : nullptr;       // (the control block was zeroed above)
            long long mwg = slots * (ctx->scan_grid_mult > 0 ? ctx->scan_grid_mult : 1);
            if (mwg > (long long)p.nt * sa.nbands) mwg = (long long)p.nt * sa.nbands;
            const dim3 mgrid((unsigned)mwg), sblock(64 * b->nwork_strips), bgrid((unsigned)items);
            if (ctx->scan_priority) {                      // (the stage's events go where the kernel goes)
                b->timer.drop_last();
                prio_begin(ctx);
                b->timer.begin(ctx, "scan");
            }
#define WB_STREAM_F32(N) \
            if (!sa.halo_rows) hipLaunchKernelGGL((wb_k_scan_stream<float, true, N, 8, 6, false>), mgrid, sblock, slds, ctx->stream, sa); \
            else switch (variant) { \
                case 1: hipLaunchKernelGGL((wb_k_scan_stream<float, true, N, 10, 5, true>), mgrid, sblock, slds, ctx->stream, sa); break; \
                case 2: if (sa.keys_only) hipLaunchKernelGGL((wb_k_scan_stream<float, true, N, 13, 4, true, true>), mgrid, sblock, slds, ctx->stream, sa); \
                        else hipLaunchKernelGGL((wb_k_scan_stream<float, true, N, 13, 4, true>), mgrid, sblock, slds, ctx->stream, sa); \
                        break; \
                case 3: hipLaunchKernelGGL((wb_k_scan_stream<float, true, N, 6, 8, true>), mgrid, sblock, slds, ctx->stream, sa); break; \
                case 4: hipLaunchKernelGGL((wb_k_scan_stream<float, true, N, 9, 5, true>), mgrid, sblock, slds, ctx->stream, sa); break; \
                case 5: hipLaunchKernelGGL((wb_k_scan_stream<float, true, N, 8, 5, true>), mgrid, sblock, slds, ctx->stream, sa); break; \
                default: hipLaunchKernelGGL((wb_k_scan_stream<float, true, N, 8, 6, true>), mgrid, sblock, slds, ctx->stream, sa); break; \
            }
#ifndef WB_F64_RING
#define WB_F64_RING 6                  // ring depth of the float64 instance with the LDS halo exchange (8 spilled 22 registers: 2.90 ms against the scalar-halo kernel's 2.35)
#endif
#define WB_STREAM_ANY(T, V, N) \
            if (sizeof(T) == 8 && V && sa.halo_rows && sa.keys_only) hipLaunchKernelGGL((wb_k_scan_stream<double, true, N, WB_F64_RING, 4, true, true>), mgrid, sblock, slds, ctx->stream, sa); \
            else hipLaunchKernelGGL((wb_k_scan_stream<T, V, N, 6, 4, false>), mgrid, sblock, slds, ctx->stream, sa)
#define WB_SCAN_N(T, V) \
            switch (p.nlevels) { case 1: WB_STREAM_ANY(T, V, 1); break; case 2: WB_STREAM_ANY(T, V, 2); break; \
                                 case 3: WB_STREAM_ANY(T, V, 3); break; default: WB_STREAM_ANY(T, V, 4); break; }
            if (p.dtype == 0 && vec) {
                switch (p.nlevels) { case 1: WB_STREAM_F32(1); break; case 2: WB_STREAM_F32(2); break;
                                     case 3: WB_STREAM_F32(3); break; default: WB_STREAM_F32(4); break; }
            } else if (p.dtype == 0) { WB_SCAN_N(float, false) }
            else { if (vec) { WB_SCAN_N(double, true) } else { WB_SCAN_N(double, false) } }
#undef WB_STREAM_F32
#undef WB_STREAM_ANY
#undef WB_SCAN_N
            { const hipError_t le = hipGetLastError(); if (le != hipSuccess) return wb_fail(ctx, std::string("wb_k_scan_stream launch failed: ") + hipGetErrorString(le)); }
            b->timer.end(ctx);
            prio_end(ctx);
            wb_scan_gate_leave(ctx);
            b->timer.begin(ctx, "scan_build");
#define WB_BUILD_LAUNCH(T, N) \
            if (sa.keys_only) hipLaunchKernelGGL((wb_k_scan_build<T, N, true>), bgrid, dim3(WB_BUILD_KO_THREADS), 0, ctx->stream, sa); \
            else hipLaunchKernelGGL((wb_k_scan_build<T, N, false>), bgrid, block, 0, ctx->stream, sa)
#define WB_BUILD_N(T) \
            switch (p.nlevels) { case 1: WB_BUILD_LAUNCH(T, 1); break; case 2: WB_BUILD_LAUNCH(T, 2); break; \
                                 case 3: WB_BUILD_LAUNCH(T, 3); break; default: WB_BUILD_LAUNCH(T, 4); break; }
            if (p.dtype == 0) { WB_BUILD_N(float) } else { WB_BUILD_N(double) }
#undef WB_BUILD_N
#undef WB_BUILD_LAUNCH
            }
        }
        if (!streamed) {
        dzero_flush(ctx);
        const dim3 grid((unsigned)nwg);
#define WB_SCAN_LAUNCH(T, V, N) hipLaunchKernelGGL((wb_k_scan<T, V, N>), grid, block, 0, ctx->stream, sa)
#define WB_SCAN_N(T, V) \
        switch (p.nlevels) { case 1: WB_SCAN_LAUNCH(T, V, 1); break; case 2: WB_SCAN_LAUNCH(T, V, 2); break; \
                             case 3: WB_SCAN_LAUNCH(T, V, 3); break; default: WB_SCAN_LAUNCH(T, V, 4); break; }
        if (p.dtype == 0) { if (vec) { WB_SCAN_N(float, true) } else { WB_SCAN_N(float, false) } }
        else { if (vec) { WB_SCAN_N(double, true) } else { WB_SCAN_N(double, false) } }
#undef WB_SCAN_N
#undef WB_SCAN_LAUNCH
        }
    }
#endif
    b->timer.end(ctx);
    if (wb_debug()) { dsync(ctx); WB_TICK("phase1 scan+sync"); }

    ta.njobs = J; ta.nlat = nlat; ta.nlon = p.nlon; ta.W = b->W; ta.cap_seg = cap; ta.hash_size = b->hash_size;
    ta.nstrips = NS; ta.cap_strip = caps;
    ta.stg_k = b->stg_k; ta.stg_v = b->stg_v; ta.stg_n = b->stg_n; ta.dtype = p.dtype; ta.nlev = p.nlevels;
    for (int l = 0; l < p.nlevels; ++l) ta.levels[l] = b->levels[l];
    ta.cpts = b->cpts; ta.cown = b->cown; ta.c_off = b->c_off; ta.c_len = b->c_len; ta.c_closed = b->c_closed;
    ta.c_nux = b->c_nux; ta.c_sumy = b->c_sumy; ta.ncont = b->ncont; ta.status = b->status; ta.ctl = b->d_ctl;
    // One kernel per job with the job resident in LDS (wb_trace_fused.h); what it declines -- counted in ctl->n_unfused, read with
    // the phase's summary -- goes through the three kernels below, which then leave the finished jobs alone.
    bool fused_all = false;
    if (ctx->trace_fused && b->W <= 16384) {
        if (ctx->fused_seen_W != b->W) { ctx->fused_seen_W = b->W; ctx->fused_seen = 0; ctx->fused_seen_c = 0; ctx->fused_seen_s = 0; }
        // What the launch's LDS admits follows what the context's batches on this grid have asked for -- segments, chains and ranking
        // splitters of the largest job so far, with a margin; a job that asks for more is declined and taken by the three-kernel chain,
        // and the next launch has grown.  The LDS of a job decides how many jobs a CU holds, and the kernel's rate is that number
        // (160 KB: three jobs at 52 KB, four at <= 40 KB): of the two margins the smaller one is taken when it admits another job per CU.
        WbFusedArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.nwork = b->nwork_strips;
        size_t lds_block = 0, lds_cu = 0;
        wb_lds_limits(ctx, &lds_block, &lds_cu);
        WbFusedLayout L;
        auto plan = [&](int tight) {
            const int seen = ctx->fused_seen, seen_c = ctx->fused_seen_c, seen_s = ctx->fused_seen_s;
            int nmax = ctx->trace_nmax > 0 ? ctx->trace_nmax : (seen > 0 ? (tight ? seen + seen / 32 + 32 : seen + seen / 16 + 64) : 3 * b->W);
            nmax = (nmax + 63) & ~63;
            if (nmax < 512) nmax = 512;
            if (nmax > 8184) nmax = 8184;
            int cmax = ctx->trace_cmax > 0 ? ctx->trace_cmax : (seen_c > 0 ? ((seen_c + seen_c / 4 + 16 + 31) & ~31) : 256);
            if (cmax > 256 && ctx->trace_cmax <= 0) cmax = 256;
            fa.nmax = nmax; fa.cmax = cmax;
            const int s_grid = nmax / 16 + 128, s_seen = (seen_s + seen_s / (tight ? 16 : 8) + 16 + 7) & ~7;
            fa.smax = (ctx->trace_nmax > 0 || seen_s <= 0) ? s_grid : (s_seen > 1016 ? 1016 : s_seen);        // (the kernel's link words hold 10-bit splitter ordinals)
            L = wb_fused_layout(fa.nmax, fa.smax, fa.cmax, p.nlon, WB_FUSED_THREADS);
            while (L.total > lds_cu / 2 && fa.nmax > 1024) {        // never more than half a CU's LDS: smaller jobs only
                fa.nmax -= 256; fa.smax = fa.nmax / 16 + 128;
                L = wb_fused_layout(fa.nmax, fa.smax, fa.cmax, p.nlon, WB_FUSED_THREADS);
            }
            return (int)(lds_cu / (L.total + 512));                // jobs per CU (512: the kernel's static LDS, rounded up)
        };
        const int per_cu_wide = plan(0);
        if (ctx->trace_nmax <= 0 && plan(1) <= per_cu_wide) (void)plan(0);
        bool ok = L.total <= lds_cu / 2;
#if !defined(WB_HOSTSIM)
        if (ok && L.total > lds_block && hipFuncSetAttribute((const void*)wb_k_trace_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total) != hipSuccess) {
            (void)hipGetLastError();
            ok = false;
        }
#endif
        if (ok) {
            unsigned char* fdone = balloc<unsigned char>(b, (size_t)J + 16);
            if (!fdone) return 1;
            sc.push_back(fdone);
            fa.t = ta; fa.fdone = fdone;
            b->timer.begin(ctx, "trace_fused");
            WB_LAUNCH_LDS(wb_k_trace_fused, J, WB_FUSED_THREADS, L.total, fa);
            b->timer.end(ctx);
            if (phase1_summary(b)) return 1;
            b->fused_declined = b->h_ctl.n_unfused; b->fused_lds = (long long)L.total; b->fused_per_cu = (int)(lds_cu / (L.total + 512));
            if (b->h_ctl.max_nseg > ctx->fused_seen) ctx->fused_seen = b->h_ctl.max_nseg;
            if (b->h_ctl.max_chains > ctx->fused_seen_c) ctx->fused_seen_c = b->h_ctl.max_chains;
            if (b->h_ctl.max_splitters > ctx->fused_seen_s) ctx->fused_seen_s = b->h_ctl.max_splitters;
            if (wb_debug()) fprintf(stderr, "[wb] fused trace: nmax %d smax %d cmax %d, %zu B of LDS (%d jobs per CU), %d of %d jobs declined, largest job %d segments, %d chains, %d splitters\n", fa.nmax, fa.smax, fa.cmax, L.total, (int)(lds_cu / (L.total + 512)), b->h_ctl.n_unfused, J, b->h_ctl.max_nseg, b->h_ctl.max_chains, b->h_ctl.max_splitters);
            if (b->h_ctl.n_unfused == 0) fused_all = true;
            else {
                ta.fdone = fdone;
                // the summary again, from zero, behind the chain below (max_nux and max_len are running maxima and stay)
                dmemset(ctx, (char*)b->d_ctl + offsetof(WbCtl, ncont_total), 0, offsetof(WbCtl, scan_cursor) - offsetof(WbCtl, ncont_total));
            }
        }
    }
    if (!fused_all) {
    if (!chain_scratch()) return wb_fail(ctx, "out of device memory (scratch of the three-kernel trace chain)");
    b->timer.begin(ctx, "trace_link");
    {   // row tables in dynamic LDS when they fit (16-bit entries: a job of the LDS paths has at most 8192 segments)
        const size_t tab = (size_t)(NS + 1) * nlat * sizeof(uint16_t) + 16;
        ta.link_rows = (ctx->link_rows && tab <= (size_t)WB_LINK_ROWTAB_BYTES) ? 1 : 0;
        WB_LAUNCH_LDS(wb_k_trace_link, J, WB_LINK_THREADS, ta.link_rows ? tab : 16, ta);
    }
    b->timer.end(ctx);
    b->timer.begin(ctx, "trace_rank");
    WB_LAUNCH(wb_k_trace_rank, J, WB_RANK_THREADS, ta);
    b->timer.end(ctx);
    b->timer.begin(ctx, "trace_finish");
    ta.slow_pass = 0;
    WB_LAUNCH(wb_k_trace_finish, J, WB_FINISH_THREADS, ta);
    b->timer.end(ctx);
    if (phase1_summary(b)) return 1;
    if (b->h_ctl.n_slow > 0) {
        // the second trace pass, for the jobs the first left out (the link kernel counted them): more segments than the LDS
        // paths hold (wb_k_trace_rank_big), or an order-dependent join at cells equal to the level (wb_k_trace_walk: skimage's
        // sequential walk, one lane each).  An ordinary batch has none and pays no launch for them.
        b->timer.begin(ctx, "trace_slow");
        WB_LAUNCH(wb_k_trace_rank_big, J, WB_RANK_THREADS, ta);
        WB_LAUNCH(wb_k_trace_walk, J, 64, ta);
        ta.slow_pass = 1;
        WB_LAUNCH(wb_k_trace_finish, J, WB_FINISH_THREADS, ta);
        // the summary again, from zero (max_nux and max_len are running maxima and stay)
        dmemset(ctx, (char*)b->d_ctl + offsetof(WbCtl, ncont_total), 0, offsetof(WbCtl, scan_cursor) - offsetof(WbCtl, ncont_total));
        b->timer.end(ctx);
        if (phase1_summary(b)) return 1;
    }
    }
    b->timer_phase1 = b->timer.names.size();
    WB_TICK("phase1 launches + sync");
    return 0;
}

extern "C" int wb_contours(wb_ctx* ctx, const wb_params* pp, wb_batch** out) {
    if (!ctx || !pp || !out) return 1;
    double wb_tc_ = wb_now_ms();
    const wb_params& p = *pp;
    if (p.nt <= 0 || p.nlat < 2 || p.nlon < 2) return wb_fail(ctx, "bad dimensions");
    if (p.nlevels < 1 || p.nlevels > WB_MAX_LEVELS) return wb_fail(ctx, "between 1 and 4 contour levels per call");
    if (!(p.dlon > 0) || !(p.dlat > 0)) return wb_fail(ctx, "dlon and dlat must be positive");
    const int padd = (int)(p.periodic_add / p.dlon);     // contour_index.py:97
    if (padd < 0 || padd > p.nlon) return wb_fail(ctx, "periodic_add must be between 0 and 360 degrees");
    const int W = p.nlon + padd;
    if (W > WB_MAX_W || p.nlat > 32767) return wb_fail(ctx, "grid too large: nlon + periodic columns <= 16384, nlat <= 32767");
    std::lock_guard<std::recursive_mutex> wb_lk_(ctx->mu);
    if (!wb_select_device(ctx)) return wb_fail(ctx, "hipSetDevice failed");
    wb_batch* b = new wb_batch();
    b->ctx = ctx;
    b->p = p;
    b->lat.assign(p.lat, p.lat + p.nlat);
    b->lon.assign(p.lon, p.lon + p.nlon);
    b->levels.assign(p.levels, p.levels + p.nlevels);
    b->p.lat = b->lat.data();
    b->p.lon = b->lon.data();
    b->p.levels = b->levels.data();
    b->W = W;
    b->padd = padd;
    b->njobs = p.nt * p.nlevels;
    const size_t esz = p.dtype == 0 ? 4 : 8;
    const size_t fbytes = (size_t)p.nt * p.nlat * p.nlon * esz;
    if (p.on_device) {
        b->d_field = p.field;
        b->d_intensity = p.intensity;
    } else {
        void* d = dmalloc(ctx, fbytes);
        if (!d) { delete b; return 1; }
        b->owned.push_back(d);
        h2d(ctx, d, p.field, fbytes);
        b->d_field = d;
        if (p.intensity) {
            void* di = dmalloc(ctx, fbytes);
            if (!di) { wb_free_batch(b); return 1; }
            b->owned.push_back(di);
            h2d(ctx, di, p.intensity, fbytes);
            b->d_intensity = di;
        }
    }
    b->nwork_strips = (p.nlon + WB_STRIP - 1) / WB_STRIP;                       // strips that are streamed
    b->nstrips = b->nwork_strips + (padd > 1 ? (padd - 1 + WB_STRIP - 1) / WB_STRIP : 0);   // + periodic copy lists
    int cap = p.cap_seg > 0 ? p.cap_seg : next_pow2(8LL * W);
    if (cap < 1024) cap = 1024;
    int caps = b->nstrips > 2 ? cap / 4 : (b->nstrips > 1 ? cap / 2 : cap);
    for (int attempt = 0; attempt < 6; ++attempt) {
        b->cap_seg = cap;
        b->cap_strip = caps;
        b->timer.clear();
        if (run_phase1(b)) { std::string e = ctx->err; wb_free_batch(b); return wb_fail(ctx, e); }
        const int mx = b->h_ctl.max_seg_job, mxs = b->h_ctl.max_seg_strip;     // non-zero only when a capacity was exceeded
        if (mx <= cap && mxs <= caps) break;
        // overflow: free everything phase 1 allocated (the field copies stay) and retry with room to spare
        std::vector<void*> keep;
        for (void* q : b->owned) {
            if (q == b->d_field || q == b->d_intensity) keep.push_back(q); else dfree(ctx, q);
        }
        b->owned = keep;
        b->phase1_scratch.clear();
        if (mx > cap) cap = next_pow2((long long)mx + mx / 4);
        if (mxs > caps) caps = next_pow2((long long)mxs + mxs / 4);
        if (caps > cap) caps = cap;
        if (attempt == 5) { wb_free_batch(b); return wb_fail(ctx, "segment capacity overflow"); }
    }
    b->h_status.assign(b->njobs, 0);
    // phase-1 scratch is dead now
    double wb_t0_ = wb_now_ms();
    for (void* q : b->phase1_scratch) bfree_one(b, q);
    b->phase1_scratch.clear();
    bfree_one(b, b->stg_k); bfree_one(b, b->stg_v); bfree_one(b, b->stg_n); bfree_one(b, b->cown);
    b->stg_k = b->cown = nullptr;
    b->stg_v = nullptr;
    b->stg_n = nullptr;
    WB_TICK("phase1 frees");
    b->ct_base.assign(b->njobs + 1, 0);
    for (int j = 0; j < b->njobs; ++j) b->ct_base[j + 1] = b->ct_base[j] + b->h_ncont[j];
    *out = b;
    if (wb_debug()) fprintf(stderr, "[wb] wb_contours total            %9.3f ms\n", wb_now_ms() - wb_tc_);
    return 0;
}

extern "C" int wb_batch_max_nux(wb_batch* b) { return b ? b->max_nux : 0; }

extern "C" int wb_get_status(wb_batch* b, const uint32_t** st, int64_t* n) {
    if (!b) return 1;
    d2h(b->ctx, b->h_status.data(), b->status, b->njobs * sizeof(uint32_t));
    *st = b->h_status.data();
    *n = b->njobs;
    return 0;
}

extern "C" int wb_get_band_hits(wb_batch* b, int64_t* geo, int64_t* cont) {
    if (!b) return 1;
    *geo = b->band[0];
    *cont = b->band[1];
    return 0;
}

extern "C" int wb_set_index_params(wb_batch* b, double geo_dis, double cont_dis, double range_group, double ot_min_exp,
                                   double co_min_exp) {
    if (!b) return 1;
    if (geo_dis == geo_dis) b->p.geo_dis = geo_dis;            // NaN = keep
    if (cont_dis == cont_dis) b->p.cont_dis = cont_dis;
    if (range_group == range_group) b->p.range_group = range_group;
    if (ot_min_exp == ot_min_exp) b->p.ot_min_exp = ot_min_exp;
    if (co_min_exp == co_min_exp) b->p.co_min_exp = co_min_exp;
    return 0;
}

extern "C" int wb_set_intensity(wb_batch* b, const void* dev_intensity) {
    if (!b) return 1;
    if (!b->p.on_device) return wb_fail(b->ctx, "wb_set_intensity: the batch owns host-side inputs (on_device = 0)");
    b->d_intensity = dev_intensity;
    return 0;
}

extern "C" int wb_get_com_rechecks(wb_batch* b, int64_t* events, int64_t* boxes, int64_t* boxes_one_lane) {
    if (!b) return 1;
    *events = b->com_rechecks[0];
    *boxes = b->com_rechecks[1];
    *boxes_one_lane = b->com_rechecks[2];
    return 0;
}

// counters of the last wb_indices by name: "box_bound" (queued overturning boxes whose centre-of-mass sums followed from the exact
// totals of their terms, no chain), "box_bound_mismatch" (test builds with -DWB_BOXCOM_BOUND=2 only: boxes on which that and the
// chained evaluation disagree -- must be 0); of wb_contours: "fused_declined" (jobs wb_k_trace_fused left to the three-kernel chain:
// a cell equal to the level, or more segments / chains / splitters than the launch's LDS admitted -- the context's next launch has
// grown), "fused_lds_bytes", "fused_jobs_per_cu" (-1: the fused kernel was not launched)
extern "C" int wb_get_counter(wb_batch* b, const char* name, int64_t* value) {
    if (!b || !name || !value) return 1;
    if (!strcmp(name, "box_bound")) { *value = b->box_bound; return 0; }
    if (!strcmp(name, "box_bound_mismatch")) { *value = b->box_bound_mismatch; return 0; }
    if (!strcmp(name, "fused_declined")) { *value = b->fused_declined; return 0; }
    if (!strcmp(name, "fused_lds_bytes")) { *value = b->fused_lds; return 0; }
    if (!strcmp(name, "fused_jobs_per_cu")) { *value = b->fused_per_cu; return 0; }
    return wb_fail(b->ctx, std::string("wb_get_counter: unknown counter ") + name);
}
extern "C" int wb_get_pair_stats(wb_batch* b, int64_t* point_pairs, int64_t* candidate_pairs) {
    if (!b) return 1;
    *point_pairs = b->pair_stats[0];
    *candidate_pairs = b->pair_stats[1];
    return 0;
}
extern "C" int wb_get_event_geometry(wb_batch* b, int kind, wb_geometry* o) {
    if (!b || !o || kind < 0 || kind > 2) return 1;
    wb_use(b->ctx);
    if (wb_geom_settle(b)) return 1;              // geometry kernels wb_indices did not wait for (context option geometry_async)
    wb_batch::Ev& t = b->ev[kind];
    WbRingSink& g = b->geo[kind];
    g = WbRingSink();
    const int64_t n = (int64_t)t.step.size();
    if (b->g_rings >= 0) {
        // cut and mapped by the device during wb_indices: this table's slice of the pass's arrays, offsets from zero
        const int64_t lo = kind == 0 ? 0 : (kind == 1 ? b->dev_n[0] : b->dev_n[0] + b->dev_n[1]);
        const long long r0 = b->g_ring_offset[lo], r1 = b->g_ring_offset[lo + n];
        if (r0 < 0 || r1 < r0 || r1 > b->g_rings)
            return wb_fail(b->ctx, "wb_get_event_geometry: ring offsets out of order (kind " + std::to_string(kind) + ", events " + std::to_string(lo) + " + " + std::to_string(n) +
                           ", rings " + std::to_string(r0) + " .. " + std::to_string(r1) + " of " + std::to_string(b->g_rings) + ")");
        const long long v0 = b->g_vert_offset[r0];
        g.ring_offset.resize((size_t)n + 1);
        for (int64_t k = 0; k <= n; ++k) g.ring_offset[(size_t)k] = b->g_ring_offset[lo + k] - r0;
        g.vert_offset.resize((size_t)(r1 - r0) + 1);
        for (long long q = 0; q <= r1 - r0; ++q) g.vert_offset[(size_t)q] = b->g_vert_offset[r0 + q] - v0;
        o->n = n;
        o->ring_offset = g.ring_offset.data();
        o->nrings = (int64_t)(r1 - r0);
        o->vert_offset = g.vert_offset.data();
        o->x = b->g_x + v0;
        o->y = b->g_y + v0;
        return 0;
    }
    // the host function (option "geometry" off, or a pass the device's arrays did not hold / an event its kernel flagged): from the
    // polygons in grid indices
    if (n > 0 && !b->poly_x && b->poly_n > 0 && wb_pack_polys_now(b)) return 1;
    g.ring_offset.push_back(0);
    g.vert_offset.push_back(0);
    for (int64_t k = 0; k < n; ++k) {
        const int64_t a = t.polyoff[(size_t)k], e = t.polyoff[(size_t)k + 1];
        wb_event_rings(b->poly_x + t.pbase + a, b->poly_y + t.pbase + a, e - a, b->lon.data(), b->lat.data(), b->p.nlon, g);
    }
    o->n = n;
    o->ring_offset = g.ring_offset.data();
    o->nrings = (int64_t)g.vert_offset.size() - 1;
    o->vert_offset = g.vert_offset.data();
    o->x = g.x.data();
    o->y = g.y.data();
    return 0;
}
extern "C" int wb_transform_polygons(wb_ctx* ctx, int64_t n, const int64_t* poly_offset, const int32_t* px, const int32_t* py,
                                     const double* lon, const double* lat, int nlon, int index_space, wb_geometry* o) {
    if (!ctx || !o || n < 0 || (n && (!poly_offset || !px || !py))) return 1;
    if (nlon < 1 || (!index_space && (!lon || !lat))) return wb_fail(ctx, "wb_transform_polygons: lon / lat / nlon missing");
    WbRingSink& g = ctx->geo;
    g = WbRingSink();
    g.ring_offset.push_back(0);
    g.vert_offset.push_back(0);
    for (int64_t k = 0; k < n; ++k)
        wb_event_rings(px + poly_offset[k], py + poly_offset[k], poly_offset[k + 1] - poly_offset[k], lon, lat, nlon, g, index_space != 0);
    o->n = n;
    o->ring_offset = g.ring_offset.data();
    o->nrings = (int64_t)g.vert_offset.size() - 1;
    o->vert_offset = g.vert_offset.data();
    o->x = g.x.data();
    o->y = g.y.data();
    return 0;
}
// the same polygons through the device kernels of wb_geom.h (what wb_indices runs for a pass's events): uploaded, cut and
// mapped on the device, read back.  An event the kernels flag (more than WB_GEOM_XMAX cut edges) sends the call to the host
// function above.
extern "C" int wb_transform_polygons_device(wb_ctx* ctx, int64_t n, const int64_t* poly_offset, const int32_t* px, const int32_t* py,
                                            const double* lon, const double* lat, int nlon, int nlat, int index_space, wb_geometry* o) {
    if (!ctx || !o || n < 0 || (n && (!poly_offset || !px || !py))) return 1;
    if (nlon < 1 || (!index_space && (!lon || !lat || nlat < 1))) return wb_fail(ctx, "wb_transform_polygons_device: lon / lat / nlon missing");
    if (!wb_select_device(ctx)) return wb_fail(ctx, "hipSetDevice failed");
    const long long NP = n ? poly_offset[n] : 0;
    if (!index_space)                             // (wb_k_geom_fill looks lat[y] up unguarded: rows outside the grid are the caller's error)
        for (long long q = 0; q < NP; ++q)
            if (py[q] < 0 || py[q] >= nlat) return wb_fail(ctx, "wb_transform_polygons_device: a polygon vertex lies outside the grid's rows (0 <= y < nlat)");
    std::vector<void*> tmp;
    auto talloc = [&](size_t bytes) { void* q = dmalloc(ctx, bytes ? bytes : 8); if (q) tmp.push_back(q); return q; };
    auto release = [&]() { for (void* q : tmp) dfree(ctx, q); tmp.clear(); };
    WbGeomArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.nev = n; ga.nlon = nlon; ga.index_space = index_space ? 1 : 0;
    long long* d_po = (long long*)talloc((size_t)(n + 1) * sizeof(long long));
    int32_t* d_px = (int32_t*)talloc((size_t)(NP + 1) * 4); int32_t* d_py = (int32_t*)talloc((size_t)(NP + 1) * 4);
    double* d_lon = (double*)talloc((size_t)(lon ? nlon : 1) * 8); double* d_lat = (double*)talloc((size_t)(lat ? nlat : 1) * 8);
    ga.cap_r = n + NP / 2 + 16; ga.cap_v = 2 * NP + 4 * n + 64;
    ga.cnt_r = (int*)talloc((size_t)(n + 2) * 4); ga.cnt_v = (int*)talloc((size_t)(n + 2) * 4); ga.gflag = (int*)talloc((size_t)(n + 2) * 4);
    ga.ring_offset = (long long*)talloc((size_t)(n + 2) * 8); ga.vert_offset = (long long*)talloc((size_t)(ga.cap_r + 2) * 8);
    ga.x = (double*)talloc((size_t)(ga.cap_v + 2) * 8); ga.y = (double*)talloc((size_t)(ga.cap_v + 2) * 8);
    if (!ga.y || !ga.x || !ga.vert_offset || !ga.ring_offset || !ga.gflag || !ga.cnt_v || !ga.cnt_r || !d_lat || !d_lon || !d_py || !d_px || !d_po) {
        release();
        return wb_fail(ctx, "out of device memory");
    }
    // a copy or launch that fails: the host function is the same statement (the device path is an optimisation of this call)
    auto host = [&]() { release(); return wb_transform_polygons(ctx, n, poly_offset, px, py, lon, lat, nlon, index_space, o); };
    std::vector<long long> po(poly_offset, poly_offset + n + 1);
    if (n == 0) po.assign(1, 0);
    int rc = h2d(ctx, d_po, po.data(), po.size() * sizeof(long long));
    if (!rc && NP) { rc = h2d(ctx, d_px, px, (size_t)NP * 4); if (!rc) rc = h2d(ctx, d_py, py, (size_t)NP * 4); }
    if (!rc && lon) rc = h2d(ctx, d_lon, lon, (size_t)nlon * 8);
    if (!rc && lat) rc = h2d(ctx, d_lat, lat, (size_t)nlat * 8);
    if (rc) return host();
    ga.poly_offset = d_po; ga.px = d_px; ga.py = d_py; ga.lon = d_lon; ga.lat = d_lat;
    WbOpTimer tm(ctx);
    WB_LAUNCH(wb_k_geom_count, n < 4096 ? (n > 0 ? n : 1) : 4096, WB_GEOM_THREADS, ga);
    WB_LAUNCH(wb_k_geom_offsets, 1, WB_LIST_THREADS, ga);
    WB_LAUNCH(wb_k_geom_fill, n < 4096 ? (n > 0 ? n : 1) : 4096, WB_GEOM_THREADS, ga);
    tm.stop();
    std::vector<int> cr((size_t)n + 1), fl((size_t)n + 1, 0);
    std::vector<int> cv((size_t)n + 1);
    rc = dsync(ctx);
    if (!rc) rc = d2h(ctx, cr.data(), ga.cnt_r, (size_t)(n + 1) * 4);
    if (!rc) rc = d2h(ctx, cv.data(), ga.cnt_v, (size_t)(n + 1) * 4);
    if (!rc && n) rc = d2h(ctx, fl.data(), ga.gflag, (size_t)n * 4);
    if (rc) return host();
    bool flagged = false;
    for (int64_t k = 0; k < n; ++k) flagged |= fl[(size_t)k] != 0;
    const long long NR = cr[(size_t)n], NV = cv[(size_t)n];
    if (flagged || NR < 0 || NV < 0 || NR > ga.cap_r || NV > ga.cap_v) return host();
    WbRingSink& g = ctx->geo;
    g = WbRingSink();
    g.ring_offset.resize((size_t)n + 1); g.vert_offset.resize((size_t)NR + 1); g.x.resize((size_t)NV); g.y.resize((size_t)NV);
    rc = d2h(ctx, g.ring_offset.data(), ga.ring_offset, (size_t)(n + 1) * 8);
    if (!rc) rc = d2h(ctx, g.vert_offset.data(), ga.vert_offset, (size_t)(NR + 1) * 8);
    if (!rc && NV) { rc = d2h(ctx, g.x.data(), ga.x, (size_t)NV * 8); if (!rc) rc = d2h(ctx, g.y.data(), ga.y, (size_t)NV * 8); }
    if (rc) return host();
    release();
    o->n = n;
    o->ring_offset = g.ring_offset.data();
    o->nrings = NR;
    o->vert_offset = g.vert_offset.data();
    o->x = g.x.data();
    o->y = g.y.data();
    return 0;
}
// ---- development: the streamer index' two lattice predicates on the device (include/wavebreaking_hip_dev.h)
WB_KERNEL(wb_k_test_chord_touches, int n, const int* x, const int* y, int nchords, const int* i1, const int* i2, int* out) {
    for (int k = WB_BLOCK_ID; k < nchords; k += WB_GRID_DIM) {
        WB_SHARED(int, hit, 1);
        WB_SINGLE
        hit[0] = 0;
        WB_END_SINGLE
        const int px = x[i1[k]], py = y[i1[k]], qx = x[i2[k]], qy = y[i2[k]];
        WB_PFOR(m, n - 1)
            if (wb_chord_hits_segment(px, py, qx, qy, x[m], y[m], x[m + 1], y[m + 1], x[0], y[0], x[n - 1], y[n - 1])) hit[0] = 1;
        WB_END_PFOR
        WB_SINGLE
        out[k] = hit[0] ? 0 : 1;
        WB_END_SINGLE
    }
}
WB_KERNEL(wb_k_test_segments_intersect, int np, const int* s8, int* out) {
    for (int b0 = WB_BLOCK_ID * 256; b0 < np; b0 += WB_GRID_DIM * 256) {
        WB_PFOR(t, 256)
            const int k = b0 + (int)t;
            if (k < np) {
                const int* s = s8 + (size_t)k * 8;
                out[k] = wb_segments_intersect(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]) ? 1 : 0;
            }
        WB_END_PFOR
    }
}
extern "C" int wb_test_chord_touches(wb_ctx* ctx, int n, const int32_t* x, const int32_t* y, int nchords, const int32_t* i1, const int32_t* i2,
                                     int32_t* touches) {
    if (!ctx || n < 2 || nchords < 0 || !x || !y || (nchords && (!i1 || !i2 || !touches))) return 1;
    if (!wb_select_device(ctx)) return wb_fail(ctx, "hipSetDevice failed");
    int* d = (int*)dmalloc(ctx, ((size_t)2 * n + (size_t)3 * nchords + 4) * sizeof(int));
    if (!d) return wb_fail(ctx, "out of device memory");
    int *dx = d, *dy = d + n, *d1 = d + 2 * n, *d2 = d1 + nchords, *dout = d2 + nchords;
    h2d(ctx, dx, x, (size_t)n * 4); h2d(ctx, dy, y, (size_t)n * 4);
    if (nchords) { h2d(ctx, d1, i1, (size_t)nchords * 4); h2d(ctx, d2, i2, (size_t)nchords * 4); }
    WB_LAUNCH(wb_k_test_chord_touches, nchords < 1024 ? (nchords > 0 ? nchords : 1) : 1024, 256, n, dx, dy, nchords, d1, d2, dout);
    int rc = dsync(ctx);
    if (!rc && nchords) rc = d2h(ctx, touches, dout, (size_t)nchords * 4);
    dfree(ctx, d);
    return rc;
}
WB_KERNEL(wb_k_test_props_columns, int n, const double* sums, const double* lon, const double* lat, int nlon, int nlat, double* out) {
    for (int b0 = WB_BLOCK_ID * 256; b0 < n; b0 += WB_GRID_DIM * 256) {
        WB_PFOR(t, 256)
            const int k = b0 + (int)t;
            if (k < n) wb_props_columns(sums + (size_t)k * 5, lon, lat, nlon, nlat, out + (size_t)k * 5);
        WB_END_PFOR
    }
}
extern "C" int wb_test_props_columns(wb_ctx* ctx, int64_t n, const double* sums, const double* lon, int nlon, const double* lat, int nlat, double* out) {
    if (!ctx || n < 0 || nlon < 1 || nlat < 1 || !lon || !lat || (n && (!sums || !out))) return 1;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!wb_select_device(ctx)) return wb_fail(ctx, "hipSetDevice failed");
    double* d = (double*)dmalloc(ctx, ((size_t)10 * n + nlon + nlat + 4) * sizeof(double));
    if (!d) return wb_fail(ctx, "out of device memory");
    double *ds = d, *dout = d + 5 * n, *dlon = dout + 5 * n, *dlat = dlon + nlon;
    int rc = n ? h2d(ctx, ds, sums, (size_t)n * 5 * 8) : 0;
    if (!rc) rc = h2d(ctx, dlon, lon, (size_t)nlon * 8);
    if (!rc) rc = h2d(ctx, dlat, lat, (size_t)nlat * 8);
    if (!rc) {
        WB_LAUNCH(wb_k_test_props_columns, (n + 255) / 256 > 0 ? ((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024) : 1, 256, (int)n, ds, dlon, dlat, nlon, nlat, dout);
        rc = dsync(ctx);
    }
    if (!rc && n) rc = d2h(ctx, out, dout, (size_t)n * 5 * 8);
    dfree(ctx, d);
    return rc;
}
extern "C" int wb_test_segments_intersect(wb_ctx* ctx, int npairs, const int32_t* seg8, int32_t* intersects) {
    if (!ctx || npairs < 0 || (npairs && (!seg8 || !intersects))) return 1;
    if (!wb_select_device(ctx)) return wb_fail(ctx, "hipSetDevice failed");
    int* d = (int*)dmalloc(ctx, ((size_t)9 * npairs + 4) * sizeof(int));
    if (!d) return wb_fail(ctx, "out of device memory");
    if (npairs) h2d(ctx, d, seg8, (size_t)npairs * 8 * 4);
    WB_LAUNCH(wb_k_test_segments_intersect, (npairs + 255) / 256 > 0 ? ((npairs + 255) / 256 < 1024 ? (npairs + 255) / 256 : 1024) : 1, 256, npairs, d, d + (size_t)8 * npairs);
    int rc = dsync(ctx);
    if (!rc && npairs) rc = d2h(ctx, intersects, d + (size_t)8 * npairs, (size_t)npairs * 4);
    dfree(ctx, d);
    return rc;
}

extern "C" int wb_set_step_offset(wb_batch* b, int64_t first_step) {
    if (!b) return 1;
    b->step_offset = first_step;
    return 0;
}
extern "C" int wb_get_device_tables(wb_batch* b, const double** rows, int64_t n[3], const int32_t** poly_x, const int32_t** poly_y,
                                    int64_t* nverts) {
    if (!b || !b->dev_rows) return 1;
    *rows = b->dev_rows;
    n[0] = b->dev_n[0]; n[1] = b->dev_n[1]; n[2] = b->dev_n[2];
    *poly_x = b->dev_px; *poly_y = b->dev_py;
    *nverts = b->dev_nv;
    return 0;
}
// the rings of the last wb_indices (context option "geometry") as they lie where the kernels wrote them: DEVICE arrays with the
// options geometry_dma and keep_device (a time-sharded run gathers them device to device beside the tables of
// wb_get_device_tables), else the pinned host arrays.  ring_offset[nev + 1] (events in table order: streamers, overturnings,
// cutoffs), vert_offset[nrings + 1], x / y[nverts] degrees.  Waits for the geometry kernels.
extern "C" int wb_get_device_geometry(wb_batch* b, const int64_t** ring_offset, int64_t* nev, const int64_t** vert_offset, int64_t* nrings,
                                      const double** x, const double** y, int64_t* nverts, int* on_device) {
    if (!b || !ring_offset || !nev || !vert_offset || !nrings || !x || !y || !nverts || !on_device) return 1;
    wb_use(b->ctx);
    if (wb_geom_settle(b)) return 1;
    if (b->g_rings < 0) return wb_fail(b->ctx, "wb_get_device_geometry: the pass has no device geometry (context option geometry off, or a pass its arrays did not hold)");
    *nev = b->dev_n[0] + b->dev_n[1] + b->dev_n[2]; *nrings = b->g_rings; *nverts = b->g_verts;
    const bool dev = b->g_dx != nullptr;
    *on_device = dev ? 1 : 0;
    *ring_offset = (const int64_t*)(dev ? b->g_dro : b->g_ring_offset);
    *vert_offset = (const int64_t*)(dev ? b->g_dvo : b->g_vert_offset);
    *x = dev ? b->g_dx : b->g_x;
    *y = dev ? b->g_dy : b->g_y;
    return 0;
}
extern "C" int wb_get_group_ties(wb_batch* b, int64_t* ties) {
    if (!b) return 1;
    *ties = b->group_ties;
    return 0;
}
extern "C" int wb_get_evaluated_pairs(wb_batch* b, int64_t* evaluated) {
    if (!b) return 1;
    *evaluated = b->pair_stats[2];
    return 0;
}

extern "C" int wb_get_timings(wb_batch* b, const char** names, const float** ms, int* n) {
    if (!b) return 1;
    b->timer.collect();
    b->timer_names.clear();
    for (auto& s : b->timer.names) { b->timer_names += s; b->timer_names.push_back('\0'); }
    *names = b->timer_names.c_str();
    *ms = b->timer.ms.data();
    *n = (int)b->timer.ms.size();
    return 0;
}

// ------------------------------------------------------------------------------------------------ contour table
struct WbPackArgs {
    int njobs, cap_seg, nlev;
    const int* ncont; const int* c_off; const int* c_len; const int* c_closed; const int* c_nux; const int* c_sumy;
    const uint32_t* cpts;
    const long long* cbase;    // [njobs] first contour row
    const long long* pbase;    // [njobs] first point
    int* o_step; int* o_level; int* o_closed; int* o_nux; int* o_npts; long long* o_sumy; long long* o_off;
    int* o_x; int* o_y;
};
WB_KERNEL(wb_k_job_npts, int njobs, int cap_seg, const int* ncont, const int* c_len, long long* npts) {
    const int job = WB_BLOCK_ID;
    if (job >= njobs) return;
    WB_SINGLE
    long long s = 0;
    for (int k = 0; k < ncont[job]; ++k) s += c_len[(size_t)job * cap_seg + k];
    npts[job] = s;
    WB_END_SINGLE
}
WB_KERNEL(wb_k_pack_contours, WbPackArgs a) {
    const int job = WB_BLOCK_ID;
    if (job >= a.njobs) return;
    const size_t jb = (size_t)job * a.cap_seg;
    const int C = a.ncont[job];
    WB_SINGLE
    long long run = a.pbase[job];
    for (int k = 0; k < C; ++k) {
        const long long row = a.cbase[job] + k;
        a.o_step[row] = job / a.nlev;
        a.o_level[row] = job % a.nlev;
        a.o_closed[row] = a.c_closed[jb + k];
        a.o_nux[row] = a.c_nux[jb + k];
        a.o_npts[row] = a.c_len[jb + k];
        a.o_sumy[row] = a.c_sumy[jb + k];
        a.o_off[row] = run;
        run += a.c_len[jb + k];
    }
    WB_END_SINGLE
    for (int k = 0; k < C; ++k) {
        const long long dst = a.o_off[a.cbase[job] + k];
        const uint32_t* src = a.cpts + 2 * jb + a.c_off[jb + k];
        WB_PFOR(i, a.c_len[jb + k])
            a.o_x[dst + i] = wb_pt_x(src[i]);
            a.o_y[dst + i] = wb_pt_y(src[i]);
        WB_END_PFOR
    }
}

static int fetch_contours(wb_batch* b) {
    if (b->have_contours) return 0;
    wb_ctx* ctx = b->ctx;
    const int J = b->njobs;
    const long long NC = b->ct_base[J];
    long long* d_npts = balloc<long long>(b, J);
    b->timer.begin(ctx, "pack_contours");
    WB_LAUNCH(wb_k_job_npts, J, 64, J, b->cap_seg, b->ncont, b->c_len, d_npts);
    std::vector<long long> npts(J), pbase(J + 1, 0), cbase(b->ct_base.begin(), b->ct_base.end());
    d2h(ctx, npts.data(), d_npts, J * sizeof(long long));
    for (int j = 0; j < J; ++j) pbase[j + 1] = pbase[j] + npts[j];
    const long long NP = pbase[J];
    long long* d_cbase = upload(b, cbase);
    long long* d_pbase = upload(b, pbase);
    WbPackArgs a;
    a.njobs = J; a.cap_seg = b->cap_seg; a.nlev = b->p.nlevels;
    a.ncont = b->ncont; a.c_off = b->c_off; a.c_len = b->c_len; a.c_closed = b->c_closed; a.c_nux = b->c_nux;
    a.c_sumy = b->c_sumy; a.cpts = b->cpts; a.cbase = d_cbase; a.pbase = d_pbase;
    a.o_step = balloc<int>(b, NC); a.o_level = balloc<int>(b, NC); a.o_closed = balloc<int>(b, NC);
    a.o_nux = balloc<int>(b, NC); a.o_npts = balloc<int>(b, NC); a.o_sumy = balloc<long long>(b, NC);
    a.o_off = balloc<long long>(b, NC); a.o_x = balloc<int>(b, NP); a.o_y = balloc<int>(b, NP);
    if (!a.o_y) return 1;
    WB_LAUNCH(wb_k_pack_contours, J, 256, a);
    b->ct_step.resize(NC); b->ct_level.resize(NC); b->ct_closed.resize(NC); b->ct_nux.resize(NC);
    b->ct_npts.resize(NC); b->ct_sumy.resize(NC); b->ct_off.resize(NC); b->ct_x.resize(NP); b->ct_y.resize(NP);
    if (NC) {
        d2h(ctx, b->ct_step.data(), a.o_step, NC * 4); d2h(ctx, b->ct_level.data(), a.o_level, NC * 4);
        d2h(ctx, b->ct_closed.data(), a.o_closed, NC * 4); d2h(ctx, b->ct_nux.data(), a.o_nux, NC * 4);
        d2h(ctx, b->ct_npts.data(), a.o_npts, NC * 4); d2h(ctx, b->ct_sumy.data(), a.o_sumy, NC * 8);
        d2h(ctx, b->ct_off.data(), a.o_off, NC * 8);
    }
    if (NP) { d2h(ctx, b->ct_x.data(), a.o_x, NP * 4); d2h(ctx, b->ct_y.data(), a.o_y, NP * 4); }
    b->timer.end(ctx);
    if (dsync(ctx)) return 1;
    if (!b->job_step.empty())
        for (int j = 0; j < J; ++j)
            for (long long r = b->ct_base[j]; r < b->ct_base[j + 1]; ++r) { b->ct_step[r] = b->job_step[j]; b->ct_level[r] = b->job_level[j]; }
    void* tmp[] = {d_npts, d_cbase, d_pbase, a.o_step, a.o_level, a.o_closed, a.o_nux, a.o_npts, a.o_sumy, a.o_off, a.o_x, a.o_y};
    for (void* q : tmp) bfree_one(b, q);
    b->have_contours = true;
    return 0;
}

extern "C" int wb_get_contours(wb_batch* b, wb_contour_table* o) {
    if (!b || !o) return 1;
    if (fetch_contours(b)) return 1;
    b->timer.collect();
    o->n = (int64_t)b->ct_step.size();
    o->step = b->ct_step.data(); o->level_idx = b->ct_level.data(); o->closed = b->ct_closed.data();
    o->nux = b->ct_nux.data(); o->sum_y = (const int64_t*)b->ct_sumy.data(); o->npts = b->ct_npts.data();
    o->offset = (const int64_t*)b->ct_off.data(); o->npoints = (int64_t)b->ct_x.size();
    o->x = b->ct_x.data(); o->y = b->ct_y.data();
    return 0;
}

// ------------------------------------------------------------------------------------------------ phase 2
struct WbEvFillArgs {
    int cap_ev, cap_ne, cap_seg;
    long long cap_poly;
    WbCtl* ctl;
    const int* sel_job; const int* sel_cont; const int* cut_job; const int* cut_cont;
    const int* ev_n; const int* ev_ind; int* sbase;   // streamers (ev_n == NULL: not requested)
    const int* ot_n; const int* ot_box; const int* ot_orient; int* obase;   // overturnings
    const int* c_len;
    int* ev;           // [cap_ne][8]
    int* plen;         // [cap_ne + 1] polygon length of an event -> exclusive scan
    long long* poff;   // [cap_ne + 1]
};
// event offsets: streamers, then overturnings, then cutoffs, each in (selected contour, event) order = frame order
WB_KERNEL(wb_k_event_offsets, WbEvFillArgs a) {
    const int nsel = a.ctl->nsel;
    WB_PFOR(s, nsel)
        a.sbase[s] = a.ev_n ? a.ev_n[s] : 0;
        a.obase[s] = a.ot_n ? a.ot_n[s] : 0;
    WB_END_PFOR
    const int NS = wb_block_scan_excl(a.sbase, nsel);
    const int NO = wb_block_scan_excl(a.obase, nsel);
    WB_SINGLE
    const long long NE = (long long)NS + NO + a.ctl->ncut;
    a.ctl->NS = NS;
    a.ctl->NO = NO;
    a.ctl->props_cursor = 0;
    if (NE > a.cap_ne) { a.ctl->overflow |= 1; a.ctl->NE = 0; a.ctl->ne_demand = (int)(NE > 0x7FFFFFFF ? 0x7FFFFFFF : NE); }
    else a.ctl->NE = (int)NE;
    WB_END_SINGLE
}
WB_KERNEL(wb_k_fill_events, WbEvFillArgs a) {
    const int s = WB_BLOCK_ID;
    const int nsel = a.ctl->nsel;
    if (a.ctl->NE == 0) return;
    const int NS = a.ctl->NS, NO = a.ctl->NO;
    if (s < nsel) {
        if (a.ev_n) {
            WB_PFOR(e, a.ev_n[s])
                const int row = a.sbase[s] + (int)e;
                int* d = a.ev + (size_t)row * 8;
                d[0] = WB_EV_STREAMER; d[1] = a.sel_job[s]; d[2] = a.sel_cont[s];
                d[3] = a.ev_ind[((size_t)s * a.cap_ev + e) * 2]; d[4] = a.ev_ind[((size_t)s * a.cap_ev + e) * 2 + 1];
                d[5] = 0; d[6] = 0; d[7] = 0;
                a.plen[row] = d[4] - d[3] + 1;
            WB_END_PFOR
        }
        if (a.ot_n) {
            WB_PFOR(e, a.ot_n[s])
                const int row = NS + a.obase[s] + (int)e;
                int* d = a.ev + (size_t)row * 8;
                const int* bx = a.ot_box + ((size_t)s * WB_OT_CAP + e) * 4;
                d[0] = WB_EV_OVERTURNING | (a.ot_orient[(size_t)s * WB_OT_CAP + e] << 8); d[1] = a.sel_job[s]; d[2] = a.sel_cont[s];
                d[3] = bx[0]; d[4] = bx[1]; d[5] = bx[2]; d[6] = bx[3];
                d[7] = s * WB_OT_CAP + (int)e;                   // slot of the box in ot_sums
                a.plen[row] = 4;
            WB_END_PFOR
        }
    } else {
        const int c = s - nsel;
        if (c < a.ctl->ncut) {
            WB_SINGLE
            const int row = NS + NO + c;
            int* d = a.ev + (size_t)row * 8;
            d[0] = WB_EV_CUTOFF; d[1] = a.cut_job[c]; d[2] = a.cut_cont[c];
            d[3] = 0; d[4] = a.c_len[(size_t)a.cut_job[c] * a.cap_seg + a.cut_cont[c]] - 1; d[5] = 0; d[6] = 0; d[7] = 0;
            a.plen[row] = d[4] + 1;
            WB_END_SINGLE
        }
    }
}
WB_KERNEL(wb_k_poly_offsets, WbEvFillArgs a) {
    const int NE = a.ctl->NE;
    const int total = wb_block_scan_excl(a.plen, NE);
    WB_PFOR(e, NE + 1)
        a.poff[e] = e < NE ? (long long)a.plen[e] : (long long)total;
    WB_END_PFOR
    WB_SINGLE
    a.ctl->poly_total = total;
    if ((long long)total > a.cap_poly) a.ctl->overflow |= 2;
    WB_END_SINGLE
}

// one row of 20 doubles per event, on the device, for the rank-0 gather of a time-sharded run (no host staging):
// step (+ the shard's first step), level, contour inside its (step, level) job, ind1, ind2, box[4], orientation, the five
// sums, cell count, kind, flags, polygon length
#define WB_ROW_WIDTH 20
WB_KERNEL(wb_k_pack_rows, const WbCtl* ctl, const int* ev, const double* sums, const int* ncells, const int* flags,
          const long long* poff, const int* job_step, const int* job_level, int nlev, long long step_offset, double* rows) {
    const int nev = ctl->NE;
    for (int e0 = WB_BLOCK_ID * 256; e0 < nev; e0 += WB_GRID_DIM * 256) {
        WB_PFOR(k, 256)
            const int e = e0 + (int)k;
            if (e < nev) {
                const int* d = ev + (size_t)e * 8;
                double* r = rows + (size_t)e * WB_ROW_WIDTH;
                const int kind = d[0] & 0xFF, job = d[1];
                r[0] = (double)((job_step ? job_step[job] : job / nlev) + step_offset);
                r[1] = (double)(job_level ? job_level[job] : job % nlev);
                r[2] = (double)d[2];
                const bool box = kind == WB_EV_OVERTURNING;
                r[3] = box ? 0.0 : (double)d[3];
                r[4] = box ? 0.0 : (double)d[4];
                for (int q = 0; q < 4; ++q) r[5 + q] = box ? (double)d[3 + q] : 0.0;
                r[9] = (double)(d[0] >> 8);
                for (int q = 0; q < 5; ++q) r[10 + q] = sums[(size_t)e * 5 + q];
                r[15] = (double)ncells[e];
                r[16] = (double)kind;
                r[17] = (double)flags[e];
                r[18] = (double)(poff[e + 1] - poff[e]);
                r[19] = 0.0;
            }
        WB_END_PFOR
    }
}

// event polygons: vertices of every event, back to back -- written straight into the batch's PINNED host arrays (the
// device reaches them over PCIe; until round 4 the vertices went to device arrays first and two copy operations of 6.7 MB
// each brought them over), and, for a time-sharded run's gather, into device arrays as well
WB_KERNEL(wb_k_pack_polys, const WbCtl* ctl, int cap_seg, const int* ev, const long long* poff, const int* c_off, const uint32_t* cpts,
          int* out_x, int* out_y, int* dev_x, int* dev_y) {
    if (ctl->overflow & 2) return;
    const int nev = ctl->NE;
    for (int e = WB_BLOCK_ID; e < nev; e += WB_GRID_DIM) {
        const int* d = ev + (size_t)e * 8;
        const size_t jb = (size_t)d[1] * cap_seg;
        const uint32_t* pts = cpts + 2 * jb + c_off[jb + d[2]];
        const int n = (int)(poff[e + 1] - poff[e]);
        int* dx = out_x ? out_x + poff[e] : nullptr;          // (NULL: the device copies only -- a gather's)
        int* dy = out_x ? out_y + poff[e] : nullptr;
        WB_PFOR(k, n)
            const uint32_t v = wb_event_vertex(d, pts, (int)k);
            if (dx) { dx[k] = wb_pt_x(v); dy[k] = wb_pt_y(v); }
            if (dev_x) { dev_x[poff[e] + k] = wb_pt_x(v); dev_y[poff[e] + k] = wb_pt_y(v); }
        WB_END_PFOR
    }
}

// the event tables of a pass -> ONE pinned host block, written by the device: control block first (the pass's counters,
// capacity flags included), then descriptors, polygon offsets, the five sums, cell counts, flags (and cell offsets).  One
// launch where seven copy operations used to follow two read-backs of the control block.
struct WbResultBlock {
    WbCtl* ctl; int* ev; long long* poff; double* sums; int* ncells; int* flags; long long* cell_off;
    double* props;                     // [n][5] the frame's property columns (index_utils.py:104-120): com_lon, com_lat, mean_var, intensity, event_area
};
static size_t wb_result_block_bytes(size_t cap_ne, bool cells) {
    const size_t n = cap_ne + 1;
    return 512 + n * (8 * sizeof(int) + sizeof(long long) + 10 * sizeof(double) + 2 * sizeof(int) + (cells ? sizeof(long long) : 0)) + 64;
}
static WbResultBlock wb_result_block(void* base, size_t cap_ne, bool cells) {
    const size_t n = cap_ne + 1;
    WbResultBlock r;
    char* q = (char*)base;
    r.ctl = (WbCtl*)q; q += 512;                      // (sizeof(WbCtl) <= 512, checked below)
    r.sums = (double*)q; q += n * 5 * sizeof(double);
    r.props = (double*)q; q += n * 5 * sizeof(double);
    r.poff = (long long*)q; q += n * sizeof(long long);
    r.cell_off = cells ? (long long*)q : nullptr; q += cells ? n * sizeof(long long) : 0;
    r.ev = (int*)q; q += n * 8 * sizeof(int);
    r.ncells = (int*)q; q += n * sizeof(int);
    r.flags = (int*)q;
    return r;
}
static_assert(sizeof(WbCtl) <= 512, "the result block reserves 512 bytes for the control block");
WB_KERNEL(wb_k_pack_results, const WbCtl* ctl, const int* ev, const long long* poff, const double* sums, const int* ncells,
          const int* flags, const long long* cell_off, WbResultBlock out, const double* lon, const double* lat, int nlon, int nlat) {
    const int nev = (ctl->overflow & 1) ? 0 : ctl->NE;
    if (WB_BLOCK_ID == 0) {
        WB_PFOR(w, sizeof(WbCtl) / 4)
            ((uint32_t*)out.ctl)[w] = ((const uint32_t*)ctl)[w];
        WB_END_PFOR
    }
    for (int e0 = WB_BLOCK_ID * 256; e0 <= nev; e0 += WB_GRID_DIM * 256) {
        WB_PFOR(k, 256)
            const int e = e0 + (int)k;
            if (e <= nev) out.poff[e] = poff[e];
            if (e < nev) {
                for (int q = 0; q < 8; ++q) out.ev[(size_t)e * 8 + q] = ev[(size_t)e * 8 + q];
                double sv[5], pc[5];
                for (int q = 0; q < 5; ++q) { sv[q] = sums[(size_t)e * 5 + q]; out.sums[(size_t)e * 5 + q] = sv[q]; }
                wb_props_columns(sv, lon, lat, nlon, nlat, pc);          // the last flops of calculate_properties (index_utils.py:104-120)
                for (int q = 0; q < 5; ++q) out.props[(size_t)e * 5 + q] = pc[q];
                out.ncells[e] = ncells[e];
                out.flags[e] = flags[e];
                if (out.cell_off) out.cell_off[e] = cell_off[e];
            }
        WB_END_PFOR
    }
}

// dynamic LDS for a workgroup of wb_k_streamer_pairs2: half a CU's LDS, so that two workgroups (16 waves) share a CU
static size_t wb_p2_lds_budget(wb_ctx* ctx) {
    if (ctx->p2_budget) return ctx->p2_budget;
    size_t per_block, per_cu;
    wb_lds_limits(ctx, &per_block, &per_cu);
    size_t want = per_cu / 2;
    if (ctx->p2_budget_kb > 0) want = (size_t)ctx->p2_budget_kb << 10;
#if !defined(WB_HOSTSIM)
    if (want > per_block) {
        // more than the default limit of a launch: ask for it
        if (hipFuncSetAttribute((const void*)wb_k_streamer_pairs2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) != hipSuccess) {
            (void)hipGetLastError();
            want = per_block;
        }
    }
#endif
    if (wb_debug()) fprintf(stderr, "[wb] LDS per block %zu, per CU %zu, pairs2 budget %zu\n", per_block, per_cu, want);
    ctx->p2_budget = want;
    return ctx->p2_budget;
}

static wb_ctx::Tables* make_tables(wb_batch* b) {
    wb_ctx* ctx = b->ctx;
    const wb_params& p = b->p;
    for (wb_ctx::Tables* t : ctx->tables)
        if (t->dlon == p.dlon && t->dlat == p.dlat && t->geo_dis == p.geo_dis && t->lat == b->lat && t->lon == b->lon) return t;
    wb_ctx::Tables* t = new wb_ctx::Tables();
    t->lat = b->lat; t->lon = b->lon; t->dlon = p.dlon; t->dlat = p.dlat; t->geo_dis = p.geo_dis;
    std::vector<double> latrad(p.nlat), sphi2(p.nlat), cphi2(p.nlat), cphi(p.nlat), wlat(p.nlat);
    std::vector<double> lonrad(p.nlon), slam2(p.nlon), clam2(p.nlon);
    // index_utils.py:58-62
    const double area_cell = pow(rint(6371.0 * 2.0 * M_PI / (360.0 / ((p.dlon + p.dlat) / 2.0))), 2.0);
    for (int y = 0; y < p.nlat; ++y) {
        latrad[y] = b->lat[y] * (M_PI / 180.0);    // np.radians
        sphi2[y] = sin(0.5 * latrad[y]);
        cphi2[y] = cos(0.5 * latrad[y]);
        cphi[y] = cos(latrad[y]);
        wlat[y] = cos(latrad[y]) * area_cell;
    }
    for (int x = 0; x < p.nlon; ++x) {
        lonrad[x] = b->lon[x] * (M_PI / 180.0);
        slam2[x] = sin(0.5 * lonrad[x]);
        clam2[x] = cos(0.5 * lonrad[x]);
    }
    std::vector<double> all;
    all.reserve((size_t)6 * p.nlat + (size_t)4 * p.nlon);
    auto put = [&](const std::vector<double>& v) { const size_t o = all.size(); all.insert(all.end(), v.begin(), v.end()); return o; };
    t->o_lat = put(b->lat); t->o_latrad = put(latrad); t->o_sphi2 = put(sphi2); t->o_cphi2 = put(cphi2); t->o_cphi = put(cphi);
    t->o_wlat = put(wlat); t->o_lonrad = put(lonrad); t->o_slam2 = put(slam2); t->o_clam2 = put(clam2); t->o_lonval = put(b->lon);
    t->base = (double*)dmalloc(ctx, all.size() * sizeof(double));
    {
        // on[k] of the streamer index (streamer_index.py:130-133) is the distance between two consecutive contour points, i.e.
        // between a grid point and one of its eight neighbours (a contour jumps further only where duplicate points were
        // dropped).  check_groups compares SUMS of them that agree to the last bits, so every on[k] has to be the value the
        // reference's math library gives -- the device's sin / cos / asin differ from it by an ulp here and there.  All
        // unit steps, evaluated here in sklearn's operation order: 4 per grid point (the formula is symmetric in its points).
        const size_t cells = (size_t)p.nlat * p.nlon;
        std::vector<double> tab(cells * 4);
        const int nth = 4;
        std::vector<std::thread> th;
        for (int w = 0; w < nth; ++w)
            th.emplace_back([&, w]() {
                for (int y = w; y < p.nlat; y += nth)
                    for (int x = 0; x < p.nlon; ++x) {
                        double* o = &tab[((size_t)y * p.nlon + x) * 4];
                        const int xe = (x + 1) % p.nlon, xw = (x + p.nlon - 1) % p.nlon, yn = y + 1 < p.nlat ? y + 1 : y;
                        o[0] = wb_haversine(latrad[y], lonrad[x], latrad[y], lonrad[xe]) * 6371.0;
                        o[1] = wb_haversine(latrad[y], lonrad[x], latrad[yn], lonrad[x]) * 6371.0;
                        o[2] = wb_haversine(latrad[y], lonrad[x], latrad[yn], lonrad[xe]) * 6371.0;
                        o[3] = wb_haversine(latrad[y], lonrad[x], latrad[yn], lonrad[xw]) * 6371.0;
                    }
            });
        for (auto& q : th) q.join();
        t->on_tab = (double*)dmalloc(ctx, tab.size() * sizeof(double));
        if (t->on_tab) h2d(ctx, t->on_tab, tab.data(), tab.size() * sizeof(double));
    }
    // streamer thresholds
    { const double sv = sin(p.geo_dis / (2.0 * 6371.0)); t->hthr = sv * sv; }
    t->dy_max = (int)floor(p.geo_dis / (6371.0 * p.dlat * (M_PI / 180.0))) + 1;
    // per latitude row: the largest |dx| (index units) at which a partner within dy_max rows can still be closer than
    // geo_dis -- haversine >= cos(phi_i) cos(phi_j) sin^2(dlambda / 2) -- capped by the reference's 120 (streamer_index.py:157)
    std::vector<int> dxl(p.nlat, 120);
    const double dlr = p.dlon * (M_PI / 180.0);
    if (120.0 * p.dlon <= 180.0) {                               // sin^2 is monotonic over the whole window
        for (int y = 0; y < p.nlat; ++y) {
            // the partner row that allows most has the smallest cosine: one of the two ends of the window
            const int ya = y - t->dy_max < 0 ? 0 : y - t->dy_max, yb = y + t->dy_max > p.nlat - 1 ? p.nlat - 1 : y + t->dy_max;
            const double cm = cphi[ya] < cphi[yb] ? cphi[ya] : cphi[yb];
            const double den = cphi[y] * cm;
            if (!(den > 0.0)) continue;
            const double ratio = t->hthr * (1.0 + 1e-9) / den;
            if (ratio >= 1.0) continue;
            const double lim = floor(2.0 * asin(sqrt(ratio)) / dlr) + 2.0;
            if (lim < 120.0) dxl[y] = (int)lim;
        }
    }
    {
        const double dlatr = p.dlat * (M_PI / 180.0);
        std::vector<double> sl((size_t)t->dy_max + 2), sx(WB_P2_MAXDX + 2);
        for (size_t k = 0; k < sl.size(); ++k) { const double v = sin(0.5 * (double)k * dlatr); sl[k] = v * v; }
        for (size_t k = 0; k < sx.size(); ++k) { const double v = sin(0.5 * (double)k * dlr); sx[k] = v * v; }
        t->sl_tab = (double*)dmalloc(ctx, sl.size() * sizeof(double));
        t->sx_tab = (double*)dmalloc(ctx, sx.size() * sizeof(double));
        if (t->sl_tab && t->sx_tab) { h2d(ctx, t->sl_tab, sl.data(), sl.size() * sizeof(double)); h2d(ctx, t->sx_tab, sx.data(), sx.size() * sizeof(double)); }
        double dev = 0.0;                                         // largest deviation of a coordinate from the regular grid, radians
        for (int y = 0; y < p.nlat; ++y) dev = std::max(dev, fabs(b->lat[y] - (b->lat[0] + (double)y * p.dlat)) * (M_PI / 180.0));
        for (int x = 0; x < p.nlon; ++x) dev = std::max(dev, fabs(b->lon[x] - (b->lon[0] + (double)x * p.dlon)) * (M_PI / 180.0));
        // |dh / d(coordinate)| <= 1.5 and four coordinates enter a pair
        t->geo_band = std::max(WB_GEO_BAND * t->hthr, 8.0 * dev);
        // lon[x % nlon] continues the grid across the seam only when the grid closes on itself; a band wider than 1e-4 of the
        // threshold would send many pairs to the exact evaluation: such grids stay with the tiled kernel
        t->p2_ok = t->sl_tab && t->sx_tab && fabs((double)p.nlon * p.dlon - 360.0) <= 1e-9 && t->geo_band <= 1e-4 * t->hthr &&
                   (size_t)p.nlat <= 8192;
    }
    t->dxlim = (int*)dmalloc(ctx, p.nlat * sizeof(int));
    if (!t->base || !t->dxlim) { delete t; return nullptr; }
    h2d(ctx, t->base, all.data(), all.size() * sizeof(double));
    h2d(ctx, t->dxlim, dxl.data(), p.nlat * sizeof(int));
    ctx->tables.push_back(t);
    return t;
}

// Phase 2 without the host in the middle: work lists, offsets and counts live on the device (WbCtl), launch sizes and
// allocations come from the phase-1 bounds, capacities that turn out too small are reported through the control block
// and the whole phase is repeated with room to spare (rare).  The host waits twice: for the first control-block snapshot
// (event and vertex counts -> sizes of the result copies) while the property kernel is still running, and for the end.

// ---- what the last wb_indices left unfinished on purpose
// the geometry kernels nobody has waited for yet (context option geometry_async): wait, judge the pass from the totals they left in
// pinned memory, free their scratch.  An overflow of the pinned arrays -- they are sized from the largest pass the context has
// seen -- or an event with more cut edges than the kernel holds leaves g_rings = -1: wb_get_event_geometry then runs the host
// function (wb_geometry.h), as it does with the option "geometry" off.
static int wb_geom_settle(wb_batch* b) {
    if (!b->g_pending) return 0;
    wb_ctx* ctx = b->ctx;
    b->g_pending = false;
#if !defined(WB_HOSTSIM)
    if (b->g_event && hipEventSynchronize(b->g_event) != hipSuccess) return wb_fail(ctx, "wb_indices: the geometry kernels failed");
#endif
    bfree_one(b, b->g_cnt_r); bfree_one(b, b->g_cnt_v); bfree_one(b, b->g_flag);
    b->g_cnt_r = b->g_cnt_v = b->g_flag = nullptr;
    const long long NR = b->g_tot[0], NV = b->g_tot[1], over = b->g_tot[2], flagged = b->g_tot[3];
    int tail_rc = 0;
#if !defined(WB_HOSTSIM)
    if (b->g_dx && NV > b->g_copied && !over && !flagged && NV <= b->g_cap_v) {
        // the DMA copy behind the kernels covered as many vertices as the context's largest pass so far held: the rest now
        std::lock_guard<std::mutex> lk(ctx->settle_mu);
        if (!ctx->settle_stream && hipStreamCreateWithFlags(&ctx->settle_stream, hipStreamNonBlocking) != hipSuccess) { ctx->settle_stream = nullptr; tail_rc = 1; }
        const size_t o = (size_t)b->g_copied, m = (size_t)(NV - b->g_copied) * sizeof(double);
        if (!tail_rc) tail_rc = hipMemcpyAsync(b->g_x + o, b->g_dx + o, m, hipMemcpyDeviceToHost, ctx->settle_stream) != hipSuccess ||
                                hipMemcpyAsync(b->g_y + o, b->g_dy + o, m, hipMemcpyDeviceToHost, ctx->settle_stream) != hipSuccess ||
                                hipStreamSynchronize(ctx->settle_stream) != hipSuccess;
    }
#endif
    // (a time-sharded run gathers the rings device to device: with keep_device they stay until the batch's next wb_indices / its end)
    if (b->g_dx && !ctx->keep_device) { bfree_one(b, b->g_dx); bfree_one(b, b->g_dy); bfree_one(b, b->g_dro); bfree_one(b, b->g_dvo); b->g_dx = b->g_dy = nullptr; b->g_dro = b->g_dvo = nullptr; }
    if (tail_rc) return wb_fail(ctx, "wb_indices: the copy of the geometry's tail failed");
    if (NR < 0 || NV < 0) return wb_fail(ctx, "wb_indices: the geometry kernels left no totals");
    if (NV > ctx->geom_seen) ctx->geom_seen = NV;
    if (NR > ctx->geom_rings_seen) ctx->geom_rings_seen = NR;
    if (over || flagged || NR > b->g_cap_r || NV > b->g_cap_v) {
        b->g_rings = -1;                                  // the host function, on demand
        for (auto& hb : b->gres) hfree(ctx, hb.first, hb.second);
        b->gres.clear();
        b->g_ring_offset = b->g_vert_offset = nullptr; b->g_x = b->g_y = nullptr; b->g_tot = nullptr;
        return 0;
    }
    b->g_rings = NR; b->g_verts = NV;
    return 0;
}
// the events' polygons in grid indices, packed into pinned host memory now (context option index_polygons = 0: wb_indices left
// them on the device as descriptors + contours).  Launches on the context's stream and waits: the context's lock is taken.
static int wb_pack_polys_now(wb_batch* b) {
    if (b->poly_x || !b->d_ev || !b->d_poff) return 0;
    wb_ctx* ctx = b->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!wb_select_device(ctx)) return wb_fail(ctx, "hipSetDevice failed");
    b->poly_x = (int32_t*)hmalloc(ctx, (b->poly_n + 1) * sizeof(int32_t), &b->poly_bytes);
    b->poly_y = (int32_t*)hmalloc(ctx, (b->poly_n + 1) * sizeof(int32_t), &b->poly_bytes);
    if (!b->poly_x || !b->poly_y) return wb_fail(ctx, "out of host memory");
    WB_LAUNCH(wb_k_pack_polys, ctx->polys_grid, 256, b->d_ctl, b->d_ev_cap_seg, b->d_ev, b->d_poff, b->c_off, b->cpts, b->poly_x, b->poly_y, (int*)nullptr, (int*)nullptr);
    return dsync(ctx);
}

extern "C" int wb_indices(wb_batch* b, int gmax, int which) {
    if (!b) return 1;
    wb_ctx* ctx = b->ctx;
    const wb_params& p = b->p;
    const int J = b->njobs, cap = b->cap_seg;
    std::lock_guard<std::recursive_mutex> wb_lk_(ctx->mu);
    if (!wb_select_device(ctx)) return wb_fail(ctx, "hipSetDevice failed");
    // what an earlier call on this batch left behind: geometry kernels still in flight, descriptors kept for them / for the polygons
    if (wb_geom_settle(b)) return 1;
    if (b->d_ev) { bfree_one(b, b->d_ev); bfree_one(b, b->d_poff); b->d_ev = nullptr; b->d_poff = nullptr; }
    if (b->g_dx) { bfree_one(b, b->g_dx); bfree_one(b, b->g_dy); bfree_one(b, b->g_dro); bfree_one(b, b->g_dvo); b->g_dx = b->g_dy = nullptr; b->g_dro = b->g_dvo = nullptr; }
    wb_ctx::Tables* T = make_tables(b);
    if (!T) return wb_fail(ctx, "out of memory (tables)");
    const WbCtl& h1 = b->h_ctl;                 // phase-1 summary
    // bounds: the selected contours are those with nux == gmax; with gmax equal to this batch's maximum they are counted
    // exactly by the phase-1 summary, with a larger one (another rank holds the full-span contour) there are none
    const bool at_max = gmax == h1.max_nux, above = gmax > h1.max_nux;
    const int nsel_b = at_max ? h1.nsel_max : (above ? 0 : h1.ncont_total);
    const long long npt_b = at_max ? h1.npt_sel_max : (above ? 0 : h1.npts_total);
    const int items_b = at_max ? h1.pair_items_max : (above ? 0 : (int)(h1.npts_total / (WB_PAIR_ROWS * WB_PAIR_TPB)) + h1.ncont_total);
    const int ncut_b = (which & 4) ? h1.nclosed : 0;
    const long long npt_cut_b = (which & 4) ? h1.npt_closed : 0;
    const bool do_st = (which & 1) && nsel_b > 0, do_ot = (which & 2) && nsel_b > 0;
    const int cap_ev = 256;
    int cap_pairs = p.cap_pairs > 0 ? p.cap_pairs : 65536;
    int cap_ne = 16 * (b->job_step.empty() ? J : 64) + 4096 + ncut_b;
    long long cap_poly = 2 * npt_b + npt_cut_b + 4LL * cap_ne + 1024;
    unsigned long long cell_cap = p.want_cells ? (unsigned long long)cap_ne * 1024ull : 0ull;
    const double r = ((p.dlon + p.dlat) / 2.0) / 2.0;     // index_utils.py:48-50

    // result buffers of the previous call
    hfree(ctx, b->poly_x, b->poly_bytes);
    hfree(ctx, b->poly_y, b->poly_bytes);
    b->poly_x = b->poly_y = nullptr;
    b->poly_n = b->poly_bytes = 0;
    for (auto& hb : b->hres) hfree(ctx, hb.first, hb.second);
    b->hres.clear();
    for (auto& hb : b->gres) hfree(ctx, hb.first, hb.second);
    b->gres.clear();
    b->g_rings = -1;
    long long cap_gv = std::min<long long>(cap_poly + cap_poly / 8 + 1024, std::max<long long>(ctx->geom_seen + ctx->geom_seen / 4 + 4096, 1LL << 21));
    long long cap_gr = std::max<long long>((long long)cap_ne + cap_ne / 4 + 1024, ctx->geom_rings_seen + ctx->geom_rings_seen / 4 + 1024);

    WbCtl fin;
    memset(&fin, 0, sizeof(fin));
    bool p2_off = false;
    double* h_props = nullptr;
    int* h_ev = nullptr; double* h_sums = nullptr; int* h_nc = nullptr; int* h_fl = nullptr; long long* h_poff = nullptr; long long* h_coff = nullptr;
    uint32_t* h_cells = nullptr;
    for (int attempt = 0; attempt < 6; ++attempt) {
        std::vector<void*> tmp;   // device scratch of this attempt
        auto talloc = [&](size_t bytes) { void* q = dmalloc(ctx, bytes); if (q) { b->owned.push_back(q); tmp.push_back(q); } return q; };
        auto release = [&]() { for (void* q : tmp) bfree_one(b, q); tmp.clear(); };
        b->timer.truncate(b->timer_phase1);
        WbCtl* ctl = b->d_ctl;
        // the phase-2 part of the control block starts from zero (phase-1 values stay): cleared by wb_k_select_count's first workgroup
#if defined(WB_STAMPS) && (WB_STAMPS == 4 || WB_STAMPS == 6 || WB_STAMPS == 7 || WB_STAMPS == 8)              // (the stamps of a phase-1 kernel survive into the snapshot)
        const size_t zero_bytes = offsetof(WbCtl, dbg) - offsetof(WbCtl, nsel);
#else
        const size_t zero_bytes = sizeof(WbCtl) - offsetof(WbCtl, nsel);
#endif
        static_assert(offsetof(WbCtl, nsel) % 4 == 0 && sizeof(WbCtl) % 4 == 0 && offsetof(WbCtl, dbg) % 4 == 0, "WbCtl is cleared in 32-bit words");
        if (J <= 0) dmemset(ctx, (char*)ctl + offsetof(WbCtl, nsel), 0, zero_bytes);

        // ---- selection
        WbSelectArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.njobs = J; sa.cap_seg = cap; sa.gmax = gmax; sa.want_cut = (which & 4) ? 1 : 0; sa.dlon = p.dlon; sa.co_min_exp = p.co_min_exp;
        sa.ncont = b->ncont; sa.c_nux = b->c_nux; sa.c_closed = b->c_closed;
        sa.zero_at = (uint32_t*)((char*)ctl + offsetof(WbCtl, nsel)); sa.zero_words = (int)(zero_bytes / 4);
        int* d_nsel = (int*)talloc(J * sizeof(int));
        int* d_ncut = (int*)talloc(J * sizeof(int));
        b->sel_job = (int*)talloc((nsel_b + 1) * sizeof(int)); b->sel_cont = (int*)talloc((nsel_b + 1) * sizeof(int));
        b->cut_job = (int*)talloc((ncut_b + 1) * sizeof(int)); b->cut_cont = (int*)talloc((ncut_b + 1) * sizeof(int));
        if (!b->cut_cont || !d_ncut) { release(); return 1; }
        sa.nsel = d_nsel; sa.ncut = d_ncut;
        b->timer.begin(ctx, "select");
        WB_LAUNCH(wb_k_select_count, J, 64, sa);
        WB_LAUNCH(wb_k_scan_jobs, 1, 256, J, d_nsel, d_ncut, &ctl->nsel);      // totals -> ctl->nsel, ctl->ncut
        sa.selbase = d_nsel; sa.cutbase = d_ncut;
        sa.sel_job = b->sel_job; sa.sel_cont = b->sel_cont; sa.cut_job = b->cut_job; sa.cut_cont = b->cut_cont;
        WB_LAUNCH(wb_k_select_fill, J, 64, sa);
        b->timer.end(ctx);

        // ---- overturnings + the centre-of-mass sums of their boxes: side stream, next to the streamer chain
        int *ot_n = nullptr, *ot_box = nullptr, *ot_orient = nullptr;
        double* ot_sums = nullptr;

        WbBoxComArgs ba;
        long long boxcom_grid = 0;
        if (do_ot) {
            ot_n = (int*)talloc(nsel_b * sizeof(int));
            ot_box = (int*)talloc((size_t)nsel_b * WB_OT_CAP * 4 * sizeof(int));
            ot_orient = (int*)talloc((size_t)nsel_b * WB_OT_CAP * sizeof(int));
            ot_sums = (double*)talloc((size_t)nsel_b * WB_OT_CAP * 3 * sizeof(double));
            int* boxq = (int*)talloc((size_t)nsel_b * WB_OT_CAP * sizeof(int));
            if (!ot_sums || !boxq) { release(); return 1; }
            const bool box_sums = r * r <= 1.0;        // (a wider rim: the property kernel rasterises the boxes like any polygon)
            WbOverturnArgs oa;
            memset(&oa, 0, sizeof(oa));
            oa.ctl = ctl; oa.cap_seg = cap; oa.nlon = p.nlon; oa.W = b->W;
            oa.rg_over_dlon = p.range_group / p.dlon; oa.minexp_over_dlon = p.ot_min_exp / p.dlon;
            oa.sel_job = b->sel_job; oa.sel_cont = b->sel_cont; oa.c_off = b->c_off; oa.c_len = b->c_len; oa.cpts = b->cpts;
            oa.lat = T->base + T->o_lat; oa.ot_n = ot_n; oa.ot_box = ot_box; oa.ot_orient = ot_orient; oa.status = b->status;
            oa.nlat = p.nlat; oa.wlat = box_sums ? T->base + T->o_wlat : nullptr; oa.ot_sums = ot_sums; oa.boxq = boxq;
            side_begin(ctx);
            b->timer.begin(ctx, "overturn");
            WB_LAUNCH_LDS(wb_k_overturn, nsel_b, 256, (size_t)b->W * 8 + 32, oa);
            b->timer.end(ctx);
            // the boxes wb_k_overturn queued (their truncated centre of mass could turn on the last bits of the reference's
            // cell-by-cell sums): a persistent grid of single-wave workgroups takes them one at a time.  WHEN it runs is an
            // option ("boxcom_after"): its twelve workgroups per CU hold 156 KB of a CU's LDS for half a millisecond, and right
            // behind wb_k_overturn that is beside wb_k_streamer_prep, which then waits for LDS (0.40 ms instead of 0.13)
            if (box_sums) {
                memset(&ba, 0, sizeof(ba));
                ba.nlat = p.nlat; ba.W = b->W; ba.ctl = ctl; ba.boxq = boxq; ba.ot_box = ot_box; ba.wlat = T->base + T->o_wlat;
                ba.ot_sums = ot_sums;
                boxcom_grid = (long long)ctx->boxcom_per_cu * wb_num_cus(ctx);
                const long long most = (long long)nsel_b * 4;
                if (boxcom_grid > most) boxcom_grid = most;
                if (boxcom_grid < 1) boxcom_grid = 1;
                if (ctx->boxcom_after == 0 || !do_st) {
                    b->timer.begin(ctx, "box_com");
                    WB_LAUNCH(wb_k_box_com, boxcom_grid, 64, ba);
                    b->timer.end(ctx);
                    boxcom_grid = 0;
                }
            }
            side_end(ctx);
        }
        // (option boxcom_after = 1 / 2: on the side stream behind wb_k_streamer_prep / wb_k_streamer_pairs2 of the main stream)
        auto boxcom_here = [&](int stage) {
            if (boxcom_grid <= 0 || ctx->boxcom_after != stage) return;
            side_begin(ctx);
            const bool masked = boxmask_begin(ctx);
            b->timer.begin(ctx, "box_com");
            WB_LAUNCH(wb_k_box_com, boxcom_grid, 64, ba);
            b->timer.end(ctx);
            if (masked) boxmask_end(ctx);
            side_end(ctx);
            boxcom_grid = 0;
        };

        // ---- streamers
        int *ev_n = nullptr, *ev_ind = nullptr;
        WbStreamerArgs st;
        memset(&st, 0, sizeof(st));
        if (do_st) {
            ev_n = (int*)talloc(nsel_b * sizeof(int));
            ev_ind = (int*)talloc((size_t)nsel_b * cap_ev * 2 * sizeof(int));
            st.ctl = ctl; st.cap_seg = cap; st.nlon = p.nlon; st.nlat = p.nlat; st.W = b->W; st.cap_ev = cap_ev;
            st.geo_dis = p.geo_dis; st.cont_dis = p.cont_dis; st.hthr = T->hthr; st.dy_max = T->dy_max; st.dxlim = T->dxlim;
            st.sel_job = b->sel_job; st.sel_cont = b->sel_cont; st.c_off = b->c_off; st.c_len = b->c_len; st.cpts = b->cpts;
            st.on_tab = T->on_tab;
            st.lat_rad = T->base + T->o_latrad; st.lon_rad = T->base + T->o_lonrad; st.sphi2 = T->base + T->o_sphi2;
            st.cphi2 = T->base + T->o_cphi2; st.cphi = T->base + T->o_cphi; st.slam2 = T->base + T->o_slam2; st.clam2 = T->base + T->o_clam2;
            const long long NPT = npt_b;
            st.npts_total = NPT + 1;
            st.poff = (long long*)talloc((nsel_b + 1) * sizeof(long long));
            st.item_off = (int*)talloc((nsel_b + 1) * sizeof(int));
            st.item_sel = (int*)talloc((size_t)(items_b + 1) * sizeof(int));
            st.item_tile = (int*)talloc((size_t)(items_b + 1) * sizeof(int));
            st.f2_off = (int*)talloc((nsel_b + 1) * sizeof(int));
            int* d_lens = (int*)talloc((nsel_b + 1) * sizeof(int));
            st.on = (double*)talloc((size_t)(NPT + 1) * sizeof(double));
            st.pre = (double*)talloc((size_t)(NPT + 1) * sizeof(double));
            // the all-pairs test: contours that fit the LDS of wb_k_streamer_pairs2 go there (regular global grids), longer
            // ones -- and every contour of other grids -- through the tiled kernel and its per-point table
            size_t p2_lds = 0;
            st.sl_tab = T->sl_tab; st.sx_tab = T->sx_tab; st.geo_band = T->geo_band;
            if (ctx->pairs_lds && T->p2_ok && !p2_off) {
                const size_t budget = wb_p2_lds_budget(ctx);
                int fit = 0;
                while (wb_p2_lds_bytes(fit + 256, p.nlat, T->dy_max) <= budget) fit += 256;
                int want = ((h1.max_len + 255) / 256) * 256;
                if (want < 512) want = 512;
                st.p2_cap = want < fit ? want : fit;
                if (st.p2_cap > 0) p2_lds = wb_p2_lds_bytes(st.p2_cap, p.nlat, T->dy_max);
                st.p2_split = ctx->p2_split > 0 ? ctx->p2_split : (nsel_b >= 1024 ? 1 : (nsel_b >= 256 ? 2 : (nsel_b >= 64 ? 4 : 8)));
                if (wb_debug()) fprintf(stderr, "[wb] pairs2: longest contour %d, capacity %d points, %zu B of LDS, %d workgroups per contour\n", h1.max_len, st.p2_cap, p2_lds, st.p2_split);
            }
            if (st.p2_cap > 0) {
                st.recheck_cap = 65536;
                st.recheck = (int*)talloc((size_t)st.recheck_cap * 4 * sizeof(int));
                if (!st.recheck) { release(); return 1; }
            }
            st.f234_cap = ((h1.max_len + 255) / 256) * 256;
            if (st.f234_cap < 512) st.f234_cap = 512;
            if (st.f234_cap > WB_F234_LPTS) st.f234_cap = WB_F234_LPTS;
            if (ctx->f234_lpts > 0 && st.f234_cap > ctx->f234_lpts) st.f234_cap = ctx->f234_lpts;
            const bool need_tiled = st.p2_cap <= 0 || h1.max_len > st.p2_cap;
            st.ptab = need_tiled ? (double*)talloc((size_t)(NPT + 1) * 5 * sizeof(double)) : nullptr;
            if (need_tiled && !st.ptab) { release(); return 1; }
            st.bbx = (int*)talloc((size_t)((NPT >> 5) + nsel_b + 1) * 4 * sizeof(int));
            st.tbx = (int*)talloc((size_t)((NPT >> 4) + nsel_b + 1) * 4 * sizeof(int));
            st.npairs = (int*)talloc(nsel_b * sizeof(int));
            st.stage = (int*)talloc(nsel_b * sizeof(int));
            st.nalive = (int*)talloc(nsel_b * sizeof(int));
            st.tie_m = (int*)talloc(nsel_b * sizeof(int));
            st.eager = ctx->eager_f2 ? 1 : 0;
            st.f1_fused = (ctx->f1_fused && !ctx->eager_f2) ? 1 : 0;
            st.rowmax = (int*)talloc((size_t)(NPT + 1) * sizeof(int));
            st.rowaux = (int*)talloc((size_t)(NPT + 1) * sizeof(int));
            st.rowmx = (int*)talloc((size_t)(NPT + 1) * sizeof(int));
            st.ev_ind = ev_ind; st.ev_n = ev_n; st.status = b->status;
            st.cap_pairs = cap_pairs;
            st.hash_size = next_pow2(2LL * cap_pairs);
            // (room for 4 words per point of the longest contour: a traced job's stride bounds it, a frame's job has its own)
            { const long long lenb = std::max((long long)cap, (long long)h1.max_len / 2 + 1);
              if (st.hash_size < next_pow2(4LL * lenb)) st.hash_size = next_pow2(4LL * lenb); }
            st.pairs = (unsigned long long*)talloc((size_t)nsel_b * cap_pairs * 8);
            st.alive = (int*)talloc((size_t)nsel_b * cap_pairs * sizeof(int));
            st.hkey = (unsigned long long*)talloc((size_t)nsel_b * st.hash_size * 8);
            st.hval = (unsigned long long*)talloc((size_t)nsel_b * st.hash_size * 8);
            st.order = (int*)talloc((size_t)(nsel_b + 1) * sizeof(int));
            if (!st.order) { release(); return 1; }
            if (!st.hval || !st.rowaux || !st.rowmx || !ev_ind || !st.bbx || !st.tbx || !d_lens || !st.item_tile) { release(); return 1; }
            (void)p2_lds;
            b->timer.begin(ctx, "streamer_lists");
            // (single workgroups of 256 threads: a 1024-thread workgroup waits for sixteen free wave slots on one CU, and beside the
            //  side stream's 256-thread workgroups, which take every slot as it frees up, it waited 0.25 ms for them)
            WB_LAUNCH(wb_k_sel_scan, 1, WB_LIST_THREADS, st, d_lens);
            WB_LAUNCH(wb_k_pair_items, nsel_b, 64, st);
            b->timer.end(ctx);
            b->timer.begin(ctx, "streamer_prep");
            WB_LAUNCH(wb_k_streamer_prep, nsel_b, 256, st);
            b->timer.end(ctx);
            boxcom_here(1);
            b->timer.begin(ctx, "streamer_pairs");
            if (st.p2_cap > 0) {
                { long long g2 = (long long)nsel_b * st.p2_split, res = (long long)ctx->p2_grid_x2 * wb_num_cus(ctx) / 2 - ctx->persist_slack;   // persistent: two workgroups of 16 waves per CU (minus option persist_slack)
                  if (res < 1) res = 1;
                  if (res < 1) res = 1;
                  WB_LAUNCH_LDS(wb_k_streamer_pairs2, g2 < res ? g2 : res, WB_P2_THREADS, p2_lds, st); }
                WB_LAUNCH(wb_k_streamer_recheck, 64, 64, st);              // pairs inside a guard band (normally none)
            }
            if (need_tiled && items_b > 0) WB_LAUNCH(wb_k_streamer_pairs, items_b, 64, st);
            b->timer.end(ctx);
            boxcom_here(2);
            b->timer.begin(ctx, "streamer_f1");
            WB_LAUNCH(wb_k_pairs_order, 1, WB_LIST_THREADS, st);
            // routine 1: inside wb_k_streamer_f234 (option f1_fused, the default), or its own persistent launch in front of the eager kernels
            if (!st.f1_fused) { long long res = (2048LL / WB_F1_THREADS) * wb_num_cus(ctx) * ctx->f1_grid_x2 / 4 - ctx->persist_slack; if (res < 1) res = 1; WB_LAUNCH(wb_k_streamer_f1, nsel_b < res ? nsel_b : res, WB_F1_THREADS, st); }
            b->timer.end(ctx);
            if (ctx->eager_f2) {
                // routine 2 for every candidate pair, then routines 3 + 4 on the survivors (option "eager_f2": the exhaustive
                // formulation, kept as a cross-check of the lazy one)
                b->timer.begin(ctx, "streamer_f2");
                WB_LAUNCH(wb_k_f2_items, 1, 1024, st);
                long long g2 = (long long)nsel_b * ((cap_pairs + WB_F2_THREADS - 1) / WB_F2_THREADS);
                if (g2 > 16384) g2 = 16384;                         // persistent: the workgroups stride over the device-built list
                WB_LAUNCH(wb_k_streamer_f2, g2, 256, st);
                b->timer.end(ctx);
                b->timer.begin(ctx, "streamer_f34");
                WB_LAUNCH(wb_k_streamer_f34, nsel_b, 256, st);
                WB_LAUNCH(wb_k_streamer_ties, 1, 64, st);
                b->timer.end(ctx);
            } else {
                b->timer.begin(ctx, "streamer_f234");
                WB_LAUNCH_LDS(wb_k_streamer_f234, nsel_b, WB_F234_THREADS, (size_t)st.f234_cap * 4 + (size_t)(((st.f234_cap + 1) & ~1) + (st.f234_cap > WB_F234_SCR ? ((st.f234_cap + 1) & ~1) : WB_F234_SCR)) * 4 + (size_t)(st.f234_cap / WB_BLK + 1) * 16 + 64, st);
                WB_LAUNCH(wb_k_streamer_ties, 1, 64, st);      // one wave: check_groups' exact ties (nearly always none)
                b->timer.end(ctx);
            }
            if (getenv("WB_DEBUG_SEL")) {      // development: the selected contours' lengths and candidate pairs (largest first)
                dsync(ctx);
                std::vector<int> hl(nsel_b), hp(nsel_b), ha(nsel_b);
                std::vector<long long> hoff(nsel_b + 1);
                d2h(ctx, hoff.data(), st.poff, (nsel_b + 1) * sizeof(long long));
                for (int q = 0; q < nsel_b; ++q) hl[q] = (int)(hoff[q + 1] - hoff[q]);
                d2h(ctx, hp.data(), st.npairs, nsel_b * sizeof(int));
                d2h(ctx, ha.data(), st.nalive, nsel_b * sizeof(int));
                std::vector<int> idx(nsel_b);
                for (int q = 0; q < nsel_b; ++q) idx[q] = q;
                std::sort(idx.begin(), idx.end(), [&](int x, int y) { return hp[x] > hp[y]; });
                fprintf(stderr, "[wb] selected contours %d, longest %d points, f234 holds %d; (points, pairs, alive) by pairs:", nsel_b, h1.max_len, st.f234_cap);
                for (int q = 0; q < nsel_b && q < 40; ++q) fprintf(stderr, " (%d %d %d)", hl[idx[q]], hp[idx[q]], ha[idx[q]]);
                long long over = 0, npair_over = 0; int nbig = 0;
                for (int q = 0; q < nsel_b; ++q) { if (hl[q] > st.f234_cap) { ++over; npair_over += hp[q]; } if (hp[q] > 4000) ++nbig; }
                fprintf(stderr, "\n[wb] contours longer than the f234 capacity: %lld (%lld pairs); with more than 4000 pairs: %d\n", over, npair_over, nbig);
            }
        }
        if (do_ot) side_join(ctx);

        // ---- event list (streamers, overturnings, cutoffs): offsets, descriptors, polygon offsets
        WbEvFillArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.cap_ev = cap_ev; fa.cap_ne = cap_ne; fa.cap_seg = cap; fa.cap_poly = cap_poly; fa.ctl = ctl;
        fa.sel_job = b->sel_job; fa.sel_cont = b->sel_cont; fa.cut_job = b->cut_job; fa.cut_cont = b->cut_cont;
        fa.ev_n = ev_n; fa.ev_ind = ev_ind; fa.ot_n = ot_n; fa.ot_box = ot_box; fa.ot_orient = ot_orient; fa.c_len = b->c_len;
        fa.sbase = (int*)talloc((nsel_b + 1) * sizeof(int));
        fa.obase = (int*)talloc((nsel_b + 1) * sizeof(int));
        fa.ev = (int*)talloc((size_t)(cap_ne + 1) * 8 * sizeof(int));
        fa.plen = (int*)talloc((size_t)(cap_ne + 1) * sizeof(int));
        fa.poff = (long long*)talloc((size_t)(cap_ne + 1) * sizeof(long long));
        WbPropsArgs pa;
        memset(&pa, 0, sizeof(pa));
        // one wave per event slot, fed from the event queue; 4096 slots (4 waves per SIMD) for a year-sized batch, fewer for a
        // few jobs: a slot owns W x nlat cells of scratch (5.5 MB at 0.25 degree)
        pa.ctl = ctl; pa.nslots = 8 * J < 4096 ? (8 * J > 64 ? 8 * J : 64) : 4096;
        pa.nlat = p.nlat; pa.nlon = p.nlon; pa.W = b->W; pa.cap_seg = cap;
        pa.nlev = p.nlevels; pa.dtype = p.dtype; pa.want_cells = p.want_cells; pa.pad = (int)ceil(r); pa.r2 = r * r;
        pa.ev = fa.ev; pa.ot_sums = ot_sums; pa.c_off = b->c_off; pa.c_len = b->c_len; pa.cpts = b->cpts; pa.data = b->d_field;
        pa.intensity = b->d_intensity; pa.wlat = T->base + T->o_wlat; pa.job_step = b->d_job_step;
        if (cap_ne < pa.nslots) pa.nslots = cap_ne > 0 ? cap_ne : 1;
        pa.cells = (int*)talloc((size_t)pa.nslots * b->W * p.nlat * sizeof(int));
        pa.rowacc = (double*)talloc((size_t)pa.nslots * 4 * p.nlat * sizeof(double));
        pa.ev_sums = (double*)talloc((size_t)(cap_ne + 1) * 5 * sizeof(double));
        pa.ev_ncells = (int*)talloc((size_t)(cap_ne + 1) * sizeof(int));
        pa.ev_flags = (int*)talloc((size_t)(cap_ne + 1) * sizeof(int));
        pa.ev_cell_off = (long long*)talloc((size_t)(cap_ne + 1) * sizeof(long long));
        pa.cell_cap = cell_cap;
        pa.cell_pool = cell_cap ? (uint32_t*)talloc(cell_cap * sizeof(uint32_t)) : nullptr;
        if (!fa.poff || !pa.ev_flags || !pa.ev_cell_off || !pa.rowacc || !pa.cells || (cell_cap && !pa.cell_pool)) { release(); return 1; }
        b->timer.begin(ctx, "fill_events");
        WB_LAUNCH(wb_k_event_offsets, 1, WB_LIST_THREADS, fa);
        if (nsel_b + ncut_b > 0) WB_LAUNCH(wb_k_fill_events, nsel_b + ncut_b, 64, fa);
        b->timer.end(ctx);
        // The events' descriptors stand: the property kernel goes out AT ONCE, on the main stream.  The polygons' offsets (one workgroup
        // scanning 19 000 lengths: 35 us), the buffers of the results and the three geometry kernels are launched behind it from the
        // point marked here, on the side stream -- until round 6 wb_k_props waited for the offsets it does not read and for the host
        // to have set all that up (70 us of a pass one batch at a time).
        side_mark(ctx);
        b->timer.begin(ctx, "props");
        WB_LAUNCH(wb_k_props, pa.nslots, 64, pa);
        b->timer.end(ctx);
        // descriptors are complete: the vertices go out on the side stream (straight into the batch's pinned host arrays,
        // sized for the capacity -- the counts are only known on the device), properties on the main one; then ONE kernel
        // writes the control block and the event tables into a pinned host block, and the host waits ONCE
        size_t got = 0;
        // index polygons: packed now into pinned host memory (context option index_polygons = 1, or the device copies a gather
        // wants), or left as descriptors + contours on the device until wb_get_events asks for them (the default)
        const bool polys_now = ctx->index_polygons != 0;
        if (polys_now) {
            b->poly_x = (int32_t*)hmalloc(ctx, (size_t)(cap_poly + 1) * sizeof(int32_t), &b->poly_bytes);
            b->poly_y = (int32_t*)hmalloc(ctx, (size_t)(cap_poly + 1) * sizeof(int32_t), &b->poly_bytes);
        }
        void* hblock = hmalloc(ctx, wb_result_block_bytes((size_t)cap_ne, p.want_cells != 0), &got);
        if (hblock) b->hres.push_back({hblock, got});
        if (!hblock || (polys_now && (!b->poly_x || !b->poly_y))) { release(); return wb_fail(ctx, "out of host memory"); }
        const WbResultBlock rb = wb_result_block(hblock, (size_t)cap_ne, p.want_cells != 0);
        int *d_px = nullptr, *d_py = nullptr;
        if (ctx->keep_device == 1) {                           // device copies of the vertices for the gather (keep_device = 2: rows and rings only)
            d_px = (int*)talloc((size_t)(cap_poly + 1) * sizeof(int));
            d_py = (int*)talloc((size_t)(cap_poly + 1) * sizeof(int));
            if (!d_py) { release(); return 1; }
        }
        // transform_polygons (index_utils.py:129-184) beside them: rings in lon / lat, cut at the last meridian, into pinned host
        // arrays (wb_geom.h; count -> offsets -> fill)
        WbGeomArgs ga;
        memset(&ga, 0, sizeof(ga));
        const bool do_geom = ctx->geometry != 0;
        const bool geom_async = do_geom && ctx->geometry_async != 0;
        if (do_geom) {
            if (cap_gr < (long long)cap_ne + 16) cap_gr = (long long)cap_ne + cap_ne / 4 + 1024;
            ga.nev = -1; ga.ctl = ctl; ga.ctl_out = ctl; ga.cap_seg = cap; ga.nlon = p.nlon; ga.index_space = 0;
            ga.ev = fa.ev; ga.poff = fa.poff; ga.c_off = b->c_off; ga.cpts = b->cpts;
            ga.lon = T->base + T->o_lonval; ga.lat = T->base + T->o_lat;
            ga.cnt_r = (int*)talloc((size_t)(cap_ne + 2) * sizeof(int)); ga.cnt_v = (int*)talloc((size_t)(cap_ne + 2) * sizeof(int));
            ga.gflag = (int*)talloc((size_t)(cap_ne + 2) * sizeof(int));
            ga.cap_r = cap_gr; ga.cap_v = cap_gv;
            size_t gg = 0;
            auto gbuf = [&](size_t bytes) -> void* { void* q = hmalloc(ctx, bytes ? bytes : 8, &gg); if (q) b->gres.push_back({q, gg}); return q; };
            ga.ring_offset = (long long*)gbuf((size_t)(cap_ne + 2 + 8) * sizeof(long long));      // (+ the pass's four totals behind the offsets)
            ga.vert_offset = (long long*)gbuf((size_t)(cap_gr + 2) * sizeof(long long));
            ga.x = (double*)gbuf((size_t)(cap_gv + 2) * sizeof(double));
            ga.y = (double*)gbuf((size_t)(cap_gv + 2) * sizeof(double));
            if (!ga.gflag || !ga.cnt_v || !ga.y || !ga.x || !ga.vert_offset || !ga.ring_offset) { release(); return wb_fail(ctx, "out of memory (event geometry)"); }
            ga.tot_out = ga.ring_offset + (cap_ne + 2) + 2;
            for (int q = 0; q < 4; ++q) ga.tot_out[q] = -1;
        }
        // context option geometry_dma: the kernels write device arrays of the same layout, a DMA copy behind them fills the pinned ones
        long long *h_ro = ga.ring_offset, *h_vo = ga.vert_offset;
        double *h_gx = ga.x, *h_gy = ga.y;
        long long* h_tot = ga.tot_out;
        bool geom_dma = false;
        long long n_copy = 0;
#if !defined(WB_HOSTSIM)
        if (do_geom && ctx->geometry_dma) {
            long long* d_ro = (long long*)talloc((size_t)(cap_ne + 2 + 8) * sizeof(long long));
            long long* d_vo = (long long*)talloc((size_t)(cap_gr + 2) * sizeof(long long));
            double* d_gx = (double*)talloc((size_t)(cap_gv + 2) * sizeof(double));
            double* d_gy = (double*)talloc((size_t)(cap_gv + 2) * sizeof(double));
            if (!d_ro || !d_vo || !d_gx || !d_gy) { release(); return wb_fail(ctx, "out of device memory (event geometry)"); }
            ga.ring_offset = d_ro; ga.vert_offset = d_vo; ga.x = d_gx; ga.y = d_gy; ga.tot_out = d_ro + (cap_ne + 2) + 2;
            geom_dma = true;
            n_copy = std::min<long long>(cap_gv, ctx->geom_seen > 0 ? ctx->geom_seen + ctx->geom_seen / 16 + 4096 : 0);
        }
#endif
        auto launch_geometry = [&]() {
            b->timer.begin(ctx, "geometry");
            WB_LAUNCH(wb_k_geom_count, ctx->polys_grid, WB_GEOM_THREADS, ga);
            WB_LAUNCH(wb_k_geom_offsets, 1, WB_LIST_THREADS, ga);
            WB_LAUNCH(wb_k_geom_fill, ctx->polys_grid, WB_GEOM_THREADS, ga);
            b->timer.end(ctx);
#if !defined(WB_HOSTSIM)
            if (geom_dma) {
                // (on the stream the kernels went to; the event tables' ring offsets with the pass's totals behind them, the rings'
                //  vertex offsets, and the first n_copy vertices -- all of them, unless this pass holds more than any before it)
                (void)hipMemcpyAsync(h_ro, ga.ring_offset, (size_t)(cap_ne + 2 + 8) * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream);
                (void)hipMemcpyAsync(h_vo, ga.vert_offset, (size_t)(cap_gr + 2) * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream);
                if (n_copy > 0) {
                    (void)hipMemcpyAsync(h_gx, ga.x, (size_t)n_copy * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
                    (void)hipMemcpyAsync(h_gy, ga.y, (size_t)n_copy * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
                }
            }
#endif
        };
        side_begin_marked(ctx);
        WB_LAUNCH(wb_k_poly_offsets, 1, WB_LIST_THREADS, fa);
        if (polys_now || d_px) {
            b->timer.begin(ctx, "polygons");
            WB_LAUNCH(wb_k_pack_polys, ctx->polys_grid, 256, ctl, cap, fa.ev, fa.poff, b->c_off, b->cpts, b->poly_x, b->poly_y, d_px, d_py);
            b->timer.end(ctx);
        }
        if (do_geom && !geom_async) launch_geometry();
        side_end(ctx);
        if (geom_async) {
            // the geometry kernels write 27 MB of doubles per year of 0.25 degree over PCIe (0.5 ms) and nothing on the device reads
            // them: nobody waits for them here -- neither the main stream nor the host.  They finish beside whatever the device runs
            // next (the next pass's HBM-bound field scan), and wb_get_event_geometry / wb_free_batch / the batch's next wb_indices wait
            // for the event recorded behind them (wb_geom_settle)
            side_begin_marked(ctx);                            // (still from the point marked above: beside wb_k_props, behind the polygons' offsets)
            launch_geometry();
#if !defined(WB_HOSTSIM)
            if (!b->g_event) (void)hipEventCreateWithFlags(&b->g_event, hipEventDisableTiming);
            if (b->g_event) (void)hipEventRecord(b->g_event, ctx->stream);
#endif
            side_leave(ctx);
        }
        side_join(ctx);                                        // (the polygons' offsets, the vertices; synchronous geometry: the control block carries its totals)
        b->timer.begin(ctx, "results");
        WB_LAUNCH(wb_k_pack_results, cap_ne < 64 * 256 ? (cap_ne + 255) / 256 + 1 : 64, 256, ctl, fa.ev, fa.poff, pa.ev_sums, pa.ev_ncells, pa.ev_flags,
                  pa.ev_cell_off, rb, T->base + T->o_lonval, T->base + T->o_lat, p.nlon, p.nlat);
        b->timer.end(ctx);
        if (ctx->keep_device) {
            // device rows for the gather, packed here; the vertex arrays are taken out of this attempt's scratch
            if (b->dev_rows) { bfree_one(b, b->dev_rows); if (b->dev_px) bfree_one(b, b->dev_px); if (b->dev_py) bfree_one(b, b->dev_py); b->dev_rows = nullptr; b->dev_px = b->dev_py = nullptr; }
            b->dev_rows = balloc<double>(b, (size_t)(cap_ne + 1) * WB_ROW_WIDTH);
            if (!b->dev_rows) { release(); return 1; }
            if (!b->job_level.empty() && !b->d_job_level) b->d_job_level = upload(b, b->job_level);
            WB_LAUNCH(wb_k_pack_rows, cap_ne < 256 * 1024 ? (cap_ne + 255) / 256 + 1 : 1024, 256, ctl, fa.ev, pa.ev_sums, pa.ev_ncells, pa.ev_flags,
                      fa.poff, b->d_job_step, b->d_job_level, p.nlevels, b->step_offset, b->dev_rows);
        }
        side_join(ctx);
        if (dsync(ctx) || (!geom_async && side_sync(ctx))) { release(); return 1; }
        fin = *rb.ctl;
        bool again = false;
        if (do_st && fin.max_pairs > cap_pairs) { cap_pairs = next_pow2((long long)fin.max_pairs + fin.max_pairs / 8); again = true; }
        if (do_st && fin.n_recheck > 65536) { p2_off = true; again = true; }    // more guard-band pairs than the recheck list holds: tiled kernel
        if (fin.overflow & 1) { cap_ne = next_pow2((long long)fin.ne_demand + fin.ne_demand / 4); again = true; }
        if (fin.overflow & 2) { cap_poly = fin.poly_total + fin.poly_total / 8 + 1024; again = true; }
        if (do_geom && !geom_async && (fin.overflow & 4)) { cap_gv = (long long)fin.geom_verts + fin.geom_verts / 8 + 1024; cap_gr = (long long)fin.geom_rings + fin.geom_rings / 8 + 1024; again = true; }
        const bool other = again;
        if (p.want_cells && !again && fin.cell_count > cell_cap) { cell_cap = fin.cell_count + 1024; again = true; }   // cell pool too small: again with the exact size
        if (again) {
#if !defined(WB_HOSTSIM)
            if (geom_async && b->g_event) (void)hipEventSynchronize(b->g_event);      // (they read this attempt's descriptors and write its arrays)
#endif
            hfree(ctx, b->poly_x, b->poly_bytes);
            hfree(ctx, b->poly_y, b->poly_bytes);
            b->poly_x = b->poly_y = nullptr;
            for (auto& hb : b->hres) hfree(ctx, hb.first, hb.second);
            b->hres.clear();
            for (auto& hb : b->gres) hfree(ctx, hb.first, hb.second);
            b->gres.clear();
            if (ctx->keep_device && b->dev_rows) { bfree_one(b, b->dev_rows); b->dev_rows = nullptr; }
            release();
            if (attempt == 5) return wb_fail(ctx, "capacity overflow in the index phase");
            if (p.want_cells && other) cell_cap = (unsigned long long)cap_ne * 1024ull;
            continue;
        }
        const int NE = fin.NE;
        const long long NV = fin.poly_total;
        b->poly_n = (size_t)NV;
        auto keep = [&](void* q) { for (size_t i = 0; i < tmp.size(); ++i) if (tmp[i] == q) { tmp.erase(tmp.begin() + i); break; } };
        if (do_geom) {
            // judged by wb_geom_settle from the totals the kernels left: at once when this call waited for them (geometry_async = 0),
            // when somebody asks otherwise -- they may still be running; the arrays they use leave this attempt's scratch
            b->g_pending = true;
            b->g_ring_offset = h_ro; b->g_vert_offset = h_vo; b->g_x = h_gx; b->g_y = h_gy; b->g_tot = h_tot;
            b->g_cap_r = cap_gr; b->g_cap_v = cap_gv;
            b->g_cnt_r = ga.cnt_r; b->g_cnt_v = ga.cnt_v; b->g_flag = ga.gflag;
            keep(ga.cnt_r); keep(ga.cnt_v); keep(ga.gflag);
            b->g_copied = geom_dma ? n_copy : cap_gv;
            if (geom_dma) { b->g_dx = ga.x; b->g_dy = ga.y; b->g_dro = ga.ring_offset; b->g_dvo = ga.vert_offset; keep(ga.x); keep(ga.y); keep(ga.ring_offset); keep(ga.vert_offset); }
            if (!geom_async && wb_geom_settle(b)) { release(); return 1; }
        }
        // the descriptors and polygon offsets stay on the device (lazy index polygons, the geometry kernels)
        b->d_ev = fa.ev; b->d_poff = fa.poff; b->d_ev_cap_seg = cap;
        keep(fa.ev); keep(fa.poff);
        h_props = rb.props;
        h_ev = rb.ev; h_poff = rb.poff; h_sums = rb.sums; h_nc = rb.ncells; h_fl = rb.flags; h_coff = rb.cell_off;
        if (ctx->keep_device) {
            b->dev_px = d_px; b->dev_py = d_py;
            for (size_t i = 0; d_px && i < tmp.size(); ++i) if (tmp[i] == (void*)d_px) { tmp.erase(tmp.begin() + i); break; }
            for (size_t i = 0; d_py && i < tmp.size(); ++i) if (tmp[i] == (void*)d_py) { tmp.erase(tmp.begin() + i); break; }
            b->dev_nv = d_px ? NV : 0;                         // (keep_device = 2: the index polygons stay descriptors, as on a single GPU)
        }
        if (p.want_cells) {
            auto hbuf = [&](size_t bytes) -> void* { void* q = hmalloc(ctx, bytes ? bytes : 8, &got); if (q) b->hres.push_back({q, got}); return q; };
            h_cells = (uint32_t*)hbuf((size_t)(fin.cell_count + 1) * sizeof(uint32_t));
            if (!h_cells) { release(); return wb_fail(ctx, "out of host memory"); }
            d2h_async(ctx, h_cells, pa.cell_pool, (size_t)fin.cell_count * sizeof(uint32_t));
            if (dsync(ctx)) { release(); return 1; }
        }
        (void)NE;
        release();
        break;
    }
    b->sel_job = b->sel_cont = b->cut_job = b->cut_cont = nullptr;
    b->nsel = fin.nsel; b->ncut = fin.ncut;
    if (getenv("WB_DEBUG_STAMPS")) {
        fprintf(stderr, "[wb] stamps (ms of workgroup time, 100 MHz clock):");
        for (int q = 0; q < 24; ++q) if (fin.dbg[q]) fprintf(stderr, " %d:%.2f", q, (double)fin.dbg[q] * 1e-5);
        fprintf(stderr, "\n");
    }
    b->band[0] = fin.band[0]; b->band[1] = fin.band[1];
    b->pair_stats[0] = fin.point_pairs; b->pair_stats[1] = fin.cand_pairs; b->pair_stats[2] = fin.evaluated_pairs;
    for (int q = 0; q < 3; ++q) b->com_rechecks[q] = fin.com_rechecks[q];
    b->box_bound = fin.box_bound; b->box_bound_mismatch = fin.dbg[23];
    b->group_ties = fin.group_ties;
    const int NS = fin.NS, NO = fin.NO, NE = fin.NE;
    if (NS < 0 || NO < 0 || NE < NS + NO)
        return wb_fail(ctx, "wb_indices: inconsistent event counts in the control block (NS " + std::to_string(NS) + ", NO " + std::to_string(NO) + ", NE " + std::to_string(NE) + ")");
    b->dev_n[0] = NS; b->dev_n[1] = NO; b->dev_n[2] = NE - NS - NO;

    // ---- host tables (sized once, filled with plain loops)
    for (int kind = 0; kind < 3; ++kind) {
        wb_batch::Ev& t = b->ev[kind];
        t = wb_batch::Ev();
        const int lo = kind == 0 ? 0 : (kind == 1 ? NS : NS + NO), hi = kind == 0 ? NS : (kind == 1 ? NS + NO : NE);
        const size_t m = (size_t)(hi - lo);
        const long long pbase = NE ? h_poff[lo] : 0, pn = NE ? h_poff[hi] - h_poff[lo] : 0;
        t.step.resize(m); t.level.resize(m); t.contour.resize(m); t.ind1.resize(m); t.ind2.resize(m);
        t.box.assign(4 * m, 0); t.orient.assign(m, 0); t.sums.resize(5 * m); t.props.resize(5 * m); t.ncells.resize(m); t.flags.resize(m);
        t.polyoff.resize(m + 1); t.pbase = (size_t)pbase;
        for (size_t k = 0; k < m; ++k) {
            const int e = lo + (int)k;
            const int* d = &h_ev[(size_t)e * 8];
            const int job = d[1];
            t.step[k] = b->job_step.empty() ? job / p.nlevels : b->job_step[job];
            t.level[k] = b->job_level.empty() ? job % p.nlevels : b->job_level[job];
            t.contour[k] = b->ct_base[job] + d[2];
            if (kind == WB_OVERTURNINGS) {
                t.ind1[k] = 0; t.ind2[k] = 0;
                for (int q = 0; q < 4; ++q) t.box[4 * k + q] = d[3 + q];
                t.orient[k] = d[0] >> 8;
            } else {
                t.ind1[k] = d[3]; t.ind2[k] = d[4];
            }
            for (int q = 0; q < 5; ++q) { t.sums[5 * k + q] = h_sums[(size_t)e * 5 + q]; t.props[5 * k + q] = h_props[(size_t)e * 5 + q]; }
            t.polyoff[k] = h_poff[e] - pbase;
            t.ncells[k] = h_nc[e];
            t.flags[k] = h_fl[e];
        }
        t.polyoff[m] = pn;

        if (p.want_cells) {
            t.celloff.resize(m + 1);
            size_t tot = 0;
            for (size_t k = 0; k < m; ++k) { t.celloff[k] = (int64_t)tot; tot += h_coff[lo + k] >= 0 ? (size_t)h_nc[lo + k] : 0; }
            t.celloff[m] = (int64_t)tot;
            t.cx.resize(tot); t.cy.resize(tot);
            for (size_t k = 0; k < m; ++k) {
                const long long off = h_coff[lo + k];
                if (off < 0) continue;
                for (int c = 0; c < h_nc[lo + k]; ++c) {
                    const uint32_t v = h_cells[(size_t)off + c];
                    t.cx[(size_t)t.celloff[k] + c] = wb_pt_x(v);
                    t.cy[(size_t)t.celloff[k] + c] = wb_pt_y(v);
                }
            }
        }
    }
    // the small pinned result blocks go back to the context's pool
    for (auto& hb : b->hres) hfree(ctx, hb.first, hb.second);
    b->hres.clear();
    b->timer.collect(!b->g_pending);          // (geometry kernels nobody waits for: their stage is collected by wb_get_timings)
    return 0;
}

static int wb_event_rows(wb_batch* b, int kind, wb_event_table* o, bool polygons) {
    if (!b || !o || kind < 0 || kind > 2) return 1;
    if (polygons && !b->poly_x && b->poly_n > 0 && wb_pack_polys_now(b)) return 1;
    wb_batch::Ev& t = b->ev[kind];
    o->n = (int64_t)t.step.size();
    o->step = t.step.data(); o->level_idx = t.level.data(); o->contour = t.contour.data();
    o->ind1 = t.ind1.data(); o->ind2 = t.ind2.data(); o->box = t.box.data(); o->orientation = t.orient.data();
    o->sums = t.sums.data(); o->ncells = t.ncells.data(); o->flags = t.flags.data(); o->props = t.props.data();
    o->cell_offset = b->p.want_cells ? t.celloff.data() : nullptr;
    o->cell_x = t.cx.data(); o->cell_y = t.cy.data();
    if (t.polyoff.empty()) t.polyoff.push_back(0);
    o->poly_offset = t.polyoff.data();
    o->poly_x = (polygons && b->poly_x) ? b->poly_x + t.pbase : nullptr; o->poly_y = (polygons && b->poly_y) ? b->poly_y + t.pbase : nullptr;
    return 0;
}
// the event table of one index.  wb_get_events: with the polygons in grid indices -- packed into pinned host memory by this call if
// wb_indices left them on the device (context option index_polygons = 0, the default: it then launches a kernel and waits, so it
// belongs to the thread that drives the context or to a moment when no pass runs on it).  wb_get_event_rows: everything but
// the vertices (poly_x = poly_y = NULL, poly_offset still holds the vertex counts): what a caller needs who takes the
// polygons in lon / lat from wb_get_event_geometry -- never launches, any thread.
extern "C" int wb_get_events(wb_batch* b, int kind, wb_event_table* o) { return wb_event_rows(b, kind, o, true); }
extern "C" int wb_get_event_rows(wb_batch* b, int kind, wb_event_table* o) { return wb_event_rows(b, kind, o, false); }

// ------------------------------------------------------------------------------------------------ user-supplied contours
// decorator_contour_calculation (contour_index.py:197-249): the index functions also accept a contour frame
// computed earlier (and possibly filtered by the user).  One job per contour row keeps the frame's row order.
extern "C" int wb_batch_from_contours(wb_ctx* ctx, const wb_params* pp, int64_t n, const int32_t* step,
                                      const int32_t* level_idx, const int32_t* closed, const int64_t* offset,
                                      const int32_t* npts, const int32_t* x, const int32_t* y, wb_batch** out) {
    if (!ctx || !pp || !out || n < 0) return 1;
    const wb_params& p = *pp;
    if (!(p.dlon > 0) || !(p.dlat > 0)) return wb_fail(ctx, "dlon and dlat must be positive");
    const int padd = (int)(p.periodic_add / p.dlon);
    if (padd < 0 || padd > p.nlon) return wb_fail(ctx, "periodic_add must be between 0 and 360 degrees");
    const int W = p.nlon + padd;
    if (W > WB_MAX_W || p.nlat > 32767) return wb_fail(ctx, "grid too large: nlon + periodic columns <= 16384, nlat <= 32767");
    std::lock_guard<std::recursive_mutex> wb_lk_(ctx->mu);
    if (!wb_select_device(ctx)) return wb_fail(ctx, "hipSetDevice failed");
    int maxn = 1;
    for (int64_t k = 0; k < n; ++k) {
        if (npts[k] < 1) return wb_fail(ctx, "empty contour in frame");
        if (step[k] < 0 || step[k] >= p.nt) return wb_fail(ctx, "contour date not found in data");
        maxn = npts[k] > maxn ? npts[k] : maxn;
        for (int i = 0; i < npts[k]; ++i) {
            const int xx = x[offset[k] + i], yy = y[offset[k] + i];
            if (xx < 0 || xx >= W || yy < 0 || yy >= p.nlat) return wb_fail(ctx, "contour point outside the (extended) grid");
        }
    }
    wb_batch* b = new wb_batch();
    b->ctx = ctx;
    b->p = p;
    b->lat.assign(p.lat, p.lat + p.nlat);
    b->lon.assign(p.lon, p.lon + p.nlon);
    b->levels.assign(p.levels, p.levels + p.nlevels);
    b->p.lat = b->lat.data(); b->p.lon = b->lon.data(); b->p.levels = b->levels.data();
    b->W = W; b->padd = padd;
    const int J = (int)n;
    b->njobs = J;
    // one job per contour row (events come back in frame-row order).  A job's slot stride is kept minimal: its one contour
    // lives in a packed point array behind the slots, addressed through c_off (the kernels add c_off to the job's base
    // whatever its value) -- a stride of the longest contour cost 150 MB of zero-filled uploads for a year's frame
    const int cap = 4;
    b->cap_seg = cap;
    (void)maxn;
    const size_t esz = p.dtype == 0 ? 4 : 8;
    const size_t fbytes = (size_t)p.nt * p.nlat * p.nlon * esz;
    if (p.on_device) {
        b->d_field = p.field;
        b->d_intensity = p.intensity;
    } else {
        void* d = dmalloc(ctx, fbytes);
        if (!d) { delete b; return 1; }
        b->owned.push_back(d);
        h2d(ctx, d, p.field, fbytes);
        b->d_field = d;
        if (p.intensity) {
            void* di = dmalloc(ctx, fbytes);
            if (!di) { wb_free_batch(b); return 1; }
            b->owned.push_back(di);
            h2d(ctx, di, p.intensity, fbytes);
            b->d_intensity = di;
        }
    }
    long long np_total = 0;
    for (int j = 0; j < J; ++j) np_total += npts[j];
    if ((long long)J * 2 * cap + np_total >= 0x7FFFFFFFLL) return wb_fail(ctx, "contour frame too large");
    std::vector<uint32_t> h_pts((size_t)J * 2 * cap + (size_t)np_total, 0);
    long long ppos = (long long)J * 2 * cap;
    std::vector<int> h_off((size_t)J * cap, 0), h_len((size_t)J * cap, 0), h_cl((size_t)J * cap, 0), h_nux((size_t)J * cap, 0),
        h_sy((size_t)J * cap, 0), h_nc(J, 1);
    std::vector<char> seen(W);
    b->job_step.resize(J);
    b->job_level.resize(J);
    b->max_nux = 0;
    for (int j = 0; j < J; ++j) {
        std::fill(seen.begin(), seen.end(), 0);
        int nux = 0;
        long long sy = 0;
        for (int i = 0; i < npts[j]; ++i) {
            const int xx = x[offset[j] + i], yy = y[offset[j] + i];
            h_pts[(size_t)ppos + i] = (uint32_t)xx | ((uint32_t)yy << 16);
            if (!seen[xx]) { seen[xx] = 1; ++nux; }
            sy += yy;
        }
        h_off[(size_t)j * cap] = (int)(ppos - (long long)j * 2 * cap);       // relative to the job's base cpts + 2 * j * cap
        ppos += npts[j];
        h_len[(size_t)j * cap] = npts[j];
        h_cl[(size_t)j * cap] = closed[j] ? 1 : 0;
        h_nux[(size_t)j * cap] = nux;
        h_sy[(size_t)j * cap] = (int)sy;
        b->job_step[j] = step[j];
        b->job_level[j] = level_idx[j];
        if (nux > b->max_nux) b->max_nux = nux;
    }
    b->cpts = upload(b, h_pts);
    b->c_off = upload(b, h_off); b->c_len = upload(b, h_len); b->c_closed = upload(b, h_cl);
    b->c_nux = upload(b, h_nux); b->c_sumy = upload(b, h_sy); b->ncont = upload(b, h_nc);
    b->d_job_step = upload(b, b->job_step);
    b->status = balloc<uint32_t>(b, J > 0 ? J : 1);
    b->d_ctl = balloc<WbCtl>(b, 1);
    if (!b->status || !b->d_ctl) { wb_free_batch(b); return 1; }
    dmemset(ctx, b->status, 0, (size_t)(J > 0 ? J : 1) * sizeof(uint32_t));
    {
        WbCtl c0;
        memset(&c0, 0, sizeof(c0));
        c0.max_nux = b->max_nux;
        h2d(ctx, b->d_ctl, &c0, sizeof(c0));
    }
    b->h_status.assign(J, 0);
    b->ct_base.assign(J + 1, 0);
    for (int j = 0; j < J; ++j) b->ct_base[j + 1] = j + 1;
    if (J > 0 ? phase1_summary(b) : dsync(ctx)) { wb_free_batch(b); return 1; }
    b->h_ncont = h_nc;
    b->max_nux = b->h_ctl.max_nux;
    *out = b;
    return 0;
}

// ------------------------------------------------------------------------------------------------ synthetic PV
struct WbSynthArgs {
    void* field; int dtype; long long t0; int nt, nlat, nlon; const double* lat; const double* lon;
    double par[7][5]; double eps; unsigned long long seed;
};
WB_HD inline unsigned long long wb_synth_key(unsigned long long seed, unsigned long long stream, unsigned long long a,
                                             unsigned long long bb) {
    unsigned long long k = wb_mix64(seed * 0x100000001B3ull + stream);
    k = wb_mix64(k ^ a);
    k = wb_mix64(k ^ bb);
    return k;
}
WB_KERNEL(wb_k_synth, WbSynthArgs a) {
    // one workgroup per (time step, latitude row); see oracle/synth.py for the formula
    const int n = WB_BLOCK_ID / a.nlat, j = WB_BLOCK_ID - n * a.nlat;
    const double t = (double)(a.t0 + n);
    const double la = a.lat[j], phi = fabs(la), phir = phi * (M_PI / 180.0);
    const double sgn = la > 0 ? 1.0 : (la < 0 ? -1.0 : 0.0);
    WB_PFOR(i, a.nlon)
        const double lam = a.lon[i] * (M_PI / 180.0);
        double D = 0.0;
        for (int m = 0; m < 7; ++m) {
            const double k = (double)(m + 3);
            const double ph = la < 0 ? a.par[m][4] : a.par[m][1];
            D += a.par[m][0] * cos(k * lam + ph + a.par[m][3] * t + a.par[m][2] * k * (phir - 0.8));
        }
        double v = 2.0 + 6.0 * tanh((phi + D - 45.0) / 6.0);
        double pv = sgn * (v > 0.0 ? v : 0.0);
        if (a.eps != 0.0) {
            const unsigned long long cell = (unsigned long long)j * (unsigned long long)a.nlon + (unsigned long long)i;
            const unsigned long long h = wb_synth_key(a.seed, 6, (unsigned long long)(a.t0 + n), cell);
            pv += a.eps * (2.0 * ((double)(h >> 11) * (1.0 / 9007199254740992.0)) - 1.0);
        }
        const size_t idx = ((size_t)n * a.nlat + j) * a.nlon + i;
        // generator contract: no cell equals +-2 exactly (moved one ulp up, as oracle/synth.py does)
        if (a.dtype == 0) {
            float f = (float)pv;
            if (f == 2.0f) f = nextafterf(2.0f, INFINITY);
            if (f == -2.0f) f = nextafterf(-2.0f, INFINITY);
            ((float*)a.field)[idx] = f;
        } else {
            if (pv == 2.0) pv = nextafter(2.0, (double)INFINITY);
            if (pv == -2.0) pv = nextafter(-2.0, (double)INFINITY);
            ((double*)a.field)[idx] = pv;
        }
    WB_END_PFOR
}

extern "C" int wb_synth_fill(wb_ctx* ctx, void* dev_field, int dtype, int64_t t0, int nt, int nlat, int nlon,
                             const double* lat, const double* lon, const double* wave_params, double eps, uint64_t seed) {
    if (!ctx || !dev_field) return 1;
    wb_use(ctx);
    double* d_lat = (double*)dmalloc(ctx, nlat * sizeof(double));
    double* d_lon = (double*)dmalloc(ctx, nlon * sizeof(double));
    if (!d_lon) return 1;
    h2d(ctx, d_lat, lat, nlat * sizeof(double));
    h2d(ctx, d_lon, lon, nlon * sizeof(double));
    WbSynthArgs a;
    a.field = dev_field; a.dtype = dtype; a.t0 = t0; a.nt = nt; a.nlat = nlat; a.nlon = nlon; a.lat = d_lat; a.lon = d_lon;
    memcpy(a.par, wave_params, sizeof(a.par));
    a.eps = eps; a.seed = seed;
    WB_LAUNCH(wb_k_synth, (long long)nt * nlat, 256, a);
    const int rc = dsync(ctx);
    dfree(ctx, d_lat);
    dfree(ctx, d_lon);
    return rc;
}

// ------------------------------------------------------------------------------------------------ pre-processing
#include "wb_spatial.h"
#include <float.h>

#if !defined(WB_HOSTSIM)
template <typename T, bool CAST, bool UNIT4, bool FAST>
static void wb_launch_smooth_plus2(wb_ctx* ctx, int P, const WbSmoothArgs& a) {
    // four strips per workgroup (measured: 1, 2, 8, 13, 16 waves per workgroup are all slower)
    const long long waves = a.nt * a.nstrips;
    const int wpb = 4;
    const dim3 grid((unsigned)((waves + wpb - 1) / wpb)), block(64 * wpb);
    switch (P) {
#define WB_SP_CASE(N) \
    case N: hipLaunchKernelGGL((wb_k_smooth_plus<T, N, CAST, UNIT4, FAST>), grid, block, 0, ctx->stream, a); break;
        WB_SP_CASE(1) WB_SP_CASE(2) WB_SP_CASE(3) WB_SP_CASE(4) WB_SP_CASE(5) WB_SP_CASE(6) WB_SP_CASE(7) WB_SP_CASE(8)
#undef WB_SP_CASE
    }
}
template <typename T, bool CAST>
static void wb_launch_smooth_plus(wb_ctx* ctx, int P, const WbSmoothArgs& a) {
    // the default weights (outer taps 1, integer sum) get the specialised instance, anything else the general one
    if (a.unit4 && a.fastdiv == 2) wb_launch_smooth_plus2<T, CAST, true, true>(ctx, P, a);
    else wb_launch_smooth_plus2<T, CAST, false, false>(ctx, P, a);
}
#endif
#define WB_SP_MAXFUSE 8

extern "C" float wb_last_op_ms(wb_ctx* ctx) { return ctx ? ctx->last_op_ms : 0.f; }

// ------------------------------------------------------------------------------------------------ fingerprint of a host buffer
// Chunks of 4 MiB hashed independently (four multiply-rotate lanes over 32-byte blocks: memory-bound), combined in chunk
// order: the same value whatever the number of threads.
static uint64_t wb_rotl64(uint64_t v, int r) { return (v << r) | (v >> (64 - r)); }
static uint64_t wb_hash_chunk(const unsigned char* p, size_t n, uint64_t seed) {
    const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full;
    uint64_t h[4] = {seed + P1 + P2, seed + P2, seed, seed - P1};
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        memcpy(w, p + i, 32);
        for (int k = 0; k < 4; ++k) h[k] = wb_rotl64(h[k] + w[k] * P2, 31) * P1;
    }
    uint64_t t = wb_rotl64(h[0], 1) + wb_rotl64(h[1], 7) + wb_rotl64(h[2], 12) + wb_rotl64(h[3], 18) + (uint64_t)n;
    for (; i < n; ++i) t = wb_rotl64(t ^ ((uint64_t)p[i] * P1), 11) * P2;
    return wb_mix64(t);
}
extern "C" uint64_t wb_fingerprint(const void* host_buffer, uint64_t bytes, int threads) {
    const unsigned char* p = (const unsigned char*)host_buffer;
    if (!p || !bytes) return 0x5742ull;
    const size_t chunk = (size_t)4 << 20, nchunks = (size_t)((bytes + chunk - 1) / chunk);
    if (threads <= 0) { threads = (int)std::thread::hardware_concurrency() / 2; if (threads > 32) threads = 32; }
    if (threads < 1) threads = 1;
    if ((size_t)threads > nchunks) threads = (int)nchunks;
    std::vector<uint64_t> part(nchunks);
    auto work = [&](int t) {
        for (size_t c = (size_t)t; c < nchunks; c += (size_t)threads) {
            const size_t o = c * chunk, n = (size_t)bytes - o < chunk ? (size_t)bytes - o : chunk;
            part[c] = wb_hash_chunk(p + o, n, (uint64_t)c);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    uint64_t h = 0x5742ull ^ (uint64_t)bytes;
    for (size_t c = 0; c < nchunks; ++c) h = wb_mix64(h ^ part[c]) + 0x9E3779B97F4A7C15ull;
    return h;
}

extern "C" int wb_set_option(wb_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return 1;
    if (!strcmp(name, "scan_split")) { ctx->scan_split = value ? 1 : 0; return 0; }
    if (!strcmp(name, "eager_f2")) { ctx->eager_f2 = value ? 1 : 0; return 0; }
    if (!strcmp(name, "keep_device")) { ctx->keep_device = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return 0; }
    if (!strcmp(name, "scan_variant")) { ctx->scan_variant = value; return 0; }
    if (!strcmp(name, "scan_keys_only")) { ctx->scan_keys_only = value ? 1 : 0; return 0; }
    if (!strcmp(name, "scan_f64_fast")) { ctx->scan_f64_fast = value ? 1 : 0; return 0; }
    if (!strcmp(name, "scan_debug")) { ctx->scan_debug = value; return 0; }
    if (!strcmp(name, "scan_grid_mult")) { ctx->scan_grid_mult = value; return 0; }
    if (!strcmp(name, "scan_wg_per_cu")) { ctx->scan_wg_per_cu = value > 0 ? (int)value : 0; return 0; }
    if (!strcmp(name, "scan_dynamic")) { ctx->scan_dynamic = value ? 1 : 0; return 0; }
    if (!strcmp(name, "pairs_lds")) { ctx->pairs_lds = value ? 1 : 0; return 0; }
    if (!strcmp(name, "p2_budget_kb")) { ctx->p2_budget_kb = value; ctx->p2_budget = 0; return 0; }
    if (!strcmp(name, "p2_split")) { ctx->p2_split = value; return 0; }
    if (!strcmp(name, "p2_grid_x2")) { ctx->p2_grid_x2 = value > 0 ? value : 4; return 0; }
    if (!strcmp(name, "f1_grid_x2")) { ctx->f1_grid_x2 = value > 0 ? (int)value : 4; return 0; }
    if (!strcmp(name, "scan_halo_lds")) { ctx->scan_halo_lds = value ? 1 : 0; return 0; }
    if (!strcmp(name, "scan_exclusive")) { ctx->scan_exclusive = value ? 1 : 0; return 0; }
    if (!strcmp(name, "scan_priority")) { ctx->scan_priority = value ? 1 : 0; return 0; }
    if (!strcmp(name, "persist_slack")) { ctx->persist_slack = value > 0 ? (int)value : 0; return 0; }
    if (!strcmp(name, "link_rows")) { ctx->link_rows = value ? 1 : 0; return 0; }
    if (!strcmp(name, "trace_fused")) { ctx->trace_fused = value ? 1 : 0; return 0; }
    if (!strcmp(name, "geometry")) { ctx->geometry = value ? 1 : 0; return 0; }
    if (!strcmp(name, "geometry_async")) { ctx->geometry_async = value ? 1 : 0; return 0; }
    if (!strcmp(name, "geometry_dma")) { ctx->geometry_dma = value ? 1 : 0; return 0; }
    if (!strcmp(name, "index_polygons")) { ctx->index_polygons = value ? 1 : 0; return 0; }
    if (!strcmp(name, "trace_nmax")) { ctx->trace_nmax = value > 0 ? value : 0; return 0; }
    if (!strcmp(name, "trace_cmax")) { ctx->trace_cmax = value > 0 ? value : 0; return 0; }
    if (!strcmp(name, "f1_fused")) { ctx->f1_fused = value ? 1 : 0; return 0; }
    if (!strcmp(name, "f234_lpts")) { ctx->f234_lpts = value >= 64 ? value : 0; return 0; }      // (tests: longer contours take wb_k_streamer_f234's path through global memory)
    if (!strcmp(name, "scan_bands")) { ctx->scan_bands = value; return 0; }
    if (!strcmp(name, "scan_pad")) { ctx->scan_pad = value >= 0 ? value : 0; return 0; }
    if (!strcmp(name, "scan_band_rows")) { ctx->scan_band_rows = value; return 0; }
    if (!strcmp(name, "polys_grid")) { ctx->polys_grid = value > 0 ? value : 2048; return 0; }
    if (!strcmp(name, "boxcom_per_cu")) { ctx->boxcom_per_cu = value > 0 ? value : 12; return 0; }
    if (!strcmp(name, "boxcom_after")) { ctx->boxcom_after = value < 0 ? 0 : (value > 2 ? 2 : value); return 0; }
    if (!strcmp(name, "timers")) { ctx->timers = value < 0 ? 0 : (value > 2 ? 2 : value); return 0; }
#if !defined(WB_HOSTSIM)
    // value = first * 1000 + count compute units (count 0: no mask)
    if (!strcmp(name, "cu_mask")) {
        if (!wb_remask_streams(ctx, value / 1000, value % 1000)) return wb_fail(ctx, "wb_set_option: cu_mask: the masked streams could not be made");
        return 0;
    }
    if (!strcmp(name, "boxcom_cu_mask")) {       // before the first wb_indices of the context (its stream is made on first use)
        if (ctx->stream4) return wb_fail(ctx, "wb_set_option: boxcom_cu_mask after the box kernel's stream was made");
        ctx->boxcom_cu_first = value / 1000; ctx->boxcom_cu_count = value % 1000;
        return 0;
    }
    if (!strcmp(name, "scan_cu_mask")) {         // before the first wb_contours of the context (its stream is made on first use)
        if (ctx->stream3) return wb_fail(ctx, "wb_set_option: scan_cu_mask after the scan's stream was made");
        ctx->scan_cu_first = value / 1000; ctx->scan_cu_count = value % 1000; ctx->scan_priority = ctx->scan_cu_count > 0 ? 1 : 0;
        return 0;
    }
#else
    if (!strcmp(name, "cu_mask") || !strcmp(name, "scan_cu_mask") || !strcmp(name, "boxcom_cu_mask")) return 0;
#endif
    ctx->err = std::string("wb_set_option: unknown option ") + name;
    return 1;
}

extern "C" int wb_smooth_field(wb_ctx* ctx, const void* field, int dtype, int on_device, int64_t nt, int nlat, int nlon,
                               int passes, const double* weights, int kh, int kw, double wsum, int mode, double* out,
                               int out_on_device, int flags) {
    if (!ctx) return 1;
    wb_use(ctx);
    if (!field || !out || !weights || nt < 0 || nlat < 1 || nlon < 1 || passes < 1 || kh < 1 || kw < 1 || kh > 7 || kw > 7 ||
        mode < 0 || mode > 4 || (dtype != 0 && dtype != 1)) {
        ctx->err = "wb_smooth_field: bad argument (passes >= 1, weights 1..7 x 1..7, mode 0..4, dtype 0/1)";
        return 1;
    }
    const int b = (int)(kh / 2.0 + 0.5);                       // spatial.py:105
    if (b > nlat) { ctx->err = "wb_smooth_field: border rows exceed the latitude axis"; return 1; }
    if (nt == 0) return 0;
    const size_t ncell = (size_t)nt * nlat * nlon, esz = dtype == 0 ? 4 : 8;
    WbSmoothArgs a;
    memset(&a, 0, sizeof(a));
    a.nt = nt; a.nlat = nlat; a.nlon = nlon; a.mode = mode; a.wsum = wsum;
    // scipy _correlate_or_convolve(convolution=True): flipped kernel, origin -origin (-1 for even sizes)
    const int ch = kh / 2 - ((kh & 1) ? 0 : 1), cw = kw / 2 - ((kw & 1) ? 0 : 1);
    for (int i = 0; i < kh; ++i)
        for (int j = 0; j < kw; ++j) {
            const double w = weights[(kh - 1 - i) * kw + (kw - 1 - j)];
            if (!(fabs(w) > DBL_EPSILON)) continue;             // ni_filters.c: only taps with |w| > eps take part
            a.tr[a.ntap] = i - ch; a.tc[a.ntap] = j - cw; a.tw[a.ntap] = w; ++a.ntap;
        }
    {   // wb_sm_div's short sequence needs wsum = k * 2^j with an odd integer |k| < 2^26 and a moderate exponent
        int ex = 0;
        const double m = frexp(wsum, &ex);
        const double k26 = ldexp(m, 26);
        a.fastdiv = (wsum == wsum && wsum != 0.0 && fabs(wsum) < 1e300 && k26 == floor(k26) && ex > -60 && ex < 60) ? 1 : 0;
        a.rsum = 1.0 / wsum;
        if (flags & 2) a.fastdiv = 0;
    }
    bool plus = false;
#if !defined(WB_HOSTSIM)
    if (kh == 3 && kw == 3 && mode == WB_SM_WRAP && !(flags & 1)) {
        auto F = [&](int i, int j) { return weights[(2 - i) * 3 + (2 - j)]; };
        auto zero = [&](double w) { return !(fabs(w) > DBL_EPSILON); };
        if (zero(F(0, 0)) && zero(F(0, 2)) && zero(F(2, 0)) && zero(F(2, 2))) {
            plus = true;
            const double pw[5] = {F(0, 1), F(1, 0), F(1, 1), F(1, 2), F(2, 1)};
            for (int k = 0; k < 5; ++k) { a.pw[k] = pw[k]; a.puse[k] = zero(pw[k]) ? 0 : 1; }
            a.unit4 = (pw[0] == 1.0 && pw[1] == 1.0 && pw[3] == 1.0 && pw[4] == 1.0) ? 1 : 0;
            // the fused kernel's unchecked short division (wb_sp_strip) also bounds |S| and the centre weight
            const double ac = fabs(pw[2]), as = fabs(wsum);
            if (a.fastdiv && as >= 0x1p-20 && as <= 0x1p20 && (zero(pw[2]) || (ac >= 0x1p-20 && ac <= 0x1p20))) a.fastdiv = 2;
        }
    }
#endif
    (void)flags;
    // device buffers: input copy (host input), output (host output), one float64 scratch cube when several launches chain
    const int nlaunch = plus ? (passes + WB_SP_MAXFUSE - 1) / WB_SP_MAXFUSE : passes;
    const void* d_in = field;
    void* d_in_own = nullptr;
    if (!on_device) {
        d_in_own = dmalloc(ctx, ncell * esz);
        if (!d_in_own) return 1;
        h2d(ctx, d_in_own, field, ncell * esz);
        d_in = d_in_own;
    }
    double* d_out = out_on_device ? out : (double*)dmalloc(ctx, ncell * 8);
    double* d_tmp = nlaunch > 1 ? (double*)dmalloc(ctx, ncell * 8) : nullptr;
    if (!d_out || (nlaunch > 1 && !d_tmp)) { ctx->err = "wb_smooth_field: out of device memory"; return 1; }
    WbOpTimer timer(ctx);
    const void* src = d_in;
    int done = 0;
    for (int l = 0; l < nlaunch; ++l) {
        const int P = plus ? ((passes - done) < WB_SP_MAXFUSE ? (passes - done) : WB_SP_MAXFUSE) : 1;
        double* dst = ((nlaunch - 1 - l) & 1) ? d_tmp : d_out;          // the last launch lands in d_out
        a.in = src; a.out = dst;
        a.in_f32 = (l == 0 && dtype == 0) ? 1 : 0;
        a.cast_f32 = a.in_f32;
        a.nan_rows = (l == nlaunch - 1) ? b : 0;
        if (plus) {
#if !defined(WB_HOSTSIM)
            a.halo = P + (P & 1);
            a.strip_w = 128 - 2 * a.halo;
            a.nstrips = (nlon + a.strip_w - 1) / a.strip_w;
            a.vec = ((nlon & 1) == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0) ? 1 : 0;
            if (a.in_f32) wb_launch_smooth_plus<float, true>(ctx, P, a);
            else wb_launch_smooth_plus<double, false>(ctx, P, a);
#endif
        } else {
            const long long nchunk = ((long long)ncell + WB_SM_CHUNK - 1) / WB_SM_CHUNK;
            WB_LAUNCH(wb_k_smooth_pass, nchunk < 8192 ? nchunk : 8192, 256, a);
        }
        src = dst;
        done += P;
    }
    timer.stop();
    int rc = 0;
    if (!out_on_device) { rc = d2h(ctx, out, d_out, ncell * 8); dfree(ctx, d_out); }
    else rc = dsync(ctx);
    if (d_tmp) dfree(ctx, d_tmp);
    if (d_in_own) dfree(ctx, d_in_own);
    return rc;
}

extern "C" int wb_momentum_flux(wb_ctx* ctx, const void* u, const void* v, int dtype, int on_device, int64_t nt, int nlat,
                                int nlon, void* out, int out_on_device) {
    if (!ctx) return 1;
    wb_use(ctx);
    if (!u || !v || !out || nt < 0 || nlat < 1 || nlon < 1 || (dtype != 0 && dtype != 1)) {
        ctx->err = "wb_momentum_flux: bad argument";
        return 1;
    }
    if (nt == 0) return 0;
    WbMfluxArgs a;
    memset(&a, 0, sizeof(a));
    a.nrows = (long long)nt * nlat; a.nlon = nlon;
    if (nlon > 4096 || wb_mf_plan(a, nlon)) { ctx->err = "wb_momentum_flux: longitude axis too long for the summation plan"; return 1; }
    const size_t bytes = (size_t)nt * nlat * nlon * (dtype == 0 ? 4 : 8);
    void *du = nullptr, *dv = nullptr;
    if (!on_device) {
        du = dmalloc(ctx, bytes); dv = dmalloc(ctx, bytes);
        if (!du || !dv) { ctx->err = "wb_momentum_flux: out of device memory"; return 1; }
        h2d(ctx, du, u, bytes); h2d(ctx, dv, v, bytes);
    }
    void* dout = out_on_device ? out : dmalloc(ctx, bytes);
    if (!dout) { ctx->err = "wb_momentum_flux: out of device memory"; return 1; }
    a.u = on_device ? u : du; a.v = on_device ? v : dv; a.out = dout;
    WbOpTimer timer(ctx);
    const long long grid = a.nrows < 16384 ? a.nrows : 16384;     // single-wave workgroups, each looping over rows
    if (dtype == 0) WB_LAUNCH(wb_k_mflux<float>, grid, 64, a);
    else WB_LAUNCH(wb_k_mflux<double>, grid, 64, a);
    timer.stop();
    int rc = 0;
    if (!out_on_device) { rc = d2h(ctx, out, dout, bytes); dfree(ctx, dout); }
    else rc = dsync(ctx);
    if (du) dfree(ctx, du);
    if (dv) dfree(ctx, dv);
    return rc;
}

#include "wb_flags.h"
#include <algorithm>
#include <numeric>

extern "C" int wb_flag_events(wb_ctx* ctx, int64_t nt, int nlat, int nlon, const double* lat, const double* lon, double radius,
                              int64_t nev, const int32_t* step, const double* value, const int64_t* ring_offset,
                              const int64_t* vert_offset, const double* vx, const double* vy, int out_dtype, void* out,
                              int out_on_device) {
    if (!ctx) return 1;
    wb_use(ctx);
    if (!out || !lat || !lon || nt < 0 || nlat < 2 || nlon < 2 || nev < 0 || out_dtype < 0 || out_dtype > 2 || !(radius >= 0.0) ||
        (nev > 0 && (!step || !ring_offset || !vert_offset || !vx || !vy))) {
        ctx->err = "wb_flag_events: bad argument";
        return 1;
    }
    const double dlat = lat[1] - lat[0], dlon = lon[1] - lon[0];
    if (!(dlat > 0.0) || !(dlon > 0.0)) { ctx->err = "wb_flag_events: lat / lon must be ascending"; return 1; }
    for (int64_t k = 0; k < nev; ++k)
        if (step[k] < 0 || step[k] >= nt) { ctx->err = "wb_flag_events: event time index out of range"; return 1; }
    const size_t esz = out_dtype == 0 ? 1 : (out_dtype == 1 ? 4 : 8);
    const size_t bytes = (size_t)nt * nlat * nlon * esz;
    void* d_out = out_on_device ? out : dmalloc(ctx, bytes ? bytes : 1);
    if (!d_out) { ctx->err = "wb_flag_events: out of device memory"; return 1; }
    dmemset(ctx, d_out, 0, bytes);
    int rc = 0;
    if (nev > 0 && nt > 0) {
        // events by time step, frame order inside a step
        std::vector<int64_t> order((size_t)nev);
        std::iota(order.begin(), order.end(), (int64_t)0);
        std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return step[x] < step[y]; });
        std::vector<int> act_step;
        std::vector<long long> act_ev0, ev_ring0((size_t)nev + 1), ring_v0;
        std::vector<double> svx, svy, sval;
        ring_v0.push_back(0);
        for (int64_t q = 0; q < nev; ++q) {
            const int64_t k = order[(size_t)q];
            if (q == 0 || step[k] != step[order[(size_t)q - 1]]) { act_step.push_back(step[k]); act_ev0.push_back(q); }
            ev_ring0[(size_t)q] = (long long)ring_v0.size() - 1;
            for (int64_t rg = ring_offset[k]; rg < ring_offset[k + 1]; ++rg) {
                for (int64_t v = vert_offset[rg]; v < vert_offset[rg + 1]; ++v) { svx.push_back(vx[v]); svy.push_back(vy[v]); }
                ring_v0.push_back((long long)svx.size());
            }
            if (value) sval.push_back(value[k]);
        }
        act_ev0.push_back(nev);
        ev_ring0[(size_t)nev] = (long long)ring_v0.size() - 1;
        WbFlagArgs a;
        memset(&a, 0, sizeof(a));
        a.nt = nt; a.nlat = nlat; a.nlon = nlon; a.dlat = dlat; a.dlon = dlon; a.r = radius; a.nact = (int)act_step.size();
        a.out_dtype = out_dtype; a.out = d_out;
        const int grid = a.nact < 1024 ? a.nact : 1024;
        auto up = [&](const void* h, size_t n) -> void* {
            void* d = dmalloc(ctx, n ? n : 8);
            if (d && n) h2d(ctx, d, h, n);
            return d;
        };
        double* d_lat = (double*)up(lat, nlat * sizeof(double));
        double* d_lon = (double*)up(lon, nlon * sizeof(double));
        int* d_as = (int*)up(act_step.data(), act_step.size() * sizeof(int));
        long long* d_ae = (long long*)up(act_ev0.data(), act_ev0.size() * sizeof(long long));
        long long* d_er = (long long*)up(ev_ring0.data(), ev_ring0.size() * sizeof(long long));
        long long* d_rv = (long long*)up(ring_v0.data(), ring_v0.size() * sizeof(long long));
        double* d_vx = (double*)up(svx.data(), svx.size() * sizeof(double));
        double* d_vy = (double*)up(svy.data(), svy.size() * sizeof(double));
        double* d_val = value ? (double*)up(sval.data(), sval.size() * sizeof(double)) : nullptr;
        int* d_tog = (int*)dmalloc(ctx, (size_t)grid * nlat * (nlon + 1) * sizeof(int));
        if (!d_lat || !d_lon || !d_as || !d_ae || !d_er || !d_rv || !d_vx || !d_vy || !d_tog || (value && !d_val)) {
            ctx->err = "wb_flag_events: out of device memory";
            rc = 1;
        } else {
            a.lat = d_lat; a.lon = d_lon; a.act_step = d_as; a.act_ev0 = d_ae; a.ev_ring0 = d_er; a.ring_v0 = d_rv;
            a.vx = d_vx; a.vy = d_vy; a.value = d_val; a.tog = d_tog;
            WbOpTimer timer(ctx);
            WB_LAUNCH(wb_k_flag_events, grid, 256, a);
            timer.stop();
        }
        for (void* d : {(void*)d_lat, (void*)d_lon, (void*)d_as, (void*)d_ae, (void*)d_er, (void*)d_rv, (void*)d_vx, (void*)d_vy,
                        (void*)d_val, (void*)d_tog})
            if (d) dfree(ctx, d);
    }
    if (!out_on_device) {
        if (!rc) rc = d2h(ctx, out, d_out, bytes);
        dfree(ctx, d_out);
    } else if (!rc) {
        rc = dsync(ctx);
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------ track_events
// the geometry of wb_track_events / wb_track_overlaps on the device, events in the order `order`
struct WbTrackGeom {
    std::vector<long long> edge0, row0, piece0, fedge0;
    std::vector<short> ex0, ey0, ex1, ey1;
    std::vector<short> fx0, fy0, fx1, fy1;          // ALL ring edges (the reach test of buffered tracking needs the horizontal ones too)
    std::vector<int> ymin, ymax, xmin, xmax;
    std::vector<void*> dev;
};
static int wb_track_geometry(wb_ctx* ctx, const char* who, int64_t n, const int64_t* order, const int64_t* ring_offset,
                             const int64_t* vert_offset, const int32_t* vx, const int32_t* vy, WbTrackGeom& g, WbTrackArgs& a) {
    g.edge0.assign((size_t)n + 1, 0); g.row0.assign((size_t)n + 1, 0); g.piece0.assign((size_t)n + 1, 0); g.fedge0.assign((size_t)n + 1, 0);
    g.ymin.assign((size_t)n, 0); g.ymax.assign((size_t)n, 0); g.xmin.assign((size_t)n, 0); g.xmax.assign((size_t)n, 0);
    for (int64_t q = 0; q < n; ++q) {
        const int64_t e = order[q];
        int y0 = INT32_MAX, y1 = INT32_MIN, x0 = INT32_MAX, x1 = INT32_MIN;
        long long pieces = 0;
        for (int64_t rg = ring_offset[e]; rg < ring_offset[e + 1]; ++rg) {
            const int64_t v0 = vert_offset[rg], nv = vert_offset[rg + 1] - v0;
            for (int64_t k = 0; k < nv; ++k) {
                const int64_t k2 = k + 1 < nv ? k + 1 : 0;                 // the closing edge is implied (zero length if repeated)
                const int ax = vx[v0 + k], ay = vy[v0 + k], bx = vx[v0 + k2], by = vy[v0 + k2];
                if (ax < -32768 || ax > 32767 || ay < -32768 || ay > 32767) {
                    ctx->err = std::string(who) + ": lattice coordinate outside the 16-bit range";
                    return 1;
                }
                y0 = ay < y0 ? ay : y0; y1 = ay > y1 ? ay : y1; x0 = ax < x0 ? ax : x0; x1 = ax > x1 ? ax : x1;
                if (a.reach > 0.0 && (ax != bx || ay != by || nv == 1)) {
                    g.fx0.push_back((short)ax); g.fy0.push_back((short)ay); g.fx1.push_back((short)bx); g.fy1.push_back((short)by);
                }
                if (ay == by) continue;                                    // horizontal edges bound no row from the side
                g.ex0.push_back((short)ax); g.ey0.push_back((short)ay); g.ex1.push_back((short)bx); g.ey1.push_back((short)by);
                pieces += ay < by ? by - ay : ay - by;
            }
        }
        if (y0 > y1) { y0 = y1 = 0; x0 = x1 = 0; }                        // an empty geometry
        g.ymin[(size_t)q] = y0; g.ymax[(size_t)q] = y1; g.xmin[(size_t)q] = x0; g.xmax[(size_t)q] = x1;
        g.edge0[(size_t)q + 1] = (long long)g.ex0.size();
        g.fedge0[(size_t)q + 1] = (long long)g.fx0.size();
        g.row0[(size_t)q + 1] = g.row0[(size_t)q] + (y1 - y0);
        g.piece0[(size_t)q + 1] = g.piece0[(size_t)q] + pieces;
        if (pieces > INT32_MAX) { ctx->err = std::string(who) + ": polygon too large"; return 1; }
    }
    auto up = [&](const void* h, size_t bytes) -> void* {
        void* d = dmalloc(ctx, bytes ? bytes : 8);
        if (d && bytes) h2d(ctx, d, h, bytes);
        g.dev.push_back(d);
        return d;
    };
    auto scratch = [&](size_t bytes) -> void* { void* d = dmalloc(ctx, bytes ? bytes : 8); g.dev.push_back(d); return d; };
    const size_t rows = (size_t)g.row0[(size_t)n], pieces = (size_t)g.piece0[(size_t)n], ne = g.ex0.size();
    a.edge0 = (const long long*)up(g.edge0.data(), g.edge0.size() * 8);
    a.row0 = (const long long*)up(g.row0.data(), g.row0.size() * 8);
    a.piece0 = (const long long*)up(g.piece0.data(), g.piece0.size() * 8);
    a.ex0 = (const short*)up(g.ex0.data(), ne * 2); a.ey0 = (const short*)up(g.ey0.data(), ne * 2);
    a.ex1 = (const short*)up(g.ex1.data(), ne * 2); a.ey1 = (const short*)up(g.ey1.data(), ne * 2);
    if (a.reach > 0.0) {
        const size_t nf = g.fx0.size();
        a.fedge0 = (const long long*)up(g.fedge0.data(), g.fedge0.size() * 8);
        a.fx0 = (const short*)up(g.fx0.data(), nf * 2); a.fy0 = (const short*)up(g.fy0.data(), nf * 2);
        a.fx1 = (const short*)up(g.fx1.data(), nf * 2); a.fy1 = (const short*)up(g.fy1.data(), nf * 2);
    }
    a.ymin = (const int*)up(g.ymin.data(), (size_t)n * 4); a.ymax = (const int*)up(g.ymax.data(), (size_t)n * 4);
    a.xmin = (const int*)up(g.xmin.data(), (size_t)n * 4); a.xmax = (const int*)up(g.xmax.data(), (size_t)n * 4);
    a.row_off = (int*)scratch(rows * 4); a.row_cnt = (int*)scratch(rows * 4);
    a.piece = (unsigned long long*)scratch(pieces * 8);
    a.row_area = (double*)scratch(rows * 8);
    a.area = (double*)scratch((size_t)n * 8);
    for (void* d : g.dev)
        if (!d) { ctx->err = std::string(who) + ": out of device memory"; return 1; }
    return 0;
}
static void wb_track_release(wb_ctx* ctx, WbTrackGeom& g) {
    for (void* d : g.dev)
        if (d) dfree(ctx, d);
    g.dev.clear();
}

static int wb_track_events_impl(wb_ctx* ctx, int64_t n, const int64_t* date_ns, const double* date_val, int method,
                                double time_range, double overlap, double distance, const double* com,
                                const int64_t* ring_offset, const int64_t* vert_offset, const int32_t* vx, const int32_t* vy,
                                int32_t* label, double* area, uint64_t* stats, double reach, double step_x, double step_y);

extern "C" int wb_track_events(wb_ctx* ctx, int64_t n, const int64_t* date_ns, const double* date_val, int method,
                               double time_range, double overlap, double distance, const double* com,
                               const int64_t* ring_offset, const int64_t* vert_offset, const int32_t* vx, const int32_t* vy,
                               int32_t* label, double* area, uint64_t* stats) {
    return wb_track_events_impl(ctx, n, date_ns, date_val, method, time_range, overlap, distance, com, ring_offset, vert_offset, vx, vy,
                                label, area, stats, 0.0, 1.0, 1.0);
}

extern "C" int wb_track_events_buffered(wb_ctx* ctx, int64_t n, const int64_t* date_ns, const double* date_val, double time_range,
                                        double buffer_deg, double step_x_deg, double step_y_deg, const int64_t* ring_offset,
                                        const int64_t* vert_offset, const int32_t* vx, const int32_t* vy, int32_t* label, uint64_t* stats) {
    if (!ctx) return 1;
    if (!(buffer_deg > 0.0) || !(step_x_deg > 0.0) || !(step_y_deg > 0.0)) { ctx->err = "wb_track_events_buffered: buffer and grid steps must be positive"; return 1; }
    return wb_track_events_impl(ctx, n, date_ns, date_val, 0, time_range, 0.0, 0.0, nullptr, ring_offset, vert_offset, vx, vy, label, nullptr,
                                stats, 2.0 * buffer_deg, step_x_deg, step_y_deg);
}

static int wb_track_events_impl(wb_ctx* ctx, int64_t n, const int64_t* date_ns, const double* date_val, int method,
                                double time_range, double overlap, double distance, const double* com,
                                const int64_t* ring_offset, const int64_t* vert_offset, const int32_t* vx, const int32_t* vy,
                                int32_t* label, double* area, uint64_t* stats, double reach, double step_x, double step_y) {
    if (!ctx) return 1;
    wb_use(ctx);
    if (n < 1 || n > INT32_MAX || (!date_ns == !date_val) || !label || (method != 0 && method != 1) ||
        (method == 1 && !com) || (method == 0 && (!ring_offset || !vert_offset || !vx || !vy))) {
        ctx->err = "wb_track_events: bad argument";
        return 1;
    }
    std::vector<int64_t> order((size_t)n);
    std::iota(order.begin(), order.end(), (int64_t)0);
    if (date_ns) std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return date_ns[x] < date_ns[y]; });
    else std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return date_val[x] < date_val[y]; });
    WbTrackArgs a;
    memset(&a, 0, sizeof(a));
    a.n = n; a.method = method; a.date_kind = date_ns ? 0 : 1; a.time_range = time_range; a.overlap = overlap; a.distance = distance;
    a.reach = reach; a.sx = step_x; a.sy = step_y;
    const int nstats = reach > 0.0 ? 6 : 4;
    WbTrackGeom g;
    int rc = 0;
    if (method == 0) rc = wb_track_geometry(ctx, "wb_track_events", n, order.data(), ring_offset, vert_offset, vx, vy, g, a);
    std::vector<long long> sns; std::vector<double> sval, scom; std::vector<int> orig((size_t)n);
    for (int64_t q = 0; q < n; ++q) orig[(size_t)q] = (int)order[(size_t)q];
    if (date_ns) { sns.resize((size_t)n); for (int64_t q = 0; q < n; ++q) sns[(size_t)q] = date_ns[order[(size_t)q]]; }
    else { sval.resize((size_t)n); for (int64_t q = 0; q < n; ++q) sval[(size_t)q] = date_val[order[(size_t)q]]; }
    if (method == 1) {
        scom.resize((size_t)n * 2);
        for (int64_t q = 0; q < n; ++q) { scom[(size_t)q * 2] = com[order[(size_t)q] * 2]; scom[(size_t)q * 2 + 1] = com[order[(size_t)q] * 2 + 1]; }
    }
    auto up = [&](const void* h, size_t bytes) -> void* {
        void* d = dmalloc(ctx, bytes ? bytes : 8);
        if (d && bytes) h2d(ctx, d, h, bytes);
        g.dev.push_back(d);
        if (!d && !rc) { ctx->err = "wb_track_events: out of device memory"; rc = 1; }
        return d;
    };
    if (!rc) {
        a.date_ns = date_ns ? (const long long*)up(sns.data(), (size_t)n * 8) : nullptr;
        a.date_val = date_ns ? nullptr : (const double*)up(sval.data(), (size_t)n * 8);
        a.orig = (const int*)up(orig.data(), (size_t)n * 4);
        a.com = method == 1 ? (const double*)up(scom.data(), (size_t)n * 16) : nullptr;
        a.parent = (int*)dmalloc(ctx, (size_t)n * 4); g.dev.push_back(a.parent);
        a.flag = (int*)dmalloc(ctx, (size_t)n * 4); g.dev.push_back(a.flag);
        a.label = (int*)dmalloc(ctx, (size_t)n * 4); g.dev.push_back(a.label);
        a.stats = (unsigned long long*)dmalloc(ctx, 6 * 8); g.dev.push_back(a.stats);
        if (!a.parent || !a.flag || !a.label || !a.stats) { ctx->err = "wb_track_events: out of device memory"; rc = 1; }
    }
    if (!rc) {
        dmemset(ctx, a.stats, 0, 6 * 8);
        WbOpTimer timer(ctx);
        const int gl = n < 4096 ? (int)((n + 255) / 256) : 1024;         // flat kernels: a contiguous slice per workgroup
        WB_LAUNCH(wb_k_track_init, gl, 256, a);
        const long long ge = n < (1 << 20) ? n : (1 << 20);
        if (method == 0) WB_LAUNCH(wb_k_track_build, ge, 256, a);
        WB_LAUNCH(wb_k_track_pairs, ge, WB_TRACK_THREADS, a);
        WB_LAUNCH(wb_k_track_roots, gl, 256, a);
        WB_LAUNCH(wb_k_track_rank, 1, 1024, a);
        WB_LAUNCH(wb_k_track_labels, gl, 256, a);
        timer.stop();
        rc = d2h(ctx, label, a.label, (size_t)n * 4);
        if (!rc && stats) rc = d2h(ctx, stats, a.stats, (size_t)nstats * 8);
        if (!rc && area && method == 0) {
            std::vector<double> sa((size_t)n);
            rc = d2h(ctx, sa.data(), a.area, (size_t)n * 8);
            for (int64_t q = 0; q < n; ++q) area[order[(size_t)q]] = sa[(size_t)q];
        }
    }
    wb_track_release(ctx, g);
    return rc;
}

extern "C" int wb_track_overlaps(wb_ctx* ctx, int64_t n, const int64_t* ring_offset, const int64_t* vert_offset, const int32_t* vx,
                                 const int32_t* vy, int64_t npairs, const int64_t* pair_i, const int64_t* pair_j, double* inter,
                                 double* area) {
    if (!ctx) return 1;
    wb_use(ctx);
    if (n < 1 || n > INT32_MAX || !ring_offset || !vert_offset || !vx || !vy || npairs < 0 || (npairs > 0 && (!pair_i || !pair_j || !inter))) {
        ctx->err = "wb_track_overlaps: bad argument";
        return 1;
    }
    for (int64_t q = 0; q < npairs; ++q)
        if (pair_i[q] < 0 || pair_i[q] >= n || pair_j[q] < 0 || pair_j[q] >= n) { ctx->err = "wb_track_overlaps: pair out of range"; return 1; }
    std::vector<int64_t> order((size_t)n);
    std::iota(order.begin(), order.end(), (int64_t)0);
    WbTrackArgs a;
    memset(&a, 0, sizeof(a));
    a.n = n;
    WbTrackGeom g;
    int rc = wb_track_geometry(ctx, "wb_track_overlaps", n, order.data(), ring_offset, vert_offset, vx, vy, g, a);
    long long *d_pi = nullptr, *d_pj = nullptr;
    double* d_out = nullptr;
    if (!rc) {
        d_pi = (long long*)dmalloc(ctx, (size_t)npairs * 8); g.dev.push_back(d_pi);
        d_pj = (long long*)dmalloc(ctx, (size_t)npairs * 8); g.dev.push_back(d_pj);
        d_out = (double*)dmalloc(ctx, (size_t)npairs * 8); g.dev.push_back(d_out);
        if (!d_pi || !d_pj || !d_out) { ctx->err = "wb_track_overlaps: out of device memory"; rc = 1; }
    }
    if (!rc) {
        if (npairs) { h2d(ctx, d_pi, pair_i, (size_t)npairs * 8); h2d(ctx, d_pj, pair_j, (size_t)npairs * 8); }
        WbOpTimer timer(ctx);
        WB_LAUNCH(wb_k_track_build, n < (1 << 20) ? n : (1 << 20), 256, a);
        if (npairs) WB_LAUNCH(wb_k_track_overlap_list, npairs < (1 << 20) ? npairs : (1 << 20), WB_TRACK_THREADS, a, (long long)npairs,
                              (const long long*)d_pi, (const long long*)d_pj, d_out);
        timer.stop();
        if (npairs) rc = d2h(ctx, inter, d_out, (size_t)npairs * 8);
        if (!rc && area) rc = d2h(ctx, area, a.area, (size_t)n * 8);
    }
    wb_track_release(ctx, g);
    return rc;
}

// ------------------------------------------------------------------------------------------------ RCCL exchanges
#if defined(WB_HOSTSIM)
extern "C" int wb_comm_unique_id(wb_ctx* ctx, void*) { return wb_fail(ctx, "the host simulation has no RCCL"); }
extern "C" int wb_comm_init(wb_ctx* ctx, const void*, int, int) { return wb_fail(ctx, "the host simulation has no RCCL"); }
extern "C" int wb_comm_allreduce_max(wb_ctx* ctx, int*) { return wb_fail(ctx, "the host simulation has no RCCL"); }
extern "C" int wb_comm_gather_tables(wb_ctx* ctx, wb_batch*, int, const double**, int64_t*, const int32_t**, const int32_t**, int64_t*, int*) {
    return wb_fail(ctx, "the host simulation has no RCCL");
}
extern "C" int wb_comm_stage_tables(wb_ctx* ctx, wb_batch*, int) { return wb_fail(ctx, "the host simulation has no RCCL"); }
extern "C" int wb_comm_gather_staged(wb_ctx* ctx, int, int, const double**, int64_t*, const int32_t**, const int32_t**, int64_t*, int*) {
    return wb_fail(ctx, "the host simulation has no RCCL");
}
extern "C" void wb_comm_destroy(wb_ctx*) {}
extern "C" int wb_comm_abort(wb_ctx* ctx) { return wb_fail(ctx, "the host simulation has no RCCL"); }
extern "C" int wb_comm_reserve(wb_ctx* ctx, int64_t, int64_t) { return wb_fail(ctx, "the host simulation has no RCCL"); }
#else
#define WB_NCCL(call) do { const ncclResult_t r_ = (call); if (r_ != ncclSuccess) return wb_fail(ctx, std::string(#call) + ": " + cm.GetErrorString(r_)); } while (0)
#define WB_HIPC(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return wb_fail(ctx, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" int wb_comm_unique_id(wb_ctx* ctx, void* id) {
    if (!ctx || !id) return 1;
    wb_use(ctx);
    if (!ctx->comm) ctx->comm = new WbComm();
    WbComm& cm = *ctx->comm;
    const std::string e = wb_comm_load(cm);
    if (!e.empty()) return wb_fail(ctx, "wb_comm_unique_id: " + e);
    ncclUniqueId u;
    WB_NCCL(cm.GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

static void wb_comm_release(wb_ctx* ctx) {
    if (!ctx || !ctx->comm) return;
    WbComm& cm = *ctx->comm;
    (void)hipSetDevice(ctx->device);
    if (cm.stream) (void)hipStreamSynchronize(cm.stream);
    if (cm.copy_stream) (void)hipStreamSynchronize(cm.copy_stream);
    if (cm.comm && !cm.aborted && cm.CommDestroy) (void)cm.CommDestroy(cm.comm);     // (an aborted communicator is gone already)
    void* dev[] = {cm.d_counts, cm.d_scalar, cm.d_rows, cm.d_px, cm.d_py};
    for (void* q : dev) if (q) (void)hipFree(q);
    for (auto& sl : cm.slot) { if (sl.rows) (void)hipFree(sl.rows); if (sl.px) (void)hipFree(sl.px); if (sl.py) (void)hipFree(sl.py); }
    void* host[] = {cm.h_counts, cm.h_scalar, cm.h_rows, cm.h_px, cm.h_py};
    for (void* q : host) if (q) (void)hipHostFree(q);
    if (cm.stream) (void)hipStreamDestroy(cm.stream);
    if (cm.copy_stream) (void)hipStreamDestroy(cm.copy_stream);
    delete ctx->comm;                       // (the RCCL library stays loaded: it may be the host program's)
    ctx->comm = nullptr;
}

static int wb_comm_init_buffers(wb_ctx* ctx, WbComm& cm, int world) {
    WB_HIPC(hipStreamCreateWithFlags(&cm.stream, hipStreamNonBlocking));
    WB_HIPC(hipStreamCreateWithFlags(&cm.copy_stream, hipStreamNonBlocking));
    WB_HIPC(hipMalloc((void**)&cm.d_counts, (size_t)(3 + 3 * world) * sizeof(long long)));
    WB_HIPC(hipHostMalloc((void**)&cm.h_counts, (size_t)(3 * world + 3) * sizeof(long long), hipHostMallocDefault));
    WB_HIPC(hipMalloc((void**)&cm.d_scalar, 2 * sizeof(int)));
    WB_HIPC(hipHostMalloc((void**)&cm.h_scalar, 2 * sizeof(int), hipHostMallocDefault));
    return 0;
}

extern "C" int wb_comm_init(wb_ctx* ctx, const void* id, int rank, int world) {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return 1;
    wb_use(ctx);
    if (!ctx->comm) ctx->comm = new WbComm();
    WbComm& cm = *ctx->comm;
    if (cm.comm) return wb_fail(ctx, "wb_comm_init: the context has a communicator already");
    const std::string e = wb_comm_load(cm);
    if (!e.empty()) return wb_fail(ctx, "wb_comm_init: " + e);
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    WB_NCCL(cm.CommInitRank(&cm.comm, world, u, rank));
    cm.rank = rank; cm.world = world;
    if (wb_comm_init_buffers(ctx, cm, world)) {          // a partial set-up is undone: the call can be repeated
        const std::string why = ctx->err;
        wb_comm_release(ctx);
        return wb_fail(ctx, "wb_comm_init: " + why);
    }
    return 0;
}

extern "C" void wb_comm_destroy(wb_ctx* ctx) { wb_comm_release(ctx); }

extern "C" int wb_comm_allreduce_max(wb_ctx* ctx, int* value) {
    if (!ctx || !value) return 1;
    if (ctx->comm && ctx->comm->aborted) return wb_fail(ctx, "wb_comm_allreduce_max: the communicator was aborted (wb_comm_abort)");
    if (!ctx->comm || !ctx->comm->comm) return wb_fail(ctx, "wb_comm_allreduce_max: wb_comm_init first");
    wb_use(ctx);
    WbComm& cm = *ctx->comm;
    {
        std::unique_lock<std::timed_mutex> lk(cm.issue);
        if (cm.aborted) return wb_fail(ctx, "wb_comm_allreduce_max: the communicator was aborted (wb_comm_abort)");
        cm.h_scalar[0] = *value;
        WB_HIPC(hipMemcpyAsync(cm.d_scalar, cm.h_scalar, sizeof(int), hipMemcpyHostToDevice, cm.stream));
        WB_NCCL(cm.AllReduce(cm.d_scalar, cm.d_scalar + 1, 1, ncclInt32, ncclMax, cm.comm, cm.stream));
        WB_HIPC(hipMemcpyAsync(cm.h_scalar + 1, cm.d_scalar + 1, sizeof(int), hipMemcpyDeviceToHost, cm.stream));
    }
    WB_HIPC(hipStreamSynchronize(cm.stream));
    if (cm.aborted) return wb_fail(ctx, "wb_comm_allreduce_max: the communicator was aborted (wb_comm_abort)");
    *value = cm.h_scalar[1];
    return 0;
}

// inside an open ncclGroupStart: a failing call closes the group before the error is reported
#define WB_NCCL_G(call) do { const ncclResult_t r_ = (call); if (r_ != ncclSuccess) { (void)cm.GroupEnd(); return wb_fail(ctx, std::string(#call) + ": " + cm.GetErrorString(r_)); } } while (0)

// room for `rows` event rows and `verts` polygon vertices of the whole group's catalogue (0: leave as it is), device and pinned
// host.  A capacity is raised only when its allocation has succeeded (a failed one leaves pointer and capacity at zero: the
// next gather tries again instead of writing through a null pointer).  Views handed out by an earlier gather (*rows, *poly_x,
// *poly_y of wb_comm_gather_*) die here when a buffer grows.
static int wb_comm_room(wb_ctx* ctx, WbComm& cm, size_t rows, size_t verts) {
    if (rows > cm.cap_rows) {
        if (cm.d_rows) (void)hipFree(cm.d_rows);
        cm.d_rows = nullptr; cm.cap_rows = 0;
        WB_HIPC(hipMalloc((void**)&cm.d_rows, rows * WB_ROW_WIDTH * sizeof(double)));
        cm.cap_rows = rows;
    }
    if (verts > cm.cap_v) {
        if (cm.d_px) (void)hipFree(cm.d_px);
        if (cm.d_py) (void)hipFree(cm.d_py);
        cm.d_px = cm.d_py = nullptr; cm.cap_v = 0;
        WB_HIPC(hipMalloc((void**)&cm.d_px, verts * sizeof(int32_t)));
        WB_HIPC(hipMalloc((void**)&cm.d_py, verts * sizeof(int32_t)));
        cm.cap_v = verts;
    }
    if (rows > cm.hcap_rows) {
        if (cm.h_rows) (void)hipHostFree(cm.h_rows);
        cm.h_rows = nullptr; cm.hcap_rows = 0;
        WB_HIPC(hipHostMalloc((void**)&cm.h_rows, rows * WB_ROW_WIDTH * sizeof(double), hipHostMallocDefault));
        cm.hcap_rows = rows;
    }
    if (verts > cm.hcap_v) {
        if (cm.h_px) (void)hipHostFree(cm.h_px);
        if (cm.h_py) (void)hipHostFree(cm.h_py);
        cm.h_px = cm.h_py = nullptr; cm.hcap_v = 0;
        WB_HIPC(hipHostMalloc((void**)&cm.h_px, verts * sizeof(int32_t), hipHostMallocDefault));
        WB_HIPC(hipHostMalloc((void**)&cm.h_py, verts * sizeof(int32_t), hipHostMallocDefault));
        cm.hcap_v = verts;
    }
    return 0;
}

// exchange (2) for tables that lie in device memory at src_*: counts, payload, the one copy to the host on dst
static int wb_comm_gather_core(wb_ctx* ctx, WbComm& cm, const double* src_rows, const int32_t* src_px, const int32_t* src_py, long long ne,
                               long long nv, int dst, const double** rows, int64_t* nrows, const int32_t** poly_x, const int32_t** poly_y,
                               int64_t* nverts, int* ranks_seen) {
    const int W = cm.world;
    // (1) counts: one all-gather of (rank + 1, events, vertices); the landing area is poisoned first, so a rank whose triple
    //     did not arrive -- or arrived in another rank's place -- is seen
    //     (every group of RCCL calls is enqueued under cm.issue and behind a look at cm.aborted: wb_comm_abort may come from
    //     another thread, WbComm in wb_comm.h)
    const char* gone = "wb_comm_gather_tables: the communicator was aborted (wb_comm_abort)";
    {
        std::unique_lock<std::timed_mutex> lk(cm.issue);
        if (cm.aborted) return wb_fail(ctx, gone);
        for (int q = 0; q < 3 * W; ++q) cm.h_counts[q] = -1;
        cm.h_counts[3 * W] = cm.rank + 1; cm.h_counts[3 * W + 1] = ne; cm.h_counts[3 * W + 2] = nv;
        WB_HIPC(hipMemcpyAsync(cm.d_counts + 3, cm.h_counts, (size_t)3 * W * sizeof(long long), hipMemcpyHostToDevice, cm.stream));
        WB_HIPC(hipMemcpyAsync(cm.d_counts, cm.h_counts + 3 * W, 3 * sizeof(long long), hipMemcpyHostToDevice, cm.stream));
        WB_NCCL(cm.AllGather(cm.d_counts, cm.d_counts + 3, 3, ncclInt64, cm.comm, cm.stream));
        WB_HIPC(hipMemcpyAsync(cm.h_counts, cm.d_counts + 3, (size_t)3 * W * sizeof(long long), hipMemcpyDeviceToHost, cm.stream));
    }
    WB_HIPC(hipStreamSynchronize(cm.stream));
    if (cm.aborted) return wb_fail(ctx, gone);
    long long tot_e = 0, tot_v = 0;
    int seen = 0;
    for (int r = 0; r < W; ++r) {
        if (cm.h_counts[3 * r] == r + 1 && cm.h_counts[3 * r + 1] >= 0 && cm.h_counts[3 * r + 2] >= 0) ++seen;
        tot_e += cm.h_counts[3 * r + 1]; tot_v += cm.h_counts[3 * r + 2];
    }
    if (ranks_seen) *ranks_seen = seen;
    if (seen != W) return wb_fail(ctx, "wb_comm_gather_tables: the all-gather of the counts returned " + std::to_string(seen) + " of " + std::to_string(W) + " ranks in place");
    if (cm.h_counts[3 * cm.rank + 1] != ne || cm.h_counts[3 * cm.rank + 2] != nv) return wb_fail(ctx, "wb_comm_gather_tables: the all-gather returned another count for this rank");
    *rows = nullptr; *poly_x = *poly_y = nullptr; *nrows = 0; *nverts = 0;
    if (cm.rank == dst) {
        // room for the whole catalogue, on the device and in pinned host memory: grown when needed, kept between passes
        if (wb_comm_room(ctx, cm, (size_t)tot_e > cm.cap_rows || (size_t)tot_e > cm.hcap_rows ? (size_t)tot_e + (size_t)tot_e / 4 + 1024 : 0,
                         (size_t)tot_v > cm.cap_v || (size_t)tot_v > cm.hcap_v ? (size_t)tot_v + (size_t)tot_v / 4 + 4096 : 0)) return 1;
    }
    // (2) payload: exact sizes, rank order = time order
    {
    std::unique_lock<std::timed_mutex> lk(cm.issue);
    if (cm.aborted) return wb_fail(ctx, gone);
    WB_NCCL(cm.GroupStart());
    if (cm.rank == dst) {
        long long oe = 0, ov = 0;
        for (int r = 0; r < W; ++r) {
            const long long ke = cm.h_counts[3 * r + 1], kv = cm.h_counts[3 * r + 2];
            if (r != dst) {
                if (ke) WB_NCCL_G(cm.Recv(cm.d_rows + (size_t)oe * WB_ROW_WIDTH, (size_t)ke * WB_ROW_WIDTH, ncclDouble, r, cm.comm, cm.stream));
                if (kv) { WB_NCCL_G(cm.Recv(cm.d_px + ov, (size_t)kv, ncclInt32, r, cm.comm, cm.stream)); WB_NCCL_G(cm.Recv(cm.d_py + ov, (size_t)kv, ncclInt32, r, cm.comm, cm.stream)); }
            }
            oe += ke; ov += kv;
        }
    } else {
        if (ne) WB_NCCL_G(cm.Send(src_rows, (size_t)ne * WB_ROW_WIDTH, ncclDouble, dst, cm.comm, cm.stream));
        if (nv) { WB_NCCL_G(cm.Send(src_px, (size_t)nv, ncclInt32, dst, cm.comm, cm.stream)); WB_NCCL_G(cm.Send(src_py, (size_t)nv, ncclInt32, dst, cm.comm, cm.stream)); }
    }
    WB_NCCL(cm.GroupEnd());
    if (cm.rank == dst) {
        long long oe = 0, ov = 0;
        for (int r = 0; r < dst; ++r) { oe += cm.h_counts[3 * r + 1]; ov += cm.h_counts[3 * r + 2]; }
        if (ne) WB_HIPC(hipMemcpyAsync(cm.d_rows + (size_t)oe * WB_ROW_WIDTH, src_rows, (size_t)ne * WB_ROW_WIDTH * sizeof(double), hipMemcpyDeviceToDevice, cm.stream));
        if (nv) {
            WB_HIPC(hipMemcpyAsync(cm.d_px + ov, src_px, (size_t)nv * sizeof(int32_t), hipMemcpyDeviceToDevice, cm.stream));
            WB_HIPC(hipMemcpyAsync(cm.d_py + ov, src_py, (size_t)nv * sizeof(int32_t), hipMemcpyDeviceToDevice, cm.stream));
        }
        // the one copy to the host
        if (tot_e) WB_HIPC(hipMemcpyAsync(cm.h_rows, cm.d_rows, (size_t)tot_e * WB_ROW_WIDTH * sizeof(double), hipMemcpyDeviceToHost, cm.stream));
        if (tot_v) {
            WB_HIPC(hipMemcpyAsync(cm.h_px, cm.d_px, (size_t)tot_v * sizeof(int32_t), hipMemcpyDeviceToHost, cm.stream));
            WB_HIPC(hipMemcpyAsync(cm.h_py, cm.d_py, (size_t)tot_v * sizeof(int32_t), hipMemcpyDeviceToHost, cm.stream));
        }
    }
    }
    WB_HIPC(hipStreamSynchronize(cm.stream));
    if (cm.aborted) return wb_fail(ctx, gone);
    if (cm.rank == dst) { *rows = cm.h_rows; *poly_x = cm.h_px; *poly_y = cm.h_py; *nrows = tot_e; *nverts = tot_v; }
    return 0;
}

extern "C" int wb_comm_gather_tables(wb_ctx* ctx, wb_batch* b, int dst, const double** rows, int64_t* nrows, const int32_t** poly_x,
                                     const int32_t** poly_y, int64_t* nverts, int* ranks_seen) {
    if (!ctx || !rows || !nrows || !poly_x || !poly_y || !nverts) return 1;
    if (ctx->comm && ctx->comm->aborted) return wb_fail(ctx, "wb_comm_gather_tables: the communicator was aborted (wb_comm_abort)");
    if (!ctx->comm || !ctx->comm->comm) return wb_fail(ctx, "wb_comm_gather_tables: wb_comm_init first");
    WbComm& cm = *ctx->comm;
    if (dst < 0 || dst >= cm.world) return wb_fail(ctx, "wb_comm_gather_tables: bad destination rank");
    if (b && !b->dev_rows) return wb_fail(ctx, "wb_comm_gather_tables: set the context option keep_device before wb_indices");
    wb_use(ctx);
    const long long ne = b ? b->dev_n[0] + b->dev_n[1] + b->dev_n[2] : 0, nv = b ? b->dev_nv : 0;
    return wb_comm_gather_core(ctx, cm, b ? b->dev_rows : nullptr, b ? b->dev_px : nullptr, b ? b->dev_py : nullptr, ne, nv, dst, rows, nrows,
                               poly_x, poly_y, nverts, ranks_seen);
}

// The tables of batch b (NULL: none) copied into slot `slot` of the communicator; when the call returns the batch may be freed.
extern "C" int wb_comm_stage_tables(wb_ctx* ctx, wb_batch* b, int slot) {
    if (!ctx) return 1;
    if (!ctx->comm || !ctx->comm->comm) return wb_fail(ctx, "wb_comm_stage_tables: wb_comm_init first");
    if (slot < 0 || slot >= WB_COMM_SLOTS) return wb_fail(ctx, "wb_comm_stage_tables: bad slot");
    if (b && !b->dev_rows) return wb_fail(ctx, "wb_comm_stage_tables: set the context option keep_device before wb_indices");
    WbComm& cm = *ctx->comm;
    WbComm::Slot& sl = cm.slot[slot];
    if (sl.full) return wb_fail(ctx, "wb_comm_stage_tables: the slot still holds tables that were not gathered");
    wb_use(ctx);
    const long long ne = b ? b->dev_n[0] + b->dev_n[1] + b->dev_n[2] : 0, nv = b ? b->dev_nv : 0;
    if ((size_t)ne > sl.cap_rows) {
        if (sl.rows) (void)hipFree(sl.rows);
        sl.rows = nullptr;
        sl.cap_rows = (size_t)ne + (size_t)ne / 4 + 1024;
        WB_HIPC(hipMalloc((void**)&sl.rows, sl.cap_rows * WB_ROW_WIDTH * sizeof(double)));
    }
    if ((size_t)nv > sl.cap_v) {
        if (sl.px) (void)hipFree(sl.px);
        if (sl.py) (void)hipFree(sl.py);
        sl.px = sl.py = nullptr;
        sl.cap_v = (size_t)nv + (size_t)nv / 4 + 4096;
        WB_HIPC(hipMalloc((void**)&sl.px, sl.cap_v * sizeof(int32_t)));
        WB_HIPC(hipMalloc((void**)&sl.py, sl.cap_v * sizeof(int32_t)));
    }
    if (ne) WB_HIPC(hipMemcpyAsync(sl.rows, b->dev_rows, (size_t)ne * WB_ROW_WIDTH * sizeof(double), hipMemcpyDeviceToDevice, cm.copy_stream));
    if (nv) {
        WB_HIPC(hipMemcpyAsync(sl.px, b->dev_px, (size_t)nv * sizeof(int32_t), hipMemcpyDeviceToDevice, cm.copy_stream));
        WB_HIPC(hipMemcpyAsync(sl.py, b->dev_py, (size_t)nv * sizeof(int32_t), hipMemcpyDeviceToDevice, cm.copy_stream));
    }
    WB_HIPC(hipStreamSynchronize(cm.copy_stream));
    sl.ne = ne; sl.nv = nv; sl.full = true;
    return 0;
}

// wb_comm_gather_tables for the tables staged in `slot`; may be called from another host thread than the one that drives the
// context's batches (it touches the communicator's own streams and buffers only).  The slot is free again afterwards.
extern "C" int wb_comm_gather_staged(wb_ctx* ctx, int slot, int dst, const double** rows, int64_t* nrows, const int32_t** poly_x,
                                     const int32_t** poly_y, int64_t* nverts, int* ranks_seen) {
    if (!ctx || !rows || !nrows || !poly_x || !poly_y || !nverts) return 1;
    if (ctx->comm && ctx->comm->aborted) return wb_fail(ctx, "wb_comm_gather_staged: the communicator was aborted (wb_comm_abort)");
    if (!ctx->comm || !ctx->comm->comm) return wb_fail(ctx, "wb_comm_gather_staged: wb_comm_init first");
    WbComm& cm = *ctx->comm;
    if (cm.aborted) return wb_fail(ctx, "wb_comm_gather_staged: the communicator was aborted (wb_comm_abort)");
    if (dst < 0 || dst >= cm.world) return wb_fail(ctx, "wb_comm_gather_staged: bad destination rank");
    (void)hipSetDevice(ctx->device);
    if (slot == -1) {
        // POISON: this rank cannot deliver a slot (an error upstream) but still takes part in the exchange of the counts -- with
        // -1 for its own --, so that EVERY rank's call fails for this slot with "... ranks in place" instead of waiting for ever
        // for a payload that will not come (the ranks' sequences of collectives stay matched)
        (void)wb_comm_gather_core(ctx, cm, nullptr, nullptr, nullptr, -1, -1, dst, rows, nrows, poly_x, poly_y, nverts, ranks_seen);
        return wb_fail(ctx, "wb_comm_gather_staged: poisoned slot (this rank reported a failure to the group)");
    }
    if (slot < 0 || slot >= WB_COMM_SLOTS || !cm.slot[slot].full) return wb_fail(ctx, "wb_comm_gather_staged: nothing staged in this slot");
    WbComm::Slot& sl = cm.slot[slot];
    const int rc = wb_comm_gather_core(ctx, cm, sl.rows, sl.px, sl.py, sl.ne, sl.nv, dst, rows, nrows, poly_x, poly_y, nverts, ranks_seen);
    sl.full = false;
    return rc;
}
// ncclCommAbort: the communicator is torn down without waiting for outstanding operations -- peers blocked in a collective with
// this rank get an error from RCCL instead of waiting for ever ("RCCL error -> abort").  Every later wb_comm_* call
// of this context fails; wb_comm_destroy releases what is left.
extern "C" int wb_comm_abort(wb_ctx* ctx) {
    if (!ctx) return 1;
    if (!ctx->comm || !ctx->comm->comm) return 0;
    WbComm& cm = *ctx->comm;
    if (cm.aborted.exchange(true)) return 0;
    (void)hipSetDevice(ctx->device);
    // no thread starts another group of RCCL calls from here on.  One that is ENQUEUEING a group right now holds cm.issue: wait for
    // it, but not for ever -- a thread that sits inside an RCCL call (a group end that sets up connections to a peer that is
    // gone) is exactly what ncclCommAbort is for, and it sees `aborted` before it touches the communicator again
    std::unique_lock<std::timed_mutex> lk(cm.issue, std::chrono::seconds(2));
    if (cm.CommAbort) (void)cm.CommAbort(cm.comm);
    if (lk.owns_lock()) cm.comm = nullptr;         // (else the handle's VALUE stays for the thread inside RCCL; nobody passes it on)
    return 0;
}
// room for `rows` event rows and `verts` polygon vertices of the WHOLE group's catalogue on this rank (the destination of the
// gathers), device and pinned host, made NOW -- before the record's batches are in flight: growing them inside a gather is a
// hipFree / hipMalloc beside the workers' kernels (device-wide synchronisations on the first passes)
extern "C" int wb_comm_reserve(wb_ctx* ctx, int64_t rows, int64_t verts) {
    if (!ctx || rows < 0 || verts < 0) return 1;
    if (!ctx->comm || !ctx->comm->comm) return wb_fail(ctx, "wb_comm_reserve: wb_comm_init first");
    WbComm& cm = *ctx->comm;
    (void)hipSetDevice(ctx->device);
    if (cm.aborted) return wb_fail(ctx, "wb_comm_reserve: the communicator was aborted (wb_comm_abort)");
    return wb_comm_room(ctx, cm, (size_t)rows, (size_t)verts);
}
#undef WB_NCCL_G
#undef WB_NCCL
#undef WB_HIPC
#endif

#if defined(WB_HOSTSIM)
extern "C" int wb_hostsim_group_order(const int* pp, const int* pq, int np, int cap, int* out) {
    std::vector<long long> ws((size_t)2 * cap);
    std::vector<unsigned char> done((size_t)np + 1);
    return wb_pyset_group_order(pp, pq, np, done.data(), ws.data(), cap, out);
}
extern "C" double wb_hostsim_np_sum(const double* a, int64_t n) { return wb_np_pairwise_sum(a, (int)n); }
extern "C" void wb_hostsim_div(const double* a, int64_t n, double S, double* fast, double* plain) {
    const double y = 1.0 / S;
    for (int64_t i = 0; i < n; ++i) { fast[i] = wb_sm_div(a[i], S, y, true); plain[i] = a[i] / S; }
}
#endif
