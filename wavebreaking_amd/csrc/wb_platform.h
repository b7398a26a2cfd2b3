// wb_platform.h -- everything in the C ABI's implementation that differs between the product build (hipcc, gfx950) and the
// host simulation of the kernel sources (-DWB_HOSTSIM, g++; TEST INFRASTRUCTURE, tests/_hostsim/, never loaded by the
// package): how a kernel is launched, the context and its device / pinned-host memory, copies, the side stream, timers.
// wb_api.hip is written against these names; what remains switched there are the places where the device runs a kernel
// the host simulation states differently (the streaming scan, the fused smoothing kernel) and the RCCL entry points.
#pragma once
#if defined(WB_HOSTSIM)
int wb_hostsim_reverse = 0;
extern "C" void wb_hostsim_set_reverse(int r) { wb_hostsim_reverse = r; }
long long wb_hostsim_path[4] = {0, 0, 0, 0};            // which rare paths the simulated kernels took (WB_HOSTSIM_PATH; tests assert they were reached)
extern "C" long long wb_hostsim_path_count(int k) { const long long v = wb_hostsim_path[k & 3]; wb_hostsim_path[k & 3] = 0; return v; }
#define WB_LAUNCH(kernel, grid, block, ...)                                       \
    do {                                                                          \
        const int wb_g_ = (int)(grid);                                            \
        for (int wb_b_ = 0; wb_b_ < wb_g_; ++wb_b_) kernel(wb_b_, wb_g_, __VA_ARGS__); \
    } while (0)
#else
#define WB_LAUNCH(kernel, grid, block, ...) \
    hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3(block), 0, ctx->stream, __VA_ARGS__)
#endif
// ... with `lds` bytes of dynamic LDS (the host simulation's kernels keep such arrays in their own storage)
#if defined(WB_HOSTSIM)
#define WB_LAUNCH_LDS(kernel, grid, block, lds, ...) WB_LAUNCH(kernel, grid, block, __VA_ARGS__)
#else
#define WB_LAUNCH_LDS(kernel, grid, block, lds, ...) \
    hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3(block), (lds), ctx->stream, __VA_ARGS__)
#endif

// ------------------------------------------------------------------------------------------------ context
#include <mutex>
struct wb_ctx {
    int device = 0;
    std::string err;
    // Two host threads may meet on a context since round 6: the one that DRIVES it (wb_contours / wb_indices, pass after pass) and
    // one that reads a finished batch's tables and frees it while the next pass is already running (wavebreaking_amd/pipeline.py).
    // `mu` is held by every call that launches on the context's streams (recursive: such calls nest); `alloc_mu` guards the two
    // allocators (device blocks, pinned host blocks) and nothing else, so wb_free_batch never waits for a running pass.
    std::recursive_mutex mu;
    std::mutex alloc_mu;
    // host buffers of the big result arrays are recycled between batches: a fresh 10 MB vector costs ~1 ms of page
    // faults and zero-fill per pass
    std::multimap<size_t, void*> hpool;
    float last_op_ms = 0.f;
    // 1 (default): scan as two kernels -- the streaming patch capture (wb_stream.h: 1.20 ms = 5.1 TB/s for 1460 steps of
    // 1440 x 721 f32) + the list build (0.24 ms); 0: the single classify-and-emit kernel wb_k_scan (1.5-1.7 ms)
    int scan_split = (getenv("WB_SCAN_SPLIT") && atoi(getenv("WB_SCAN_SPLIT")) == 0) ? 0 : 1;
    int scan_debug = 0;
    int scan_halo_lds = 1;                    // streaming scan: halo columns through LDS (0: scalar loads)
    int scan_grid_mult = 1;                   // workgroups of the streaming scan = resident slots x this (capped at the work items)
    int scan_dynamic = 1;                     // streaming scan: a workgroup takes the next work item when it is free (0: items dealt out by workgroup number)
    int scan_keys_only = 1;                   // streaming scan (13 x 4 float kernel): a captured patch is stored as its key alone, the build kernel reads the values from the field (0: key + ten values, 44 B)
    int scan_f64_fast = 1;                    // streaming scan, float64 fields: 8 x 4 kernel with the LDS halo exchange and patches as keys only (0: the scalar-halo 6 x 4 kernel with key + ten values, rounds 2-5)
    int scan_variant = 2;                     // streaming scan, float fields: ring depth x waves per SIMD -- 0: 8 x 6, 1: 10 x 5, 2: 13 x 4 (default), 3: 6 x 8, 4: 9 x 5, 5: 8 x 5
    int boxcom_per_cu = 12;                   // persistent grid of wb_k_box_com: single-wave workgroups per CU (what its LDS admits)
    int boxcom_after = 1;                     // when wb_k_box_com runs on the side stream: 0 right behind wb_k_overturn (beside wb_k_streamer_prep), 1 behind wb_k_streamer_prep, 2 behind wb_k_streamer_pairs2 of the main stream
    int persist_slack = 0;                    // workgroup places the persistent 1024-thread launches (all-pairs, routine 1) leave free: a full grid holds EVERY wave slot of the device until its queue is empty, and the other batch's small kernels wait for that
    int scan_priority = 0;                    // 1: the field scan is launched on a stream of its own (stream3): the device's highest priority, or -- option "scan_cu_mask" -- a CU-masked one
    int cu_first = 0, cu_count = 0;           // option "cu_mask": the context's main and side streams own the compute units [cu_first, cu_first + cu_count) only (0: the whole device)
    int boxcom_cu_first = 0, boxcom_cu_count = 0;   // option "boxcom_cu_mask": wb_k_box_com runs on those compute units only (its single waves otherwise take slots on every CU from the 1024-thread workgroups of the streamer chain beside it)
    int scan_cu_first = 0, scan_cu_count = 0; // option "scan_cu_mask": ... and the field scan's stream those
    int scan_wg_per_cu = 0;                   // option "scan_wg_per_cu": resident workgroups of the streaming scan per CU (0: as many as fit -- two at 0.25 degree)
    int scan_exclusive = 0;                   // 1: a context's field scan waits (on the device) for the scan another context of the same device launched before it
    int index_polygons = 0;                   // 1: wb_indices packs the events' polygons in grid indices into pinned host memory (13.5 MB per year of 0.25 degree over PCIe); 0 (default since round 6): wb_get_events packs them when it is first asked for them -- the drop-in functions build their frames from the lon / lat rings and never ask
    int geometry_async = 1;                   // 1: the geometry kernels of wb_indices (PCIe-bound: 27 MB of doubles per year of 0.25 degree) are not waited for by wb_indices -- they finish beside the next pass's field scan, wb_get_event_geometry waits for them; 0: wb_indices returns when they are done
    int geometry_dma = 1;                     // 1: the geometry kernels write DEVICE arrays and a DMA copy behind them moves the rings to pinned host memory (the copy engine, not the compute units, waits for PCIe; as many vertices as the context's largest pass so far held, the rest -- if any -- when the geometry is asked for); 0: the kernels store into pinned host memory themselves (round 5: while 2048 waves trickle 27 MB through PCIe every kernel beside them slows down -- wb_k_props 0.55 -> 0.9 ms)
    int geometry = 1;                         // 1: wb_indices also leaves the events' polygons in lon / lat (transform_polygons, index_utils.py:129-184) in pinned host memory, cut and mapped by the device (wb_geom.h); 0: wb_get_event_geometry computes them on the host when asked
    long long geom_seen = 0;                  // vertices of the largest pass's geometry so far (sizes the pinned arrays)
    long long geom_rings_seen = 0;            // ... and its rings
    int trace_fused = 1;                      // 1: the contours of a job by ONE kernel with the job resident in LDS (wb_trace_fused.h); 0: link -> rank -> finish (wb_trace.h), which also take the jobs the fused kernel declines
    int f1_fused = 0;                         // routine 1 of the streamer index inside wb_k_streamer_f234 (0: wb_k_streamer_f1 in front of it)
    int f234_lpts = 0;                        // wb_k_streamer_f234: contour points its LDS holds at most (0: WB_F234_LPTS)
    int trace_nmax = 0, trace_cmax = 0;       // fused kernel: segments / chains per job its LDS admits (0: from the grid's width / 256 chains, then from the largest job the context has seen)
    int fused_seen = 0, fused_seen_W = 0;     // largest segment count of a job in the context's batches so far on the grid of that width
    int fused_seen_c = 0, fused_seen_s = 0;   // ... largest number of chains / of ranking splitters
    int link_rows = 1;                        // wb_k_trace_link: merge by row tables in LDS (0: by binary searches, the path of jobs / grids the tables do not hold)
    int scan_band_rows = 0;                   // streaming scan: rows of squares per band (0: the rows divided evenly over the bands)
    int scan_pad = 0;                         // streaming scan: elements added to the stride of the patch lists' arrays (WbScanArgs.cap_vs)
    int scan_bands = 0;                       // streaming scan: row bands per time step (0: chosen from the grid and the device)
    int polys_grid = 2048;                    // workgroups of wb_k_pack_polys (it writes over PCIe: more of them only hold wave slots)
    int timers = 1;                           // stage timers (WbTimer): 2 every stage, 1 the field scan only (the default since round 6: a pair of event packets per stage and
                                              // the dependency gaps behind them cost 2.7 % of a pass with two batches in flight, 5 % one batch at a time), 0 none
    int keep_device = 0;                      // 1: wb_indices keeps packed event rows + polygon vertices on the device (wb_get_device_tables); 2: the rows only (and, with
                                              // geometry, the rings: wb_get_device_geometry) -- the index polygons stay descriptors, as they do on a single GPU since round 6
    int eager_f2 = 0;                         // 1: chord test for every candidate pair (wb_k_streamer_f2 + f34) instead of the lazy sweep
    int pairs_lds = 1;                        // 1: all-pairs test with the contour in LDS (wb_k_streamer_pairs2) where it applies; 0: tiled kernel only
    int ncu = 0;                              // compute units of the device (found once)
    size_t p2_budget = 0;                     // dynamic LDS a workgroup of wb_k_streamer_pairs2 may use (found once)
    int p2_grid_x2 = 4;                       // persistent grid of wb_k_streamer_pairs2 in half workgroups per CU (4 = two: every wave slot)
    int f1_grid_x2 = 4;                       // ... of wb_k_streamer_f1 (4 = two workgroups of 1024 threads per CU)
    int p2_budget_kb = 0, p2_split = 0;       // development: overrides of that budget / of the workgroups per contour
#if !defined(WB_HOSTSIM)
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;            // side stream: result transfers that overlap the last kernel
    hipStream_t stream3 = nullptr;            // option scan_priority: a stream of the device's highest priority for the field scan (made on first use)
    hipEvent_t fork3 = nullptr, join3 = nullptr;
    hipStream_t stream4 = nullptr;            // option boxcom_cu_mask: a CU-masked stream for wb_k_box_com (made on first use)
    hipEvent_t fork4 = nullptr, join4 = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    hipEvent_t scan_done = nullptr;           // recorded behind this context's field scan (option scan_exclusive)
    hipStream_t settle_stream = nullptr;      // wb_geom_settle: the tail of a pass's geometry that the DMA copy behind its kernels did not cover (made on first use)
    std::mutex settle_mu;
    // caching device allocator: hipMalloc after hipFree of GB-sized blocks costs 100s of ms on this stack
    // (measured: 0.7 ms for the first pass's ~40 allocations, 350 ms for every later pass), so blocks are
    // kept for the life of the context and handed out again by size
    std::multimap<size_t, void*> pool;        // free blocks by size
    std::map<void*, size_t> live;             // blocks handed out
    // pinned staging buffer for large device<->host copies.  Copying straight into pageable memory makes the
    // runtime register those pages with the GPU; when the caller frees them the kernel driver's MMU notifier
    // evicts and restores the process's queues (measured: ~20 ms stall on the next dispatch).
    void* pin = nullptr;
    size_t pin_bytes = 0;
    void* zq_p[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // queued zero fills (dzero_q / dzero_flush)
    size_t zq_bytes[6] = {0, 0, 0, 0, 0, 0};
    int zq_n = 0;
    void* h2d_pin[2] = {nullptr, nullptr};    // double-buffered staging of large uploads
    hipEvent_t h2d_ev[2] = {nullptr, nullptr};
#endif
    // per-grid device tables (trigonometric tables, area weights, the all-pairs reach per latitude row): built once per
    // (lat, lon, resolution, geo_dis) and shared by every batch of the context
    struct Tables {
        std::vector<double> lat, lon;
        double dlon = 0, dlat = 0, geo_dis = 0;
        double* base = nullptr;
        size_t o_lat = 0, o_latrad = 0, o_sphi2 = 0, o_cphi2 = 0, o_cphi = 0, o_wlat = 0, o_lonrad = 0, o_slam2 = 0, o_clam2 = 0, o_lonval = 0;
        int* dxlim = nullptr;
        double* on_tab = nullptr;             // [nlat][nlon][4] distance (km) of a grid point to its E, N, NE, NW neighbour: sklearn's
                                              // haversine evaluated with the HOST's math library (wb_k_streamer_prep)
        int dy_max = 0;
        double hthr = 0;
        // wb_k_streamer_pairs2: sin^2 of the multiples of dlat / 2 and dlon / 2 (host math library), whether the grid is the
        // regular global one the tables stand for, and the absolute guard band on h that covers what its coordinates
        // deviate from lat[0] + j dlat / lon[0] + i dlon
        double* sl_tab = nullptr;
        double* sx_tab = nullptr;
        bool p2_ok = false;
        double geo_band = 0;
    };
    std::vector<Tables*> tables;
    WbCtl* h_ctl = nullptr;                   // pinned: where the control block of a batch lands (two snapshots)
    WbRingSink geo;                           // result of the last wb_transform_polygons
#if !defined(WB_HOSTSIM)
    WbComm* comm = nullptr;                   // RCCL communicator of a time-sharded run (wb_comm_init)
#endif
};

#include <chrono>
static double wb_now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static bool wb_debug() { static int v = -1; if (v < 0) v = getenv("WB_DEBUG_TIMING") ? 1 : 0; return v == 1; }
#define WB_TICK(label) do { if (wb_debug()) { double t_ = wb_now_ms(); fprintf(stderr, "[wb] %-28s %9.3f ms\n", label, t_ - wb_t0_); wb_t0_ = t_; } } while (0)

// Host buffers of the big result arrays (polygon vertices).  Product build: PINNED memory, so the device-to-host copy
// lands in them directly (through the staging buffer it costs a second pass over the data on the host, ~0.3 ms per
// array and batch); pinning is expensive, so blocks are rounded up to a power of two and recycled by the context.
static void* hmalloc(wb_ctx* c, size_t bytes, size_t* got) {
    size_t want = (size_t)1 << 20;
    while (want < bytes) want <<= 1;
    *got = want;
    std::lock_guard<std::mutex> lk(c->alloc_mu);
    auto it = c->hpool.find(want);
    if (it != c->hpool.end()) {
        void* p = it->second;
        c->hpool.erase(it);
        return p;
    }
#if defined(WB_HOSTSIM)
    return malloc(want);
#else
    void* p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
#endif
}
static void hrelease(void* p) {
#if defined(WB_HOSTSIM)
    free(p);
#else
    (void)hipHostFree(p);
#endif
}
static void hfree(wb_ctx* c, void* p, size_t bytes) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(c->alloc_mu);
    if (c->hpool.size() < 24) c->hpool.insert({bytes, p}); else hrelease(p);      // (24: the three index calls on a contour frame use three size classes of vertex arrays each; pinning a fresh 16-64 MB block costs tens of ms)
}

// every entry point that touches the device selects the context's device first: the current device is per host thread
// (worker threads of a pipeline, ranks driving GPUs other than 0)
static inline void wb_use(wb_ctx* c) {
#if !defined(WB_HOSTSIM)
    if (c) (void)hipSetDevice(c->device);
#else
    (void)c;
#endif
}

static int wb_fail(wb_ctx* c, const std::string& m) {
    if (c) c->err = m;
    return 1;
}

#if defined(WB_HOSTSIM)
static void* dmalloc(wb_ctx*, size_t bytes) { return malloc(bytes ? bytes : 1); }
static void dfree(wb_ctx*, void* p) { free(p); }
static int dmemset(wb_ctx*, void* p, int v, size_t bytes) { memset(p, v, bytes); return 0; }
static void dzero_q(wb_ctx*, void* p, size_t bytes) { memset(p, 0, bytes); }     // (product build: queued, one launch at dzero_flush)
static void dzero_flush(wb_ctx*) {}
static int h2d(wb_ctx*, void* d, const void* s, size_t b) { memcpy(d, s, b); return 0; }
static int d2h(wb_ctx*, void* d, const void* s, size_t b) { memcpy(d, s, b); return 0; }
static int d2h_pinned(wb_ctx*, void* d, const void* s, size_t b) { memcpy(d, s, b); return 0; }
static int dsync(wb_ctx*) { return 0; }
#else
static void* dmalloc(wb_ctx* c, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    std::lock_guard<std::mutex> lk(c->alloc_mu);
    auto it = c->pool.lower_bound(bytes);
    if (it != c->pool.end() && it->first <= bytes + bytes / 4 + 4096) {   // close enough: reuse
        void* p = it->second;
        c->live[p] = it->first;
        c->pool.erase(it);
        return p;
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        // out of memory: give the cached blocks back and try once more
        for (auto& kv : c->pool) (void)hipFree(kv.second);
        c->pool.clear();
        if (hipMalloc(&p, bytes) != hipSuccess) {
            c->err = "hipMalloc failed (" + std::to_string(bytes) + " bytes)";
            return nullptr;
        }
    }
    c->live[p] = bytes;
    return p;
}
static void dfree(wb_ctx* c, void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(c->alloc_mu);
    auto it = c->live.find(p);
    if (it == c->live.end()) { (void)hipFree(p); return; }
    c->pool.insert({it->second, p});
    c->live.erase(it);
}
// zero / byte fill as our own kernel on the context's stream (hipMemsetAsync showed ~20 ms stalls on the first
// call of every pass on this stack)
__global__ void wb_k_fill(unsigned char* p, int v, size_t bytes) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < bytes; i += stride) p[i] = (unsigned char)v;
}
static int dmemset(wb_ctx* c, void* p, int v, size_t bytes) {
    if (!bytes) return 0;
    size_t blocks = (bytes + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(wb_k_fill, dim3((unsigned)blocks), dim3(256), 0, c->stream, (unsigned char*)p, v, bytes);
    return 0;
}
// several zero fills in ONE launch: dzero_q queues (pointer, bytes), dzero_flush launches (grid.y = the queued segments).
// Phase 1 clears four small arrays before its first kernel; as four launches they were four dispatches on the stream.
#define WB_ZERO_SEGS 6
struct WbZeroSegs { unsigned char* p[WB_ZERO_SEGS]; size_t bytes[WB_ZERO_SEGS]; };
__global__ void wb_k_zero_segs(WbZeroSegs z) {
    unsigned char* p = z.p[blockIdx.y];
    const size_t bytes = z.bytes[blockIdx.y];
    const size_t stride = (size_t)gridDim.x * blockDim.x * 16, head = ((16 - ((uintptr_t)p & 15)) & 15) < bytes ? ((16 - ((uintptr_t)p & 15)) & 15) : bytes;
    const size_t body = (bytes - head) & ~(size_t)15;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < body; i += stride) *(uint4*)(p + head + i) = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0) {
        for (size_t i = threadIdx.x; i < head; i += blockDim.x) p[i] = 0;
        for (size_t i = head + body + threadIdx.x; i < bytes; i += blockDim.x) p[i] = 0;
    }
}
static void dzero_flush(wb_ctx* c);
static void dzero_q(wb_ctx* c, void* p, size_t bytes) {
    if (!bytes) return;
    if (c->zq_n == WB_ZERO_SEGS) dzero_flush(c);
    c->zq_p[c->zq_n] = p; c->zq_bytes[c->zq_n] = bytes; ++c->zq_n;
}
static void dzero_flush(wb_ctx* c) {
    if (!c->zq_n) return;
    WbZeroSegs z;
    size_t mx = 0;
    for (int k = 0; k < WB_ZERO_SEGS; ++k) {
        z.p[k] = k < c->zq_n ? (unsigned char*)c->zq_p[k] : nullptr; z.bytes[k] = k < c->zq_n ? c->zq_bytes[k] : 0;
        if (z.bytes[k] > mx) mx = z.bytes[k];
    }
    // few workgroups, each with up to 64 stores per thread: the launch is over when its LAST workgroup has found a place, and
    // beside the other batch's persistent kernels a hundred workgroups of 4 KB each waited 0.3 ms on average for theirs
    size_t blocks = (mx + 256 * 16 * 64 - 1) / (256 * 16 * 64);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wb_k_zero_segs, dim3((unsigned)blocks, (unsigned)c->zq_n), dim3(256), 0, c->stream, z);
    c->zq_n = 0;
}
#define WB_PIN_THRESHOLD (256u << 10)
static void* wb_pin(wb_ctx* c, size_t bytes) {
    if (c->pin_bytes >= bytes) return c->pin;
    if (c->pin) (void)hipHostFree(c->pin);
    c->pin = nullptr;
    c->pin_bytes = 0;
    size_t want = (size_t)32 << 20;
    while (want < bytes && want < ((size_t)256 << 20)) want <<= 1;
    if (hipHostMalloc(&c->pin, want, hipHostMallocDefault) != hipSuccess) { c->pin = nullptr; return nullptr; }
    c->pin_bytes = want;
    return c->pin;
}
// Host -> device copy of a large pageable buffer: chunks go through TWO pinned staging buffers, the host memcpy of chunk
// k + 1 (split over a few threads: one core copies ~10 GB/s, PCIe Gen5 takes ~55) overlapping the DMA of chunk k.
#define WB_H2D_CHUNK ((size_t)64 << 20)
#define WB_H2D_THREADS 8
static int wb_h2d_threads() {
    // one core copies ~10 GB/s into the staging buffer, the link takes ~55: eight copiers where the host has the cores
    // (WB_H2D_THREADS overrides)
    static int n = 0;
    if (!n) {
        const char* e = getenv("WB_H2D_THREADS");
        int v = e ? atoi(e) : (int)std::thread::hardware_concurrency() / 2;
        n = v < 1 ? 1 : (v > WB_H2D_THREADS ? WB_H2D_THREADS : v);
    }
    return n;
}
static void wb_par_memcpy(char* dst, const char* src, size_t n) {
    const int nth = wb_h2d_threads();
    if (n < ((size_t)8 << 20) || nth == 1) { memcpy(dst, src, n); return; }
    std::thread th[WB_H2D_THREADS - 1];
    const size_t part = (n / nth + 4095) & ~(size_t)4095;
    for (int t = 1; t < nth; ++t) {
        const size_t o = (size_t)t * part;
        if (o >= n) break;
        const size_t m = n - o < part ? n - o : part;
        th[t - 1] = std::thread([=]() { memcpy(dst + o, src + o, m); });
    }
    memcpy(dst, src, part < n ? part : n);
    for (int t = 1; t < nth; ++t) if (th[t - 1].joinable()) th[t - 1].join();
}
static int h2d(wb_ctx* c, void* d, const void* s, size_t b) {
    if (b > WB_PIN_THRESHOLD) {
        if (!c->h2d_pin[0]) {
            for (int k = 0; k < 2; ++k) {
                if (hipHostMalloc(&c->h2d_pin[k], WB_H2D_CHUNK, hipHostMallocDefault) != hipSuccess) { c->h2d_pin[k] = nullptr; break; }
                (void)hipEventCreateWithFlags(&c->h2d_ev[k], hipEventDisableTiming);
            }
        }
        if (c->h2d_pin[0] && c->h2d_pin[1]) {
            int k = 0;
            bool used[2] = {false, false};
            for (size_t o = 0; o < b; o += WB_H2D_CHUNK, k ^= 1) {
                const size_t n = b - o < WB_H2D_CHUNK ? b - o : WB_H2D_CHUNK;
                if (used[k] && hipEventSynchronize(c->h2d_ev[k]) != hipSuccess) return 1;      // the buffer's previous DMA is done
                wb_par_memcpy((char*)c->h2d_pin[k], (const char*)s + o, n);
                if (hipMemcpyAsync((char*)d + o, c->h2d_pin[k], n, hipMemcpyHostToDevice, c->stream) != hipSuccess) return 1;
                (void)hipEventRecord(c->h2d_ev[k], c->stream);
                used[k] = true;
            }
            return hipStreamSynchronize(c->stream) != hipSuccess;
        }
    }
    return hipMemcpyAsync(d, s, b, hipMemcpyHostToDevice, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess;
}
static int d2h(wb_ctx* c, void* d, const void* s, size_t b) {
    if (b > WB_PIN_THRESHOLD && wb_pin(c, b < ((size_t)256 << 20) ? b : ((size_t)256 << 20))) {
        for (size_t o = 0; o < b; o += c->pin_bytes) {
            const size_t n = b - o < c->pin_bytes ? b - o : c->pin_bytes;
            if (hipMemcpyAsync(c->pin, (const char*)s + o, n, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess) return 1;
            memcpy((char*)d + o, c->pin, n);
        }
        return 0;
    }
    return hipMemcpyAsync(d, s, b, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess;
}
// destination is pinned host memory: no staging
static int d2h_pinned(wb_ctx* c, void* d, const void* s, size_t b) {
    return hipMemcpyAsync(d, s, b, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess;
}
static int dsync(wb_ctx* c) {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { c->err = std::string("HIP error: ") + hipGetErrorString(e); return 1; }
    return 0;
}
#endif

// asynchronous device-to-host copy into PINNED memory on the context's current stream (no synchronisation)
static int d2h_async(wb_ctx* c, void* dst_pinned, const void* src, size_t bytes) {
    if (!bytes) return 0;
#if defined(WB_HOSTSIM)
    (void)c;
    memcpy(dst_pinned, src, bytes);
    return 0;
#else
    return hipMemcpyAsync(dst_pinned, src, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess;
#endif
}
static WbCtl* host_ctl(wb_ctx* c) {
    if (c->h_ctl) return c->h_ctl;
#if defined(WB_HOSTSIM)
    c->h_ctl = (WbCtl*)calloc(2, sizeof(WbCtl));
#else
    void* q = nullptr;
    if (hipHostMalloc(&q, 2 * sizeof(WbCtl), hipHostMallocDefault) != hipSuccess) return nullptr;
    c->h_ctl = (WbCtl*)q;
#endif
    return c->h_ctl;
}
// side stream: the launches between side_begin and side_end run next to whatever follows on the main stream
static void side_begin(wb_ctx* ctx) {
#if !defined(WB_HOSTSIM)
    (void)hipEventRecord(ctx->fork, ctx->stream);
    (void)hipStreamWaitEvent(ctx->stream2, ctx->fork, 0);
    std::swap(ctx->stream, ctx->stream2);
#else
    (void)ctx;
#endif
}
// the fork in two steps: side_mark records the point of the main stream the side stream will start from, side_begin_marked -- after
// MORE has been launched on the main stream -- starts the side section there (what was launched in between does not hold it up)
static void side_mark(wb_ctx* ctx) {
#if !defined(WB_HOSTSIM)
    (void)hipEventRecord(ctx->fork, ctx->stream);
#else
    (void)ctx;
#endif
}
static void side_begin_marked(wb_ctx* ctx) {
#if !defined(WB_HOSTSIM)
    (void)hipStreamWaitEvent(ctx->stream2, ctx->fork, 0);
    std::swap(ctx->stream, ctx->stream2);
#else
    (void)ctx;
#endif
}
static void side_end(wb_ctx* ctx) {
#if !defined(WB_HOSTSIM)
    (void)hipEventRecord(ctx->join, ctx->stream);
    std::swap(ctx->stream, ctx->stream2);
#else
    (void)ctx;
#endif
}
// ... back to the main stream WITHOUT moving the join event: what was launched on the side stream since the last side_begin is
// waited for by nobody on the device (the caller has recorded an event of its own behind it: the deferred geometry kernels)
static void side_leave(wb_ctx* ctx) {
#if !defined(WB_HOSTSIM)
    std::swap(ctx->stream, ctx->stream2);
#else
    (void)ctx;
#endif
}
#if !defined(WB_HOSTSIM)
static bool wb_masked_stream(hipStream_t* out, int first, int count);
#endif
// option "boxcom_cu_mask": what is launched between boxmask_begin and boxmask_end (inside side_begin / side_end) goes to a
// CU-masked stream, in stream order with the side stream before and behind it
static bool boxmask_begin(wb_ctx* ctx) {
#if !defined(WB_HOSTSIM)
    if (ctx->boxcom_cu_count <= 0) return false;
    if (!ctx->stream4) {
        if (!wb_masked_stream(&ctx->stream4, ctx->boxcom_cu_first, ctx->boxcom_cu_count) ||
            hipEventCreateWithFlags(&ctx->fork4, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->join4, hipEventDisableTiming) != hipSuccess) {
            ctx->boxcom_cu_count = 0;
            return false;
        }
    }
    (void)hipEventRecord(ctx->fork4, ctx->stream);
    (void)hipStreamWaitEvent(ctx->stream4, ctx->fork4, 0);
    std::swap(ctx->stream, ctx->stream4);
    return true;
#else
    (void)ctx;
    return false;
#endif
}
static void boxmask_end(wb_ctx* ctx) {
#if !defined(WB_HOSTSIM)
    (void)hipEventRecord(ctx->join4, ctx->stream);
    std::swap(ctx->stream, ctx->stream4);
    (void)hipStreamWaitEvent(ctx->stream, ctx->join4, 0);
#else
    (void)ctx;
#endif
}
static void side_join(wb_ctx* ctx) {          // the main stream waits for the side stream's last side_end
#if !defined(WB_HOSTSIM)
    (void)hipStreamWaitEvent(ctx->stream, ctx->join, 0);
#else
    (void)ctx;
#endif
}
// A stream whose kernels run on the compute units [first, first + count) only (hipExtStreamCreateWithCUMask).  Two batches in
// flight otherwise share the device by whole kernels, each sized to fill every wave slot: partitioning is the alternative
// the round-4 review asked to be measured ("CU masks").
#if !defined(WB_HOSTSIM)
static bool wb_masked_stream(hipStream_t* out, int first, int count) {
    uint32_t mask[16];
    memset(mask, 0, sizeof(mask));
    for (int cu = first; cu < first + count && cu < 512; ++cu) if (cu >= 0) mask[cu >> 5] |= 1u << (cu & 31);
    return hipExtStreamCreateWithCUMask(out, 16, mask) == hipSuccess;
}
// option "cu_mask": the context's two streams are made again, masked (count == 0: plain streams)
static bool wb_remask_streams(wb_ctx* c, int first, int count) {
    if (hipSetDevice(c->device) != hipSuccess) return false;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    hipStream_t a = nullptr, b = nullptr;
    bool ok;
    if (count > 0) ok = wb_masked_stream(&a, first, count) && wb_masked_stream(&b, first, count);
    else ok = hipStreamCreate(&a) == hipSuccess && hipStreamCreateWithFlags(&b, hipStreamNonBlocking) == hipSuccess;
    if (!ok) { if (a) (void)hipStreamDestroy(a); if (b) (void)hipStreamDestroy(b); return false; }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    c->stream = a; c->stream2 = b;
    c->cu_first = first; c->cu_count = count;
    c->ncu = 0;                               // (found again: wb_num_cus)
    return true;
}
#endif
// option "scan_priority": the launches between prio_begin and prio_end go to a stream of the device's highest priority, in
// stream order with what came before and what follows on the main stream.  The field scan is one fat workgroup per CU (12
// waves of 128 registers): beside another batch's thin workgroups the dispatcher otherwise hands every freed slot to those.
static void prio_begin(wb_ctx* ctx) {
#if !defined(WB_HOSTSIM)
    if (!ctx->scan_priority) return;
    if (!ctx->stream3) {
        int lo = 0, hi = 0;
        bool ok;
        if (ctx->scan_cu_count > 0) ok = wb_masked_stream(&ctx->stream3, ctx->scan_cu_first, ctx->scan_cu_count);
        else ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hipStreamCreateWithPriority(&ctx->stream3, hipStreamNonBlocking, hi) == hipSuccess;
        if (!ok || hipEventCreateWithFlags(&ctx->fork3, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->join3, hipEventDisableTiming) != hipSuccess) {
            ctx->scan_priority = 0;
            return;
        }
    }
    (void)hipEventRecord(ctx->fork3, ctx->stream);
    (void)hipStreamWaitEvent(ctx->stream3, ctx->fork3, 0);
    std::swap(ctx->stream, ctx->stream3);
#else
    (void)ctx;
#endif
}
static void prio_end(wb_ctx* ctx) {
#if !defined(WB_HOSTSIM)
    if (!ctx->scan_priority || !ctx->stream3) return;
    (void)hipEventRecord(ctx->join3, ctx->stream);
    std::swap(ctx->stream, ctx->stream3);
    (void)hipStreamWaitEvent(ctx->stream, ctx->join3, 0);
#else
    (void)ctx;
#endif
}
static int side_sync(wb_ctx* ctx) {           // the HOST waits for the side stream
#if !defined(WB_HOSTSIM)
    return hipStreamSynchronize(ctx->stream2) != hipSuccess;
#else
    (void)ctx;
    return 0;
#endif
}

// option "scan_exclusive": the HBM-bound field scans of the contexts of one device run one after the other -- ordered ON THE
// DEVICE (a stream waits for the event behind the scan launched before it; no host thread waits) -- so that a scan shares the
// memory system with the other batch's latency-bound kernels instead of with its twin
#if !defined(WB_HOSTSIM)
#include <mutex>
static std::mutex wb_scan_gate_mutex;
static hipEvent_t wb_scan_gate_last[16] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
static void wb_scan_gate_enter(wb_ctx* c) {
    if (!c->scan_exclusive || c->device < 0 || c->device >= 16) return;
    std::lock_guard<std::mutex> g(wb_scan_gate_mutex);
    hipEvent_t last = wb_scan_gate_last[c->device];
    if (last && last != c->scan_done) (void)hipStreamWaitEvent(c->stream, last, 0);
}
static void wb_scan_gate_leave(wb_ctx* c) {
    if (!c->scan_exclusive || c->device < 0 || c->device >= 16) return;
    std::lock_guard<std::mutex> g(wb_scan_gate_mutex);
    if (!c->scan_done && hipEventCreateWithFlags(&c->scan_done, hipEventDisableTiming) != hipSuccess) { c->scan_done = nullptr; return; }
    (void)hipEventRecord(c->scan_done, c->stream);
    wb_scan_gate_last[c->device] = c->scan_done;
}
static void wb_scan_gate_forget(wb_ctx* c) {
    std::lock_guard<std::mutex> g(wb_scan_gate_mutex);

    if (c->device >= 0 && c->device < 16 && wb_scan_gate_last[c->device] == c->scan_done) wb_scan_gate_last[c->device] = nullptr;
    if (c->scan_done) (void)hipEventDestroy(c->scan_done);
    c->scan_done = nullptr;
}
#else
static void wb_scan_gate_enter(wb_ctx*) {}
static void wb_scan_gate_leave(wb_ctx*) {}
static void wb_scan_gate_forget(wb_ctx*) {}
#endif

// stage timer: HIP events on the context's stream (host simulation: zeros)
struct WbTimer {
    std::vector<std::string> names;
    std::vector<float> ms;
#if !defined(WB_HOSTSIM)
    std::vector<hipEvent_t> ev0, ev1;   // parallel to names; nullptr once collected
#endif
    void begin(wb_ctx* c, const char* name) {
        names.push_back(name);
        ms.push_back(0.f);
#if !defined(WB_HOSTSIM)
        hipEvent_t a = nullptr, b = nullptr;
        // option "timers": 2 every stage, 1 the scan kernel only (the default: what the roofline is quoted on), 0 none -- an event
        // pair is two more packets on the stream per stage; the stage keeps its name and reads 0 ms
        if (c->timers >= 2 || (c->timers == 1 && !strcmp(name, "scan"))) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipEventRecord(a, c->stream);
        }
        ev0.push_back(a);
        ev1.push_back(b);
#else
        (void)c;
#endif
    }
    void end(wb_ctx* c) {
#if !defined(WB_HOSTSIM)
        if (ev1.back()) (void)hipEventRecord(ev1.back(), c->stream);
#else
        (void)c;
#endif
    }
    void drop_last() {   // forget the most recent stage (a retried launch)
#if !defined(WB_HOSTSIM)
        if (ev0.back()) { (void)hipEventSynchronize(ev1.back()); (void)hipEventDestroy(ev0.back()); (void)hipEventDestroy(ev1.back()); }
        ev0.pop_back();
        ev1.pop_back();
#endif
        names.pop_back();
        ms.pop_back();
    }
    // wait = false: stages whose end event has not been reached yet (kernels nobody waited for: the deferred geometry) keep
    // their events and are collected by a later call
    void collect(bool wait = true) {
#if !defined(WB_HOSTSIM)
        for (size_t i = 0; i < ev0.size(); ++i) {
            if (!ev0[i]) continue;
            if (!wait && hipEventQuery(ev1[i]) != hipSuccess) { (void)hipGetLastError(); continue; }
            (void)hipEventSynchronize(ev1[i]);
            float t = 0.f;
            (void)hipEventElapsedTime(&t, ev0[i], ev1[i]);
            ms[i] = t;
            (void)hipEventDestroy(ev0[i]);
            (void)hipEventDestroy(ev1[i]);
            ev0[i] = nullptr;
            ev1[i] = nullptr;
        }
#endif
    }
    void truncate(size_t n) {               // forget the stages after the first n (a repeated phase)
        collect();
        if (n >= names.size()) return;
        names.resize(n);
        ms.resize(n);
#if !defined(WB_HOSTSIM)
        ev0.resize(n);
        ev1.resize(n);
#endif
    }
    void clear() {
        collect();
        names.clear();
        ms.clear();
#if !defined(WB_HOSTSIM)
        ev0.clear();
        ev1.clear();
#endif
    }
};

// ------------------------------------------------------------------------------------------------ small differences
static const char* wb_backend_name() {
#if defined(WB_HOSTSIM)
    return "hostsim";
#else
    return "hip";
#endif
}
// the calling thread's current device <- the context's (the current device is per host thread)
static bool wb_select_device(wb_ctx* c) {
#if defined(WB_HOSTSIM)
    (void)c;
    return true;
#else
    return hipSetDevice(c->device) == hipSuccess;
#endif
}
// stream, side stream and events of a new context; false: no such device
static bool wb_platform_create(wb_ctx* c) {
#if defined(WB_HOSTSIM)
    (void)c;
    return true;
#else
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= c->device) return false;
    if (hipSetDevice(c->device) != hipSuccess) return false;
    // WB_SIDE_PRIORITY=1 (development): the side stream at the lowest priority the device offers
    int lo = 0, hi = 0;
    const char* sp = getenv("WB_SIDE_PRIORITY");
    const bool low = sp && atoi(sp) == 1 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;
    return hipStreamCreate(&c->stream) == hipSuccess &&
           (low ? hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, lo) : hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking)) == hipSuccess &&
           hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&c->join, hipEventDisableTiming) == hipSuccess;
#endif
}
static void wb_platform_destroy(wb_ctx* c) {
    if (c->h_ctl) {
#if defined(WB_HOSTSIM)
        free(c->h_ctl);
#else
        (void)hipHostFree(c->h_ctl);
#endif
    }
#if !defined(WB_HOSTSIM)
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    wb_scan_gate_forget(c);
    for (auto& kv : c->pool) (void)hipFree(kv.second);
    for (auto& kv : c->live) (void)hipFree(kv.first);
    if (c->pin) (void)hipHostFree(c->pin);
    for (int k = 0; k < 2; ++k) { if (c->h2d_pin[k]) (void)hipHostFree(c->h2d_pin[k]); if (c->h2d_ev[k]) (void)hipEventDestroy(c->h2d_ev[k]); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream4) (void)hipStreamDestroy(c->stream4);
    if (c->settle_stream) (void)hipStreamDestroy(c->settle_stream);
    if (c->fork4) (void)hipEventDestroy(c->fork4);
    if (c->join4) (void)hipEventDestroy(c->join4);
    if (c->fork3) (void)hipEventDestroy(c->fork3);
    if (c->join3) (void)hipEventDestroy(c->join3);
    if (c->fork) (void)hipEventDestroy(c->fork);
    if (c->join) (void)hipEventDestroy(c->join);
#endif
}
// what follows goes to the side stream / back to the main one, host side only (side_begin / side_end also order the two)
static void wb_swap_streams(wb_ctx* c) {
#if defined(WB_HOSTSIM)
    (void)c;
#else
    std::swap(c->stream, c->stream2);
#endif
}
// LDS a workgroup may use by default / a CU holds
static void wb_lds_limits(wb_ctx* c, size_t* per_block, size_t* per_cu) {
    *per_block = 64 << 10; *per_cu = 160 << 10;
#if !defined(WB_HOSTSIM)
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) {
        *per_block = prop.sharedMemPerBlock;
        *per_cu = prop.maxSharedMemoryPerMultiProcessor > prop.sharedMemPerBlock ? prop.maxSharedMemoryPerMultiProcessor : prop.sharedMemPerBlock;
    }
#else
    (void)c;
#endif
}
// device time of one operation (wb_last_op_ms): HIP events on the context's stream; zero in the host simulation
struct WbOpTimer {
    wb_ctx* c;
#if !defined(WB_HOSTSIM)
    hipEvent_t e0 = nullptr, e1 = nullptr;
#endif
    explicit WbOpTimer(wb_ctx* ctx) : c(ctx) {
#if !defined(WB_HOSTSIM)
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, c->stream);
#endif
    }
    void stop() {
#if !defined(WB_HOSTSIM)
        (void)hipEventRecord(e1, c->stream);
        (void)hipEventSynchronize(e1);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e0, e1);
        c->last_op_ms = t;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        e0 = e1 = nullptr;
#else
        c->last_op_ms = 0.f;
#endif
    }
};
