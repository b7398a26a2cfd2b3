// wb_spmd.h -- the thin SPMD layer every kernel body is written against.
//
// Product build (hipcc, gfx950): a kernel is a __global__ function, WB_PFOR is a
// block-strided loop followed by __syncthreads(), atomics are HIP atomics, scans use
// 64-wide wave shuffles + LDS.
//
// Host-simulation build (-DWB_HOSTSIM, g++; one variant with -fsanitize=address,undefined run by
// tests/test_hostsim_sanitizers.py; TEST INFRASTRUCTURE ONLY, built into tests/_hostsim/ and never
// loaded by the package): a kernel is a plain
// function run once per block index, WB_PFOR runs its iterations one after another in
// ascending -- or, with WB_HOSTSIM_REVERSE, descending -- order (a phase whose result
// depends on that order has a race), atomics are plain read-modify-writes.  This is how
// the irregular kernels are debugged in the GPU-less build container.
#pragma once
#include <stdint.h>

#if defined(WB_HOSTSIM)
// ---------------------------------------------------------------- host simulation
#include <math.h>
#include <string.h>
#include <stdlib.h>
#define WB_HD
#define WB_DEV
#define WB_NOINLINE
#define WB_LAMBDA_INLINE
#define WB_KERNEL(name, ...) void name(int wb_block_id, int wb_grid_dim, __VA_ARGS__)
#define WB_KERNEL_LB(name, threads, waves, ...) void name(int wb_block_id, int wb_grid_dim, __VA_ARGS__)
#define WB_BLOCK_ID wb_block_id
#define WB_GRID_DIM wb_grid_dim
#define WB_NT 1
#define WB_TID 0
extern int wb_hostsim_reverse;
extern long long wb_hostsim_path[4];
#define WB_HOSTSIM_PATH(k) (++wb_hostsim_path[k])          // 0: routine 1 in parts, 1: routine 1 with the table in global memory
#define WB_PFOR(i, n)                                                                    \
    for (int64_t wb_k_##i = 0, wb_n_##i = (int64_t)(n); wb_k_##i < wb_n_##i; ++wb_k_##i) { \
        const int64_t i = wb_hostsim_reverse ? (wb_n_##i - 1 - wb_k_##i) : wb_k_##i;
#define WB_END_PFOR }
// the same with a 32-bit index (extents that fit an int: a per-job kernel's loops; the 64-bit index costs two instructions per
// add / compare / address on the device)
#define WB_PFOR32(i, n)                                                          \
    for (int wb_k_##i = 0, wb_n_##i = (int)(n); wb_k_##i < wb_n_##i; ++wb_k_##i) { \
        const int i = wb_hostsim_reverse ? (wb_n_##i - 1 - wb_k_##i) : wb_k_##i;
#define WB_PFOR_B32(i, ic, u, n)                                                 \
    for (int wb_k_##i = 0, wb_n_##i = (int)(n); wb_k_##i < wb_n_##i; ++wb_k_##i) { \
        const int i = wb_hostsim_reverse ? (wb_n_##i - 1 - wb_k_##i) : wb_k_##i;   \
        const int ic = i;                                                          \
        const int u = 0;                                                           \
        (void)ic; (void)u;                                                         \
        {
// WB_PFOR_B(i, ic, u, n) <requests> WB_PFOR_B_USE(i, u) <the rest> WB_END_PFOR_B: a parallel loop whose body is cut in two -- the
// first part only REQUESTS what element i needs (into arrays indexed [u], WB_PFOR_B_WIDTH long, declared in front of the loop;
// ic = i clamped to n - 1), the second part uses it.  On the device a thread requests for WB_PFOR_B_WIDTH of its elements before it
// uses any: one memory latency per WB_PFOR_B_WIDTH elements instead of one per element (the compiler does not move the loads
// of a later iteration of WB_PFOR in front of the branches of an earlier one).  Here: one element after the other, u = 0.
#define WB_PFOR_B_WIDTH 4
#define WB_PFOR_B(i, ic, u, n)                                                            \
    for (int64_t wb_k_##i = 0, wb_n_##i = (int64_t)(n); wb_k_##i < wb_n_##i; ++wb_k_##i) { \
        const int64_t i = wb_hostsim_reverse ? (wb_n_##i - 1 - wb_k_##i) : wb_k_##i;       \
        const int64_t ic = i;                                                              \
        const int u = 0;                                                                   \
        (void)ic; (void)u;                                                                 \
        {
#define WB_PFOR_B_USE(i, u) } {
#define WB_PFOR_B32_USE(i, u) } {
#define WB_END_PFOR_B } }
#define WB_SINGLE {
#define WB_END_SINGLE }
#define WB_SYNC() ((void)0)
#define WB_WAVE 1                     // lanes that run in lockstep (work a wave does by itself, between two workgroup barriers)
static inline bool wb_wave_any(bool v) { return v; }
#define WB_SHARED(type, name, n) type name[n]
// arrays in the launch's dynamic LDS (WB_LAUNCH_LDS), carved in the order of their declarations, 8-byte aligned
#include <vector>
#define WB_DYN_SHARED_BEGIN
#define WB_DYN_SHARED(type, name, n) std::vector<type> name##_store((size_t)(n) + 1); type* name = name##_store.data()
template <class T> static inline T wb_atomic_add(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T wb_atomic_min(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T wb_atomic_max(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T wb_atomic_or(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T wb_atomic_and(T* p, T v) { T o = *p; *p = o & v; return o; }
template <class T> static inline T wb_atomic_cas(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
static inline int wb_popc32(uint32_t v) { return __builtin_popcount(v); }
static inline int wb_ctz32(uint32_t v) { return __builtin_ctz(v); }
static inline int wb_clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }
static inline int wb_popc64(uint64_t v) { return __builtin_popcountll(v); }
// base[idx] += val (no return value); on the GPU lanes of a wave that hit the same idx are combined first
static inline void wb_atomic_add_agg(int* base, int idx, int val) { base[idx] += val; }
static inline void wb_atomic_add_wave(int* base, int idx, int val) { base[idx] += val; }
static inline void wb_atomic_min_first(int* addr, bool want, int val) { if (want && val < *addr) *addr = val; }
// the block's sum of v, returned to every thread, added in a fixed order (butterfly inside a wave, then the waves in turn):
// the same bits on every run; lw: LDS scratch of 16 doubles; all threads of the block call it
static inline double wb_block_sum_f64(double v, double* lw) { (void)lw; return v; }
// every thread's v added to *acc (LDS) -- one atomic per wave; all threads of the block call it, the sum stands after it
static inline void wb_block_add_u64(unsigned long long* acc, unsigned long long v) { *acc += v; }
// slot of an append to a list whose length counter all calling lanes share (-1 for lanes with want == false)
static inline int wb_append_slot(int* counter, bool want) { return want ? (*counter)++ : -1; }
// approximate reciprocal (the device's v_rcp_f32: 1 ulp); for decisions that carry their own margin only
static inline float wb_rcp_f32(float v) { return 1.0f / v; }
// first slot of n (0 <= n < 32) consecutive entries appended by the calling lane
static inline int wb_append_slots(int* counter, int n) { const int b = *counter; *counter += n; return b; }

// in-place exclusive scan of a[0..n); returns the total
static inline int wb_block_scan_excl(int* a, int n) {
    int run = 0;
    for (int i = 0; i < n; ++i) { int v = a[i]; a[i] = run; run += v; }
    return run;
}

// in-place exclusive running maximum of a[0..n): a[i] <- max(init, a[0..i-1])
static inline void wb_block_scan_excl_max(int* a, int n, int init) {
    int run = init;
    for (int i = 0; i < n; ++i) { int v = a[i]; a[i] = run; if (v > run) run = v; }
}

// the RECORDS of a[lo..hi) above `floor`, in ascending order of their index: the i with a[i] > max(floor, a[lo..i)).
// out[0..cap) <- the first cap of them; returns how many were written (== cap: there may be more behind out[cap - 1])
static inline int wb_block_records(const int* a, int lo, int hi, int floor, int* out, int cap) {
    int run = floor, cnt = 0;
    for (int i = lo; i < hi && cnt < cap; ++i)
        if (a[i] > run) { run = a[i]; out[cnt++] = i; }
    return cnt;
}

// in-place inclusive running sum of a[0..n) (doubles) on top of `carry`; returns the new carry.  (On the GPU the terms are
// combined as a parallel scan combines them: a running sum that differs from the left-to-right one in its last bits.)
static inline double wb_block_cumsum_f64(double* a, int n, double carry) {
    for (int i = 0; i < n; ++i) { carry = carry + a[i]; a[i] = carry; }
    return carry;
}

#else
// ---------------------------------------------------------------- HIP / gfx950
#include <hip/hip_runtime.h>
#define WB_HD __host__ __device__
#define WB_HOSTSIM_PATH(k) ((void)0)
#define WB_DEV __device__
#define WB_NOINLINE __attribute__((noinline))      // rare slow paths: kept out of the hot kernels' register allocation
#define WB_LAMBDA_INLINE __attribute__((always_inline))
#define WB_KERNEL(name, ...) __global__ void name(__VA_ARGS__)
// waves = minimum waves per SIMD the register allocator must leave room for
#define WB_KERNEL_LB(name, threads, waves, ...) __global__ __launch_bounds__(threads, waves) void name(__VA_ARGS__)
#define WB_BLOCK_ID ((int)blockIdx.x)
#define WB_GRID_DIM ((int)gridDim.x)
#define WB_NT ((int)blockDim.x)
#define WB_TID ((int)threadIdx.x)
#define WB_PFOR(i, n)                                                              \
    {                                                                              \
        for (int64_t i = WB_TID, wb_n_##i = (int64_t)(n); i < wb_n_##i; i += WB_NT) {
#define WB_END_PFOR \
        }           \
        __syncthreads(); \
    }
#define WB_PFOR32(i, n)                                                            \
    {                                                                              \
        for (int i = WB_TID, wb_n_##i = (int)(n); i < wb_n_##i; i += WB_NT) {
#define WB_PFOR_B32(i, ic, u, n)                                                                                          \
    {                                                                                                                      \
        const int wb_n_##i = (int)(n);                                                                                     \
        for (int wb_b_##i = WB_TID; wb_b_##i < wb_n_##i; wb_b_##i += WB_PFOR_B_WIDTH * WB_NT) {                            \
            _Pragma("unroll") for (int u = 0; u < WB_PFOR_B_WIDTH; ++u) {                                                  \
                const int i = wb_b_##i + u * WB_NT, ic = i < wb_n_##i ? i : wb_n_##i - 1;                                  \
                (void)i;
#define WB_PFOR_B32_USE(i, u)                                                                                              \
            }                                                                                                              \
            _Pragma("unroll") for (int u = 0; u < WB_PFOR_B_WIDTH; ++u) {                                                  \
                const int i = wb_b_##i + u * WB_NT;                                                                        \
                if (i >= wb_n_##i) break;
#define WB_PFOR_B_WIDTH 4
#define WB_PFOR_B(i, ic, u, n)                                                                                            \
    {                                                                                                                      \
        const int64_t wb_n_##i = (int64_t)(n);                                                                             \
        for (int64_t wb_b_##i = WB_TID; wb_b_##i < wb_n_##i; wb_b_##i += (int64_t)WB_PFOR_B_WIDTH * WB_NT) {               \
            _Pragma("unroll") for (int u = 0; u < WB_PFOR_B_WIDTH; ++u) {                                                  \
                const int64_t i = wb_b_##i + (int64_t)u * WB_NT, ic = i < wb_n_##i ? i : wb_n_##i - 1;                     \
                (void)i;
#define WB_PFOR_B_USE(i, u)                                                                                                \
            }                                                                                                              \
            _Pragma("unroll") for (int u = 0; u < WB_PFOR_B_WIDTH; ++u) {                                                  \
                const int64_t i = wb_b_##i + (int64_t)u * WB_NT;                                                           \
                if (i >= wb_n_##i) break;
#define WB_END_PFOR_B \
            }         \
        }             \
        __syncthreads(); \
    }
#define WB_SINGLE if (WB_TID == 0) {
#define WB_END_SINGLE \
    }                 \
    __syncthreads();
#define WB_SYNC() __syncthreads()
#define WB_WAVE 64
__device__ inline bool wb_wave_any(bool v) { return __ballot(v) != 0ull; }      // (over the lanes that are at the call)
#define WB_SHARED(type, name, n) __shared__ type name[n]
// arrays in the launch's dynamic LDS (WB_LAUNCH_LDS), carved in the order of their declarations, 8-byte aligned
#define WB_DYN_SHARED_BEGIN extern __shared__ __attribute__((aligned(16))) unsigned char wb_dyn_lds_[]; size_t wb_dyn_off_ = 0;
#define WB_DYN_SHARED(type, name, n) type* name = (type*)(wb_dyn_lds_ + wb_dyn_off_); wb_dyn_off_ += ((size_t)(n) * sizeof(type) + 7) & ~(size_t)7
template <class T> __device__ inline T wb_atomic_add(T* p, T v) { return atomicAdd(p, v); }
template <class T> __device__ inline T wb_atomic_min(T* p, T v) { return atomicMin(p, v); }
template <class T> __device__ inline T wb_atomic_max(T* p, T v) { return atomicMax(p, v); }
template <class T> __device__ inline T wb_atomic_or(T* p, T v) { return atomicOr(p, v); }
template <class T> __device__ inline T wb_atomic_and(T* p, T v) { return atomicAnd(p, v); }
template <class T> __device__ inline T wb_atomic_cas(T* p, T cmp, T v) { return atomicCAS(p, cmp, v); }
__device__ inline int wb_popc32(uint32_t v) { return __popc(v); }
__device__ inline int wb_ctz32(uint32_t v) { return __ffs((int)v) - 1; }
__device__ inline int wb_clz32(uint32_t v) { return __clz((int)v); }
__device__ inline int wb_popc64(uint64_t v) { return __popcll(v); }
// base[idx] += val without return value.  Many lanes of a wave usually target the same idx (all points of one
// long contour): they are summed inside the wave first, one atomic per distinct idx (atomics on one address
// serialise at ~12 ns each on this chip).  Must be called by converged code paths only in the sense that every
// the block's sum of v, returned to every thread, added in a fixed order (butterfly inside a wave, then the waves in turn):
// the same bits on every run; lw: LDS scratch of 16 doubles; all threads of the block call it
__device__ inline double wb_block_sum_f64(double v, double* lw) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    const int nw = ((int)blockDim.x + 63) >> 6;
    if (nw == 1) return v;
    if ((threadIdx.x & 63) == 0) lw[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += lw[w];
    __syncthreads();
    return t;
}
// every thread's v added to *acc (LDS) -- one atomic per wave; all threads of the block call it, the sum stands after it
__device__ inline void wb_block_add_u64(unsigned long long* acc, unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(acc, v);
    __syncthreads();
}
// atomicMin(addr, val) by the lowest lane that wants it -- for values that ascend with the lane (an index search): one
// atomic per wave instead of one per hit
__device__ inline void wb_atomic_min_first(int* addr, bool want, int val) {
    const unsigned long long m = __ballot(want);
    if (want && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicMin(addr, val);
}
// base[idx] += val for every lane of a wave, ALL of which must be at the call (top level of a kernel, behind a loop): one atomic
// for the wave when its lanes name the same entry -- sixty-four atomics on one address are served one after the other
__device__ inline void wb_atomic_add_wave(int* base, int idx, int val) {
    const int first = __builtin_amdgcn_readfirstlane(idx);
    if (__ballot(idx != first) == 0ull) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) val += __shfl_xor(val, d, 64);
        if ((threadIdx.x & 63) == 0 && val) atomicAdd(&base[first], val);
    } else if (val) atomicAdd(&base[idx], val);
}
// base[idx] += val (0 <= val < 65536), the lanes that name the same entry combined into ONE atomic.  Call from code every
// participating lane reaches it; inactive lanes are simply not in the ballot.
__device__ inline void wb_atomic_add_agg(int* base, int idx, int val) {
    unsigned long long todo = __ballot(1);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lidx = __shfl(idx, leader, 64);
        const unsigned long long same = __ballot(idx == lidx) & todo;
        // sum of val (0 <= val < 65536) over the lanes in `same`, bit-sliced with ballots: well defined whatever
        // lanes are inactive (a shuffle would read undefined registers of lanes that left the loop)
        int sum = 0;
#pragma unroll
        for (int bit = 0; bit < 16; ++bit) sum += __popcll(__ballot((val >> bit) & 1) & same) << bit;
        if (lane == leader) atomicAdd(&base[lidx], sum);
        todo &= ~same;
    }
}

// slot of an append to a list whose length counter ALL calling lanes share (-1 for lanes with want == false): one atomic
// per wave instead of one per appending lane (same-address atomics serialise).  Call from code every lane of interest
// reaches together; lanes that left the enclosing loop are simply not in the ballot.
__device__ inline float wb_rcp_f32(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ inline int wb_append_slot(int* counter, bool want) {
    const unsigned long long m = __ballot(want);
    if (!m) return -1;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader, 64);
    return want ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

// first slot of n (0 <= n < 32) consecutive entries appended by the calling lane: the wave's lanes reserve their entries
// with ONE returning atomic (a returning atomic is a round trip to L2: one per lane and entry, inside a loop, sets the
// loop's pace).  Prefix and total by bit-sliced ballots: well defined whatever lanes are inactive.
__device__ inline int wb_append_slots(int* counter, int n) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int prefix = 0, total = 0;
#pragma unroll
    for (int bit = 0; bit < 5; ++bit) {
        const unsigned long long m = __ballot((n >> bit) & 1);
        prefix += __popcll(m & lt) << bit;
        total += __popcll(m) << bit;
    }
    if (total == 0) return 0;
    const int leader = __ffsll((long long)__ballot(1)) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, total);
    base = __shfl(base, leader, 64);
    return base + prefix;
}

// 64-wide inclusive wave scan (gfx950 wavefront = 64)
__device__ inline int wb_wave_scan_incl(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// in-place exclusive scan of a[0..n) (global or LDS), all threads of the block call it;
// returns the total to every thread.  PRECONDITIONS (the function does not start with a barrier of its own): (1) a[0..n) as the
// block wrote it is visible to the whole block -- the caller ends a WB_PFOR / WB_SYNC behind its last write to a[] (every caller
// does); (2) blockDim.x is a multiple of 64: a wave's total is written by its lane 63, a partial last wave would leave its row
// entry unwritten (every launch that reaches this uses 256, 512 or 1024 threads).  A thread owns 8 consecutive elements of a chunk of 8 * blockDim.x (local prefix,
// wave scan of the thread sums, running carry): a per-job array of a few thousand entries is two or three chunks, each
// one memory round trip and three barriers.
__device__ inline int wb_block_scan_excl(int* a, int n) {
    // The waves' totals of a chunk go through one of two LDS rows in turn and the running carry stays in registers (every thread
    // adds up the same totals): ONE barrier per chunk and one behind the last -- it was three per chunk and two around them, and the
    // single-workgroup kernels that lay out a pass's lists (20 000 events: ten chunks) are nothing but these barriers.
    __shared__ int wb_scan_wave[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
    if (n <= (int)blockDim.x) {
        // at most one element per thread (a per-job table of a few hundred entries)
        const int v = tid < n ? a[tid] : 0;
        const int inc = wb_wave_scan_incl(v);
        if (lane == 63) wb_scan_wave[0][wave] = inc;
        __syncthreads();
        int woff = 0, total = 0;
        for (int w = 0; w < nw; ++w) { const int t = wb_scan_wave[0][w]; if (w < wave) woff += t; total += t; }
        if (tid < n) a[tid] = woff + inc - v;
        __syncthreads();                               // (a[] stands; the rows are free for the next call)
        return total;
    }
    int carry = 0, row = 0;                            // the same in every thread
    for (int base = 0; base < n; base += 8 * blockDim.x) {
        const int i0 = base + 8 * tid;
        int v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = (i0 + k < n) ? a[i0 + k] : 0; sum += v[k]; }
        const int inc = wb_wave_scan_incl(sum);
        if (lane == 63) wb_scan_wave[row][wave] = inc;
        __syncthreads();
        int woff = 0, total = 0;
        for (int w = 0; w < nw; ++w) { const int t = wb_scan_wave[row][w]; if (w < wave) woff += t; total += t; }
        int run = carry + woff + inc - sum;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (i0 + k < n) a[i0 + k] = run;
            run += v[k];
        }
        carry += total;
        row ^= 1;                                      // (the row just read is written again two chunks on, behind the next chunk's barrier)
    }
    __syncthreads();                                   // (a[] stands; every thread has read its totals before the next call writes row 0)
    return carry;
}

// in-place exclusive running maximum of a[0..n): a[i] <- max(init, a[0..i-1]); all threads of the block call it.
// Eight consecutive elements per thread and chunk, like wb_block_scan_excl.
__device__ inline void wb_block_scan_excl_max(int* a, int n, int init) {
    __shared__ int wb_smax_wave[16];
    __shared__ int wb_smax_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
    if (tid == 0) wb_smax_carry = init;
    __syncthreads();
    for (int base = 0; base < n; base += 8 * blockDim.x) {
        const int i0 = base + 8 * tid;
        int v[8], mx = init;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = (i0 + k < n) ? a[i0 + k] : init; mx = v[k] > mx ? v[k] : mx; }
        int inc = mx;                                  // inclusive max over the wave prefix of the thread maxima
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d && o > inc) inc = o;
        }
        if (lane == 63) wb_smax_wave[wave] = inc;
        int excl = __shfl_up(inc, 1, 64);              // exclusive within the wave
        if (lane == 0) excl = init;
        __syncthreads();
        int run = wb_smax_carry;
        for (int w = 0; w < wave; ++w) run = wb_smax_wave[w] > run ? wb_smax_wave[w] : run;
        run = excl > run ? excl : run;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (i0 + k < n) a[i0 + k] = run;
            run = v[k] > run ? v[k] : run;
        }
        __syncthreads();
        if (tid == 0) {
            int tot = wb_smax_carry;
            for (int w = 0; w < nw; ++w) tot = wb_smax_wave[w] > tot ? wb_smax_wave[w] : tot;
            wb_smax_carry = tot;
        }
        __syncthreads();
    }
}
// the RECORDS of a[lo..hi) above `floor`, in ascending order of their index: the i with a[i] > max(floor, a[lo..i)).
// out[0..cap) (LDS) <- the first cap of them; returns how many were written (== cap: there may be more behind out[cap - 1]).
// All threads of the block call it with the same arguments and get the same result; eight consecutive elements per thread and
// chunk, running maximum and count across the wave by shuffles and across the waves through LDS: two barriers per chunk of
// 8 * blockDim.x elements and one behind it (out[] is complete and visible when the call returns).
__device__ inline int wb_block_records(const int* a, int lo, int hi, int floor, int* out, int cap) {
    __shared__ int wb_rec_max[16];
    __shared__ int wb_rec_cnt[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
    const int none = (int)0x80000000;
    int runmax = floor, runcnt = 0;                    // the same in every thread
    for (int base = lo; base < hi && runcnt < cap; base += 8 * (int)blockDim.x) {
        const int i0 = base + 8 * tid;
        int v[8], mx = none;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = (i0 + k < hi) ? a[i0 + k] : none; mx = v[k] > mx ? v[k] : mx; }
        int inc = mx;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d && o > inc) inc = o;
        }
        int excl = __shfl_up(inc, 1, 64);
        if (lane == 0) excl = none;
        if (lane == 63) wb_rec_max[wave] = inc;
        __syncthreads();
        int run = runmax;
        for (int w = 0; w < wave; ++w) run = wb_rec_max[w] > run ? wb_rec_max[w] : run;
        run = excl > run ? excl : run;
        int cnt = 0;
        uint32_t mask = 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (v[k] > run) { run = v[k]; mask |= 1u << k; ++cnt; }
        const int cinc = wb_wave_scan_incl(cnt);
        if (lane == 63) wb_rec_cnt[wave] = cinc;
        __syncthreads();
        int off = runcnt + cinc - cnt;
        for (int w = 0; w < wave; ++w) off += wb_rec_cnt[w];
        while (mask) {
            const int k = __ffs((int)mask) - 1;
            mask &= mask - 1u;
            if (off < cap) out[off] = i0 + k;
            ++off;
        }
        for (int w = 0; w < nw; ++w) { runmax = wb_rec_max[w] > runmax ? wb_rec_max[w] : runmax; runcnt += wb_rec_cnt[w]; }
        __syncthreads();
    }
    return runcnt < cap ? runcnt : cap;
}
// in-place inclusive running sum of a[0..n) (doubles, LDS or global; n <= 8 * blockDim.x) on top of `carry`; all threads of
// the block call it, every thread gets the new carry.  A thread owns 8 consecutive terms; thread sums are scanned inside
// the wave, wave totals across the block: the result differs from the left-to-right running sum in its last bits.
__device__ inline double wb_block_cumsum_f64(double* a, int n, double carry) {
    __shared__ double wb_cs_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
    const int i0 = 8 * tid;
    double v[8], sum = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = (i0 + k < n) ? a[i0 + k] : 0.0; sum += v[k]; }
    double inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    __syncthreads();                                     // (the previous call's readers of wb_cs_wave are done)
    if (lane == 63) wb_cs_wave[wave] = inc;
    __syncthreads();
    double run = carry;
    for (int w = 0; w < wave; ++w) run += wb_cs_wave[w];
    double total = carry;
    for (int w = 0; w < nw; ++w) total += wb_cs_wave[w];
    run += inc - sum;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        run += v[k];
        if (i0 + k < n) a[i0 + k] = run;
    }
    __syncthreads();
    return total;
}
#endif
