// wb_comm.h -- the two exchanges of a time-sharded run inside the library, on RCCL directly.
//
// One process per GPU; every rank holds a wb_ctx.  The launcher makes ONE unique id on rank 0 (wb_comm_unique_id), hands
// its 128 bytes to every rank by whatever means it has (the reference has none -- INTEGRATION.md uses the torch.distributed
// store, bench.py a broadcast_object_list), and every rank calls wb_comm_init.  After that
//   (1) wb_comm_allreduce_max   contours.exp_lon.max() over the WHOLE frame (streamer_index.py:106): one int32;
//   (2) wb_comm_gather_tables   the event rows and polygon vertices wb_indices left on the device (context option
//                               "keep_device") to rank dst: the counts by one ncclAllGather, the payload by grouped
//                               ncclSend / ncclRecv of the exact sizes into device buffers of dst that persist between
//                               passes, one copy from there into pinned host buffers that persist too.
//       wb_comm_stage_tables + wb_comm_gather_staged: the same exchange off the caller's critical path -- the batch's
//                               tables are copied (device to device, a few MB) into one of WB_COMM_SLOTS buffers the
//                               communicator owns, the batch can be released at once, and the gather of the staged slot
//                               may run on another host thread while the next batches are computed.
// RCCL is looked up at run time (dlopen; the copy a host program such as PyTorch already loaded is preferred): the library
// neither links librccl nor needs it for single-GPU use.  Everything runs on a stream of its own (non-blocking: the NULL
// stream would order it against every kernel the process has queued).
#pragma once
#if !defined(WB_HOSTSIM)
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <mutex>

struct WbComm {
    void* lib = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;
    // entry points
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    // wb_comm_abort may come from another host thread than the one inside a gather: `aborted` is checked before every group of
    // RCCL calls, and `issue` is held while such a group is ENQUEUED (never across a stream wait), so the communicator is not
    // torn down between two calls that use it; the abort itself waits for `issue` for a bounded time only -- a thread that sits
    // INSIDE an RCCL call is what ncclCommAbort exists for
    std::atomic<bool> aborted{false};     // wb_comm_abort: every later exchange fails at once
    std::timed_mutex issue;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    hipStream_t copy_stream = nullptr;  // wb_comm_stage_tables (the caller's thread); `stream` may be driven by another thread
    // staged tables of batches that have been released (wb_comm_stage_tables)
    struct Slot { double* rows = nullptr; size_t cap_rows = 0; int32_t *px = nullptr, *py = nullptr; size_t cap_v = 0; long long ne = 0, nv = 0; bool full = false; };
    Slot slot[WB_COMM_SLOTS];
    // persistent buffers: counts (device + pinned), the catalogue on dst (device + pinned)
    long long* d_counts = nullptr;      // [3 + 3 * world]: (rank + 1, events, vertices) of this rank, then everybody's
    long long* h_counts = nullptr;      // pinned [3 * world + 3]
    int* d_scalar = nullptr;
    int* h_scalar = nullptr;            // pinned
    double* d_rows = nullptr; size_t cap_rows = 0;          // rows of all ranks (dst only)
    int32_t *d_px = nullptr, *d_py = nullptr; size_t cap_v = 0;
    double* h_rows = nullptr; size_t hcap_rows = 0;         // pinned
    int32_t *h_px = nullptr, *h_py = nullptr; size_t hcap_v = 0;
};

static std::string wb_comm_load(WbComm& c) {
    if (c.lib) return "";
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { c.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (c.lib) break; }     // a copy the process already holds
    if (!c.lib) for (const char* n : names) { c.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (c.lib) break; }
    if (!c.lib) return std::string("librccl not found: ") + (dlerror() ? dlerror() : "");
#define WB_NCCL_SYM(field, name) \
    c.field = (decltype(c.field))dlsym(c.lib, name); \
    if (!c.field) return std::string("librccl lacks ") + name;
    WB_NCCL_SYM(GetUniqueId, "ncclGetUniqueId") WB_NCCL_SYM(CommInitRank, "ncclCommInitRank") WB_NCCL_SYM(CommDestroy, "ncclCommDestroy") WB_NCCL_SYM(CommAbort, "ncclCommAbort")
    WB_NCCL_SYM(GetErrorString, "ncclGetErrorString") WB_NCCL_SYM(AllReduce, "ncclAllReduce") WB_NCCL_SYM(AllGather, "ncclAllGather")
    WB_NCCL_SYM(Send, "ncclSend") WB_NCCL_SYM(Recv, "ncclRecv") WB_NCCL_SYM(GroupStart, "ncclGroupStart") WB_NCCL_SYM(GroupEnd, "ncclGroupEnd")
#undef WB_NCCL_SYM
    return "";
}
#endif
