// lfg_comm.cpp -- the one exchange of the hot path behind the C-ABI: the batch's shared previous frame travels from
// the rank that owns it to every other rank as ONE ncclBroadcast (RCCL over xGMI) per frame (SURVEY.md section 5 and
// 8(e); the reference has one queue and no communication at all: src/vulkan_context.cpp:130-151).
//
// RCCL is opened at run time (dlopen("librccl.so")) the first time a communicator is asked for: a single-GPU user of
// the library never loads it, and the library has no link-time dependency on it.  The broadcast runs on the context's
// own communication stream, ordered against the compute stream with events:
//     lfg_broadcast_frame   comm stream waits for what EVERY lane's stream has been given so far (the kernels that still
//                           read the frame on a receiver, the kernels that produce it on the root), broadcasts, records;
//     lfg_comm_wait         the SELECTED lane's stream waits (on the device) for every broadcast issued so far -- the
//                           communication stream is in order, so the one event, re-recorded behind each broadcast,
//                           stands for all of them; a caller that wants the older of two broadcasts waits for both.
// Between the two calls the caller enqueues the kernels of the step before: that is the overlap.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include <rccl/rccl.h>

#include "lfg_internal.hpp"

#define LFG_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

// (std::call_once: the header allows one thread per GPU, and two of them may ask for their first communicator together)
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.error = std::string("cannot load librccl.so: ") + dlerror(); return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.handle, n); if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

int fail(lfg_context *ctx, int code, const std::string &msg) {
    if (ctx) ctx->error = msg;
    return code;
}

int fail_nccl(lfg_context *ctx, ncclResult_t e, const char *what) {
    return fail(ctx, LFG_ERR_DEVICE, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error"));
}

static_assert(sizeof(ncclUniqueId) == LFG_COMM_ID_BYTES, "lfg_comm_id must hold an ncclUniqueId");

}  // namespace

LFG_EXPORT int lfg_comm_unique_id(lfg_comm_id *out_id) {
    if (!out_id) return LFG_ERR_INVALID;
    Rccl &r = rccl();
    if (!r.error.empty()) return LFG_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return LFG_ERR_DEVICE;
    std::memcpy(out_id->bytes, id.internal, LFG_COMM_ID_BYTES);
    return LFG_OK;
}

LFG_EXPORT int lfg_comm_init(lfg_context *ctx, int nranks, int rank, const lfg_comm_id *id) {
    if (!ctx) return LFG_ERR_INVALID;
    if (!id || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(ctx, LFG_ERR_INVALID, "lfg_comm_init: need 0 <= rank < nranks and a communicator id");
    if (ctx->comm) return fail(ctx, LFG_ERR_INVALID, "lfg_comm_init: this context already has a communicator");
    Rccl &r = rccl();
    if (!r.error.empty()) return fail(ctx, LFG_ERR_UNSUPPORTED, "lfg_comm_init: " + r.error);
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess && !ctx->comm_stream) {
        // highest priority: the broadcast's few workgroups must not queue behind the compute lanes' kernels for the CUs
        // that the persistent prefilter workgroups give back one by one
        int least = 0, greatest = 0;
        e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->comm_stream, hipStreamNonBlocking, greatest);
    }
    if (e == hipSuccess && !ctx->comm_ready) e = hipEventCreateWithFlags(&ctx->comm_ready, hipEventDisableTiming);
    if (e == hipSuccess && !ctx->comm_done) e = hipEventCreateWithFlags(&ctx->comm_done, hipEventDisableTiming);
    if (e != hipSuccess) return fail(ctx, LFG_ERR_DEVICE, std::string("lfg_comm_init: ") + hipGetErrorString(e));
    ncclUniqueId nid;
    std::memcpy(nid.internal, id->bytes, LFG_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = r.CommInitRank(&comm, nranks, nid, rank);       // collective: returns once every rank has called
    if (rc != ncclSuccess) return fail_nccl(ctx, rc, "ncclCommInitRank");
    ctx->comm = comm; ctx->comm_ranks = nranks; ctx->comm_rank = rank; ctx->comm_pending = false;
    return LFG_OK;
}

LFG_EXPORT int lfg_comm_rank(const lfg_context *ctx) { return ctx && ctx->comm ? ctx->comm_rank : -1; }
LFG_EXPORT int lfg_comm_ranks(const lfg_context *ctx) { return ctx && ctx->comm ? ctx->comm_ranks : 0; }

LFG_EXPORT int lfg_broadcast_frame(lfg_context *ctx, lfg_frame *frame, int root) {
    if (!ctx) return LFG_ERR_INVALID;
    if (!ctx->comm) return fail(ctx, LFG_ERR_INVALID, "lfg_broadcast_frame: no communicator (lfg_comm_init)");
    if (!frame || !frame->data || frame->width == 0 || frame->height == 0)
        return fail(ctx, LFG_ERR_INVALID, "lfg_broadcast_frame: empty frame");
    if (root < 0 || root >= ctx->comm_ranks) return fail(ctx, LFG_ERR_INVALID, "lfg_broadcast_frame: root out of range");
    const uint32_t bpp = frame->format == LFG_FORMAT_RGBA8_UNORM ? 4u : frame->format == LFG_FORMAT_MV_S8X2 ? 2u : 0u;
    if (!bpp) return fail(ctx, LFG_ERR_INVALID, "lfg_broadcast_frame: unknown format");
    if (frame->pitch != frame->width * bpp)                     // one contiguous message per frame, as every rank allocates it
        return fail(ctx, LFG_ERR_UNSUPPORTED, "lfg_broadcast_frame: the frame must be tightly packed (pitch == width * bytes per pixel)");
    const size_t bytes = (size_t)frame->pitch * frame->height;
    // The broadcast starts once EVERY lane has finished what it has been given so far: the kernels that still read the
    // frame on a receiver (or produce it on the root) may sit on any of them.  One event, recorded and waited for lane by
    // lane (a wait refers to the record that precedes it).
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipEventRecord(ctx->comm_ready, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ready, 0);
#ifndef LFG_DIAG_COMM_SELECTED_LANE_ONLY   // (diagnostic build: round 2's ordering, to see tests/test_gpu_comm.py's ordering test fail)
    for (size_t j = 0; j < ctx->lanes.size() && e == hipSuccess; ++j) {
        if ((int)j == ctx->lane || !ctx->lanes[j].stream) continue;       // (the selected lane's stream is ctx->stream, above)
        e = hipEventRecord(ctx->comm_ready, ctx->lanes[j].stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ready, 0);
    }
#endif
    if (e != hipSuccess) return fail(ctx, LFG_ERR_DEVICE, std::string("lfg_broadcast_frame: ") + hipGetErrorString(e));
    const ncclResult_t rc = rccl().Broadcast(frame->data, frame->data, bytes, ncclUint8, root, (ncclComm_t)ctx->comm, ctx->comm_stream);
    if (rc != ncclSuccess) return fail_nccl(ctx, rc, "ncclBroadcast");
    e = hipEventRecord(ctx->comm_done, ctx->comm_stream);
    if (e != hipSuccess) return fail(ctx, LFG_ERR_DEVICE, std::string("lfg_broadcast_frame: ") + hipGetErrorString(e));
    ctx->comm_pending = true;
    return LFG_OK;
}

LFG_EXPORT int lfg_comm_wait(lfg_context *ctx) {
    if (!ctx) return LFG_ERR_INVALID;
    if (!ctx->comm) return fail(ctx, LFG_ERR_INVALID, "lfg_comm_wait: no communicator (lfg_comm_init)");
    if (ctx->comm_pending) {               // (never cleared: another lane may still have to wait for the same broadcast,
                                           //  and waiting for an event that has fired costs nothing on the device)
        const hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->comm_done, 0);
        if (e != hipSuccess) return fail(ctx, LFG_ERR_DEVICE, std::string("lfg_comm_wait: ") + hipGetErrorString(e));
    }
    return LFG_OK;
}

LFG_EXPORT int lfg_comm_destroy(lfg_context *ctx) {
    if (!ctx) return LFG_ERR_INVALID;
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
        (void)rccl().CommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr; ctx->comm_ranks = 0; ctx->comm_rank = 0; ctx->comm_pending = false;
    }
    if (ctx->comm_ready) { (void)hipEventDestroy(ctx->comm_ready); ctx->comm_ready = nullptr; }
    if (ctx->comm_done) { (void)hipEventDestroy(ctx->comm_done); ctx->comm_done = nullptr; }
    if (ctx->comm_stream) { (void)hipStreamDestroy(ctx->comm_stream); ctx->comm_stream = nullptr; }
    return LFG_OK;
}
