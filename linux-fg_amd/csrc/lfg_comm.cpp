// lfg_comm.cpp -- the one exchange of the hot path behind the C-ABI: the batch's shared previous frame travels from
// the rank that owns it to every other rank as ONE ncclBroadcast (RCCL over xGMI) per frame (SURVEY.md section 5 and
// 8(e); the reference has one queue and no communication at all: src/vulkan_context.cpp:130-151).
//
// RCCL is opened at run time (dlopen("librccl.so")) the first time a communicator is asked for: a single-GPU user of
// the library never loads it, and the library has no link-time dependency on it.  The broadcast runs on the context's
// own communication stream, ordered against the compute stream with events:
//     lfg_broadcast_frame   comm stream waits for what EVERY lane's stream has been given so far (the kernels that still
//                           read the frame on a receiver, the kernels that produce it on the root), broadcasts, records;
//     lfg_broadcast_frame_lane   ... for what the SELECTED lane has been given so far, only: for a caller that has ordered
//                           that lane behind the frame's last readers itself (lfg_lane_wait) -- with frames in flight the safe
//                           call makes the broadcast for step k + 1 wait for ALL of step k - 1, and step k + 1 for it;
//     lfg_comm_wait         the SELECTED lane's stream waits (on the device) for every broadcast issued so far -- the
//                           communication stream is in order, so the one event, re-recorded behind each broadcast,
//                           stands for all of them; a caller that wants the older of two broadcasts waits for both.
// Between the two calls the caller enqueues the kernels of the step before: that is the overlap.
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <vector>

#include <rccl/rccl.h>

#include "lfg_internal.hpp"

#define LFG_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitRankConfig)(ncclComm_t *, int, ncclUniqueId, int, ncclConfig_t *) = nullptr;
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
        r.CommInitRankConfig = reinterpret_cast<decltype(r.CommInitRankConfig)>(dlsym(r.handle, "ncclCommInitRankConfig"));      // (optional: RCCL >= 2.18)
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
    const int reserve = (ctx->knobs.commCus > 0 && ctx->device_cus >= 8 * ctx->knobs.commCus) ? ctx->knobs.commCus : 0;    // (at most an eighth of the device)
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess && !ctx->comm_stream) {
        if (reserve > 0) {
            // The communication stream may use the reserved CUs and no others (and the library's compute streams every CU but those:
            // lfg_own_stream_create).  Measured on the MI355X (tools/probe_cu_reserve.hip): while a chip-filling launch of the persistent
            // kernel's footprint is resident, a kernel of RCCL's footprint on a HIGH-PRIORITY stream that may use any CU still waits for
            // that launch to end although eight CUs stand empty -- its workgroups are handed to shader engines in turn, not to where
            // there is room -- and on a stream masked to the empty CUs it starts at once (54 us for 8 workgroups of 50 us).
            std::vector<uint32_t> mask((size_t)(ctx->device_cus + 31) / 32, 0u);
            for (int i = 0; i < reserve; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
            e = hipExtStreamCreateWithCUMask(&ctx->comm_stream, (uint32_t)mask.size(), mask.data());
        } else {
            // no reservation: highest priority, so that the broadcast's few workgroups at least do not queue behind the compute lanes'
            // kernels for the CUs that the persistent prefilter workgroups give back
            int least = 0, greatest = 0;
            e = hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->comm_stream, hipStreamNonBlocking, greatest);
        }
    }
    if (e == hipSuccess && !ctx->comm_ready) e = hipEventCreateWithFlags(&ctx->comm_ready, hipEventDisableTiming);
    if (e == hipSuccess && !ctx->comm_done) e = hipEventCreateWithFlags(&ctx->comm_done, hipEventDisableTiming);
    if (e != hipSuccess) return fail(ctx, LFG_ERR_DEVICE, std::string("lfg_comm_init: ") + hipGetErrorString(e));
    ncclUniqueId nid;
    std::memcpy(nid.internal, id->bytes, LFG_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    // The communicator's kernel gets at most as many workgroups (one per channel) as there are CUs kept for it: two of them do not fit a
    // CU (comm_probe.hip: 261 - 280 of a SIMD's 512 registers each), and a ninth would wait for one of the eight to leave.
    ncclConfig_t config = NCCL_CONFIG_INITIALIZER;
    if (reserve > 0) { config.minCTAs = 1; config.maxCTAs = reserve; }
    // (collective: returns once every rank has called.  An RCCL without ncclCommInitRankConfig cannot be told how many channels to use:
    //  the reservation stands, a kernel of more channels than reserved CUs runs some of them in turn.)
    const ncclResult_t rc = r.CommInitRankConfig ? r.CommInitRankConfig(&comm, nranks, nid, rank, &config) : r.CommInitRank(&comm, nranks, nid, rank);
    if (rc != ncclSuccess) return fail_nccl(ctx, rc, r.CommInitRankConfig ? "ncclCommInitRankConfig" : "ncclCommInitRank");
    ctx->comm = comm; ctx->comm_ranks = nranks; ctx->comm_rank = rank; ctx->comm_pending = false;
    // From here on the library's own streams leave `commCus` CUs alone (lfg_own_stream_create): a broadcast that becomes
    // ready while a full persistent grid runs -- noise, uncorrelated content, a scene cut: launches of 0.7 - 7 ms -- starts at once
    // instead of when that grid's first workgroup leaves, which is when its last one does.
    ctx->comm_cus = reserve;
    return ctx->comm_cus > 0 ? lfg_restream(ctx) : LFG_OK;
}

LFG_EXPORT int lfg_comm_reserved_cus(const lfg_context *ctx) { return ctx && ctx->comm ? ctx->comm_cus : 0; }

LFG_EXPORT int lfg_comm_cu_mask(const lfg_context *ctx, uint32_t *out_words, int words) {
    if (!ctx || !out_words || words < 1) return LFG_ERR_INVALID;
    const int cus = ctx->device_cus;
    if (words * 32 < cus) return LFG_ERR_INVALID;
    for (int w = 0; w < words; ++w) out_words[w] = 0;
    for (int i = (ctx->comm ? ctx->comm_cus : 0); i < cus; ++i) out_words[i / 32] |= 1u << (i % 32);
    return LFG_OK;
}

LFG_EXPORT int lfg_comm_rank(const lfg_context *ctx) { return ctx && ctx->comm ? ctx->comm_rank : -1; }
LFG_EXPORT int lfg_comm_ranks(const lfg_context *ctx) { return ctx && ctx->comm ? ctx->comm_ranks : 0; }

namespace {
// The communication stream waits for what EVERY lane has been given so far: the kernels that still read the frame on a receiver (or
// produce it on the root) may sit on any of them.  One event, recorded and waited for lane by lane (a wait refers to the record
// that precedes it).
// (everyLane false: the selected lane only -- lfg_broadcast_frame_lane, whose caller has ordered that lane behind the frame's readers.)
hipError_t behind_lanes(lfg_context *ctx, bool everyLane) {
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipEventRecord(ctx->comm_ready, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ready, 0);
#ifndef LFG_DIAG_COMM_SELECTED_LANE_ONLY   // (diagnostic build: round 2's ordering, to see tests/test_gpu_comm.py's ordering test fail)
    for (size_t j = 0; everyLane && j < ctx->lanes.size() && e == hipSuccess; ++j) {
        if ((int)j == ctx->lane || !ctx->lanes[j].stream) continue;       // (the selected lane's stream is ctx->stream, above)
        e = hipEventRecord(ctx->comm_ready, ctx->lanes[j].stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ready, 0);
    }
#endif
    return e;
}
}  // namespace

namespace {
int broadcast(lfg_context *ctx, lfg_frame *frame, int root, bool everyLane) {
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
    hipError_t e = behind_lanes(ctx, everyLane);
    if (e != hipSuccess) return fail(ctx, LFG_ERR_DEVICE, std::string("lfg_broadcast_frame: ") + hipGetErrorString(e));
    const ncclResult_t rc = rccl().Broadcast(frame->data, frame->data, bytes, ncclUint8, root, (ncclComm_t)ctx->comm, ctx->comm_stream);
    if (rc != ncclSuccess) return fail_nccl(ctx, rc, "ncclBroadcast");
    e = hipEventRecord(ctx->comm_done, ctx->comm_stream);
    if (e != hipSuccess) return fail(ctx, LFG_ERR_DEVICE, std::string("lfg_broadcast_frame: ") + hipGetErrorString(e));
    ctx->comm_pending = true;
    return LFG_OK;
}
}  // namespace

LFG_EXPORT int lfg_broadcast_frame(lfg_context *ctx, lfg_frame *frame, int root) { return broadcast(ctx, frame, root, true); }
LFG_EXPORT int lfg_broadcast_frame_lane(lfg_context *ctx, lfg_frame *frame, int root) { return broadcast(ctx, frame, root, false); }

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

LFG_EXPORT int lfg_comm_sync(lfg_context *ctx) {
    if (!ctx) return LFG_ERR_INVALID;
    if (!ctx->comm) return fail(ctx, LFG_ERR_INVALID, "lfg_comm_sync: no communicator (lfg_comm_init)");
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->comm_stream);
    if (e != hipSuccess) return fail(ctx, LFG_ERR_DEVICE, std::string("lfg_comm_sync: ") + hipGetErrorString(e));
    return LFG_OK;
}

// Diagnostic: where a broadcast would run -- ordered like one, on the communication stream -- a kernel of RCCL's device kernel's
// footprint (comm_probe.hip), `workgroups` of them staying `microseconds` each.  A communicator of one rank launches nothing for a
// broadcast, so this is how one GPU shows whether a broadcast finds a CU while the compute lanes hold the chip.
LFG_EXPORT int lfg_comm_probe(lfg_context *ctx, int workgroups, int microseconds, int every_lane) {
    if (!ctx) return LFG_ERR_INVALID;
    if (!ctx->comm) return fail(ctx, LFG_ERR_INVALID, "lfg_comm_probe: no communicator (lfg_comm_init)");
    hipError_t e = behind_lanes(ctx, every_lane != 0);
    if (e == hipSuccess && !ctx->probe_begin) e = hipEventCreate(&ctx->probe_begin);
    if (e == hipSuccess && !ctx->probe_end) e = hipEventCreate(&ctx->probe_end);
    if (e == hipSuccess) e = hipEventRecord(ctx->probe_begin, ctx->comm_stream);      // reached when the waits above are over: "ready"
    if (e == hipSuccess) e = lfg::launch_comm_probe(ctx->comm_stream, workgroups, microseconds);
    if (e == hipSuccess) e = hipEventRecord(ctx->probe_end, ctx->comm_stream);
    if (e == hipSuccess) e = hipEventRecord(ctx->comm_done, ctx->comm_stream);
    if (e != hipSuccess) return fail(ctx, LFG_ERR_DEVICE, std::string("lfg_comm_probe: ") + hipGetErrorString(e));
    ctx->comm_pending = true;
    return LFG_OK;
}

LFG_EXPORT int lfg_comm_probe_ms(lfg_context *ctx, float *out_ms) {
    if (!ctx || !out_ms) return LFG_ERR_INVALID;
    if (!ctx->probe_end) return fail(ctx, LFG_ERR_INVALID, "lfg_comm_probe_ms: no probe has been issued");
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipEventSynchronize(ctx->probe_end);
    if (e == hipSuccess) e = hipEventElapsedTime(out_ms, ctx->probe_begin, ctx->probe_end);
    if (e != hipSuccess) return fail(ctx, LFG_ERR_DEVICE, std::string("lfg_comm_probe_ms: ") + hipGetErrorString(e));
    return LFG_OK;
}

LFG_EXPORT int lfg_comm_destroy(lfg_context *ctx) {
    if (!ctx) return LFG_ERR_INVALID;
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
        (void)rccl().CommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr; ctx->comm_ranks = 0; ctx->comm_rank = 0; ctx->comm_pending = false;
        if (ctx->comm_cus > 0) { ctx->comm_cus = 0; (void)lfg_restream(ctx); }      // the library's streams have the whole device again
    }
    if (ctx->comm_ready) { (void)hipEventDestroy(ctx->comm_ready); ctx->comm_ready = nullptr; }
    if (ctx->comm_done) { (void)hipEventDestroy(ctx->comm_done); ctx->comm_done = nullptr; }
    if (ctx->probe_begin) { (void)hipEventDestroy(ctx->probe_begin); ctx->probe_begin = nullptr; }
    if (ctx->probe_end) { (void)hipEventDestroy(ctx->probe_end); ctx->probe_end = nullptr; }
    if (ctx->comm_stream) { (void)hipStreamDestroy(ctx->comm_stream); ctx->comm_stream = nullptr; }
    return LFG_OK;
}
