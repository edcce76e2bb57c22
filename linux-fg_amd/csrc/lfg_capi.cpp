// lfg_capi.cpp -- implementation of the C-ABI in include/linuxfg_hip.h.
// The only translation units that touch HIP are this file and the three kernel files.
// There is no CPU fallback anywhere: every entry point needs a live HIP device.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "lfg_internal.hpp"

#define LFG_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

std::string g_create_error;       // lfg_last_error(NULL)

int fail(lfg_context *ctx, int code, const std::string &msg) {
    if (ctx) ctx->error = msg; else g_create_error = msg;
    return code;
}

int fail_hip(lfg_context *ctx, hipError_t e, const char *what) {
    return fail(ctx, e == hipErrorOutOfMemory ? LFG_ERR_NOMEM : LFG_ERR_DEVICE,
                std::string(what) + ": " + hipGetErrorString(e));
}

#define LFG_HIP(ctx, call)                                            \
    do {                                                              \
        hipError_t e_ = (call);                                       \
        if (e_ != hipSuccess) return fail_hip((ctx), e_, #call);      \
    } while (0)

uint32_t bytes_per_pixel(uint32_t format) {
    switch (format) {
        case LFG_FORMAT_RGBA8_UNORM: return 4;
        case LFG_FORMAT_MV_S8X2: return 2;
        default: return 0;
    }
}

bool frame_ok(const lfg_frame *f, uint32_t format) {
    return f && f->data && f->width > 0 && f->height > 0 && f->format == format &&
           f->pitch >= f->width * bytes_per_pixel(format);
}

bool same_size(const lfg_frame *a, const lfg_frame *b) { return a->width == b->width && a->height == b->height; }

// ---- Lanczos tables: the per-axis part of shaders/scale.comp:16-41, same fp32 operation order.

float lanczos_ref(float x) {                                   // scale.comp:16-20
    if (x == 0.0f) return 1.0f;
    const float px = 3.14159265359f * x;
    const float s1 = (float)std::sin((double)px);
    const float s2 = (float)std::sin((double)(px / 3.0f));
    return 3.0f * s1 * s2 / (px * px);
}

int build_axis_table(lfg_context *ctx, int in_size, int out_size, lfg::AxisTable **out) {
    for (auto &t : ctx->tables)
        if (t.in_size == in_size && t.out_size == out_size) { *out = &t; return LFG_OK; }

    std::vector<int> start((size_t)out_size);
    std::vector<float> weight((size_t)out_size * 6u);
    const float ts = 1.0f / (float)in_size;                    // scale.comp:23
    for (int p = 0; p < out_size; ++p) {
        const float uv = ((float)p + 0.5f) / (float)out_size;  // scale.comp:57
        const float pp = uv * (float)in_size - 0.5f;           // :24
        const float fl = std::floor(pp);
        const float f = pp - fl;                               // :25 fract
        const float s = fl - 2.0f;                             // :26
        double raw[6], sum = 0.0;
        for (int k = 0; k < 6; ++k) {
            const float sp = (s + (float)k + 0.5f) * ts;       // :33
            const bool skip = sp < 0.0f || sp > 1.0f;          // :34-37
            raw[k] = skip ? 0.0 : (double)lanczos_ref((float)k - f - 2.0f);   // :39-41
            sum += raw[k];
        }
        start[(size_t)p] = (int)s;
        for (int k = 0; k < 6; ++k) weight[(size_t)p * 6u + (size_t)k] = (float)(raw[k] / sum);   // :48
    }
    lfg::AxisTable t;
    t.in_size = in_size; t.out_size = out_size;
    t.pattern_2x = (out_size == 2 * in_size);
    if (t.pattern_2x)
        for (int k = 0; k < in_size; ++k)
            if (start[(size_t)(2 * k)] != k - 3 || start[(size_t)(2 * k + 1)] != k - 2) { t.pattern_2x = false; break; }

    // The distinct weight rows of a 2x table and each column's class (lfg_internal.hpp: AxisTable).
    std::vector<uint8_t> cls;
    std::vector<float> palette;
    if (t.pattern_2x) {
        cls.assign(((size_t)out_size + 3u) & ~(size_t)3u, 0);
        int rows = 0;
        for (int p = 0; p < out_size && rows >= 0; ++p) {
            const float *w = &weight[(size_t)p * 6u];
            int c = 0;
            while (c < rows && std::memcmp(&palette[(size_t)c * 8u], w, 6 * sizeof(float)) != 0) ++c;
            if (c == rows) {
                if (rows == 255) { rows = -1; break; }                       // not a palette any more: leave it out
                palette.insert(palette.end(), w, w + 6);
                palette.push_back(0.0f); palette.push_back(0.0f);
                ++rows;
            }
            cls[(size_t)p] = (uint8_t)c;
        }
        t.palette_rows = rows > 0 ? rows : 0;
    }

    if (t.pattern_2x) t.strips_per_xcd = lfg::scale_2x_strips_per_xcd(in_size);

    auto release = [&]() {
        (void)hipFree(t.d_start); (void)hipFree(t.d_weight); (void)hipFree(t.d_class); (void)hipFree(t.d_palette);
    };
    LFG_HIP(ctx, hipMalloc((void **)&t.d_start, start.size() * sizeof(int)));
    hipError_t e = hipMalloc((void **)&t.d_weight, weight.size() * sizeof(float));
    if (e == hipSuccess && t.palette_rows) e = hipMalloc((void **)&t.d_class, cls.size());
    if (e == hipSuccess && t.palette_rows) e = hipMalloc((void **)&t.d_palette, palette.size() * sizeof(float));
    if (e != hipSuccess) { release(); return fail_hip(ctx, e, "hipMalloc(weights)"); }
    // Synchronous copies: pageable host vectors go out of scope when this function returns.
    e = hipMemcpy(t.d_start, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t.d_weight, weight.data(), weight.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && t.palette_rows) e = hipMemcpy(t.d_class, cls.data(), cls.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess && t.palette_rows) e = hipMemcpy(t.d_palette, palette.data(), palette.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { release(); return fail_hip(ctx, e, "hipMemcpy(tables)"); }
    ctx->tables.push_back(t);
    *out = &ctx->tables.back();
    return LFG_OK;
}

// Everything the context has enqueued, on every lane (a resource shared by the lanes is about to go or to be read).
hipError_t sync_lanes(lfg_context *ctx) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    for (size_t j = 0; j < ctx->lanes.size() && e == hipSuccess; ++j)
        if ((int)j != ctx->lane && ctx->lanes[j].stream) e = hipStreamSynchronize(ctx->lanes[j].stream);
    return e;
}

// uv table of one axis length (lfg_internal.hpp: UvTable); a handful of sizes per context, kept until it goes.
// Bounded like the axis tables: trim_uv_tables runs at the top of the interpolate entry points, BEFORE any pointer is taken,
// so that the two lookups of a call (width, height) can never free each other's table.
void trim_uv_tables(lfg_context *ctx) {
    while (ctx->uv_tables.size() > 14) {
        (void)sync_lanes(ctx);                                 // a queued kernel may still read it
        (void)hipFree(ctx->uv_tables.front().d_uv);
        ctx->uv_tables.erase(ctx->uv_tables.begin());
    }
}
int build_uv_table(lfg_context *ctx, int size, const lfg::UvTable **out) {
    for (auto &t : ctx->uv_tables)
        if (t.size == size) { *out = &t; return LFG_OK; }
    const size_t n4 = ((size_t)size + 3u) & ~(size_t)3u, quads = n4 / 4u;
    std::vector<float> uv(n4, 0.0f);
    std::vector<uint8_t> centre(n4, 0);
    for (int p = 0; p < size; ++p) {
        uv[(size_t)p] = ((float)p + 0.5f) / (float)size;                                  // interpolate.comp:30
        // texture() at uv[p] itself (csrc/lfg_interp.hpp: texture_bilinear, the same two roundings): texel p, fraction 0?
        const float u = uv[(size_t)p] * (float)size - 0.5f;
        const float fu = floorf(u);
        centre[(size_t)p] = (fu == (float)p && u - fu == 0.0f) ? 1 : 0;
    }
    lfg::UvTable t;
    t.size = size;
    t.blocks = (int)((quads + 63u) / 64u);
    std::vector<uint64_t> good((size_t)t.blocks, 0ull);
    for (size_t q = 0; q < quads; ++q)
        if (4 * q + 3 < (size_t)size && centre[4 * q] && centre[4 * q + 1] && centre[4 * q + 2] && centre[4 * q + 3])
            good[q / 64u] |= 1ull << (q % 64u);
    // the three-instruction uv of the kernels (lfg_internal.hpp: UvTable::rcp): exact for every p of this axis?
    t.rcp = 1.0f / (float)size;
    t.rcpExact = true;
    for (int p = 0; p < size && t.rcpExact; ++p) {
        const float x = (float)p + 0.5f, q0 = x * t.rcp;
        t.rcpExact = fmaf(fmaf(-q0, (float)size, x), t.rcp, q0) == uv[(size_t)p];
    }
    const size_t c64 = ((size_t)size + 63u) & ~(size_t)63u;
    const size_t offCentre = n4 * sizeof(float), offGood = (offCentre + c64 + 15u) & ~(size_t)15u;
    const size_t bytes = offGood + good.size() * sizeof(uint64_t) + 16u;
    std::vector<uint8_t> host(bytes, 0);
    memcpy(host.data(), uv.data(), n4 * sizeof(float));
    memcpy(host.data() + offCentre, centre.data(), n4);
    memcpy(host.data() + offGood, good.data(), good.size() * sizeof(uint64_t));
    uint8_t *d = nullptr;
    LFG_HIP(ctx, hipMalloc((void **)&d, bytes));
    const hipError_t e = hipMemcpy(d, host.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return fail_hip(ctx, e, "hipMemcpy(uv table)"); }
    t.d_uv = reinterpret_cast<float *>(d);
    t.d_centre = d + offCentre;
    t.d_goodMask = reinterpret_cast<uint64_t *>(d + offGood);
    ctx->uv_tables.push_back(t);
    *out = &ctx->uv_tables.back();
    return LFG_OK;
}

// Both axes' tables of one interpolate call.  (The x table is looked up again after the y table has been built: the push may
// have moved the vector.)
int interp_tables(lfg_context *ctx, int width, int height, lfg::InterpTables *tb) {
    trim_uv_tables(ctx);
    const lfg::UvTable *tx = nullptr, *ty = nullptr;
    int rc = build_uv_table(ctx, width, &tx);
    if (rc == LFG_OK) rc = build_uv_table(ctx, height, &ty);
    if (rc == LFG_OK) rc = build_uv_table(ctx, width, &tx);
    if (rc != LFG_OK) return rc;
    tb->uvx = tx->d_uv; tb->uvy = ty->d_uv;
    tb->centreX = tx->d_centre; tb->centreY = ty->d_centre;
    tb->goodMask = tx->d_goodMask;
    tb->rcpW = tx->rcp; tb->rcpH = ty->rcp; tb->rcpExact = tx->rcpExact && ty->rcpExact ? 1 : 0;
    return LFG_OK;
}

// Bounded cache: called at the top of lfg_scale, before any table pointer is taken, so the two
// lookups that follow can never evict each other.
void trim_axis_tables(lfg_context *ctx) {
    while (ctx->tables.size() > 14) {
        (void)sync_lanes(ctx);                                 // a queued kernel may still read it
        lfg::AxisTable &old = ctx->tables.front();
        (void)hipFree(old.d_start); (void)hipFree(old.d_weight); (void)hipFree(old.d_class); (void)hipFree(old.d_palette);
        ctx->tables.erase(ctx->tables.begin());
    }
}

// ---- candidate tables of the 8/16 motion paths, both tie-break rules (tiny; built at the first lfg_motion)

int ensure_motion_tables(lfg_context *ctx) {
    if (ctx->motion_tables) return LFG_OK;
    std::vector<uint32_t> host(7 * lfg::kMotionTableWords, 0u);
    for (int sem = 0; sem < 2; ++sem)
        lfg::motion_tables(sem != 0, host.data() + (3 * sem) * lfg::kMotionTableWords,
                           host.data() + (3 * sem + 1) * lfg::kMotionTableWords,
                           host.data() + (3 * sem + 2) * lfg::kMotionTableWords, host.data() + 6 * lfg::kMotionTableWords);
    // Published only once the copy has succeeded: a later call must never find a half-initialised table.
    uint32_t *d = nullptr;
    LFG_HIP(ctx, hipMalloc((void **)&d, host.size() * sizeof(uint32_t)));
    const hipError_t e = hipMemcpy(d, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return fail_hip(ctx, e, "hipMemcpy(motion tables)"); }
    ctx->motion_tables = d;
    return LFG_OK;
}

// ---- scratch of the prefiltered motion path (per frame size; kept between calls)

// Parts of the candidate order per rim segment (motion_plan.hip: prefilter_plan), measured at 4K (frames/s):
//   4   the default with frames in flight (lfg_lanes >= 2), where the sum of all units' times is what counts: pan 2,838;
//   48  four, and eight for the segments whose position rows leave the image at its top or bottom -- the strip a vertical pan
//       exposes is the longest unit of a frame, and the launch is as long as its longest unit when ONE frame runs at a time:
//       the default there.  One frame at a time: pan 1,868 -> 2,307 (motion 0.49 -> 0.38 ms), noisy 889 -> 981, stills
//       3,880 -> 3,976, occluded 815 -> 810, moving objects 976 -> 971; with three frames in flight it costs 1 - 2 %
//       (134 more workgroups that stage a window each), which is why it is not used there;
//   8   every rim segment in eight parts: one frame at a time the pan gains less (2,050) and everything else loses 5 - 10 %;
//       9 % slower with three frames in flight.
// LFG_MOTION_RIM_SPLIT=4|8|48 overrides (read ONCE, when the context is created).  The plan therefore changes only where the lane
// count does -- inside lfg_lanes(), never silently between two lfg_motion calls -- and lfg_motion_plan() reports it.
int motion_rim_split(const lfg_context *ctx) {
    if (ctx->rim_split_env) return ctx->rim_split_env;
    return ctx->lanes.size() >= 2 ? 4 : 48;
}
// Round 4: with frames in flight a call whose content suits it -- the order kernel's verdict on the lane's previous call -- sends
// its whole interior tiles through the lean kernel first (motion_lean.hip).  What is left to the persistent kernel then is the
// rim, its longest units set the launch's length again, and 48 wins there as well (pan 3,250 -> 3,345 frames/s with three
// frames in flight; noise -5 %, stills -4 %: which is why it is this plan only for those calls).  Both plans are resident.
int motion_rim_split_lean(const lfg_context *ctx) {
    return (ctx->motion_lean && ctx->lanes.size() >= 2 && !ctx->rim_split_env) ? 48 : 0;
}

// Workgroups of the persistent kernel a launch of this context may have: what the device holds at once -- less the CUs a communicator
// keeps, where the library's streams cannot place any (lfg_own_stream_create).
int persistent_grid_most(const lfg_context *ctx) {
    if (ctx->comm_cus <= 0 || ctx->device_cus <= ctx->comm_cus) return ctx->motion_slots;
    return ctx->motion_slots / ctx->device_cus * (ctx->device_cus - ctx->comm_cus);
}

int ensure_motion_workspace(lfg_context *ctx, uint32_t width, uint32_t height) {
    const int rimSplit = motion_rim_split(ctx), rimSplit2 = motion_rim_split_lean(ctx);
    if (ctx->motion_ws && ctx->motion_ws_w == width && ctx->motion_ws_h == height && ctx->motion_ws_layout.rimSplit == rimSplit &&
        ctx->motion_ws_layout.rimSplit2 == rimSplit2) return LFG_OK;
    lfg::MotionWorkspaceLayout layout;
    if (ctx->motion_slots == 0) {
        ctx->motion_slots = lfg::prefilter_slots();
        if (ctx->knobs.debug) fprintf(stderr, "lfg: motion prefilter: %d workgroups resident at once\n", ctx->motion_slots);
    }
    const size_t bytes = lfg::motion_workspace_bytes(width, height, ctx->motion_slots, rimSplit, rimSplit2, &layout);
    if (bytes > ctx->motion_ws_bytes) {
        LFG_HIP(ctx, hipStreamSynchronize(ctx->stream));          // a queued kernel may still use the old one
        if (ctx->motion_ws) (void)hipFree(ctx->motion_ws);
        ctx->motion_ws = nullptr; ctx->motion_ws_bytes = 0; ctx->motion_ws_w = ctx->motion_ws_h = 0;
        LFG_HIP(ctx, hipMalloc((void **)&ctx->motion_ws, bytes));
        ctx->motion_ws_bytes = bytes;
    } else {
        LFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    // work-unit tables of this frame size: unitMap | unitAux | tileMap
    const lfg::PrefilterPlanHost plan = lfg::prefilter_plan(width, height, ctx->motion_slots, rimSplit);
    std::vector<uint32_t> tables;
    tables.insert(tables.end(), plan.unitMap.begin(), plan.unitMap.end());
    tables.insert(tables.end(), plan.unitAux.begin(), plan.unitAux.end());
    tables.insert(tables.end(), plan.tileMap.begin(), plan.tileMap.end());
    LFG_HIP(ctx, hipMemcpy(ctx->motion_ws + layout.plan, tables.data(), tables.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (rimSplit2) {        // the plan of the calls that go through the lean kernel, and that kernel's tiles
        const lfg::PrefilterPlanHost plan2 = lfg::prefilter_plan(width, height, ctx->motion_slots, rimSplit2);
        std::vector<uint32_t> t2;
        t2.insert(t2.end(), plan2.unitMap.begin(), plan2.unitMap.end());
        t2.insert(t2.end(), plan2.unitAux.begin(), plan2.unitAux.end());
        t2.insert(t2.end(), plan2.tileMap.begin(), plan2.tileMap.end());
        LFG_HIP(ctx, hipMemcpy(ctx->motion_ws + layout.plan2, t2.data(), t2.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        if (!plan2.leanTiles.empty()) {
            std::vector<uint32_t> listed = plan2.leanTiles;
            listed.insert(listed.end(), plan2.leanPartial.begin(), plan2.leanPartial.end());
            LFG_HIP(ctx, hipMemcpy(ctx->motion_ws + layout.leanTiles, listed.data(), listed.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    }
    // what the kernels expect to find between calls: a cleared control area (the hint kernel's counter of finished
    // workgroups lies in it, and that kernel is what clears the rest per call) and the merge words of the flagged tiles all ones
    LFG_HIP(ctx, hipMemset(ctx->motion_ws + layout.tileFlags, 0, layout.order - layout.tileFlags));
    LFG_HIP(ctx, hipMemset(ctx->motion_ws + layout.merge, 0xFF, layout.mergeBytes));
    ctx->motion_units = plan.units;
    layout.lastLean = 0;
    ctx->motion_ws_layout = layout;
    ctx->motion_ws_w = width; ctx->motion_ws_h = height;
    return LFG_OK;
}

// ---- profiling

struct StageTimer {
    lfg_context *ctx;
    hipEvent_t b = nullptr, e = nullptr;
    int stage;
    StageTimer(lfg_context *c, int st) : ctx(c), stage(st) {
        if (!ctx->profile) return;
        if (!ctx->prof_free.empty()) { b = ctx->prof_free.back().first; e = ctx->prof_free.back().second; ctx->prof_free.pop_back(); }
        else if (hipEventCreate(&b) != hipSuccess || hipEventCreate(&e) != hipSuccess) { b = e = nullptr; return; }
        (void)hipEventRecord(b, ctx->stream);
    }
    ~StageTimer() {
        if (!b) return;
        (void)hipEventRecord(e, ctx->stream);
        lfg::ProfileSlot s; s.begin = b; s.end = e; s.stage = stage;
        ctx->prof_pending.push_back(s);
    }
};

int drain_profile(lfg_context *ctx) {
    if (ctx->prof_pending.empty()) return LFG_OK;
    LFG_HIP(ctx, sync_lanes(ctx));
    for (auto &s : ctx->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.begin, s.end) == hipSuccess) { ctx->prof_ms[s.stage] += ms; ctx->prof_n[s.stage] += 1; }
        ctx->prof_free.emplace_back(s.begin, s.end);
    }
    ctx->prof_pending.clear();
    return LFG_OK;
}

}  // namespace

struct lfg_ring {
    lfg_context *ctx = nullptr;
    uint32_t slots = 0;
    size_t slot_bytes = 0;
    uint8_t *base = nullptr;                 // pinned host memory, slots * slot_bytes
    std::vector<hipEvent_t> done;            // last transfer touching each slot
    std::vector<uint8_t> busy;
    uint32_t next = 0;
    hipStream_t copy = nullptr;              // transfers run here, next to the kernels on ctx->stream
    hipEvent_t ready = nullptr;              // scratch event: "ctx->stream has reached this point"
};

// ================================================================== library / context

// A stream of the library's own.  While a communicator exists (ctx->comm_cus > 0, lfg_comm.cpp) it carries a CU mask that leaves
// the first comm_cus CUs to the communication stream, which is masked to exactly those: RCCL's device kernel needs CUs without a
// persistent prefilter workgroup on them (comm_probe.hip), and only the pair of masks guarantees it gets them (lfg_comm_init).  Mask bit i is CU i / 8 of XCD i % 8 on an MI355X (tools/probe_cu_mask.hip
// on the device: clearing bits 0 - 7 takes one CU out of each of the eight XCDs), so 8 | 16 | 24 | 32 bits are 1 - 4 CUs per XCD, where
// the hardware's round-robin of workgroups over the XCDs puts the communicator's 8 - 32 channels.
// (A stream made by hipExtStreamCreateWithCUMask has default flags: it synchronises with the NULL stream, which the library never uses.)
hipError_t lfg_own_stream_create(const lfg_context *ctx, hipStream_t *out) {
    if (!ctx || ctx->comm_cus <= 0) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    int cus = 0;
    hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    if (e != hipSuccess) return e;
    if (cus <= ctx->comm_cus) return hipErrorInvalidValue;
    std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0xffffffffu);
    if (cus % 32) mask.back() = (1u << (cus % 32)) - 1u;
    for (int i = 0; i < ctx->comm_cus; ++i) mask[(size_t)i / 32] &= ~(1u << (i % 32));
    return hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data());
}

// Every stream the library owns is made again under the current reservation (the communicator has just come or gone).  A stream the
// caller supplied (lfg_context_set_stream) is the caller's to mask: lfg_comm_cu_mask says how.
int lfg_restream(lfg_context *ctx) {
    LFG_HIP(ctx, hipSetDevice(ctx->device));
    auto again = [&](hipStream_t &own, hipStream_t &used) -> hipError_t {
        if (!own) return hipSuccess;
        hipError_t e = hipStreamSynchronize(own);
        hipStream_t fresh = nullptr;
        if (e == hipSuccess) e = lfg_own_stream_create(ctx, &fresh);
        if (e != hipSuccess) return e;
        (void)hipStreamDestroy(own);
        if (used == own) used = fresh;
        own = fresh;
        return hipSuccess;
    };
    LFG_HIP(ctx, again(ctx->own_stream, ctx->stream));
    for (size_t j = 0; j < ctx->lanes.size(); ++j)
        if ((int)j != ctx->lane) LFG_HIP(ctx, again(ctx->lanes[j].own_stream, ctx->lanes[j].stream));     // (the selected lane's are the context's own fields)
    return LFG_OK;
}

LFG_EXPORT int lfg_abi_version(void) { return LFG_ABI_VERSION; }

LFG_EXPORT int lfg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

LFG_EXPORT int lfg_context_create(int device_ordinal, lfg_context **out_ctx) {
    if (!out_ctx) return fail(nullptr, LFG_ERR_INVALID, "lfg_context_create: out_ctx is NULL");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, LFG_ERR_DEVICE, std::string("lfg_context_create: no HIP device (") +
                                                 (e != hipSuccess ? hipGetErrorString(e) : "count 0") + "); there is no CPU fallback");
    const int dev = device_ordinal < 0 ? 0 : device_ordinal;
    if (dev >= n) return fail(nullptr, LFG_ERR_INVALID, "lfg_context_create: device ordinal out of range");
    LFG_HIP(nullptr, hipSetDevice(dev));
    lfg_context *ctx = new (std::nothrow) lfg_context();
    if (!ctx) return fail(nullptr, LFG_ERR_NOMEM, "lfg_context_create: out of host memory");
    ctx->device = dev;
    (void)hipDeviceGetAttribute(&ctx->device_cus, hipDeviceAttributeMultiprocessorCount, dev);
    e = lfg_own_stream_create(ctx, &ctx->own_stream);
    if (e != hipSuccess) { delete ctx; return fail_hip(nullptr, e, "hipStreamCreate"); }
    ctx->stream = ctx->own_stream;
    ctx->tables.reserve(17);                 // AxisTable pointers handed out stay valid
    if (const char *m = getenv("LFG_MOTION_HINTS")) ctx->motion_hints = atoi(m) != 0;
    if (const char *m = getenv("LFG_MOTION_LEAN")) ctx->motion_lean = atoi(m) != 0;
    if (const char *m = getenv("LFG_MOTION_RIM_SPLIT")) { const int v = atoi(m); if (v == 4 || v == 8 || v == 48) ctx->rim_split_env = v; }
    if (const char *m = getenv("LFG_FUSED_INTERPOLATE_SCALE")) ctx->fuse_interpolate_scale = atoi(m) != 0;
    if (const char *m = getenv("LFG_FUSED_MOTION_INTERPOLATE")) ctx->fuse_motion_interpolate = atoi(m) != 0;
    if (const char *m = getenv("LFG_MOTION_MODE")) ctx->motion_mode = atoi(m) == 1 ? LFG_MOTION_EXACT_ONLY : LFG_MOTION_PREFILTERED;
    // measurement knobs (lfg_internal.hpp: MotionKnobs): here and nowhere else -- no call reads the environment
    if (const char *m = getenv("LFG_LEAN_FORCE")) ctx->knobs.leanForce = atoi(m) & 1;
    if (getenv("LFG_FALLBACK_FULL")) ctx->knobs.fallbackFull = 1;
    if (const char *m = getenv("LFG_DYN_PARTS_RT")) ctx->knobs.dynParts = atoi(m);
    if (const char *m = getenv("LFG_PREF_GROUPS")) ctx->knobs.prefGroups = atoi(m);
    if (const char *m = getenv("LFG_RESOLVE_GROUPS")) ctx->knobs.resolveGroups = atoi(m);
    if (const char *m = getenv("LFG_MOTION_STRIP")) ctx->knobs.strips = atoi(m) != 0;
    if (const char *m = getenv("LFG_STRIP_PAD")) ctx->knobs.stripPad = atoi(m);
    if (getenv("LFG_DEBUG")) ctx->knobs.debug = 1;
    if (getenv("LFG_DEBUG_DYN")) ctx->knobs.debugDyn = 1;
    if (const char *m = getenv("LFG_DEBUG_DYN_DEEP")) ctx->knobs.debugDynDeep = atoi(m);
    if (const char *m = getenv("LFG_TIER_FORCE")) ctx->knobs.tierForce = atoi(m) & 1;
    if (const char *m = getenv("LFG_COMM_CUS")) { const int v = atoi(m); if (v == 0 || v == 8 || v == 16 || v == 24 || v == 32) ctx->knobs.commCus = v; }
    *out_ctx = ctx;
    return LFG_OK;
}

// (lanes, further down: what a lane owns moves between the context's fields and its table entry)
namespace {
void lane_store(lfg_context *ctx, lfg_lane_state &l) {
    l.own_stream = ctx->own_stream; l.stream = ctx->stream; l.mv_tmp = ctx->mv_tmp; l.mid_tmp = ctx->mid_tmp;
    l.motion_ws = ctx->motion_ws; l.motion_ws_bytes = ctx->motion_ws_bytes; l.motion_ws_w = ctx->motion_ws_w; l.motion_ws_h = ctx->motion_ws_h;
    l.motion_ws_layout = ctx->motion_ws_layout; l.motion_units = ctx->motion_units; l.mark = ctx->mark; l.marked = ctx->marked;
    l.lean_flag = ctx->lean_flag; l.lean_ev = ctx->lean_ev; l.lean_ev_pending = ctx->lean_ev_pending; l.lean_predict = ctx->lean_predict; l.lean_seen = ctx->lean_seen; l.lean_request_guess = ctx->lean_request_guess;
}
void lane_load(lfg_context *ctx, const lfg_lane_state &l) {
    ctx->own_stream = l.own_stream; ctx->stream = l.stream; ctx->mv_tmp = l.mv_tmp; ctx->mid_tmp = l.mid_tmp;
    ctx->motion_ws = l.motion_ws; ctx->motion_ws_bytes = l.motion_ws_bytes; ctx->motion_ws_w = l.motion_ws_w; ctx->motion_ws_h = l.motion_ws_h;
    ctx->motion_ws_layout = l.motion_ws_layout; ctx->motion_units = l.motion_units; ctx->mark = l.mark; ctx->marked = l.marked;
    ctx->lean_flag = l.lean_flag; ctx->lean_ev = l.lean_ev; ctx->lean_ev_pending = l.lean_ev_pending; ctx->lean_predict = l.lean_predict; ctx->lean_seen = l.lean_seen; ctx->lean_request_guess = l.lean_request_guess;
}
void lane_release(lfg_lane_state &l) {
    if (l.stream) (void)hipStreamSynchronize(l.stream);
    if (l.own_stream && l.own_stream != l.stream) (void)hipStreamSynchronize(l.own_stream);
    if (l.mv_tmp.data && l.mv_tmp.owned) (void)hipFree(l.mv_tmp.data);
    if (l.mid_tmp.data && l.mid_tmp.owned) (void)hipFree(l.mid_tmp.data);
    if (l.motion_ws) (void)hipFree(l.motion_ws);
    if (l.mark) (void)hipEventDestroy(l.mark);
    if (l.lean_ev) (void)hipEventDestroy(l.lean_ev);
    if (l.lean_flag) (void)hipHostFree(l.lean_flag);
    if (l.own_stream) (void)hipStreamDestroy(l.own_stream);
    l = lfg_lane_state{};
}
}  // namespace

LFG_EXPORT void lfg_context_destroy(lfg_context *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)lfg_comm_destroy(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream != ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    for (size_t j = 0; j < ctx->lanes.size(); ++j)
        if ((int)j != ctx->lane) lane_release(ctx->lanes[j]);      // (the selected lane's resources are the context's own fields)
    if (ctx->mark) (void)hipEventDestroy(ctx->mark);
    if (ctx->lean_ev) (void)hipEventDestroy(ctx->lean_ev);
    if (ctx->lean_flag) (void)hipHostFree(ctx->lean_flag);
    for (auto &s : ctx->prof_pending) { (void)hipEventDestroy(s.begin); (void)hipEventDestroy(s.end); }
    for (auto &p : ctx->prof_free) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &t : ctx->tables) { (void)hipFree(t.d_start); (void)hipFree(t.d_weight); (void)hipFree(t.d_class); (void)hipFree(t.d_palette); }
    for (auto &t : ctx->uv_tables) (void)hipFree(t.d_uv);
    if (ctx->mv_tmp.data && ctx->mv_tmp.owned) (void)hipFree(ctx->mv_tmp.data);
    if (ctx->mid_tmp.data && ctx->mid_tmp.owned) (void)hipFree(ctx->mid_tmp.data);
    if (ctx->motion_tables) (void)hipFree(ctx->motion_tables);
    if (ctx->motion_ws) (void)hipFree(ctx->motion_ws);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

LFG_EXPORT int lfg_context_set_stream(lfg_context *ctx, void *hip_stream) {
    if (!ctx) return LFG_ERR_INVALID;
    int rc = drain_profile(ctx);
    if (rc != LFG_OK) return rc;
    LFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return LFG_OK;
}

LFG_EXPORT void *lfg_context_get_stream(lfg_context *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// ------------------------------------------------------------------ lanes: several frames in flight on one GPU

LFG_EXPORT int lfg_lanes(lfg_context *ctx, int count) {
    if (!ctx) return LFG_ERR_INVALID;
    if (count < 1 || count > LFG_MAX_LANES) return fail(ctx, LFG_ERR_INVALID, "lfg_lanes: count must be 1 .. LFG_MAX_LANES");
    LFG_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->lanes.empty()) { ctx->lanes.resize(1); ctx->lane = 0; }
    if (ctx->lane >= count) {                  // the selected lane is about to go: back to lane 0 first
        int rc = lfg_lane_select(ctx, 0);
        if (rc != LFG_OK) return rc;
    }
    while ((int)ctx->lanes.size() > count) { lane_release(ctx->lanes.back()); ctx->lanes.pop_back(); }
    while ((int)ctx->lanes.size() < count) {
        lfg_lane_state l;
        hipError_t e = lfg_own_stream_create(ctx, &l.own_stream);
        if (e != hipSuccess) return fail_hip(ctx, e, "hipStreamCreate (lane)");
        l.stream = l.own_stream;
        ctx->lanes.push_back(l);
    }
    return LFG_OK;
}

LFG_EXPORT int lfg_lane_count(const lfg_context *ctx) { return ctx ? std::max<int>(1, (int)ctx->lanes.size()) : 0; }
LFG_EXPORT int lfg_lane_current(const lfg_context *ctx) { return ctx ? ctx->lane : -1; }

LFG_EXPORT int lfg_lane_select(lfg_context *ctx, int lane) {
    if (!ctx) return LFG_ERR_INVALID;
    if (lane < 0 || lane >= std::max<int>(1, (int)ctx->lanes.size())) return fail(ctx, LFG_ERR_INVALID, "lfg_lane_select: no such lane (lfg_lanes first)");
    if (lane == ctx->lane) return LFG_OK;
    lane_store(ctx, ctx->lanes[(size_t)ctx->lane]);
    lane_load(ctx, ctx->lanes[(size_t)lane]);
    ctx->lane = lane;
    return LFG_OK;
}

LFG_EXPORT int lfg_lane_mark(lfg_context *ctx) {
    if (!ctx) return LFG_ERR_INVALID;
    LFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->mark) LFG_HIP(ctx, hipEventCreateWithFlags(&ctx->mark, hipEventDisableTiming));
    LFG_HIP(ctx, hipEventRecord(ctx->mark, ctx->stream));
    ctx->marked = true;
    return LFG_OK;
}

LFG_EXPORT int lfg_lane_wait(lfg_context *ctx, int other) {
    if (!ctx) return LFG_ERR_INVALID;
    if (other < 0 || other >= std::max<int>(1, (int)ctx->lanes.size())) return fail(ctx, LFG_ERR_INVALID, "lfg_lane_wait: no such lane");
    if (other == ctx->lane) return LFG_OK;     // a stream is in order with itself
    const lfg_lane_state &o = ctx->lanes[(size_t)other];
    if (!o.marked) return LFG_OK;              // nothing to wait for yet
    LFG_HIP(ctx, hipStreamWaitEvent(ctx->stream, o.mark, 0));
    return LFG_OK;
}

LFG_EXPORT int lfg_lane_sync(lfg_context *ctx) {
    if (!ctx) return LFG_ERR_INVALID;
    LFG_HIP(ctx, hipSetDevice(ctx->device));
    LFG_HIP(ctx, hipStreamSynchronize(ctx->stream));      // the selected lane's stream IS the context's current one
    return LFG_OK;
}

LFG_EXPORT int lfg_context_device(const lfg_context *ctx) { return ctx ? ctx->device : -1; }

LFG_EXPORT int lfg_sync(lfg_context *ctx) {
    if (!ctx) return LFG_ERR_INVALID;
    LFG_HIP(ctx, sync_lanes(ctx));             // every lane: "the device is idle" is what callers mean
    return LFG_OK;
}

LFG_EXPORT const char *lfg_last_error(const lfg_context *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

// ================================================================== frames

LFG_EXPORT int lfg_frame_create(lfg_context *ctx, uint32_t width, uint32_t height, uint32_t format, lfg_frame *out) {
    if (!ctx || !out) return fail(ctx, LFG_ERR_INVALID, "lfg_frame_create: NULL argument");
    const uint32_t bpp = bytes_per_pixel(format);
    if (!bpp || width == 0 || height == 0 || width > 32768u || height > 32768u)
        return fail(ctx, LFG_ERR_INVALID, "lfg_frame_create: bad size or format");
    LFG_HIP(ctx, hipSetDevice(ctx->device));
    void *p = nullptr;
    const size_t bytes = (size_t)width * (size_t)height * (size_t)bpp;
    LFG_HIP(ctx, hipMalloc(&p, bytes));
    out->data = p; out->width = width; out->height = height; out->pitch = width * bpp;
    out->format = format; out->owned = 1; out->reserved = 0;
    return LFG_OK;
}

LFG_EXPORT void lfg_frame_destroy(lfg_context *ctx, lfg_frame *frame) {
    if (!frame) return;
    if (frame->data && frame->owned) {
        if (ctx) { (void)hipSetDevice(ctx->device); (void)sync_lanes(ctx); }
        (void)hipFree(frame->data);
    }
    frame->data = nullptr; frame->width = frame->height = frame->pitch = 0; frame->owned = 0;
}

LFG_EXPORT int lfg_frame_wrap(void *device_ptr, uint32_t width, uint32_t height, uint32_t pitch, uint32_t format,
                              lfg_frame *out) {
    const uint32_t bpp = bytes_per_pixel(format);
    if (!out || !device_ptr || !bpp || width == 0 || height == 0 || pitch < width * bpp) return LFG_ERR_INVALID;
    out->data = device_ptr; out->width = width; out->height = height; out->pitch = pitch;
    out->format = format; out->owned = 0; out->reserved = 0;
    return LFG_OK;
}

LFG_EXPORT int lfg_frame_copy(lfg_context *ctx, const lfg_frame *src, lfg_frame *dst) {
    if (!ctx || !src || !dst || !src->data || !dst->data) return fail(ctx, LFG_ERR_INVALID, "lfg_frame_copy: NULL frame");
    if (!same_size(src, dst) || src->format != dst->format)                  // frame_manager.cpp:84-87
        return fail(ctx, LFG_ERR_INVALID, "Source and destination frame dimensions do not match");
    const size_t row = (size_t)src->width * bytes_per_pixel(src->format);
    LFG_HIP(ctx, hipMemcpy2DAsync(dst->data, dst->pitch, src->data, src->pitch, row, src->height,
                                  hipMemcpyDeviceToDevice, ctx->stream));
    return LFG_OK;
}

LFG_EXPORT int lfg_staging_create(lfg_context *ctx, size_t bytes, void **out_host_ptr) {
    if (!ctx || !out_host_ptr || bytes == 0) return fail(ctx, LFG_ERR_INVALID, "lfg_staging_create: bad argument");
    LFG_HIP(ctx, hipSetDevice(ctx->device));
    LFG_HIP(ctx, hipHostMalloc(out_host_ptr, bytes, hipHostMallocDefault));
    return LFG_OK;
}

LFG_EXPORT void lfg_staging_destroy(lfg_context *ctx, void *host_ptr) {
    if (!host_ptr) return;
    if (ctx) (void)sync_lanes(ctx);
    (void)hipHostFree(host_ptr);
}

LFG_EXPORT int lfg_frame_upload(lfg_context *ctx, lfg_frame *dst, const void *host, size_t bytes) {
    if (!ctx || !dst || !dst->data || !host) return fail(ctx, LFG_ERR_INVALID, "lfg_frame_upload: NULL argument");
    const size_t row = (size_t)dst->width * bytes_per_pixel(dst->format);
    if (bytes < row * dst->height) {                                          // window_capture.cpp:478-481
        char msg[160];
        snprintf(msg, sizeof msg, "Captured image size (%zu) smaller than expected (%zu)", bytes, row * dst->height);
        return fail(ctx, LFG_ERR_INVALID, msg);
    }
    LFG_HIP(ctx, hipMemcpy2DAsync(dst->data, dst->pitch, host, row, row, dst->height, hipMemcpyHostToDevice, ctx->stream));
    return LFG_OK;
}

LFG_EXPORT int lfg_frame_download(lfg_context *ctx, const lfg_frame *src, void *host, size_t bytes) {
    if (!ctx || !src || !src->data || !host) return fail(ctx, LFG_ERR_INVALID, "lfg_frame_download: NULL argument");
    const size_t row = (size_t)src->width * bytes_per_pixel(src->format);
    if (bytes < row * src->height) return fail(ctx, LFG_ERR_INVALID, "lfg_frame_download: host buffer too small");
    LFG_HIP(ctx, hipMemcpy2DAsync(host, row, src->data, src->pitch, row, src->height, hipMemcpyDeviceToHost, ctx->stream));
    return LFG_OK;
}

// ================================================================== pinned-host frame ring

LFG_EXPORT int lfg_ring_create(lfg_context *ctx, uint32_t slots, size_t slot_bytes, lfg_ring **out_ring) {
    if (!ctx || !out_ring || slots == 0 || slots > 64 || slot_bytes == 0)
        return fail(ctx, LFG_ERR_INVALID, "lfg_ring_create: bad argument");
    lfg_ring *r = new (std::nothrow) lfg_ring();
    if (!r) return fail(ctx, LFG_ERR_NOMEM, "lfg_ring_create: out of host memory");
    r->ctx = ctx; r->slots = slots; r->slot_bytes = (slot_bytes + 4095u) & ~(size_t)4095u;
    hipError_t e = hipHostMalloc((void **)&r->base, r->slot_bytes * slots, hipHostMallocDefault);
    if (e != hipSuccess) { delete r; return fail_hip(ctx, e, "hipHostMalloc(ring)"); }
    r->done.resize(slots, nullptr); r->busy.assign(slots, 0);
    e = hipStreamCreateWithFlags(&r->copy, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ready, hipEventDisableTiming);
    if (e != hipSuccess) { lfg_ring_destroy(r); return fail_hip(ctx, e, "hipStreamCreate(ring)"); }
    for (uint32_t i = 0; i < slots; ++i) {
        e = hipEventCreateWithFlags(&r->done[i], hipEventDisableTiming);
        if (e != hipSuccess) { lfg_ring_destroy(r); return fail_hip(ctx, e, "hipEventCreate(ring)"); }
    }
    *out_ring = r;
    return LFG_OK;
}

LFG_EXPORT void lfg_ring_destroy(lfg_ring *ring) {
    if (!ring) return;
    if (ring->copy) (void)hipStreamSynchronize(ring->copy);
    for (uint32_t i = 0; i < ring->slots; ++i)
        if (ring->done[i]) { if (ring->busy[i]) (void)hipEventSynchronize(ring->done[i]); (void)hipEventDestroy(ring->done[i]); }
    if (ring->ready) (void)hipEventDestroy(ring->ready);
    if (ring->copy) (void)hipStreamDestroy(ring->copy);
    if (ring->base) (void)hipHostFree(ring->base);
    delete ring;
}

LFG_EXPORT int lfg_ring_acquire(lfg_ring *ring, void **out_host_ptr, uint32_t *out_slot) {
    if (!ring || !out_host_ptr || !out_slot) return LFG_ERR_INVALID;
    const uint32_t s = ring->next;
    ring->next = (ring->next + 1) % ring->slots;
    if (ring->busy[s]) { LFG_HIP(ring->ctx, hipEventSynchronize(ring->done[s])); ring->busy[s] = 0; }
    *out_host_ptr = ring->base + (size_t)s * ring->slot_bytes;
    *out_slot = s;
    return LFG_OK;
}

// Transfers run on the ring's own stream so they overlap the kernels:
//   upload    copy stream waits for what ctx->stream has been given so far (earlier readers of `dst`),
//             copies, and ctx->stream then waits for the copy -- kernels enqueued next see the pixels;
//   download  copy stream waits for what ctx->stream has been given so far (the producers of `src`) and copies;
//             ctx->stream does NOT wait: before a kernel overwrites `src` again, call lfg_ring_fence_slot.
static int ring_transfer(lfg_ring *ring, uint32_t slot, const lfg_frame *f, bool upload) {
    lfg_context *ctx = ring->ctx;
    if (!f || !f->data) return fail(ctx, LFG_ERR_INVALID, "lfg_ring transfer: NULL frame");
    const size_t row = (size_t)f->width * bytes_per_pixel(f->format), need = row * f->height;
    if (ring->slot_bytes < need) return fail(ctx, LFG_ERR_INVALID, "lfg_ring transfer: slot smaller than the frame");
    LFG_HIP(ctx, hipSetDevice(ctx->device));
    uint8_t *host = ring->base + (size_t)slot * ring->slot_bytes;
    LFG_HIP(ctx, hipEventRecord(ring->ready, ctx->stream));
    LFG_HIP(ctx, hipStreamWaitEvent(ring->copy, ring->ready, 0));
    if (upload) LFG_HIP(ctx, hipMemcpy2DAsync(f->data, f->pitch, host, row, row, f->height, hipMemcpyHostToDevice, ring->copy));
    else LFG_HIP(ctx, hipMemcpy2DAsync(host, row, f->data, f->pitch, row, f->height, hipMemcpyDeviceToHost, ring->copy));
    LFG_HIP(ctx, hipEventRecord(ring->done[slot], ring->copy));
    ring->busy[slot] = 1;
    if (upload) LFG_HIP(ctx, hipStreamWaitEvent(ctx->stream, ring->done[slot], 0));
    return LFG_OK;
}

LFG_EXPORT int lfg_ring_upload(lfg_ring *ring, uint32_t slot, lfg_frame *dst) {
    if (!ring || slot >= ring->slots) return LFG_ERR_INVALID;
    return ring_transfer(ring, slot, dst, true);
}

LFG_EXPORT int lfg_ring_download(lfg_ring *ring, uint32_t slot, const lfg_frame *src) {
    if (!ring || slot >= ring->slots) return LFG_ERR_INVALID;
    return ring_transfer(ring, slot, src, false);
}

LFG_EXPORT int lfg_ring_wait(lfg_ring *ring, uint32_t slot) {
    if (!ring || slot >= ring->slots) return LFG_ERR_INVALID;
    if (ring->busy[slot]) { LFG_HIP(ring->ctx, hipEventSynchronize(ring->done[slot])); ring->busy[slot] = 0; }
    return LFG_OK;
}

LFG_EXPORT int lfg_ring_fence_slot(lfg_ring *ring, uint32_t slot) {
    if (!ring || slot >= ring->slots) return LFG_ERR_INVALID;
    if (ring->busy[slot]) LFG_HIP(ring->ctx, hipStreamWaitEvent(ring->ctx->stream, ring->done[slot], 0));
    return LFG_OK;
}

// ================================================================== stages

LFG_EXPORT int lfg_scale(lfg_context *ctx, const lfg_frame *in, lfg_frame *out) {
    if (!ctx) return LFG_ERR_INVALID;
    LFG_HIP(ctx, hipSetDevice(ctx->device));          // the stream belongs to this device (multi-GPU hosts)
    if (!frame_ok(in, LFG_FORMAT_RGBA8_UNORM) || !frame_ok(out, LFG_FORMAT_RGBA8_UNORM))
        return fail(ctx, LFG_ERR_INVALID, "lfg_scale: frames must be non-empty RGBA8");
    if (in->data == out->data) return fail(ctx, LFG_ERR_INVALID, "lfg_scale: in-place scaling is not supported");
    trim_axis_tables(ctx);
    lfg::AxisTable *tx = nullptr, *ty = nullptr;               // vector capacity is reserved: pointers stay valid
    int rc = build_axis_table(ctx, (int)in->width, (int)out->width, &tx);
    if (rc != LFG_OK) return rc;
    rc = build_axis_table(ctx, (int)in->height, (int)out->height, &ty);
    if (rc != LFG_OK) return rc;
    const bool fast = tx->pattern_2x && ty->pattern_2x && tx->palette_rows > 0 && ty->strips_per_xcd > 0 && lfg::scale_2x_supported(*in, *out);
    StageTimer timer(ctx, LFG_STAGE_SCALE);
    hipError_t e = fast ? lfg::launch_scale_2x(ctx->stream, *in, *out, *tx, *ty)
                        : lfg::launch_scale_generic(ctx->stream, *in, *out, *tx, *ty);
    if (e != hipSuccess) return fail_hip(ctx, e, "scale kernel launch");
    return LFG_OK;
}

// lfg_motion, and -- fused != nullptr -- the motion stage of lfg_interpolate_frames in the north-star order: the kernels
// write the generated frame themselves (lfg_internal.hpp: FusedOut).  *fusedDone tells the caller whether they did (the
// generic kernel, for other block sizes, radii and frames of 2 GiB, does not).
static int motion_run(lfg_context *ctx, const lfg_frame *prev, const lfg_frame *curr, lfg_frame *mv,
                      int block_size, float search_radius, const lfg::FusedOut *fused, bool *fusedDone) {
    if (fusedDone) *fusedDone = false;
    if (!ctx) return LFG_ERR_INVALID;
    LFG_HIP(ctx, hipSetDevice(ctx->device));          // the stream belongs to this device (multi-GPU hosts)
    if (!frame_ok(prev, LFG_FORMAT_RGBA8_UNORM) || !frame_ok(curr, LFG_FORMAT_RGBA8_UNORM) ||
        !frame_ok(mv, LFG_FORMAT_MV_S8X2))
        return fail(ctx, LFG_ERR_INVALID, "lfg_motion: prev/curr must be RGBA8 and mv MV_S8X2, all non-empty");
    if (!same_size(prev, curr) || !same_size(curr, mv))
        return fail(ctx, LFG_ERR_INVALID, "lfg_motion: prev, curr and mv differ in size");
    if (block_size < 1 || block_size > 64) return fail(ctx, LFG_ERR_UNSUPPORTED, "lfg_motion: blockSize must be in [1,64]");
    if (!(search_radius >= 0.0f) || search_radius > 127.0f || search_radius != std::floor(search_radius))
        return fail(ctx, LFG_ERR_UNSUPPORTED, "lfg_motion: searchRadius must be a whole number in [0,127]");
    if ((prev->pitch | curr->pitch) % 4u || ((uintptr_t)prev->data | (uintptr_t)curr->data) % 4u)
        return fail(ctx, LFG_ERR_INVALID, "lfg_motion: RGBA8 frames must be 4-byte aligned");
    const int R = (int)search_radius;
    // The 8/16 kernels address rows with 32-bit byte offsets and lean on the buffer range check for rows outside
    // the image ("negative" offsets wrap to >= 2^31): both hold only below 2 GiB per frame.  Larger frames take the
    // generic kernel, which indexes with size_t.
    const bool fits32 = (uint64_t)prev->height * prev->pitch < 0x7fffffffull && (uint64_t)curr->height * curr->pitch < 0x7fffffffull &&
                        (uint64_t)mv->height * mv->pitch < 0x7fffffffull;
    const bool tiled = block_size == 8 && R == 16 && fits32;
    const uint32_t *rank2scan = nullptr, *order32 = nullptr;
    if (tiled) {
        int rc = ensure_motion_tables(ctx);
        if (rc != LFG_OK) return rc;
        rank2scan = ctx->motion_tables + (3 * ctx->semantics) * lfg::kMotionTableWords;
        order32 = rank2scan + lfg::kMotionTableWords;
    }
    if (tiled && ctx->motion_mode == LFG_MOTION_PREFILTERED) {
        int rc = ensure_motion_workspace(ctx, curr->width, curr->height);
        if (rc != LFG_OK) return rc;
    }
    StageTimer timer(ctx, LFG_STAGE_MOTION);
    hipError_t e;
    const lfg::FusedOut fo = (fused && tiled) ? *fused : lfg::FusedOut();
    if (tiled && ctx->motion_mode == LFG_MOTION_PREFILTERED) {
        // The lean kernel (motion_lean.hip) and the plan that goes with it: with frames in flight only (one frame at a time it sits
        // in front of the rim's long units: 2,620 -> 2,360 frames/s), under either tie order (round 5: the intended order's ranks go
        // through rank2scan), and for content that suits it -- which the ORDER kernel knows (a pan, an object's motion: most sample blocks
        // match nearly but not exactly) and the host learns one call late: the verdict of the lane's last finished call decides.
        // A kernel that finds out on the device that it has nothing to do still has to be placed, 2,144 workgroups of 48 KB of LDS
        // behind the other lanes' persistent kernels: -7 % on noisy frames, measured.
        const bool leanPossible = ctx->motion_ws_layout.rimSplit2 != 0 && ctx->motion_hints;
        // The same word also says whether most sample blocks had a match at all.  With frames in flight the persistent kernel then
        // runs with 5/8 of the workgroups the device holds: each draws more units, fewer slots idle in a launch's tail, and the other
        // lanes' kernels find room beside it -- pan 3,660 -> 3,800 frames/s, stills +8 %, noise +3.7 %, occlusions and moving objects
        // +2 %; frames without a match anywhere (every segment searched in full: the slots are what they need) keep the full grid
        // (5/8 there: -3.4 %).
        const bool flagWanted = ctx->lanes.size() >= 2 && ctx->motion_hints;
        if (flagWanted && !ctx->lean_flag) {
            LFG_HIP(ctx, hipHostMalloc((void **)&ctx->lean_flag, sizeof(uint32_t), hipHostMallocDefault));
            *ctx->lean_flag = 0u;
            LFG_HIP(ctx, hipEventCreateWithFlags(&ctx->lean_ev, hipEventDisableTiming));
        }
        if (ctx->lean_ev_pending && hipEventQuery(ctx->lean_ev) == hipSuccess) {
            ctx->lean_predict = (int)*ctx->lean_flag; ctx->lean_ev_pending = false; ctx->lean_seen = true;
            // (lfg_motion_prediction_stats) the call that delivered this word had been launched on lean_request_guess -- bit 0: it went
            // through the lean kernel, bit 1: that kernel was available to it, bit 31: its persistent grid was sized for "most sample
            // blocks match", bit 30: its second pass was the small looping grid -- and its own content said:
            const uint32_t said = (uint32_t)ctx->lean_predict, guess = ctx->lean_request_guess;
            ctx->pred_verdicts += 1;
            if ((guess & 2u) && ((guess ^ said) & 1u)) ctx->pred_lean_wrong += 1;
            if (((guess ^ said) >> 31) || (((guess ^ said) >> 29) & 1u)) ctx->pred_grid_wrong += 1;      // (the persistent launch: its grid, or its variant)
            if (((guess >> 30) & 1u) && ((said >> 30) & 1u)) ctx->pred_second_wrong += 1;       // (small grid, and tiles were flagged: the costly direction)
        }
        if (ctx->knobs.leanForce >= 0) ctx->lean_predict = (ctx->lean_predict & ~1) | ctx->knobs.leanForce;          // (measurement: 1 = every call, 0 = none)
        ctx->motion_ws_layout.lastLean = (leanPossible && ctx->motion_ws_layout.leanCount > 0 && (ctx->lean_predict & 1) != 0 && !fo.data && lfg::lean_frames_ok(*prev, *curr, *mv)) ? 1 : 0;
        // (only while another lane has work queued or running -- a stream query each: a call that has the device to itself takes the
        //  full grid, and is as long as on a context without lanes)
        bool othersBusy = false;
        for (size_t j = 0; j < ctx->lanes.size() && !othersBusy; ++j)
            if ((int)j != ctx->lane && ctx->lanes[j].stream && hipStreamQuery(ctx->lanes[j].stream) == hipErrorNotReady) othersBusy = true;
        int groupsCap = (flagWanted && othersBusy && ((uint32_t)ctx->lean_predict >> 31) != 0u) ? std::max(1, ctx->motion_slots * 5 / 8) : 0;
        // ... and bit 30 whether that call sent a tile through the literal kernel (flat content under a fade, exact ties): if not, this
        // call's fallback launch is 64 workgroups instead of 2,048 -- they take whatever it flags after all, in turns (1.4 % of the
        // frame rate under a pan: workgroups of 42 KB of LDS that read a count and leave still have to be placed)
        if (ctx->comm_cus > 0) groupsCap = groupsCap ? std::min(groupsCap, persistent_grid_most(ctx)) : persistent_grid_most(ctx);
        // ... and bit 29 which variant of the persistent kernel: the one with the eight-point walk by SADs where half the sample blocks matched
        // moderately well (sensor noise of +-3 .. +-6 levels at the input; motion_prefilter.hip, kTier)
        const int tier = ctx->knobs.tierForce >= 0 ? ctx->knobs.tierForce : (flagWanted && ctx->lean_seen && (((uint32_t)ctx->lean_predict >> 29) & 1u)) ? 1 : 0;
        ctx->motion_last_tier = fo.data ? 0 : tier;
        const bool expectNoFallback = flagWanted && ctx->lean_seen && (((uint32_t)ctx->lean_predict >> 30) & 1u) == 0u && !ctx->knobs.fallbackFull /* (measurement) */;
        e = lfg::launch_motion_prefiltered_8_16(ctx->stream, *prev, *curr, *mv, ctx->motion_ws, ctx->motion_ws_layout, ctx->motion_units,
                                                rank2scan, order32, order32 + lfg::kMotionTableWords,
                                                ctx->motion_tables + 6 * lfg::kMotionTableWords, ctx->motion_hints, ctx->lanes.size() >= 2, fo,
                                                ctx->motion_ws_layout.lastLean != 0, (flagWanted && !ctx->lean_ev_pending) ? ctx->lean_flag : nullptr, groupsCap, expectNoFallback, ctx->knobs, ctx->semantics == 0, tier);
        if (e == hipSuccess && flagWanted && !ctx->lean_ev_pending && !fo.data) {
            e = hipEventRecord(ctx->lean_ev, ctx->stream); ctx->lean_ev_pending = true;
            ctx->lean_request_guess = (ctx->motion_ws_layout.lastLean ? 1u : 0u) | ((leanPossible && ctx->motion_ws_layout.leanCount > 0) ? 2u : 0u) |
                                      ((uint32_t)ctx->lean_predict & 0x80000000u) | (expectNoFallback ? 1u << 30 : 0u) | (tier ? 1u << 29 : 0u);
        }
    }
    else if (tiled) e = lfg::launch_motion_tiled_8_16(ctx->stream, *prev, *curr, *mv, nullptr, rank2scan, nullptr, nullptr, fo);
    else e = lfg::launch_motion_generic(ctx->stream, *prev, *curr, *mv, block_size, R, ctx->semantics != 0);
    if (e != hipSuccess) return fail_hip(ctx, e, "motion kernel launch");
    if (fusedDone) *fusedDone = fo.data != nullptr;
    return LFG_OK;
}

LFG_EXPORT int lfg_motion_last_variant(const lfg_context *ctx) { return ctx ? ctx->motion_last_tier : -1; }

LFG_EXPORT int lfg_motion(lfg_context *ctx, const lfg_frame *prev, const lfg_frame *curr, lfg_frame *mv,
                          int block_size, float search_radius) {
    return motion_run(ctx, prev, curr, mv, block_size, search_radius, nullptr, nullptr);
}

LFG_EXPORT int lfg_set_semantics(lfg_context *ctx, int semantics) {
    if (!ctx || (semantics != LFG_SEMANTICS_REFERENCE && semantics != LFG_SEMANTICS_INTENDED))
        return fail(ctx, LFG_ERR_INVALID, "lfg_set_semantics: unknown semantics");
    ctx->semantics = semantics;
    return LFG_OK;
}

LFG_EXPORT int lfg_set_motion_mode(lfg_context *ctx, int mode) {
    if (!ctx || (mode != LFG_MOTION_PREFILTERED && mode != LFG_MOTION_EXACT_ONLY))
        return fail(ctx, LFG_ERR_INVALID, "lfg_set_motion_mode: unknown mode");
    ctx->motion_mode = mode;
    return LFG_OK;
}

LFG_EXPORT int lfg_motion_workspace_size(lfg_context *ctx, uint32_t width, uint32_t height, uint64_t *out_bytes) {
    if (!ctx) return LFG_ERR_INVALID;
    if (!out_bytes || width == 0 || height == 0 || width > 32768u || height > 32768u)
        return fail(ctx, LFG_ERR_INVALID, "lfg_motion_workspace_size: null output or a frame size outside 1..32768");
    if (ctx->motion_slots == 0) ctx->motion_slots = lfg::prefilter_slots();
    *out_bytes = (uint64_t)lfg::motion_workspace_bytes(width, height, ctx->motion_slots, motion_rim_split(ctx), motion_rim_split_lean(ctx), nullptr);
    return LFG_OK;
}

LFG_EXPORT int lfg_motion_plan(const lfg_context *ctx, int *out_rim_split, int *out_workgroups) {
    if (!ctx) return LFG_ERR_INVALID;
    if (out_rim_split) *out_rim_split = motion_rim_split(ctx);
    if (out_workgroups) *out_workgroups = persistent_grid_most(ctx);
    return LFG_OK;
}

// Reporting only: how many 16-row segments the prefilter of the last lfg_motion left to the resolve kernel, of how many the
// frame has (64 x 64 tiles x 4).  A segment is listed once; tests hold the list to that.
LFG_EXPORT int lfg_motion_open_segments(lfg_context *ctx, uint32_t *out_open, uint32_t *out_segments) {
    if (!ctx) return LFG_ERR_INVALID;
    if (!out_open || !out_segments) return fail(ctx, LFG_ERR_INVALID, "lfg_motion_open_segments: NULL argument");
    if (!ctx->motion_ws || ctx->motion_ws_w == 0) return fail(ctx, LFG_ERR_INVALID, "lfg_motion_open_segments: the prefiltered path has not run");
    LFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t tx = (ctx->motion_ws_w + 55u) / 56u, ty = (ctx->motion_ws_h + 63u) / 64u;      // (prefilter tiles: 56 x 64 pixels)
    uint32_t open = 0;
    LFG_HIP(ctx, hipMemcpy(&open, ctx->motion_ws + ctx->motion_ws_layout.ctrl + lfg::kCtrlOpenCount * sizeof(uint32_t), sizeof(uint32_t), hipMemcpyDeviceToHost));
    *out_open = open;
    *out_segments = tx * ty * 4u;
    return LFG_OK;
}

// Reporting only: did the selected lane's last lfg_motion go through the lean kernel (csrc/motion_lean.hip), how many whole
// interior tiles are listed for it at this frame size, and in how many of them it left a segment to the persistent kernel.
LFG_EXPORT int lfg_motion_lean_stats(lfg_context *ctx, int *out_used, uint32_t *out_tiles, uint32_t *out_tiles_left) {
    if (!ctx) return LFG_ERR_INVALID;
    if (!out_used || !out_tiles || !out_tiles_left) return fail(ctx, LFG_ERR_INVALID, "lfg_motion_lean_stats: NULL argument");
    if (!ctx->motion_ws || ctx->motion_ws_w == 0) return fail(ctx, LFG_ERR_INVALID, "lfg_motion_lean_stats: the prefiltered path has not run");
    LFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t left = 0;
    LFG_HIP(ctx, hipMemcpy(&left, ctx->motion_ws + ctx->motion_ws_layout.ctrl + lfg::kCtrlHardCount * sizeof(uint32_t), sizeof(uint32_t), hipMemcpyDeviceToHost));
    *out_used = ctx->motion_ws_layout.lastLean;
    *out_tiles = (uint32_t)ctx->motion_ws_layout.leanCount;
    *out_tiles_left = ctx->motion_ws_layout.lastLean ? left : 0u;
    return LFG_OK;
}

LFG_EXPORT int lfg_motion_strip_stats(lfg_context *ctx, uint32_t *out_rows, uint32_t *out_columns) {
    if (!ctx) return LFG_ERR_INVALID;
    if (!out_rows || !out_columns) return fail(ctx, LFG_ERR_INVALID, "lfg_motion_strip_stats: NULL argument");
    if (!ctx->motion_ws || ctx->motion_ws_w == 0) return fail(ctx, LFG_ERR_INVALID, "lfg_motion_strip_stats: the prefiltered path has not run");
    LFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> t((size_t)ctx->motion_ws_h + ctx->motion_ws_w);
    LFG_HIP(ctx, hipMemcpy(t.data(), ctx->motion_ws + ctx->motion_ws_layout.colBand, t.size() * 4, hipMemcpyDeviceToHost));
    uint32_t rows = 0, cols = 0;
    for (uint32_t y = 0; y < ctx->motion_ws_h; ++y) rows += t[y] != 0u;
    for (uint32_t x = 0; x < ctx->motion_ws_w; ++x) cols += t[ctx->motion_ws_h + x] != 0u;
    *out_rows = rows; *out_columns = cols;
    return LFG_OK;
}

LFG_EXPORT int lfg_motion_prediction_stats(const lfg_context *ctx, uint64_t *out_verdicts, uint64_t *out_lean_wrong,
                                           uint64_t *out_grid_wrong, uint64_t *out_second_pass_wrong) {
    if (!ctx) return LFG_ERR_INVALID;
    if (out_verdicts) *out_verdicts = ctx->pred_verdicts;
    if (out_lean_wrong) *out_lean_wrong = ctx->pred_lean_wrong;
    if (out_grid_wrong) *out_grid_wrong = ctx->pred_grid_wrong;
    if (out_second_pass_wrong) *out_second_pass_wrong = ctx->pred_second_wrong;
    return LFG_OK;
}

LFG_EXPORT int lfg_motion_last_stats(lfg_context *ctx, uint32_t *out_tiles, uint32_t *out_fallback_tiles,
                                     double *out_mean_recorded) {
    if (!ctx) return LFG_ERR_INVALID;
    if (!ctx->motion_ws || ctx->motion_ws_w == 0) return fail(ctx, LFG_ERR_INVALID, "lfg_motion_last_stats: the prefiltered path has not run");
    LFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t tx = (ctx->motion_ws_w + 63u) / 64u, ty = (ctx->motion_ws_h + 63u) / 64u;
    std::vector<uint32_t> flags((size_t)tx * ty);
    LFG_HIP(ctx, hipMemcpy(flags.data(), ctx->motion_ws + ctx->motion_ws_layout.tileFlags, flags.size() * 4, hipMemcpyDeviceToHost));
    uint32_t fb = 0;
    for (uint32_t f : flags) fb += f != 0u;
    if (ctx->knobs.debug) {
        uint32_t handed[2] = {0, 0};
        LFG_HIP(ctx, hipMemcpy(handed, ctx->motion_ws + ctx->motion_ws_layout.queueCount, 8, hipMemcpyDeviceToHost));
        fprintf(stderr, "lfg: motion prefilter: %u requests to hand a segment over (room for %d), %u tiles flagged for the exact kernel\n",
                handed[0], ctx->motion_ws_layout.queueCap, handed[1]);
        uint32_t lean[2] = {0, 0};
        LFG_HIP(ctx, hipMemcpy(lean, ctx->motion_ws + ctx->motion_ws_layout.ctrl + lfg::kCtrlLeanSettled * sizeof(uint32_t), 8, hipMemcpyDeviceToHost));
        uint32_t flags[3] = {0, 0, 0};
        LFG_HIP(ctx, hipMemcpy(flags, ctx->motion_ws + ctx->motion_ws_layout.orderFlags, 12, hipMemcpyDeviceToHost));
        fprintf(stderr, "lfg: lean kernel: %d tiles listed, %u segments settled, %u left to the generic kernel (counted in -DLFG_LEAN_STATS builds); order flags: hand-over %u, hints %u, lean %u (sample blocks with a close match %u, with an exact one %u)\n",
                ctx->motion_ws_layout.leanCount, lean[0], lean[1], flags[0], flags[1], flags[2] & 1u, (flags[2] >> 1) & 0x7FFu, (flags[2] >> 12) & 0x7FFu);
    }
    if (ctx->knobs.debugDyn) {       // the deepest private lists of the handed-over segments: block (4 x queue slot + wave), pixel, records
        uint32_t handed[2] = {0, 0};
        LFG_HIP(ctx, hipMemcpy(handed, ctx->motion_ws + ctx->motion_ws_layout.queueCount, 8, hipMemcpyDeviceToHost));
        const size_t blocks = (size_t)std::min<uint32_t>(handed[0], (uint32_t)ctx->motion_ws_layout.queueCap) * 4u, per = 16u * 56u;
        std::vector<uint32_t> dc(blocks * per);
        if (!dc.empty()) LFG_HIP(ctx, hipMemcpy(dc.data(), ctx->motion_ws + ctx->motion_ws_layout.dynCount, dc.size() * 4, hipMemcpyDeviceToHost));
        size_t hist[40] = {0};
        int dumped = 0;
        const uint32_t deep = (uint32_t)ctx->knobs.debugDynDeep;
        for (size_t i = 0; i < dc.size(); ++i) {
            ++hist[std::min<uint32_t>(dc[i], 39u)];
            if (dc[i] > deep) {
                fprintf(stderr, "lfg: dyn list block %zu (part %zu of its segment) pixel %zu (row %zu, column %zu): %u records\n", i / per, (i / per) % 8u, i % per, (i % per) / 56u, (i % per) % 56u, dc[i]);
                if (dumped++ < 6) {      // the records themselves: lower bound of the cost and the candidate's rank, in the order they were recorded
                    const size_t blk = i / per, row = (i % per) / 56u, col = (i % per) % 56u;
                    const int K = ctx->motion_ws_layout.listDyn;
                    for (int k = 0; k < std::min<int>((int)dc[i], K); ++k) {
                        uint32_t rec = 0; float thr = 0.f;
                        (void)hipMemcpy(&rec, ctx->motion_ws + ctx->motion_ws_layout.dynList + (((blk * 16u + row) * (size_t)K + (size_t)k) * 56u + col) * 4u, 4, hipMemcpyDeviceToHost);
                        (void)hipMemcpy(&thr, ctx->motion_ws + ctx->motion_ws_layout.dynUmin + ((blk * 16u + row) * 56u + col) * 4u, 4, hipMemcpyDeviceToHost);
                        const uint32_t bits = (rec >> 11) << 10; float c; memcpy(&c, &bits, 4);
                        fprintf(stderr, "      record %2d: cost >= %.3f rank %4u (dx %+d, dy %+d)%s\n", k, c, rec & 0x7FFu, (int)((rec & 0x7FFu) % 33u) - 16, (int)((rec & 0x7FFu) / 33u) - 16, k == 0 ? "" : "");
                        if (k + 1 == std::min<int>((int)dc[i], K)) fprintf(stderr, "      final threshold %.3f\n", thr);
                    }
                }
            }
        }
        {   // which segments under the flagged tiles were handed over
            const uint32_t ptx = (ctx->motion_ws_w + 55u) / 56u, pty = (ctx->motion_ws_h + 63u) / 64u;
            std::vector<uint32_t> sm((size_t)ptx * pty * 4u);
            LFG_HIP(ctx, hipMemcpy(sm.data(), ctx->motion_ws + ctx->motion_ws_layout.segMap, sm.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < flags.size(); ++i) if (flags[i]) {
                const uint32_t fx = (uint32_t)(i % tx), fy = (uint32_t)(i / tx);
                for (uint32_t px = fx * 64u / 56u; px <= std::min(ptx - 1u, (fx * 64u + 63u) / 56u); ++px)
                    for (uint32_t sg = 0; sg < 4u; ++sg)
                        fprintf(stderr, "lfg: flagged tile (%u,%u) flag %#x: prefilter tile (%u,%u) segment %u segMap %#x\n", fx, fy, flags[i], px, fy, sg, sm[((size_t)fy * ptx + px) * 4u + sg]);
            }
        }
        for (size_t b = 0; b < blocks; ++b) {      // blocks nobody wrote counts into: their workgroup gave up (or the segment settled)
            bool any = false;
            for (size_t i = 0; i < per && !any; ++i) any = dc[b * per + i] != 0u;
            if (!any) fprintf(stderr, "lfg: dyn block %zu (part %zu) holds no counts\n", b, b % 8u);
        }
        {
            double sum[8] = {0}; size_t n8[8] = {0};
            for (size_t i = 0; i < dc.size(); ++i) { sum[(i / per) % 8u] += dc[i]; ++n8[(i / per) % 8u]; }
            fprintf(stderr, "lfg: mean records per pixel by part:");
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %.2f", n8[k] ? sum[k] / n8[k] : 0.0);
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "lfg: dyn list depths over %zu pixel-parts:", dc.size());
        for (int k = 0; k < 40; ++k) if (hist[k]) fprintf(stderr, " %d:%zu", k, hist[k]);
        fprintf(stderr, "\n");
    }
    if (ctx->knobs.debug)
        for (size_t i = 0; i < flags.size(); ++i)
            if (flags[i]) fprintf(stderr, "lfg: motion fallback tile (%zu, %zu)\n", i % tx, i / tx);
    if (out_tiles) *out_tiles = tx * ty;
    if (out_fallback_tiles) *out_fallback_tiles = fb;
    if (out_mean_recorded) {
        const size_t px = (size_t)ctx->motion_ws_w * ctx->motion_ws_h;
        std::vector<uint32_t> cnt(px);
        LFG_HIP(ctx, hipMemcpy(cnt.data(), ctx->motion_ws + ctx->motion_ws_layout.count, px * 4, hipMemcpyDeviceToHost));
        // (tiles whose candidates were shared between several workgroups keep their counts elsewhere: left out)
        const lfg::PrefilterPlanHost plan = lfg::prefilter_plan(ctx->motion_ws_w, ctx->motion_ws_h, ctx->motion_slots,
                                                                ctx->motion_ws_layout.lastLean ? ctx->motion_ws_layout.rimSplit2 : ctx->motion_ws_layout.rimSplit);
        // (a segment that settled all of its pixels in the prefilter wrote no counts: its pixels hold at most two records,
        //  counted as none here)
        std::vector<uint32_t> segDone((size_t)plan.tiles * 4u);
        LFG_HIP(ctx, hipMemcpy(segDone.data(), ctx->motion_ws + ctx->motion_ws_layout.segDone, segDone.size() * 4, hipMemcpyDeviceToHost));
        double sum = 0; size_t n = 0;
        for (uint32_t y = 0; y < ctx->motion_ws_h; ++y)
            for (uint32_t x = 0; x < ctx->motion_ws_w; ++x) {
                const size_t ptile = (size_t)(y / 64u) * (size_t)plan.tilesX + x / 56u;
                if (!flags[(size_t)(y / 64u) * tx + x / 64u] && plan.tileMap[ptile] == 0xFFFFFFFFu) {
                    sum += segDone[ptile * 4u + (y % 64u) / 16u] ? 0.0 : (double)cnt[(size_t)y * ctx->motion_ws_w + x]; ++n;
                }
            }
        *out_mean_recorded = n ? sum / (double)n : 0.0;
    }
    return LFG_OK;
}

LFG_EXPORT int lfg_interpolate(lfg_context *ctx, const lfg_frame *prev, const lfg_frame *curr, const lfg_frame *mv,
                               lfg_frame *out, float factor) {
    if (!ctx) return LFG_ERR_INVALID;
    LFG_HIP(ctx, hipSetDevice(ctx->device));          // the stream belongs to this device (multi-GPU hosts)
    if (!frame_ok(prev, LFG_FORMAT_RGBA8_UNORM) || !frame_ok(curr, LFG_FORMAT_RGBA8_UNORM) ||
        !frame_ok(mv, LFG_FORMAT_MV_S8X2) || !frame_ok(out, LFG_FORMAT_RGBA8_UNORM))
        return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate: bad frame (NULL, empty or wrong format)");
    if (!same_size(prev, curr) || !same_size(curr, mv) || !same_size(curr, out))
        return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate: frames differ in size");
    if ((prev->pitch | curr->pitch | out->pitch) % 4u || mv->pitch % 2u)
        return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate: row pitch not a multiple of the pixel size");
    if (out->data == prev->data || out->data == curr->data)
        return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate: output aliases an input");
    lfg::InterpTables tb;
    int rc = interp_tables(ctx, (int)curr->width, (int)curr->height, &tb);
    if (rc != LFG_OK) return rc;
    StageTimer timer(ctx, LFG_STAGE_INTERPOLATE);
    hipError_t e = lfg::launch_interpolate(ctx->stream, *prev, *curr, *mv, *out, factor, ctx->semantics != 0, tb);
    if (e != hipSuccess) return fail_hip(ctx, e, "interpolate kernel launch");
    return LFG_OK;
}

LFG_EXPORT int lfg_interpolate_frames(lfg_context *ctx, const lfg_frame *prev, const lfg_frame *curr,
                                      lfg_frame *out, float factor) {
    if (!ctx || !curr) return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_frames: NULL argument");
    lfg_frame &mv = ctx->mv_tmp;
    if (!mv.data || mv.width != curr->width || mv.height != curr->height) {   // frame_manager.cpp:226-230
        lfg_frame_destroy(ctx, &mv);
        int rc = lfg_frame_create(ctx, curr->width, curr->height, LFG_FORMAT_MV_S8X2, &mv);
        if (rc != LFG_OK) return fail(ctx, rc, "Failed to create motion vectors frame");
    }
    if (ctx->fuse_motion_interpolate) {
        // The north-star order: the motion kernels write the generated frame from the vectors while they
        // hold them; the vector frame -- this call's temporary -- is not written at all.  `out` is checked as lfg_interpolate
        // checks it; whatever the fused path does not cover (see motion_run) takes the two stages below.
        if (!frame_ok(prev, LFG_FORMAT_RGBA8_UNORM) || !frame_ok(out, LFG_FORMAT_RGBA8_UNORM) || !same_size(curr, out) || out->pitch % 4u)
            return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_frames: bad output frame (NULL, empty, wrong format, size or pitch)");
        if (out->data == prev->data || out->data == curr->data)
            return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_frames: the output aliases an input");
        lfg::FusedOut fo;
        fo.data = (uint8_t *)out->data; fo.pitch = (int)out->pitch; fo.t = factor; fo.intended = ctx->semantics != 0 ? 1 : 0; fo.storeMv = 0;
        bool done = false;
        int rc = motion_run(ctx, prev, curr, &mv, 8, 16.0f, &fo, &done);
        if (rc != LFG_OK || done) return rc;
        return lfg_interpolate(ctx, prev, curr, &mv, out, factor);     // (the generic kernel ran: it wrote the vectors)
    }
    int rc = lfg_motion(ctx, prev, curr, &mv, 8, 16.0f);                      // frame_manager.cpp:332-333
    if (rc != LFG_OK) return rc;
    return lfg_interpolate(ctx, prev, curr, &mv, out, factor);
}

LFG_EXPORT int lfg_interpolate_multi(lfg_context *ctx, const lfg_frame *prev, const lfg_frame *curr, const lfg_frame *mv,
                                     lfg_frame *const *outs, const float *factors, uint32_t count) {
    if (!ctx) return LFG_ERR_INVALID;
    LFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!outs || !factors || count == 0 || count > LFG_MAX_FACTORS)
        return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_multi: count must be in [1, LFG_MAX_FACTORS] and outs/factors non-NULL");
    if (!frame_ok(prev, LFG_FORMAT_RGBA8_UNORM) || !frame_ok(curr, LFG_FORMAT_RGBA8_UNORM) || !frame_ok(mv, LFG_FORMAT_MV_S8X2))
        return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_multi: bad frame (NULL, empty or wrong format)");
    if (!same_size(prev, curr) || !same_size(curr, mv))
        return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_multi: frames differ in size");
    if ((prev->pitch | curr->pitch) % 4u || mv->pitch % 2u)
        return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_multi: row pitch not a multiple of the pixel size");
    for (uint32_t i = 0; i < count; ++i) {
        const lfg_frame *o = outs[i];
        if (!frame_ok(o, LFG_FORMAT_RGBA8_UNORM) || !same_size(curr, o) || o->pitch % 4u)
            return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_multi: bad output frame (NULL, empty, wrong format, size or pitch)");
        if (o->data == prev->data || o->data == curr->data)
            return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_multi: an output aliases an input");
        for (uint32_t j = 0; j < i; ++j)
            if (outs[j]->data == o->data) return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_multi: two outputs alias each other");
    }
    lfg::InterpTables tb;
    int rc = interp_tables(ctx, (int)curr->width, (int)curr->height, &tb);
    if (rc != LFG_OK) return rc;
    StageTimer timer(ctx, LFG_STAGE_INTERPOLATE);
    hipError_t e = lfg::launch_interpolate_multi(ctx->stream, *prev, *curr, *mv, outs, factors, (int)count, ctx->semantics != 0, tb);
    if (e != hipSuccess) return fail_hip(ctx, e, "interpolate kernel launch");
    return LFG_OK;
}

LFG_EXPORT int lfg_interpolate_frames_multi(lfg_context *ctx, const lfg_frame *prev, const lfg_frame *curr,
                                            lfg_frame *const *outs, const float *factors, uint32_t count) {
    if (!ctx || !curr) return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_frames_multi: NULL argument");
    lfg_frame &mv = ctx->mv_tmp;
    if (!mv.data || mv.width != curr->width || mv.height != curr->height) {   // frame_manager.cpp:226-230
        lfg_frame_destroy(ctx, &mv);
        int rc = lfg_frame_create(ctx, curr->width, curr->height, LFG_FORMAT_MV_S8X2, &mv);
        if (rc != LFG_OK) return fail(ctx, rc, "Failed to create motion vectors frame");
    }
    int rc = lfg_motion(ctx, prev, curr, &mv, 8, 16.0f);                      // frame_manager.cpp:332-333
    if (rc != LFG_OK) return rc;
    return lfg_interpolate_multi(ctx, prev, curr, &mv, outs, factors, count);
}

LFG_EXPORT int lfg_set_fused_motion_interpolate(lfg_context *ctx, int enabled) {
    if (!ctx) return LFG_ERR_INVALID;
    ctx->fuse_motion_interpolate = enabled != 0;
    return LFG_OK;
}

LFG_EXPORT int lfg_set_fused_interpolate_scale(lfg_context *ctx, int enabled) {
    if (!ctx) return LFG_ERR_INVALID;
    ctx->fuse_interpolate_scale = enabled != 0;
    return LFG_OK;
}

LFG_EXPORT int lfg_interpolate_scale(lfg_context *ctx, const lfg_frame *prev, const lfg_frame *curr, const lfg_frame *mv,
                                     lfg_frame *out, float factor) {
    if (!ctx) return LFG_ERR_INVALID;
    LFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!frame_ok(prev, LFG_FORMAT_RGBA8_UNORM) || !frame_ok(curr, LFG_FORMAT_RGBA8_UNORM) ||
        !frame_ok(mv, LFG_FORMAT_MV_S8X2) || !frame_ok(out, LFG_FORMAT_RGBA8_UNORM))
        return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_scale: bad frame (NULL, empty or wrong format)");
    if (!same_size(prev, curr) || !same_size(curr, mv))
        return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_scale: prev, curr and mv differ in size");
    if ((prev->pitch | curr->pitch) % 4u || mv->pitch % 2u)
        return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_scale: row pitch not a multiple of the pixel size");
    if (out->data == prev->data || out->data == curr->data)
        return fail(ctx, LFG_ERR_INVALID, "lfg_interpolate_scale: output aliases an input");
    trim_axis_tables(ctx);
    lfg::AxisTable *tx = nullptr, *ty = nullptr;
    int rc = build_axis_table(ctx, (int)curr->width, (int)out->width, &tx);
    if (rc != LFG_OK) return rc;
    rc = build_axis_table(ctx, (int)curr->height, (int)out->height, &ty);
    if (rc != LFG_OK) return rc;
    const bool fused = ctx->fuse_interpolate_scale && tx->pattern_2x && ty->pattern_2x && tx->palette_rows > 0 && ty->strips_per_xcd > 0 && lfg::scale_2x_supported(*curr, *out) &&
                       (uint64_t)prev->height * prev->pitch < 0x7fffffffull;
    if (fused) {
        // one kernel: the generated frame is interpolated row by row inside the 2x scale kernel and never stored at
        // input resolution; accounted to the scale stage
        StageTimer timer(ctx, LFG_STAGE_SCALE);
        hipError_t e = lfg::launch_interpolate_scale_2x(ctx->stream, *prev, *curr, *mv, *out, *tx, *ty, factor, ctx->semantics != 0);
        if (e != hipSuccess) return fail_hip(ctx, e, "fused interpolate + scale kernel launch");
        return LFG_OK;
    }
    // any other size ratio: the two stages, through a context-owned frame at input resolution
    lfg_frame &mid = ctx->mid_tmp;
    if (!mid.data || mid.width != curr->width || mid.height != curr->height) {
        lfg_frame_destroy(ctx, &mid);
        rc = lfg_frame_create(ctx, curr->width, curr->height, LFG_FORMAT_RGBA8_UNORM, &mid);
        if (rc != LFG_OK) return rc;
    }
    rc = lfg_interpolate(ctx, prev, curr, mv, &mid, factor);
    if (rc != LFG_OK) return rc;
    return lfg_scale(ctx, &mid, out);
}

LFG_EXPORT int lfg_mv_export_rgba32f(lfg_context *ctx, const lfg_frame *mv, void *device_rgba32f) {
    if (!ctx || !frame_ok(mv, LFG_FORMAT_MV_S8X2) || !device_rgba32f || (uintptr_t)device_rgba32f % 16u)
        return fail(ctx, LFG_ERR_INVALID, "lfg_mv_export_rgba32f: bad argument");
    hipError_t e = lfg::launch_mv_export(ctx->stream, *mv, (float *)device_rgba32f);
    if (e != hipSuccess) return fail_hip(ctx, e, "mv export kernel launch");
    return LFG_OK;
}

// ================================================================== diagnostics

LFG_EXPORT int lfg_selftest_sqrt(lfg_context *ctx, uint32_t lo_bits, uint32_t hi_bits, uint64_t *out_mismatches) {
    if (!ctx || !out_mismatches || lo_bits > hi_bits) return fail(ctx, LFG_ERR_INVALID, "lfg_selftest_sqrt: bad argument");
    unsigned long long *d = nullptr;
    LFG_HIP(ctx, hipMalloc((void **)&d, sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) e = lfg::launch_sqrt_selftest(ctx->stream, lo_bits, hi_bits, d);
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail_hip(ctx, e, "sqrt selftest");
    *out_mismatches = (uint64_t)h;
    return LFG_OK;
}

LFG_EXPORT int lfg_diag_scale_2x_strip(uint32_t in_height, uint32_t xcd, uint32_t index, uint32_t *out_strips_per_xcd,
                                       int32_t *out_first_step, int32_t *out_steps) {
    if (in_height == 0 || in_height > 32768u || xcd >= 8u) return LFG_ERR_INVALID;
    const int per = lfg::scale_2x_strips_per_xcd((int)in_height);
    if (out_strips_per_xcd) *out_strips_per_xcd = (uint32_t)per;
    if ((int)index >= per) return LFG_ERR_INVALID;
    int first = 0, steps = 0;
    lfg::scale_2x_strip_host((int)in_height, (int)xcd, (int)index, first, steps);
    if (out_first_step) *out_first_step = first;
    if (out_steps) *out_steps = steps;
    return LFG_OK;
}

// ================================================================== measurement

LFG_EXPORT int lfg_profile_enable(lfg_context *ctx, int enabled) {
    if (!ctx) return LFG_ERR_INVALID;
    int rc = drain_profile(ctx);
    ctx->profile = enabled != 0;
    return rc;
}

LFG_EXPORT int lfg_profile_reset(lfg_context *ctx) {
    if (!ctx) return LFG_ERR_INVALID;
    int rc = drain_profile(ctx);
    for (int i = 0; i < LFG_STAGE_COUNT; ++i) { ctx->prof_ms[i] = 0.0; ctx->prof_n[i] = 0; }
    return rc;
}

LFG_EXPORT int lfg_profile_get(lfg_context *ctx, int stage, double *out_total_ms, uint64_t *out_launches) {
    if (!ctx || stage < 0 || stage >= LFG_STAGE_COUNT) return LFG_ERR_INVALID;
    int rc = drain_profile(ctx);
    if (out_total_ms) *out_total_ms = ctx->prof_ms[stage];
    if (out_launches) *out_launches = ctx->prof_n[stage];
    return rc;
}
