/*
 * linuxfg_hip.h -- C-ABI of the MI355X-native linux-fg hot path
 *                  (upscale -> motion -> interpolate).
 *
 * This is the drop-in boundary.  The reference (xXJSONDeruloXx/linux-fg) has no plugin
 * or FFI layer: its boundary is the C++ class surface VulkanContext / FrameManager /
 * Scaler.  Everything those classes do on the hot path through
 * Vulkan is provided here through plain C: opaque context, POD frame descriptor, plain
 * pointers and sizes, int return codes.  The C++ mirror of the reference classes in
 * linux-fg_amd/host/ and the ctypes binding in linux-fg_amd/capi.py call nothing else.
 *
 * Conventions
 *   - every fallible call returns 0 on success, a negative lfg_status otherwise, and
 *     latches a message readable with lfg_last_error() (the reference returns bool and
 *     latches Logger::GetLastError, src/logger.hpp:33-41); nothing throws across this ABI;
 *   - a context is bound to ONE GPU and ONE HIP stream and is not thread-safe (the
 *     reference is single-threaded, one VkQueue: src/vulkan_context.cpp:130-151);
 *   - compute calls ENQUEUE on the context's stream and return; lfg_sync() waits
 *     (the reference waits with vkQueueWaitIdle after every submit, src/scaler.cpp:393);
 *   - frames are row-major, `pitch` bytes per row, RGBA8 = 4 bytes per pixel in memory
 *     order R,G,B,A (channel order is irrelevant to the kernels: all four are treated alike);
 *   - there is no CPU fallback: without a usable GPU lfg_context_create() fails.
 *
 * All citations are file:line under the reference checkout.
 */
#ifndef LINUXFG_HIP_H
#define LINUXFG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFG_ABI_VERSION 1

typedef enum lfg_status {
    LFG_OK = 0,
    LFG_ERR_INVALID = -1,     /* bad argument (null, size mismatch, unsupported parameter) */
    LFG_ERR_DEVICE = -2,      /* HIP runtime error; text in lfg_last_error() */
    LFG_ERR_NOMEM = -3,
    LFG_ERR_UNSUPPORTED = -4
} lfg_status;

typedef enum lfg_format {
    /* VK_FORMAT_R8G8B8A8_UNORM, the only colour format the reference creates
     * (src/frame_manager.hpp:15). 4 bytes per pixel. */
    LFG_FORMAT_RGBA8_UNORM = 0,
    /* Motion vectors, two signed bytes per pixel (x, y), whole pixels in [-127,127].
     * The reference declares the MV image rgba32f (shaders/motion.comp:7) and stores
     * vec4(best, 0, 1) (:56); best is always integer-valued for the integer search radii
     * the host pushes (src/frame_manager.cpp:333), so two int8 hold it losslessly at 1/8
     * of the bytes.  lfg_mv_export_rgba32f() reproduces the reference's image verbatim. */
    LFG_FORMAT_MV_S8X2 = 1
} lfg_format;

/* Replaces `struct Frame` {VkImage, VkDeviceMemory, VkImageView, width, height, format}
 * (src/frame_manager.hpp:9-16): a device pointer plus geometry.  Caller-owned POD. */
typedef struct lfg_frame {
    void    *data;      /* device memory; NULL = empty frame (VK_NULL_HANDLE) */
    uint32_t width;
    uint32_t height;
    uint32_t pitch;     /* bytes per row, >= width * bytes-per-pixel */
    uint32_t format;    /* lfg_format */
    uint32_t owned;     /* 1: allocated by lfg_frame_create and freed by lfg_frame_destroy */
    uint32_t reserved;
} lfg_frame;

typedef struct lfg_context lfg_context;
typedef struct lfg_ring lfg_ring;

/* Kernel stages, for lfg_profile_get(). */
typedef enum lfg_stage {
    LFG_STAGE_SCALE = 0,
    LFG_STAGE_MOTION = 1,
    LFG_STAGE_INTERPOLATE = 2,
    LFG_STAGE_COUNT = 3
} lfg_stage;

/* ---------------------------------------------------------------- library / context */

int         lfg_abi_version(void);
/* Number of HIP devices visible to this process (0 if none); does not create a context. */
int         lfg_device_count(void);

/* Replaces VulkanContext::Initialize (src/vulkan_context.cpp:3-23: instance, physical
 * device pick :88-105, logical device + one compute queue :117-151).  device_ordinal < 0
 * picks device 0 (the reference prefers a discrete GPU, else index 0).  The context owns
 * one HIP stream (the compute queue) unless lfg_context_set_stream() adopts another. */
int         lfg_context_create(int device_ordinal, lfg_context **out_ctx);
/* Replaces VulkanContext::Cleanup / FrameManager::Cleanup / Scaler::Cleanup.  NULL is a no-op. */
void        lfg_context_destroy(lfg_context *ctx);
/* Adopt an externally owned hipStream_t (e.g. PyTorch's current stream) as the compute
 * queue, so the caller's events and collectives order against these kernels.  NULL restores
 * the context's own stream. */
int         lfg_context_set_stream(lfg_context *ctx, void *hip_stream);
void       *lfg_context_get_stream(lfg_context *ctx);
int         lfg_context_device(const lfg_context *ctx);
/* vkQueueWaitIdle (src/scaler.cpp:393, src/frame_manager.cpp:194): every lane of the context. */
int         lfg_sync(lfg_context *ctx);

/* Lanes: several frames in flight on one GPU.  The reference has one queue and waits for it after every
 * submission (src/scaler.cpp:389-393, src/frame_manager.cpp:190-194); here a frame's last long motion units leave
 * most CUs idle for a third of its time, and the next frame's scale, hints and first units can run there.  A lane
 * is a stream plus the temporaries and the motion workspace (0.9 GB at 4K, 1.0 GB for the one lane of a context without lanes: lfg_motion_workspace_size) of the calls made while it is selected;
 * frames are plain device memory and may be used from any lane -- the caller orders producers and consumers:
 *   lfg_lanes(ctx, n)       1 <= n <= LFG_MAX_LANES lanes (lane 0 is the context's own stream); shrinking waits
 *                           for the lanes that go and frees what they own
 *   lfg_lane_select(ctx, j) the calls that follow enqueue on lane j
 *   lfg_lane_mark(ctx)      remember "here" on the selected lane
 *   lfg_lane_wait(ctx, i)   the selected lane's later work waits until lane i has reached its last mark
 *   lfg_lane_sync(ctx)      the HOST waits for the selected lane alone (a frame's buffers are free again; lfg_sync waits for
 *                           every lane) -- what vkQueueWaitIdle after each submission (src/scaler.cpp:389-393) becomes when
 *                           n frames are in flight: wait for frame k before frame k + n goes onto its lane
 * e.g. frame k on lane k % 2:  select; scale(curr_k); mark; wait(other lane: scale(curr_k-1)); motion; interpolate.
 * lfg_context_set_stream applies to the selected lane. */
#define LFG_MAX_LANES 4
int         lfg_lanes(lfg_context *ctx, int count);
int         lfg_lane_count(const lfg_context *ctx);
int         lfg_lane_current(const lfg_context *ctx);
int         lfg_lane_select(lfg_context *ctx, int lane);
int         lfg_lane_mark(lfg_context *ctx);
int         lfg_lane_wait(lfg_context *ctx, int other);
int         lfg_lane_sync(lfg_context *ctx);
/* Logger::GetLastError (src/logger.hpp:38).  ctx == NULL reads the creation-time error. */
const char *lfg_last_error(const lfg_context *ctx);

/* ---------------------------------------------------------------- frames */

/* FrameManager::CreateFrame (src/frame_manager.cpp:30-69): device-local storage for a
 * width x height image; pitch is tight (width * bpp rounded up to 16 bytes is NOT applied:
 * rows are tightly packed so a frame is one contiguous upload). */
int  lfg_frame_create(lfg_context *ctx, uint32_t width, uint32_t height, uint32_t format, lfg_frame *out);
/* FrameManager::DestroyFrame (src/frame_manager.cpp:71-81): idempotent, NULL-safe. */
void lfg_frame_destroy(lfg_context *ctx, lfg_frame *frame);
/* Describe caller-owned device memory (e.g. a torch tensor) as a frame; never freed here. */
int  lfg_frame_wrap(void *device_ptr, uint32_t width, uint32_t height, uint32_t pitch, uint32_t format,
                    lfg_frame *out);
/* FrameManager::CopyFrameData (src/frame_manager.cpp:83-145): same-size check, then a
 * device-to-device copy on the stream. */
int  lfg_frame_copy(lfg_context *ctx, const lfg_frame *src, lfg_frame *dst);

/* FrameManager::CreateStagingBuffer / DestroyStagingBuffer (src/frame_manager.cpp:199-214):
 * host-visible memory the device can DMA from; here pinned host memory. */
int  lfg_staging_create(lfg_context *ctx, size_t bytes, void **out_host_ptr);
void lfg_staging_destroy(lfg_context *ctx, void *host_ptr);

/* Upload = WindowCapture::CopyToStagingBuffer (src/window_capture.cpp:472-568): `bytes` must be
 * >= width*height*bpp (the reference's size check, :478-481); rows tightly packed.  Download =
 * the readback in Scaler::ProcessFrame (src/scaler.cpp:479-536).  Both are asynchronous on the
 * stream when `host` is pinned (lfg_staging_create / ring memory); call lfg_sync() before
 * touching the host bytes. */
int  lfg_frame_upload(lfg_context *ctx, lfg_frame *dst, const void *host, size_t bytes);
int  lfg_frame_download(lfg_context *ctx, const lfg_frame *src, void *host, size_t bytes);

/* Pinned-host frame ring: replaces the per-frame staging alloc/free of the reference
 * (src/window_capture.cpp:474-487,564; src/scaler.cpp:480-487,614).  `slots` buffers of
 * `slot_bytes` each; acquire blocks until the slot's last transfer has completed. */
int  lfg_ring_create(lfg_context *ctx, uint32_t slots, size_t slot_bytes, lfg_ring **out_ring);
void lfg_ring_destroy(lfg_ring *ring);
int  lfg_ring_acquire(lfg_ring *ring, void **out_host_ptr, uint32_t *out_slot);
/* Upload slot -> frame (or download frame -> slot) asynchronously and mark the slot busy
 * until that transfer finishes.  Transfers run on the ring's own stream, next to the kernels:
 * an upload is ordered after everything enqueued so far and before everything enqueued later;
 * a download is ordered after everything enqueued so far only -- later kernels do not wait for it. */
int  lfg_ring_upload(lfg_ring *ring, uint32_t slot, lfg_frame *dst);
int  lfg_ring_download(lfg_ring *ring, uint32_t slot, const lfg_frame *src);
/* Block the host until the slot's last transfer has finished (the vkQueueWaitIdle before the
 * reference maps its staging buffer, src/scaler.cpp:532-536), then the slot's pixels may be read. */
int  lfg_ring_wait(lfg_ring *ring, uint32_t slot);
/* Make kernels enqueued from now on wait (on the device, not the host) for the slot's last transfer:
 * call it before overwriting a frame whose download into `slot` may still be running. */
int  lfg_ring_fence_slot(lfg_ring *ring, uint32_t slot);

/* ---------------------------------------------------------------- the three stages */

/* shaders/scale.comp as dispatched by Scaler::ScaleFrame (src/scaler.cpp:260-395):
 * Lanczos-3 resample of `in` to `out`'s size; push constants inputSize/outputSize are taken
 * from the frames (src/scaler.cpp:348-351).  Any sizes; out == 2 x in takes the LDS-tiled path. */
int  lfg_scale(lfg_context *ctx, const lfg_frame *in, lfg_frame *out);

/* shaders/motion.comp as dispatched by FrameManager::InterpolateFrames
 * (src/frame_manager.cpp:325-344): per-pixel full-search block match of `curr` against `prev`.
 * `mv` is an LFG_FORMAT_MV_S8X2 frame of the same size.  The reference pushes blockSize = 8,
 * searchRadius = 16.0f (:332-333); other values are accepted when searchRadius is a whole
 * number in [0,127] and 1 <= blockSize <= 64. */
int  lfg_motion(lfg_context *ctx, const lfg_frame *prev, const lfg_frame *curr, lfg_frame *mv,
                int block_size, float search_radius);

/* How lfg_motion evaluates blockSize 8 / searchRadius 16.  Both modes return bit-identical motion
 * vectors for every input; they differ in run time only.
 *   LFG_MOTION_PREFILTERED (default): a cheap value that provably brackets every candidate's cost (integer
 *       squared distances, pairwise-tree sums; within 3.6e-5 relative of the shader's sequential fp32 sum
 *       for any data) selects the few candidates that can still be the minimum; a single survivor is the
 *       answer, several get the literal 64-term chain.  Tiles where many candidates tie at one non-zero cost
 *       fall back to the exact kernel, so such content costs what LFG_MOTION_EXACT_ONLY costs (plus the filter).
 *   LFG_MOTION_EXACT_ONLY: the literal chain for every (pixel, candidate); content-independent time. */
typedef enum lfg_motion_mode { LFG_MOTION_PREFILTERED = 0, LFG_MOTION_EXACT_ONLY = 1 } lfg_motion_mode;
int  lfg_set_motion_mode(lfg_context *ctx, int mode);
/* After a prefiltered lfg_motion: number of 64x64 tiles, how many of them fell back to the exact kernel,
 * and the mean number of candidates recorded per pixel in the others (synchronises; reporting only). */
int  lfg_motion_last_stats(lfg_context *ctx, uint32_t *out_tiles, uint32_t *out_fallback_tiles,
                           double *out_mean_recorded);
/* After a prefiltered lfg_motion: how many 16-row segments of its 56 x 64 work tiles the prefilter left to the resolve
 * kernel, and how many the frame has; every open segment is listed exactly once (synchronises; reporting and tests only).
 * No reference counterpart: motion.comp (shaders/motion.comp:27-52) has one pass and no work lists. */
int  lfg_motion_open_segments(lfg_context *ctx, uint32_t *out_open, uint32_t *out_segments);
/* With frames in flight (lfg_lanes >= 2) a call whose content suits it sends its whole interior tiles through a lean kernel
 * before the general one (same vectors; the choice goes by the lane's previous call).  After an lfg_motion: whether the selected
 * lane's last call did (1/0), how many tiles are listed for that kernel at this frame size, and in how many it left work to the
 * general kernel (synchronises; reporting and tests only).  No reference counterpart (shaders/motion.comp:27-52 is one pass). */
int  lfg_motion_lean_stats(lfg_context *ctx, int *out_used, uint32_t *out_tiles, uint32_t *out_tiles_left);
/* The strips a frame's motion exposes along its edges -- sixteen pixel columns, eight rows: pixels without any match -- are
 * searched by a kernel of their own before the general one (same vectors).  After an lfg_motion: how many pixel rows had their
 * left or right band decided that way and how many pixel columns their top or bottom band (synchronises; reporting and tests
 * only).  No reference counterpart (shaders/motion.comp:27-52 treats every pixel alike). */
int  lfg_motion_strip_stats(lfg_context *ctx, uint32_t *out_rows, uint32_t *out_columns);
/* The persistent motion kernel exists in two variants with identical results: the default, and one whose lattice walks go by sums
 * of absolute differences first where a match costs a few hundred to a thousand -- sensor noise of +-3 levels and more at a
 * 1080p input (4K, three frames in flight: +15 % at +-3, +27 % at +-4 .. +-12; -3 % at +-2 and below, which is why it is not the
 * only one).  With frames in flight the lane's previous call decides (half its sample blocks matched with a SAD of 310 - 2,200);
 * LFG_TIER_FORCE=0|1 at context creation overrides.  Returns the variant the context's last lfg_motion launched (0 / 1). */
int  lfg_motion_last_variant(const lfg_context *ctx);
/* With frames in flight four launch decisions of an lfg_motion go by what the lane's PREVIOUS finished call found (its
 * verdict word, stored into pinned host memory by that call's last launch): the lean kernel and the plan that goes with it,
 * the size of the persistent grid and the variant of its kernel (counted together), the size of the second pass.  A wrong guess changes no result, only the call's duration.
 * Counters since the context was created, over all lanes: calls whose own verdict has been read back, and how many of them
 * had been launched on a guess that this verdict contradicts, per decision (does not synchronise; reporting only).
 * No reference counterpart: one queue, one dispatch per stage (src/frame_manager.cpp:342-366). */
int  lfg_motion_prediction_stats(const lfg_context *ctx, uint64_t *out_verdicts, uint64_t *out_lean_wrong,
                                 uint64_t *out_grid_wrong, uint64_t *out_second_pass_wrong);
/* Bytes of device memory the prefiltered lfg_motion keeps for frames of this size (allocated on the first such call, kept
 * until the size changes or the context goes; one per lane).  No reference counterpart -- the reference's motion pass keeps
 * nothing between its two images (src/frame_manager.cpp:262-300); a host budgets lanes with it.  Needs no GPU work.
 * The figure is for the context's CURRENT lane count (the work-unit plan of a context with frames in flight differs from
 * that of a context without, see lfg_motion_plan): call it after lfg_lanes(). */
int  lfg_motion_workspace_size(lfg_context *ctx, uint32_t width, uint32_t height, uint64_t *out_bytes);
/* The work-unit plan the prefiltered lfg_motion uses on this context: *out_rim_split = parts of the candidate order a
 * segment on the image's rim is searched in -- 4 with frames in flight (lfg_lanes >= 2: the sum of the units' times counts),
 * 48 = four, and eight for the segments at the top and bottom border, when one frame runs at a time (the longest unit counts);
 * LFG_MOTION_RIM_SPLIT=4|8|48 in the environment at context creation overrides -- and *out_workgroups = the persistent
 * workgroups the device holds for its prefilter launch (0 before the first call; with frames in flight a call launches 5/8 of
 * them when the lane's previous call found most of its sample blocks matched and another lane is busy: room for their kernels).  The plan changes only inside lfg_lanes(); the first lfg_motion
 * after such a change re-plans (it waits for the lane's stream once).  Reporting only; either pointer may be NULL. */
int  lfg_motion_plan(const lfg_context *ctx, int *out_rim_split, int *out_workgroups);

/* Which arithmetic lfg_motion and lfg_interpolate follow.  No reference counterpart: SURVEY.md 8(f) rank 4.
 *   LFG_SEMANTICS_REFERENCE (default, the parity contract): the shaders as written -- equal block-match costs
 *       resolve to the first candidate in scan order, so flat areas report (-16,-16) (shaders/motion.comp:27-28,49;
 *       F6), and the pixel-unit motion vector is added to normalised uv unscaled (shaders/interpolate.comp:17,34-35; F5).
 *   LFG_SEMANTICS_INTENDED (opt-in): equal costs resolve to the shortest vector (then scan order), so flat areas
 *       report (0,0); interpolate divides the vector by the image size before adding it to uv, so it displaces
 *       by pixels.  Everything else (costs, sampling, signs, rounding) is unchanged; the oracle has the same switch. */
typedef enum lfg_semantics { LFG_SEMANTICS_REFERENCE = 0, LFG_SEMANTICS_INTENDED = 1 } lfg_semantics;
int  lfg_set_semantics(lfg_context *ctx, int semantics);

/* shaders/interpolate.comp (src/frame_manager.cpp:351-366): MV-displaced bilinear fetch of prev
 * and curr, blended by `factor`.  Literal reference semantics (MV in pixels added to normalised
 * UV). */
int  lfg_interpolate(lfg_context *ctx, const lfg_frame *prev, const lfg_frame *curr, const lfg_frame *mv,
                     lfg_frame *out, float factor);

/* FrameManager::InterpolateFrames(previous, current, output, factor)
 * (src/frame_manager.cpp:216-372): motion (blockSize 8, searchRadius 16) then interpolate, with
 * the motion-vector image as a context-owned temporary (the reference creates and destroys it
 * per call, :226-230,369; here it persists between calls and is invisible to the caller). */
int  lfg_interpolate_frames(lfg_context *ctx, const lfg_frame *prev, const lfg_frame *curr,
                            lfg_frame *out, float factor);

/* Several generated frames per pair -- the 60 -> 240 fps cadence of BASELINE config 5 (t = 1/4, 1/2, 3/4) -- from
 * ONE pass over prev, curr and the motion vectors (the vectors and both sources are read once
 * per pair instead of once per factor).  No reference counterpart beyond the single `factor` of
 * FrameManager::InterpolateFrames (src/frame_manager.cpp:216) and ScalerConfig::interpolationFactor
 * (src/scaler.hpp:17).  outs[i] receives the frame for factors[i]; every frame is identical, byte for byte, to
 * lfg_interpolate(ctx, prev, curr, mv, outs[i], factors[i]).  1 <= count <= LFG_MAX_FACTORS; the outputs must
 * not alias each other or an input. */
#define LFG_MAX_FACTORS 16
int  lfg_interpolate_multi(lfg_context *ctx, const lfg_frame *prev, const lfg_frame *curr, const lfg_frame *mv,
                           lfg_frame *const *outs, const float *factors, uint32_t count);
/* lfg_interpolate_frames in the north-star ORDER (the reference dispatches motion and then
 * interpolate on the same grid, src/frame_manager.cpp:342-366, the second reading what the first wrote): with 1, the
 * motion kernels (blockSize 8, searchRadius 16 paths) write the generated frame from each vector at the moment it is
 * decided -- the same per-pixel function the interpolate kernel is made of, so the bytes are identical -- the interpolate
 * dispatch is gone and the motion-vector temporary is never written.  Default 0: the two stages (measured: DESIGN.md
 * section 4.4).  Also set by LFG_FUSED_MOTION_INTERPOLATE=1 in the environment at context creation.  lfg_interpolate_frames_multi
 * always runs the two stages. */
int  lfg_set_fused_motion_interpolate(lfg_context *ctx, int enabled);

/* lfg_interpolate_frames for several factors: motion (blockSize 8, searchRadius 16) ONCE, then
 * lfg_interpolate_multi with the context-owned motion-vector temporary. */
int  lfg_interpolate_frames_multi(lfg_context *ctx, const lfg_frame *prev, const lfg_frame *curr,
                                  lfg_frame *const *outs, const float *factors, uint32_t count);

/* The reference's own data flow keeps prev / curr at INPUT resolution (src/scaler.cpp:443,451): there the generated
 * frame is interpolated at input resolution and then upscaled like a captured one.  This does both in one call --
 * identical, byte for byte, to lfg_interpolate into a temporary followed by lfg_scale of that temporary -- and where
 * `out` is exactly twice the size of the inputs optionally in ONE kernel (lfg_set_fused_interpolate_scale): each input
 * row of the 2x scale kernel is then interpolated on the fly (shaders/interpolate.comp:15-40 arithmetic, rounded to
 * bytes as the stage would store it), so the generated frame never exists at input resolution in memory (SURVEY.md
 * 8(f) rank 1).  prev, curr, mv: same size; out: any size. */
int  lfg_interpolate_scale(lfg_context *ctx, const lfg_frame *prev, const lfg_frame *curr, const lfg_frame *mv,
                           lfg_frame *out, float factor);
/* Which of the two lfg_interpolate_scale uses at exactly 2x.  Default 0: the two stages through a context-owned frame
 * -- measured FASTER on MI355X at 1080p -> 4K (interpolate 5 us + scale 13 us against 33 us for the fused kernel: the
 * 16 MB the intermediate frame moves cost less than re-interpolating the five warm-up rows of every strip inside a
 * kernel that is bound by instruction issue, not by memory).  1: the fused kernel.  Results
 * are identical.  Also set by LFG_FUSED_INTERPOLATE_SCALE=1 in the environment at context creation. */
int  lfg_set_fused_interpolate_scale(lfg_context *ctx, int enabled);

/* Write the motion vectors as the reference's rgba32f image: vec4(mv.x, mv.y, 0, 1) per pixel
 * (shaders/motion.comp:56) into `device_rgba32f` (width*height*16 bytes, device memory). */
int  lfg_mv_export_rgba32f(lfg_context *ctx, const lfg_frame *mv, void *device_rgba32f);

/* ---------------------------------------------------------------- multi-GPU: the one exchange of the path */

/* Frame pairs are independent, so a batch shards one pair per GPU with no communication -- except that the batch can
 * share its previous frame, which then travels from the rank that owns it to every other rank as ONE broadcast per
 * frame (RCCL ncclBroadcast of width*height*bpp bytes as ncclUint8, over xGMI).  The
 * reference has a single queue and no communication (src/vulkan_context.cpp:130-151): no counterpart.
 *
 * One process (or thread) per GPU, one context each.  Rank 0 makes the id and hands its bytes to the other ranks by
 * any means it likes (a file, a socket, MPI, a torch.distributed store); then EVERY rank calls lfg_comm_init with the
 * same id -- a collective call that returns once all ranks have arrived.  librccl.so is opened the first time one of
 * these calls is made; without it they return LFG_ERR_UNSUPPORTED. */
#define LFG_COMM_ID_BYTES 128
typedef struct lfg_comm_id { char bytes[LFG_COMM_ID_BYTES]; } lfg_comm_id;
int  lfg_comm_unique_id(lfg_comm_id *out_id);
int  lfg_comm_init(lfg_context *ctx, int nranks, int rank, const lfg_comm_id *id);
int  lfg_comm_rank(const lfg_context *ctx);       /* -1 without a communicator */
int  lfg_comm_ranks(const lfg_context *ctx);      /*  0 without a communicator */
/* Broadcast a tightly packed frame (every rank passes its own frame of the same size and format) from `root`,
 * asynchronously on the context's communication stream: it starts once everything enqueued so far on EVERY lane of
 * the context has finished (the kernels still reading the frame on a receiver, the kernels producing it on the root,
 * whichever lane they were given to) and runs next to whatever is enqueued afterwards.  Nothing enqueued later, on any
 * lane, may touch the frame before that lane has passed an lfg_comm_wait() (or waits, lfg_lane_wait, for a lane that has). */
int  lfg_broadcast_frame(lfg_context *ctx, lfg_frame *frame, int root);
/* The same, ordered behind what the SELECTED lane has been given so far and behind nothing else.  For a caller that has
 * already ordered the selected lane behind the frame's last readers (on a receiver) or producers (on the root), e.g. with
 * lfg_lane_wait for the lane that marked (lfg_lane_mark) after reading it -- which is what a step with frames in flight
 * does anyway (linux-fg_amd/sharding.py, bench.py: step k waits for step k - 1's upscales, the last readers of the slot
 * that step k + 1's frame lands in).  With three frames in flight the call above makes the broadcast for step k + 1 wait
 * for ALL of step k - 1 and step k + 1 for the broadcast: two frames in flight and a bubble (measured on one MI355X
 * with a 170 us stand-in for the broadcast, NOTES_r05.md section 7).  Misuse corrupts frames; when in doubt use the call above. */
int  lfg_broadcast_frame_lane(lfg_context *ctx, lfg_frame *frame, int root);
/* Make everything enqueued on the SELECTED lane from now on wait (on the device) for ALL broadcasts issued so far.
 * Contract: the communication stream is in order and one event, re-recorded behind each broadcast, stands for every
 * broadcast before it -- so with two broadcasts in flight this waits for both, never for the older one alone (a caller
 * that double-buffers, like linux-fg_amd/sharding.py, waits for slot k before it issues k + 1 and loses nothing).  Other
 * lanes are not gated: they call lfg_comm_wait themselves or order behind this lane with lfg_lane_mark / lfg_lane_wait. */
int  lfg_comm_wait(lfg_context *ctx);
/* The HOST waits until every broadcast issued so far has completed (the counterpart of lfg_lane_sync). */
int  lfg_comm_sync(lfg_context *ctx);
/* Compute units kept free for the communicator.  RCCL's device kernel (gfx950: 256 threads, 261 - 280 vector registers
 * per lane, 19.7 KB of LDS per channel) cannot share a CU with a workgroup of the persistent motion prefilter kernel
 * (two of them fill a CU's register files), and those stay for the whole launch: 0.3 ms under a pan, 0.7 - 7 ms on noise,
 * uncorrelated content or a scene cut.  So while a context has a communicator, every stream the LIBRARY owns (the
 * context's own, the lanes') carries a CU mask that leaves 8 CUs -- one in each XCD -- alone, the communication stream
 * is masked to exactly those 8 (a high-priority stream that may use any CU does NOT find them: tools/probe_cu_reserve.hip),
 * RCCL's kernel is limited to as many channels (ncclConfig_t::maxCTAs), and the persistent grid is sized for the other 248.
 * lfg_comm_init makes those streams again (it waits for them first); lfg_comm_destroy gives the CUs back.
 * LFG_COMM_CUS=0|8|16|24|32 in the environment at context creation changes the number (0: no reservation).
 * A stream the CALLER supplied (lfg_context_set_stream) is not touched: create it with
 * hipExtStreamCreateWithCUMask and the mask lfg_comm_cu_mask returns (`words` 32-bit words, at least CUs / 32;
 * bit i set = CU i may be used) or accept that a broadcast waits for a persistent launch to end.
 * (A masked stream has default flags: it synchronises with the NULL stream, which the library itself never uses.) */
int  lfg_comm_reserved_cus(const lfg_context *ctx);      /* 0 without a communicator */
int  lfg_comm_cu_mask(const lfg_context *ctx, uint32_t *out_words, int words);
/* Diagnostic: enqueue, ordered like a broadcast and on the same stream, `workgroups` (1 .. 64) workgroups of
 * the footprint of RCCL's device kernel that stay `microseconds` each (csrc/comm_probe.hip).  A communicator of ONE
 * rank launches nothing for a broadcast; this is how a single GPU shows whether a broadcast would find a CU while the
 * lanes hold the chip (tests/test_gpu_comm.py).  lfg_comm_wait / lfg_comm_sync treat it as a broadcast. */
int  lfg_comm_probe(lfg_context *ctx, int workgroups, int microseconds, int every_lane /* ordered like lfg_broadcast_frame (1) or lfg_broadcast_frame_lane (0) */);
/* Device time of the LAST probe from "everything it was ordered behind has finished" to its own end, in milliseconds
 * (waits for it): its `microseconds` plus however long its workgroups waited for a CU. */
int  lfg_comm_probe_ms(lfg_context *ctx, float *out_ms);
/* Collective teardown (also done by lfg_context_destroy).  Idempotent. */
int  lfg_comm_destroy(lfg_context *ctx);

/* ---------------------------------------------------------------- diagnostics */

/* The motion kernel uses a hand-written correctly rounded sqrt (csrc/lfg_motion_common.hpp: exact_sqrt).  This
 * compares it on the device with the compiler's IEEE sqrtf for every float whose bit pattern lies
 * in [lo_bits, hi_bits] and returns the number of mismatches (expected 0).  Test-suite use only. */
int  lfg_selftest_sqrt(lfg_context *ctx, uint32_t lo_bits, uint32_t hi_bits, uint64_t *out_mismatches);

/* The 2x scale kernel cuts the in_height + 1 row steps (2 .. in_height + 2; step r emits output rows 2r-5 and 2r-4)
 * into one contiguous band per XCD and each band into strips of three lengths (csrc/scale.hip: scale_2x_strip_of, the
 * same function the kernel evaluates).  This returns strip `index` of XCD `xcd`: its first step and its number of
 * steps (0 = nothing left of the band) and how many strips an XCD has.  Host arithmetic only, no GPU needed.
 * Test-suite use: every step must belong to exactly one strip. */
int  lfg_diag_scale_2x_strip(uint32_t in_height, uint32_t xcd, uint32_t index, uint32_t *out_strips_per_xcd,
                             int32_t *out_first_step, int32_t *out_steps);

/* ---------------------------------------------------------------- measurement */

/* When enabled, every stage launch is bracketed by HIP events on the context's stream; the
 * accumulated device time and launch count per stage are read with lfg_profile_get() (which
 * synchronises).  Disabled by default; lfg_profile_reset() clears the accumulators. */
int  lfg_profile_enable(lfg_context *ctx, int enabled);
int  lfg_profile_reset(lfg_context *ctx);
int  lfg_profile_get(lfg_context *ctx, int stage, double *out_total_ms, uint64_t *out_launches);

#ifdef __cplusplus
}
#endif
#endif /* LINUXFG_HIP_H */
