// Internal declarations shared by the C-ABI implementation and the kernel launchers.
// Not part of the public boundary (that is include/linuxfg_hip.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "linuxfg_hip.h"

namespace lfg {

// Per-axis Lanczos tables for one (inSize -> outSize) resample, as shaders/scale.comp:24-41
// computes them per output coordinate: first tap index (floor(pixelPos) - 2, unclamped) and six
// weights, pre-normalised by the sum of the in-range weights (the shader divides by totalWeight,
// scale.comp:48; the filter is exactly separable) and zeroed for taps
// the shader skips (scale.comp:34-37).
struct AxisTable {
    int in_size = 0, out_size = 0;
    bool pattern_2x = false;      // start[2k] == k-3 and start[2k+1] == k-2 for every k
    int *d_start = nullptr;       // [out_size]
    float *d_weight = nullptr;    // [out_size][6]
    // pattern_2x tables only.  At exact 2x the shader's per-column fp32 arithmetic yields a handful of DISTINCT weight
    // rows (the nominal phase row for ~97 % of the columns, a few rounding variants, the renormalised border rows:
    // 31 rows at 1920 -> 3840), so the 2x kernel fetches a one-byte class per output column and the row from a small
    // palette that stays in L1, instead of 96 bytes of table per lane: d_class[p] indexes d_palette[class][8]
    // (six weights + two pad floats, 32-byte rows).  palette_rows == 0: too many distinct rows, not built.
    uint8_t *d_class = nullptr;   // [out_size rounded up to 4]
    float *d_palette = nullptr;   // [palette_rows][8]
    int palette_rows = 0;
    // pattern_2x tables of the VERTICAL axis: strips per XCD of the 2x kernel (scale.hip: scale_2x_strip_of).
    int strips_per_xcd = 0;
};

// uv[p] = ((float)p + 0.5f) / (float)size for p = 0 .. size - 1 (shaders/interpolate.comp:30): the normalised coordinate of a
// pixel centre depends on its column (row) alone, so the interpolate kernels read it from a table built once per size on the
// host -- the same fp32 division, bit for bit -- instead of five IEEE divisions per thread (the kernel was VALU-bound on them).
//
// Round 4: where nothing displaces the sample (a zero motion vector -- under the literal semantics the only vector whose samples
// stay inside the image) texture() is asked for uv[p] itself, and fl(uv[p] * size - 0.5) is texel p with fraction 0
// for most p but not for all: in fp32 102 of 3840 columns and 85 of 2160 rows miss by an ulp and get real bilinear weights.
// `centre` marks the p that hit (the sample IS texel p: no coordinate arithmetic, no weights).  For the x axis the groups of four
// pixels a thread works on ("quads") have one bit each in `goodMask`, set where all four pixels are centres: a wave covers 64
// consecutive quads and learns with one scalar load which of its lanes need the two-texel blends of interpolate.hip.
struct UvTable {
    int size = 0;
    float *d_uv = nullptr;          // [size rounded up to 4]; the base of ONE allocation that also holds the arrays below
    uint8_t *d_centre = nullptr;    // [size rounded up to 64]: 1 where the undisplaced sample of p is exactly texel p
    uint64_t *d_goodMask = nullptr; // [blocks]: bit l of word b: quad 64 b + l lies inside the axis and is all centres
    int blocks = 0;                 // ceil(quads / 64)
    // uv[p] without the table and without an IEEE division: q = (p + 0.5) * rcp, then one correction step,
    // q + fma(-q, size, p + 0.5) * rcp -- three instructions.  rcpExact: the host has checked that this reproduces uv[p] bit for bit
    // for EVERY p of this axis (it does for all the benchmark's sizes); otherwise the kernels read the table.
    float rcp = 0.0f;
    bool rcpExact = false;
};

// What the interpolate kernels take besides the frames (interpolate.hip).
struct InterpTables {
    const float *uvx = nullptr, *uvy = nullptr;
    const uint8_t *centreX = nullptr, *centreY = nullptr;
    const uint64_t *goodMask = nullptr;        // of the x axis
    float rcpW = 0.0f, rcpH = 0.0f;            // UvTable::rcp of the two axes
    int rcpExact = 0;                          // both axes' three-instruction uv is exact (UvTable::rcpExact)
};

// lfg_interpolate_frames in the north-star order (the reference: motion dispatch, then interpolate
// dispatch on the same grid, src/frame_manager.cpp:342-366): the motion kernels (8 / 16 paths) write the GENERATED frame from the
// vector they have just decided, while it is in a register -- csrc/lfg_interp.hpp: interpolate_pixel, the very function the
// interpolate kernel is made of -- and the motion-vector frame becomes an optional output.  data == nullptr: off.
struct FusedOut {
    uint8_t *data = nullptr;      // RGBA8, the size of curr
    int pitch = 0;
    float t = 0.5f;               // interpolationFactor
    int intended = 0;             // lfg_set_semantics: the vector divided by the image size before it displaces uv
    int storeMv = 1;              // 0: the caller has no use for the vectors (lfg_interpolate_frames' temporary)
};

// Words of a call's control area (MotionWorkspaceLayout::ctrl), cleared by the hint kernel with the rest of the area.
enum MotionCtrlWord : int {
    kCtrlNextUnit = 0,        // next plan unit to draw
    kCtrlUnitsDone = 1,       // plan units finished (only a running plan unit can push onto the queue)
    kCtrlNextSlot = 2,        // next queue slot to draw
    kCtrlOpenCount = 4,       // segments left to the resolve kernel (the length of openList)
    kCtrlHardCount = 5,       // tiles in which the lean kernel left a segment (the length of hardTiles)
    kCtrlLeanSettled = 6,     // segments the lean kernel settled / left: counted in -DLFG_LEAN_STATS builds only
    kCtrlLeanLeft = 7,
};

// Measurement knobs, read ONCE from the environment when a context is created (lfg_context_create) and never again: a call's
// launch geometry cannot change between two lfg_motion calls because somebody called setenv.  All of them are for A/B runs of
// tools/; none changes a result.
struct MotionKnobs {
    int leanForce = -1;       // LFG_LEAN_FORCE = 1: every call through the lean kernel whatever the verdict, 0: none, unset: by the verdict
    int fallbackFull = 0;     // LFG_FALLBACK_FULL: the second pass always on its full grid
    int dynParts = 0;         // LFG_DYN_PARTS_RT = 4 | 8: parts of a handed-over segment, whatever the lane count
    int prefGroups = 0;       // LFG_PREF_GROUPS: workgroups of the persistent kernel
    int resolveGroups = 0;    // LFG_RESOLVE_GROUPS: workgroups of the resolve kernel
    int strips = 0;           // LFG_MOTION_STRIP = 1: the exposed strips through a kernel of their own (motion_strip.hip: exact, measured slower)
    int stripPad = -1;        // LFG_STRIP_PAD: bytes of dynamic LDS a strip workgroup asks for on top of its own (-1: the launcher's choice)
    int debug = 0;            // LFG_DEBUG: reporting calls print what they read
    int debugDyn = 0;         // LFG_DEBUG_DYN: lfg_motion_last_stats prints the deepest private lists of the handed-over segments,
    int debugDynDeep = 14;    // LFG_DEBUG_DYN_DEEP: ... deeper than this
    int tierForce = -1;       // LFG_TIER_FORCE = 0 | 1: the persistent kernel's variant whatever the verdict (-1: by the verdict)
    int commCus = 8;          // LFG_COMM_CUS = 0 | 8 | 16 | 24 | 32: CUs a communicator keeps free of the library's own kernels (lfg_comm.cpp)
};

struct MotionWorkspaceLayout { size_t colBand, rowBand /* [height], [width] words of the strip kernel (motion_strip.hip), inside the control area the hint kernel clears */; size_t verdict /* byte offset of the call's verdict word, order32[kCand + 2] of its own order table; orderFlags: of [kCand] */, orderFlags; size_t list, umin, count, tileFlags, segDone, segMap, queueCount, ctrl, order, plan, auxList, auxUmin, auxCount,
                               queue, dynList, dynUmin, dynCount, dynInit, openList, merge, mergeBytes, leanTiles, hardTiles, plan2, total; int queueCap, slots, rimSplit, listMain, listAux, listDyn, leanCount, leanLaunch /* with the partial tiles behind them */, rimSplit2, units, units2, units2Static /* the second plan's units without the lean kernel's tiles, which come last in its table */, tiles, lastLean /* the lane's last call went by the second plan */; };
// Work units of the motion prefilter (motion_plan.hip: prefilter_plan).  A unit is a 56 x 64 tile, or one of nChunks
// contiguous parts of a tile's candidate order, or one 16-row segment of a tile with its four waves on four parts of
// the order; parts have private lists in the aux arrays (merged by the resolve kernel).
struct PrefilterPlan {               // passed by value to the kernels
    int tilesX, units;
    const uint32_t *unitMap;         // per unit: tile | (first) chunk << 20 | nChunks << 24 | segment unit << 28 | segment << 29
    const uint32_t *unitAux;         // per unit: index of its tile's first 56 x 64 block in the aux arrays (whole tiles: 0xFFFFFFFF)
    const uint32_t *tileMap;         // per tile: 0xFFFFFFFF (whole) or first aux index | nChunks << 24
    uint32_t *auxList;               // records: csrc/lfg_motion_common.hpp, Rec
    float *auxUmin;
    uint32_t *auxCount;
    // Segments handed over at run time (a whole tile's segment that finds no match after the first batches): a queue
    // of unitMap-style entries and their private lists in 16-row blocks; see motion_prefilter_kernel.
    uint32_t *segMap;                // per (tile, segment): 0, or first block | parts << 24 | 1 << 31
    uint32_t *queueCount;            // [0] entries pushed this call (may exceed queueCap: the excess was not handed over), [1] tiles flagged
    uint32_t *queue;                 // [queueCap] unit entries (0: not pushed yet); slot h's lists are blocks 4h .. 4h+3
    int queueCap;
    uint32_t *dynList;
    float *dynUmin;
    uint32_t *dynCount;
    uint32_t *dynInit;               // per handed-over segment: the 16 x 56 thresholds of the wave that handed it over
    int dynParts;                    // parts of the candidate order a handed-over segment is searched in: 8 (two workgroups) one frame at a time, 4 with frames in flight
    uint32_t *openList, *openCount;  // the segments left to the resolve kernel (tile * 4 + segment), appended as units end
    // Pixels the strip kernel has decided before this launch (motion_strip.hip; nullptr: that kernel did not run): colBand[y] bit 0 /
    // bit 1 = the left / right kStripCols columns of row y, rowBand[x] = the top / bottom kStripRows rows of column x
    // (lfg_motion_tile.hpp: strip_decided).  Rim units and the resolve kernel leave them alone.
    const uint32_t *colBand, *rowBand;
    // Calls that went through the lean kernel first (motion_lean.hip): the table's last units -- that kernel's tiles -- are not drawn
    // from the table; instead the tiles in which it LEFT a segment come from the list it wrote (hardCount: nullptr = no such call).
    int unitsStatic;
    const uint32_t *hardTiles, *hardCount;
    FusedOut fused;                  // the generated frame, written where the vectors are decided (off: data == nullptr)
};
struct PrefilterPlanHost {
    int tilesX = 0, tiles = 0, units = 0, auxUnits = 0;
    std::vector<uint32_t> unitMap, unitAux, tileMap;
    std::vector<uint32_t> leanTiles;     // the whole tiles the lean kernel may take first (motion_lean.hip: lean_tile_ok)
    std::vector<uint32_t> leanPartial;   // ... and rim tiles of which it takes the segments that lie inside the image: tile | mask of segments << 24
};

struct ProfileSlot {
    hipEvent_t begin = nullptr, end = nullptr;
    int stage = 0;
};

}  // namespace lfg

// What a lane (lfg_lanes: one of several frames in flight on a GPU) owns apart from the context: its stream, the temporaries
// and the motion workspace of its calls, and the event other lanes wait for.  The context's own fields ARE the selected
// lane's; lfg_lane_select swaps them with the entry of this table.
struct lfg_lane_state {
    hipStream_t own_stream = nullptr, stream = nullptr;
    lfg_frame mv_tmp{}, mid_tmp{};
    uint8_t *motion_ws = nullptr;
    size_t motion_ws_bytes = 0;
    uint32_t motion_ws_w = 0, motion_ws_h = 0;
    lfg::MotionWorkspaceLayout motion_ws_layout{};
    int motion_units = 0;
    hipEvent_t mark = nullptr;
    bool marked = false;
    uint32_t *lean_flag = nullptr;             // pinned: the order kernel's verdict on the lane's last call ("content for the lean kernel")
    hipEvent_t lean_ev = nullptr;              // ... recorded behind its copy
    bool lean_ev_pending = false;
    int lean_predict = 0;                      // the verdict word the next call goes by (bit 0: lean kernel; bit 30: a tile went through the literal kernel; bit 31: most sample blocks have a match)
    bool lean_seen = false;                    // ... and whether any call's word has arrived yet
    uint32_t lean_request_guess = 0;           // what the call that carries the pending verdict request was launched on (motion_run)
};

struct lfg_context {
    int device = 0;
    int device_cus = 0;                        // its compute units
    std::vector<lfg_lane_state> lanes;         // empty until lfg_lanes(); entry `lane` is stale while that lane is selected
    int lane = 0;
    hipEvent_t mark = nullptr;                 // the selected lane's (see lfg_lane_state)
    bool marked = false;
    uint32_t *lean_flag = nullptr;             // likewise
    hipEvent_t lean_ev = nullptr;
    bool lean_ev_pending = false;
    int lean_predict = 0;
    bool lean_seen = false;
    uint32_t lean_request_guess = 0;
    // lfg_motion_prediction_stats: calls whose verdict came back, and the guesses it contradicted (all lanes together)
    uint64_t pred_verdicts = 0, pred_lean_wrong = 0, pred_grid_wrong = 0, pred_second_wrong = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string error;
    std::vector<lfg::AxisTable> tables;       // small cache, linear search
    std::vector<lfg::UvTable> uv_tables;      // likewise (interpolate)
    lfg_frame mv_tmp{};                        // temporary of lfg_interpolate_frames
    lfg_frame mid_tmp{};                       // temporary of lfg_interpolate_scale where the fused kernel does not apply
    // prefiltered motion path: scratch for one frame size, grown on demand
    uint8_t *motion_ws = nullptr;
    size_t motion_ws_bytes = 0;
    uint32_t motion_ws_w = 0, motion_ws_h = 0;
    lfg::MotionWorkspaceLayout motion_ws_layout{};
    int motion_units = 0;                      // work units of the prefilter for the current workspace size
    int motion_slots = 0;                      // prefilter workgroups resident at once on this device (0 = not queried yet)
    int rim_split_env = 0;                     // LFG_MOTION_RIM_SPLIT at context creation (0: unset -- the plan follows the lane count)
    int motion_mode = 0;                       // 0: prefilter + exact fallback, 1: exact kernel only
    int semantics = 0;                         // 0: the shaders as written, 1: "intended" (lfg_set_semantics)
    uint32_t *motion_tables = nullptr;         // device: [semantics][rank2scan | order32 | entryOfScan], then baseScan
    bool fuse_interpolate_scale = false;       // lfg_interpolate_scale: one fused kernel instead of the two stages (measured slower)
    bool fuse_motion_interpolate = false;      // lfg_interpolate_frames: the motion kernels write the generated frame themselves
    bool motion_hints = true;                  // per-call visiting order from sample-block hints (LFG_MOTION_HINTS=0: off)
    bool motion_lean = true;                   // whole interior tiles go through the lean kernel first (LFG_MOTION_LEAN=0: off)
    lfg::MotionKnobs knobs;                    // measurement knobs, read once at creation
    // the one exchange of the path (lfg_comm.cpp): an RCCL communicator, its stream and two events
    void *comm = nullptr;                      // ncclComm_t
    int comm_ranks = 0, comm_rank = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t comm_ready = nullptr, comm_done = nullptr;
    bool comm_pending = false;                 // some broadcast has been issued on this communicator (comm_done has been recorded)
    hipEvent_t probe_begin = nullptr, probe_end = nullptr;     // lfg_comm_probe: device timestamps, ready and done
    int motion_last_tier = 0;                  // the persistent kernel's variant of the last lfg_motion on any lane (lfg_motion_last_variant)
    int comm_cus = 0;                          // CUs the library's own streams leave to the communicator's kernels (0: none reserved)
    // profiling
    bool profile = false;
    std::vector<lfg::ProfileSlot> prof_pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_free;
    double prof_ms[LFG_STAGE_COUNT] = {0, 0, 0};
    uint64_t prof_n[LFG_STAGE_COUNT] = {0, 0, 0};
};

// A stream of the library's own (lfg_capi.cpp): non-blocking; with `ctx->comm_cus` CUs reserved for a communicator, a stream whose
// CU mask leaves those free.  lfg_restream: every stream the library owns is made again under the current reservation.
hipError_t lfg_own_stream_create(const lfg_context *ctx, hipStream_t *out);
int lfg_restream(lfg_context *ctx);

namespace lfg {

// A kernel of the footprint of RCCL's device kernel (comm_probe.hip): what lfg_comm_probe puts where a broadcast would run.
hipError_t launch_comm_probe(hipStream_t s, int workgroups, int microseconds);

// Kernel launchers (scale.hip, motion_*.hip, interpolate.hip).  All enqueue on `s` and return the
// launch status; arguments have been validated by the caller.
hipError_t launch_scale_generic(hipStream_t s, const lfg_frame &in, const lfg_frame &out,
                                const AxisTable &tx, const AxisTable &ty);
hipError_t launch_scale_2x(hipStream_t s, const lfg_frame &in, const lfg_frame &out,
                           const AxisTable &tx, const AxisTable &ty);
bool scale_2x_supported(const lfg_frame &in, const lfg_frame &out);
hipError_t launch_interpolate_scale_2x(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv,
                                       const lfg_frame &out, const AxisTable &tx, const AxisTable &ty, float factor, bool intended);
int scale_2x_strips_per_xcd(int inH);
void scale_2x_strip_host(int inH, int xcd, int index, int &first, int &steps);
hipError_t launch_motion_tiled_8_16(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr,
                                    const lfg_frame &mv, const uint32_t *tileFlags, const uint32_t *rank2scan,
                                    unsigned long long *merge = nullptr, uint32_t *flaggedTiles = nullptr, const FusedOut &fused = FusedOut(),
                                    bool expectNothing = false, uint32_t *verdictWord = nullptr, uint32_t *hostWord = nullptr);
// Candidate tables of the blockSize 8 / searchRadius 16 paths for one tie-break rule (motion_order.hip: motion_tables).
constexpr int kMotionTableWords = 1092;     // 33 * 33 candidates + the sentinel, padded to a multiple of 4
void motion_tables(bool intended, uint32_t *rank2scan, uint32_t *order32, uint32_t *entryOfScan, uint32_t *baseScan);
// Prefiltered motion path (motion_plan.hip, motion_prefilter.hip): MotionWorkspaceLayout = byte offsets of its scratch arrays.
size_t motion_workspace_bytes(uint32_t width, uint32_t height, int slots, int rimSplit, int rimSplit2, MotionWorkspaceLayout *layout);   // rimSplit2: the plan used beside the lean kernel (0: none)
PrefilterPlanHost prefilter_plan(uint32_t width, uint32_t height, int slots, int rimSplit);   // rimSplit: 4 or 8 parts of the order per rim segment
int prefilter_slots();      // workgroups of the prefilter kernel the current device holds at once
hipError_t launch_motion_prefiltered_8_16(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr,
                                          const lfg_frame &mv, uint8_t *workspace, const MotionWorkspaceLayout &layout, int units,
                                          const uint32_t *rank2scan, const uint32_t *order32,
                                          const uint32_t *entryOfScan, const uint32_t *baseScan, bool useHints, bool framesInFlight,
                                          const FusedOut &fused = FusedOut(), bool lean = false, uint32_t *leanFlagHost = nullptr,
                                          int groupsCap = 0 /* persistent workgroups at most (0: as many as the device holds) */,
                                          bool expectNoFallback = false /* the lane's previous call flagged no tile: a small fallback launch */,
                                          const MotionKnobs &knobs = MotionKnobs(), bool rankIsScan = true,
                                          int tier = 0 /* 1: the persistent kernel's variant for moderate sensor noise (motion_prefilter_kernel<false, 1>) */);
// This call's visiting order (motion_order.hip): hint kernel (which also clears the call's control area) + order kernel.
hipError_t launch_motion_order(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, uint32_t *hints, uint32_t *callOrder,
                               const uint32_t *entryOfScan, const uint32_t *baseScan, uint32_t *clearFrom, int clearWords, bool framesInFlight);
// The literal chain for what the prefilter left open (motion_resolve.hip); list / umin / count: the image-shaped arrays.
hipError_t launch_motion_resolve(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv, const uint32_t *list,
                                 const float *umin, const uint32_t *count, const uint32_t *tileFlags, int tilesX, const PrefilterPlan &sp,
                                 const uint32_t *rank2scan, const uint32_t *segDone, int groups);
// The lean kernel for whole interior tiles (motion_lean.hip): runs between the order kernel and the generic prefilter, marks the
// segments it settles in segDone; the generic kernel skips those.
bool lean_tile_ok(int tile, int tilesX, int W, int H);
bool lean_segment_ok(int tile, int seg, int tilesX, int W, int H);
bool lean_frames_ok(const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv);
hipError_t launch_motion_lean(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv,
                              const uint32_t *order32, const uint32_t *rank2scan, bool rankIsScan /* the shaders' own tie order: ranks are scan indices */,
                              const uint32_t *leanTiles, int nTiles, int tilesX, uint32_t *segDone,
                              uint32_t *hardTiles, uint32_t *hardCount, uint32_t *stats, bool whateverTheVerdict);
// The strips a pan exposes (motion_strip.hip): runs behind the order kernel, decides its pixels completely and lists them in
// colBand / rowBand (cleared with the call's control area); the persistent kernel's rim units and the resolve kernel skip them.
bool strip_frames_ok(const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv);
hipError_t launch_motion_strip(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv, const uint32_t *order32,
                               const uint32_t *rank2scan, bool rankIsScan, uint32_t *colBand, uint32_t *rowBand,
                               uint32_t *tileFlags, uint32_t *flagged, int flagTilesX, int ldsPad);
hipError_t launch_motion_generic(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr,
                                 const lfg_frame &mv, int block_size, int radius, bool intended);
hipError_t launch_interpolate(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr,
                              const lfg_frame &mv, const lfg_frame &out, float factor, bool intended, const InterpTables &tb);
hipError_t launch_interpolate_multi(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv,
                                    const lfg_frame *const *outs, const float *factors, int count, bool intended,
                                    const InterpTables &tb);
hipError_t launch_mv_export(hipStream_t s, const lfg_frame &mv, float *rgba32f);
hipError_t launch_sqrt_selftest(hipStream_t s, uint32_t lo_bits, uint32_t hi_bits, unsigned long long *d_mismatch);

}  // namespace lfg
