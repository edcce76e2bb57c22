#include "scaler.hpp"

#include <algorithm>

namespace {
lfg_context* Ctx() { return HipContext::Get().GetDevice(); }
}

bool Scaler::Initialize(const ScalerConfig& config) {
    if (m_initialized) return true;
    m_config = config;
    if (!Ctx()) {
        LOG_ERROR("Scaler::Initialize: HipContext is not initialized");
        return false;
    }
    if (config.inputWidth == 0 || config.inputHeight == 0 || config.outputWidth == 0 || config.outputHeight == 0) {
        LOG_ERROR("Scaler::Initialize: input and output sizes must be non-zero");
        return false;
    }
    if (!m_source) m_source = std::make_unique<SyntheticCapture>();
    if (!m_source->Initialize(config.inputWidth, config.inputHeight)) {
        LOG_ERROR("Failed to initialize frame source");
        return false;
    }
    if (m_sharedSource) {
        if (HipContext::Get().GetRanks() < 1) {
            LOG_ERROR("Scaler::Initialize: a shared previous frame needs HipContext::InitComm first");
            return false;
        }
        if (HipContext::Get().GetRank() == 0 && !m_sharedSource->Initialize(config.inputWidth, config.inputHeight)) {
            LOG_ERROR("Failed to initialize the shared frame source");
            return false;
        }
    }
    if (m_factors.empty()) m_factors.push_back(config.interpolationFactor);
    if (m_factors.size() > LFG_MAX_FACTORS) {
        LOG_ERROR("Scaler::Initialize: at most ", LFG_MAX_FACTORS, " interpolation factors per pair");
        return false;
    }
    m_lanes = (m_sharedSource || !m_pipelined || !config.enableInterpolation) ? 1 : m_framesInFlight;
    if (m_lanes != m_framesInFlight)
        LOG_WARN("Scaler::Initialize: frames in flight need a stream of frames, interpolation and pipelined presentation; using one lane");
    if (lfg_lanes(Ctx(), m_lanes) != LFG_OK) {
        LOG_ERROR("Failed to create the lanes: ", lfg_last_error(Ctx()));
        return false;
    }
    const size_t inBytes = (size_t)config.inputWidth * config.inputHeight * 4;
    const size_t outBytes = (size_t)config.outputWidth * config.outputHeight * 4;
    if (lfg_ring_create(Ctx(), (uint32_t)(2 + m_lanes), inBytes, &m_uploadRing) != LFG_OK ||
        lfg_ring_create(Ctx(), (uint32_t)((1 + m_lanes) * (m_factors.size() + 1) + 2), outBytes, &m_readbackRing) != LFG_OK) {   // the calls in flight + the one being presented + margin
        LOG_ERROR("Failed to create pinned frame rings: ", lfg_last_error(Ctx()));
        Cleanup();
        return false;
    }
    m_initialized = true;
    return true;
}

bool Scaler::CreateFrameResources() {
    auto& fm = FrameManager::Get();
    if (!m_currentFrame.data) {
        LOG_DEBUG("Creating current frame buffer");
        if (!fm.CreateFrame(m_currentFrame, m_config.inputWidth, m_config.inputHeight)) {
            LOG_ERROR("Failed to create current frame");
            return false;
        }
    }
    if (m_config.enableInterpolation && !m_previousFrame.data) {
        LOG_DEBUG("Creating previous frame buffer");
        if (!fm.CreateFrame(m_previousFrame, m_config.inputWidth, m_config.inputHeight)) {
            LOG_ERROR("Failed to create previous frame");
            return false;
        }
    }
    if (!m_outputFrame.data) {
        LOG_DEBUG("Creating output frame buffer");
        if (!fm.CreateFrame(m_outputFrame, m_config.outputWidth, m_config.outputHeight)) {
            LOG_ERROR("Failed to create output frame");
            return false;
        }
    }
    if (m_sharedSource && !m_sharedIn[0].data) {
        if (!fm.CreateFrame(m_sharedIn[0], m_config.inputWidth, m_config.inputHeight) ||
            !fm.CreateFrame(m_sharedIn[1], m_config.inputWidth, m_config.inputHeight)) {
            LOG_ERROR("Failed to create the shared previous frames");
            return false;
        }
    }
    if (m_config.enableInterpolation && !m_previousOutput.data) {
        bool ok = fm.CreateFrame(m_previousOutput, m_config.outputWidth, m_config.outputHeight);
        m_laneInterpolated.assign((size_t)m_lanes, std::vector<Frame>(m_factors.size()));
        for (auto& lane : m_laneInterpolated)
            for (Frame& f : lane) ok = ok && fm.CreateFrame(f, m_config.outputWidth, m_config.outputHeight);
        m_interpolatedFrames = m_laneInterpolated[0];
        // frames in flight: with n lanes the upscaled frame written by call k was last read by call k - n (as its
        // current frame) and k - n + 1 (as its previous one): n + 1 buffers in rotation; inputs: one per lane
        for (int i = 0; i + 1 < m_lanes && ok; ++i) {
            Frame extraOut;
            ok = fm.CreateFrame(extraOut, m_config.outputWidth, m_config.outputHeight);
            if (ok) m_outputPool.push_back(extraOut);
        }
        if (!ok) {
            LOG_ERROR("Failed to create interpolation frames");
            return false;
        }
    }
    if (m_lanes > 1 && m_inputPool.empty()) {
        m_inputPool.push_back(m_currentFrame);
        m_inputPool.push_back(m_previousFrame);
        for (int i = 2; i < m_lanes; ++i) {
            Frame extra;
            if (!fm.CreateFrame(extra, m_config.inputWidth, m_config.inputHeight)) {
                LOG_ERROR("Failed to create an input frame");
                return false;
            }
            m_inputPool.push_back(extra);
        }
    }
    return true;
}

bool Scaler::CaptureFrame(Frame& frame) { return CaptureFrom(*m_source, frame); }

// Root: the next shared previous frame goes into m_sharedIn[slot]; every rank: its broadcast starts (asynchronously,
// ordered after everything enqueued so far -- the kernels that read the slot two calls ago).
bool Scaler::IssueSharedPrevious(int slot) {
    if (HipContext::Get().GetRank() == 0 && !CaptureFrom(*m_sharedSource, m_sharedIn[slot])) return false;
    return FrameManager::Get().BroadcastFrame(m_sharedIn[slot], 0);
}

bool Scaler::CaptureFrom(FrameSource& source, Frame& frame) {
    void* host = nullptr;
    uint32_t slot = 0;
    if (lfg_ring_acquire(m_uploadRing, &host, &slot) != LFG_OK) {
        LOG_ERROR("Failed to acquire an upload slot");
        return false;
    }
    if (!source.NextFrame(static_cast<uint8_t*>(host))) {
        LOG_ERROR("Failed to capture frame");
        return false;
    }
    lfg_frame f = frame.AsAbi();
    if (lfg_ring_upload(m_uploadRing, slot, &f) != LFG_OK) {
        LOG_ERROR("Failed to upload captured frame: ", lfg_last_error(Ctx()));
        return false;
    }
    return true;
}

bool Scaler::ScaleFrame(const Frame& input, Frame& output) {
    LOG_DEBUG("ScaleFrame - Input: ", input.width, "x", input.height, " Output: ", output.width, "x", output.height);
    const lfg_frame in = input.AsAbi();
    lfg_frame out = output.AsAbi();
    if (lfg_scale(Ctx(), &in, &out) != LFG_OK) {
        LOG_ERROR("Failed to submit scale kernel: ", lfg_last_error(Ctx()));
        return false;
    }
    return true;      // enqueued; ProcessFrame waits where it needs the pixels
}

// Enqueue the read-back of `frame` into a ring slot (copy stream) and remember it for presentation.
bool Scaler::QueueReadback(Frame& frame, bool interpolated) {
    void* host = nullptr;
    uint32_t slot = 0;
    const lfg_frame f = frame.AsAbi();
    if (lfg_ring_acquire(m_readbackRing, &host, &slot) != LFG_OK || lfg_ring_download(m_readbackRing, slot, &f) != LFG_OK) {
        LOG_ERROR("Failed to read back frame: ", lfg_last_error(Ctx()));
        return false;
    }
    m_lastReadback[frame.data] = slot;
    m_pending.push_back(Pending{host, slot, frame.width, frame.height, interpolated});
    return true;
}

// Present the oldest read-backs until only `keep` are left in flight.
bool Scaler::PresentPending(size_t keep) {
    while (m_pending.size() > keep) {
        const Pending p = m_pending.front();
        m_pending.pop_front();
        if (lfg_ring_wait(m_readbackRing, p.slot) != LFG_OK) {               // the reference's wait-idle before mapping
            LOG_ERROR("Failed to wait for a read-back: ", lfg_last_error(Ctx()));
            return false;
        }
        if (m_presenter) m_presenter(static_cast<const uint8_t*>(p.host), p.width, p.height, p.interpolated);
        ++m_presented;
    }
    return true;
}

bool Scaler::Flush() { return m_readbackRing ? PresentPending(0) : true; }

bool Scaler::ProcessFrame() {
    if (!m_initialized) {
        LOG_ERROR("Scaler not initialized");
        return false;
    }
    // FPS meter: 60-sample sliding window, as src/scaler.cpp:428-439.
    const auto now = std::chrono::steady_clock::now();
    m_frameTimings.push_back(now);
    while (m_frameTimings.size() > 60) m_frameTimings.pop_front();
    if (m_frameTimings.size() >= 2) {
        const auto ms = std::chrono::duration_cast<std::chrono::microseconds>(m_frameTimings.back() - m_frameTimings.front()).count();
        if (ms > 0) m_currentFps = 1e6f * (float)(m_frameTimings.size() - 1) / (float)ms;
    }

    if (!CreateFrameResources()) return false;

    if (m_sharedSource) {
        // this call's shared previous frame: its broadcast was issued during the last call (now, for the first);
        // the next call's starts right away and runs next to this call's kernels
        const int slot = (int)(m_calls & 1u);
        if (m_calls == 0 && !IssueSharedPrevious(0)) return false;
        if (!FrameManager::Get().WaitBroadcasts()) return false;
        if (m_config.enableInterpolation) {
            auto fence = [&](const Frame& f) {
                const auto it = m_lastReadback.find(f.data);
                if (it != m_lastReadback.end()) lfg_ring_fence_slot(m_readbackRing, it->second);
            };
            fence(m_previousOutput);
            if (!ScaleFrame(m_sharedIn[slot], m_previousOutput)) return false;
            m_havePrevious = true;
        }
        if (!IssueSharedPrevious(slot ^ 1)) return false;
    }
    // Frames in flight: this call's uploads, kernels and read-backs go to lane k % n.  It waits -- on the device -- for
    // the previous call's upscale (the other lane's mark): that is this call's previous frame, and everything that
    // lane did before it is what last read the buffers this call is about to overwrite.
    const int lane = (int)(m_calls % (uint64_t)m_lanes);
    if (m_lanes > 1) {
        if (lfg_lane_select(Ctx(), lane) != LFG_OK || (m_calls > 0 && lfg_lane_wait(Ctx(), (int)((m_calls - 1) % (uint64_t)m_lanes)) != LFG_OK)) {
            LOG_ERROR("Failed to switch lanes: ", lfg_last_error(Ctx()));
            return false;
        }
        m_currentFrame = m_inputPool[(size_t)lane];
        m_interpolatedFrames = m_laneInterpolated[(size_t)lane];
    }
    ++m_calls;
    if (!CaptureFrame(m_currentFrame)) {
        LOG_ERROR("Failed to capture frame");
        return false;
    }
    // A frame about to be overwritten may still be on its way to the host (pipelined presentation): the
    // kernels wait for that transfer on the device.
    auto fenceBeforeWrite = [&](const Frame& f) {
        const auto it = m_lastReadback.find(f.data);
        if (it != m_lastReadback.end()) lfg_ring_fence_slot(m_readbackRing, it->second);
    };
    fenceBeforeWrite(m_outputFrame);
    if (!ScaleFrame(m_currentFrame, m_outputFrame)) {
        LOG_ERROR("Failed to scale frame");
        return false;
    }
    if (m_lanes > 1 && lfg_lane_mark(Ctx()) != LFG_OK) {
        LOG_ERROR("Failed to mark the lane: ", lfg_last_error(Ctx()));
        return false;
    }
    const size_t inFlightBefore = m_pending.size();
    if (m_config.enableInterpolation && m_havePrevious) {
        for (Frame& f : m_interpolatedFrames) fenceBeforeWrite(f);
        auto& fm = FrameManager::Get();
        bool ok;
        if (m_factors.size() == 1) {                          // the reference's own entry point
            ok = m_pipelined ? fm.InterpolateFramesAsync(m_previousOutput, m_outputFrame, m_interpolatedFrames[0], m_factors[0])
                             : fm.InterpolateFrames(m_previousOutput, m_outputFrame, m_interpolatedFrames[0], m_factors[0]);
        } else {                                              // motion once, every factor in one pass
            std::vector<Frame*> outs;
            for (Frame& f : m_interpolatedFrames) outs.push_back(&f);
            ok = m_pipelined ? fm.InterpolateFramesMultiAsync(m_previousOutput, m_outputFrame, outs, m_factors)
                             : fm.InterpolateFramesMulti(m_previousOutput, m_outputFrame, outs, m_factors);
        }
        if (!ok) {
            LOG_ERROR("Failed to interpolate frame");
            return false;
        }
        // Presentation order: previous real frame (presented by the last call), generated frames t1 .. tN, this real frame.
        for (Frame& f : m_interpolatedFrames)
            if (!QueueReadback(f, true)) return false;
    }
    if (!QueueReadback(m_outputFrame, false)) return false;
    // Pipelined: present what earlier calls queued while this call's work runs (one call back; with n > 2 lanes the
    // last n - 1 calls stay in flight); otherwise everything now.
    m_queuedPerCall.push_back(m_pending.size() - inFlightBefore);
    while (m_queuedPerCall.size() > (size_t)std::max(1, m_lanes - 1)) m_queuedPerCall.pop_front();
    size_t keep = 0;
    for (size_t q : m_queuedPerCall) keep += q;
    if (!PresentPending(m_pipelined ? keep : 0)) return false;

    if (m_config.enableInterpolation && !m_sharedSource) {
        // previous <- current: move the handles instead of copying the image (src/scaler.cpp:616-621); the upscaled
        // frames rotate through the pool (empty with one lane: a plain swap)
        if (m_lanes == 1) std::swap(m_previousFrame, m_currentFrame);
        else m_previousFrame = m_currentFrame;
        m_outputPool.push_back(m_previousOutput);
        m_previousOutput = m_outputFrame;
        m_outputFrame = m_outputPool.front();
        m_outputPool.pop_front();
        m_havePrevious = true;
    }
    return true;
}

void Scaler::Cleanup() {
    if (m_initialized) Flush();
    m_pending.clear();
    m_lastReadback.clear();
    if (Ctx()) HipContext::Get().WaitIdle();
    if (m_uploadRing) { lfg_ring_destroy(m_uploadRing); m_uploadRing = nullptr; }
    if (m_readbackRing) { lfg_ring_destroy(m_readbackRing); m_readbackRing = nullptr; }
    auto& fm = FrameManager::Get();
    if (m_lanes > 1) {
        // (the handles in m_currentFrame / m_previousFrame / m_interpolatedFrames are copies of pool entries)
        for (Frame& f : m_inputPool) fm.DestroyFrame(f);
        m_currentFrame = Frame{}; m_previousFrame = Frame{};
    } else {
        fm.DestroyFrame(m_currentFrame);
        fm.DestroyFrame(m_previousFrame);
    }
    m_inputPool.clear();
    fm.DestroyFrame(m_outputFrame);
    fm.DestroyFrame(m_previousOutput);
    for (Frame& f : m_outputPool) fm.DestroyFrame(f);
    m_outputPool.clear();
    m_queuedPerCall.clear();
    if (Ctx() && m_sharedSource && HipContext::Get().GetRanks() > 0) (void)fm.WaitBroadcasts();   // the look-ahead broadcast of the last call
    fm.DestroyFrame(m_sharedIn[0]);
    fm.DestroyFrame(m_sharedIn[1]);
    m_calls = 0;
    for (auto& lane : m_laneInterpolated)
        for (Frame& f : lane) fm.DestroyFrame(f);
    m_laneInterpolated.clear();
    m_interpolatedFrames.clear();
    if (Ctx()) { (void)lfg_lane_select(Ctx(), 0); (void)lfg_lanes(Ctx(), 1); }
    m_lanes = 1;
    m_factors.clear();
    m_havePrevious = false;
    m_frameTimings.clear();
    m_initialized = false;
}
