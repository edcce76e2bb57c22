#include "frame_manager.hpp"

namespace {
lfg_context* Ctx() { return HipContext::Get().GetDevice(); }
void FromAbi(const lfg_frame& f, Frame& out) {
    out.data = f.data; out.width = f.width; out.height = f.height; out.pitch = f.pitch;
    out.format = f.format; out.owned = f.owned != 0;
}
}  // namespace

bool FrameManager::Initialize(uint32_t /*width*/, uint32_t /*height*/) {
    // The reference only creates a command pool here and ignores its arguments
    // (src/frame_manager.cpp:3-11); the stream already exists in HipContext.
    if (!Ctx()) {
        LOG_ERROR("FrameManager::Initialize: HipContext is not initialized");
        return false;
    }
    m_initialized = true;
    return true;
}

void FrameManager::Cleanup() { m_initialized = false; }

bool FrameManager::CreateFrame(Frame& frame, uint32_t width, uint32_t height) {
    lfg_frame f{};
    const uint32_t format = frame.format;                    // RGBA8 unless the caller asked otherwise
    if (lfg_frame_create(Ctx(), width, height, format, &f) != LFG_OK) {
        LOG_ERROR("Failed to create frame image: ", Ctx() ? lfg_last_error(Ctx()) : "no device context");
        return false;
    }
    FromAbi(f, frame);
    return true;
}

void FrameManager::DestroyFrame(Frame& frame) {
    if (!frame.data) return;                                  // idempotent, like src/frame_manager.cpp:71-80
    lfg_frame f = frame.AsAbi();
    lfg_frame_destroy(Ctx(), &f);
    frame.data = nullptr;
    frame.width = frame.height = frame.pitch = 0;
    frame.owned = false;
}

bool FrameManager::CopyFrameData(const Frame& source, Frame& destination) {
    if (source.width != destination.width || source.height != destination.height) {
        LOG_ERROR("Source and destination frame dimensions do not match");   // src/frame_manager.cpp:84-87
        return false;
    }
    const lfg_frame s = source.AsAbi();
    lfg_frame d = destination.AsAbi();
    if (lfg_frame_copy(Ctx(), &s, &d) != LFG_OK) {
        LOG_ERROR("Failed to copy frame: ", lfg_last_error(Ctx()));
        return false;
    }
    return HipContext::Get().WaitIdle();                      // the reference waits idle (:143)
}

bool FrameManager::InterpolateFramesAsync(const Frame& previous, const Frame& current, Frame& output, float factor) {
    const lfg_frame p = previous.AsAbi(), c = current.AsAbi();
    lfg_frame o = output.AsAbi();
    if (lfg_interpolate_frames(Ctx(), &p, &c, &o, factor) != LFG_OK) {
        LOG_ERROR("Failed to interpolate frames: ", lfg_last_error(Ctx()));
        return false;
    }
    return true;
}

bool FrameManager::InterpolateFrames(const Frame& previous, const Frame& current, Frame& output, float factor) {
    if (!InterpolateFramesAsync(previous, current, output, factor)) return false;
    return HipContext::Get().WaitIdle();                      // EndSingleTimeCommands waits idle (:183-197)
}

bool FrameManager::InterpolateFramesMultiAsync(const Frame& previous, const Frame& current, const std::vector<Frame*>& outputs,
                                               const std::vector<float>& factors) {
    if (outputs.empty() || outputs.size() != factors.size()) {
        LOG_ERROR("InterpolateFramesMulti: one output frame per factor is required");
        return false;
    }
    const lfg_frame p = previous.AsAbi(), c = current.AsAbi();
    std::vector<lfg_frame> abi(outputs.size());
    std::vector<lfg_frame*> ptrs(outputs.size());
    for (size_t i = 0; i < outputs.size(); ++i) { abi[i] = outputs[i]->AsAbi(); ptrs[i] = &abi[i]; }
    if (lfg_interpolate_frames_multi(Ctx(), &p, &c, ptrs.data(), factors.data(), (uint32_t)factors.size()) != LFG_OK) {
        LOG_ERROR("Failed to interpolate frames: ", lfg_last_error(Ctx()));
        return false;
    }
    return true;
}

bool FrameManager::InterpolateFramesMulti(const Frame& previous, const Frame& current, const std::vector<Frame*>& outputs,
                                          const std::vector<float>& factors) {
    if (!InterpolateFramesMultiAsync(previous, current, outputs, factors)) return false;
    return HipContext::Get().WaitIdle();
}

bool FrameManager::BroadcastFrame(Frame& frame, int root) {
    lfg_frame f = frame.AsAbi();
    if (lfg_broadcast_frame(Ctx(), &f, root) != LFG_OK) {
        LOG_ERROR("Failed to broadcast frame: ", lfg_last_error(Ctx()));
        return false;
    }
    return true;
}

bool FrameManager::WaitBroadcasts() {
    if (lfg_comm_wait(Ctx()) != LFG_OK) {
        LOG_ERROR("Failed to wait for broadcasts: ", lfg_last_error(Ctx()));
        return false;
    }
    return true;
}

bool FrameManager::CreateStagingBuffer(void*& buffer, size_t size) {
    if (lfg_staging_create(Ctx(), size, &buffer) != LFG_OK) {
        LOG_ERROR("Failed to create staging buffer: ", lfg_last_error(Ctx()));
        return false;
    }
    return true;
}

void FrameManager::DestroyStagingBuffer(void* buffer) { lfg_staging_destroy(Ctx(), buffer); }

bool FrameManager::UploadFrame(Frame& frame, const void* host, size_t size) {
    lfg_frame f = frame.AsAbi();
    if (lfg_frame_upload(Ctx(), &f, host, size) != LFG_OK) {
        LOG_ERROR(lfg_last_error(Ctx()));                     // "Captured image size (..) smaller than expected (..)"
        return false;
    }
    return true;
}

bool FrameManager::DownloadFrame(const Frame& frame, void* host, size_t size) {
    const lfg_frame f = frame.AsAbi();
    if (lfg_frame_download(Ctx(), &f, host, size) != LFG_OK) {
        LOG_ERROR("Failed to read back frame: ", lfg_last_error(Ctx()));
        return false;
    }
    return true;
}
