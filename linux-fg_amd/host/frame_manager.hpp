// Frame / FrameManager -- the reference's src/frame_manager.hpp:9-93 re-founded on the C-ABI.
// Same class and method names, same bool + LOG_ERROR error convention, same ownership rules
// (Frame is a caller-owned POD created and destroyed through FrameManager; DestroyFrame is
// idempotent).  VkImage/VkDeviceMemory/VkImageView become one device pointer with a row pitch.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "hip_context.hpp"

// struct Frame (src/frame_manager.hpp:9-16).  `format` defaults to RGBA8, as VK_FORMAT_R8G8B8A8_UNORM there.
struct Frame {
    void* data = nullptr;             // device memory (VkImage + VkDeviceMemory + VkImageView)
    uint32_t width = 0;
    uint32_t height = 0;
    uint32_t pitch = 0;               // bytes per row
    uint32_t format = LFG_FORMAT_RGBA8_UNORM;
    bool owned = false;

    lfg_frame AsAbi() const {
        lfg_frame f{};
        f.data = data; f.width = width; f.height = height; f.pitch = pitch; f.format = format; f.owned = owned ? 1u : 0u;
        return f;
    }
};

// Parameter blocks of the two stages InterpolateFrames dispatches.  The reference's host structs
// (src/frame_manager.hpp:18-27) are kept for their VALUES; InterpolatePushConstants there does not
// match the shader's std430 block, so nothing depends on its layout here.
struct MotionPushConstants {
    int32_t imageSize[2];
    int32_t blockSize;
    float searchRadius;
};
struct InterpolatePushConstants {
    float interpolationFactor;
    int32_t imageSize[2];
};

class FrameManager {
public:
    static FrameManager& Get() {
        static FrameManager instance;
        return instance;
    }

    bool Initialize(uint32_t width, uint32_t height);         // src/frame_manager.cpp:3-11
    void Cleanup();

    // Frame management (src/frame_manager.cpp:30-145)
    bool CreateFrame(Frame& frame, uint32_t width, uint32_t height);
    void DestroyFrame(Frame& frame);
    bool CopyFrameData(const Frame& source, Frame& destination);

    // Frame interpolation (src/frame_manager.cpp:216-372): motion (blockSize 8, searchRadius 16) then
    // interpolate; the motion-vector image is a temporary owned by the device context.
    bool InterpolateFrames(const Frame& previous, const Frame& current, Frame& output, float factor);
    // The same two stages enqueued without the wait (the Scaler pipelines presentation behind them).
    bool InterpolateFramesAsync(const Frame& previous, const Frame& current, Frame& output, float factor);

    // Several generated frames per pair (the 60 -> 240 fps cadence: factors 1/4, 1/2, 3/4): motion once, then one pass
    // that writes outputs[i] for factors[i].  Extension of InterpolateFrames, which takes a single factor.
    bool InterpolateFramesMulti(const Frame& previous, const Frame& current, const std::vector<Frame*>& outputs,
                                const std::vector<float>& factors);
    bool InterpolateFramesMultiAsync(const Frame& previous, const Frame& current, const std::vector<Frame*>& outputs,
                                     const std::vector<float>& factors);

    // The one exchange of the multi-GPU path: broadcast a frame from rank `root` to every rank (asynchronous, on the
    // device context's communication stream) and make later work on the compute queue wait for it.
    bool BroadcastFrame(Frame& frame, int root);
    bool WaitBroadcasts();

    // Buffer management (src/frame_manager.cpp:199-214): pinned host memory instead of a
    // host-visible VkBuffer.
    bool CreateStagingBuffer(void*& buffer, size_t size);
    void DestroyStagingBuffer(void* buffer);

    // The copies the reference records by hand around its staging buffers
    // (src/window_capture.cpp:472-568 upload, src/scaler.cpp:479-536 readback).  Asynchronous on the
    // compute queue when `host` is pinned; WaitIdle() before touching the bytes.
    bool UploadFrame(Frame& frame, const void* host, size_t size);
    bool DownloadFrame(const Frame& frame, void* host, size_t size);

    // Parameters InterpolateFrames pushes (src/frame_manager.cpp:332-333).
    static constexpr int kBlockSize = 8;
    static constexpr float kSearchRadius = 16.0f;

private:
    FrameManager() = default;
    ~FrameManager() { Cleanup(); }
    FrameManager(const FrameManager&) = delete;
    FrameManager& operator=(const FrameManager&) = delete;
    FrameManager(FrameManager&&) = delete;
    FrameManager& operator=(FrameManager&&) = delete;

    bool m_initialized = false;
};
