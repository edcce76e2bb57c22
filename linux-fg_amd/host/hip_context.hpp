// HipContext -- takes the place of the reference's VulkanContext (src/vulkan_context.hpp:9-60):
// the process-wide device context every other class reaches through Get().  Instead of a Vulkan
// instance / physical device / logical device / compute queue it owns one lfg_context: a HIP device
// ordinal and one stream (the "compute queue").  It calls only the C-ABI in include/linuxfg_hip.h.
#pragma once
#include <cstdint>

#include "linuxfg_hip.h"
#include "logger.hpp"

class HipContext {
public:
    static HipContext& Get() {
        static HipContext instance;
        return instance;
    }

    // VulkanContext::Initialize() (src/vulkan_context.cpp:3-23).  device < 0: first device, as the
    // reference falls back to physical device 0 (:88-105).
    bool Initialize(int device = -1);
    void Cleanup();

    lfg_context* GetDevice() const { return m_ctx; }          // VulkanContext::GetDevice
    void* GetComputeQueue() const;                            // the HIP stream (VulkanContext::GetComputeQueue)
    int GetDeviceOrdinal() const;
    bool WaitIdle();                                          // vkQueueWaitIdle

private:
    HipContext() = default;
    ~HipContext() { Cleanup(); }
    HipContext(const HipContext&) = delete;
    HipContext& operator=(const HipContext&) = delete;

    lfg_context* m_ctx = nullptr;
};
