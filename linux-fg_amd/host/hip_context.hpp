// HipContext -- takes the place of the reference's VulkanContext (src/vulkan_context.hpp:9-60):
// the process-wide device context every other class reaches through Get().  Instead of a Vulkan
// instance / physical device / logical device / compute queue it owns one lfg_context: a HIP device
// ordinal and one stream (the "compute queue").  It calls only the C-ABI in include/linuxfg_hip.h.
#pragma once
#include <cstdint>
#include <string>

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

    // Several GPUs, one process each (no reference counterpart: one queue, no communication,
    // src/vulkan_context.cpp:130-151).  Rank 0 removes `idFile`, makes the communicator id and leaves it there (magic,
    // `nonce`, id; written under a temporary name, then renamed); the other ranks wait for a file that carries the same
    // nonce -- the launcher hands every rank of one run the same number, so an id left by an earlier run is never joined.
    // Collective: returns when all ranks have joined, and rank 0 then removes the file.  A join that does not complete
    // within `joinTimeoutSeconds` ends the process with exit code 3 (ncclCommInitRank itself never gives up).
    bool InitComm(int ranks, int rank, const std::string& idFile, uint64_t nonce = 0, int joinTimeoutSeconds = 120);
    int GetRank() const { return m_ctx ? lfg_comm_rank(m_ctx) : -1; }
    int GetRanks() const { return m_ctx ? lfg_comm_ranks(m_ctx) : 0; }

private:
    HipContext() = default;
    ~HipContext() { Cleanup(); }
    HipContext(const HipContext&) = delete;
    HipContext& operator=(const HipContext&) = delete;

    lfg_context* m_ctx = nullptr;
};
