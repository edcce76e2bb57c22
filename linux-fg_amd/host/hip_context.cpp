#include "hip_context.hpp"

bool HipContext::Initialize(int device) {
    if (m_ctx) return true;
    if (lfg_context_create(device, &m_ctx) != LFG_OK) {
        LOG_ERROR("Failed to create HIP context: ", lfg_last_error(nullptr));
        m_ctx = nullptr;
        return false;
    }
    LOG_INFO("HIP context initialized on device ", lfg_context_device(m_ctx));
    return true;
}

void HipContext::Cleanup() {
    if (m_ctx) {
        lfg_context_destroy(m_ctx);
        m_ctx = nullptr;
    }
}

void* HipContext::GetComputeQueue() const { return m_ctx ? lfg_context_get_stream(m_ctx) : nullptr; }

int HipContext::GetDeviceOrdinal() const { return m_ctx ? lfg_context_device(m_ctx) : -1; }

bool HipContext::WaitIdle() {
    if (!m_ctx) return false;
    if (lfg_sync(m_ctx) != LFG_OK) {
        LOG_ERROR("Device wait failed: ", lfg_last_error(m_ctx));
        return false;
    }
    return true;
}
