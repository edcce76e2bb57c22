#include "hip_context.hpp"

#include <chrono>
#include <cstdio>
#include <thread>

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

bool HipContext::InitComm(int ranks, int rank, const std::string& idFile) {
    if (!m_ctx) {
        LOG_ERROR("HipContext::InitComm: context not initialized");
        return false;
    }
    lfg_comm_id id{};
    if (rank == 0) {
        if (lfg_comm_unique_id(&id) != LFG_OK) {
            LOG_ERROR("Failed to create a communicator id (is librccl.so available?)");
            return false;
        }
        const std::string tmp = idFile + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id.bytes, 1, sizeof id.bytes, f) != sizeof id.bytes) {
            LOG_ERROR("Cannot write communicator id to ", tmp);
            if (f) fclose(f);
            return false;
        }
        fclose(f);
        if (rename(tmp.c_str(), idFile.c_str()) != 0) {
            LOG_ERROR("Cannot publish communicator id as ", idFile);
            return false;
        }
    } else {
        bool got = false;
        for (int attempt = 0; attempt < 6000 && !got; ++attempt) {      // up to a minute
            if (FILE* f = fopen(idFile.c_str(), "rb")) {
                got = fread(id.bytes, 1, sizeof id.bytes, f) == sizeof id.bytes;
                fclose(f);
            }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (!got) {
            LOG_ERROR("Timed out waiting for the communicator id in ", idFile);
            return false;
        }
    }
    if (lfg_comm_init(m_ctx, ranks, rank, &id) != LFG_OK) {
        LOG_ERROR("Failed to join the communicator: ", lfg_last_error(m_ctx));
        return false;
    }
    LOG_INFO("Rank ", rank, " of ", ranks, " joined the communicator");
    return true;
}
