#include "hip_context.hpp"

#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
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

// The id file: 8 bytes of magic, the launcher's 8-byte run nonce, then the 128-byte id.  A file left behind by an earlier run
// (same name, rank 0 of THIS run not there yet) must not be taken for this run's: rank 0 removes the name before it makes
// the id and again once every rank has joined, and a receiver only accepts a file that carries its own nonce
// (--comm-nonce / LFG_COMM_NONCE: the launcher gives every rank of one run the same number).  A rank that joined with
// a wrong id would sit in ncclCommInitRank for ever, holding its GPU: a watchdog ends the process instead.
namespace {
constexpr char kIdMagic[8] = {'L', 'F', 'G', 'C', 'O', 'M', 'M', '1'};
struct IdFile { char magic[8]; uint64_t nonce; char id[LFG_COMM_ID_BYTES]; };
}  // namespace

bool HipContext::InitComm(int ranks, int rank, const std::string& idFile, uint64_t nonce, int joinTimeoutSeconds) {
    if (!m_ctx) {
        LOG_ERROR("HipContext::InitComm: context not initialized");
        return false;
    }
    // (several ranks need a nonce of their run: with the default 0 a rank that opens an earlier run's nonce-0 file before rank 0
    //  has removed it would join a communicator that no longer exists and sit in ncclCommInitRank until the watchdog)
    if (ranks > 1 && nonce == 0) {
        LOG_ERROR("HipContext::InitComm: ", ranks, " ranks need a non-zero run nonce (--comm-nonce / LFG_COMM_NONCE: the same number for every rank of one run)");
        return false;
    }
    lfg_comm_id id{};
    if (rank == 0) {
        (void)remove(idFile.c_str());                                  // a stale id of an earlier run
        if (lfg_comm_unique_id(&id) != LFG_OK) {
            LOG_ERROR("Failed to create a communicator id (is librccl.so available?)");
            return false;
        }
        IdFile rec{};
        memcpy(rec.magic, kIdMagic, sizeof rec.magic);
        rec.nonce = nonce;
        memcpy(rec.id, id.bytes, sizeof rec.id);
        const std::string tmp = idFile + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(&rec, 1, sizeof rec, f) != sizeof rec) {
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
        bool got = false, foreign = false;
        for (int attempt = 0; attempt < 6000 && !got; ++attempt) {      // up to a minute
            if (FILE* f = fopen(idFile.c_str(), "rb")) {
                IdFile rec{};
                if (fread(&rec, 1, sizeof rec, f) == sizeof rec && memcmp(rec.magic, kIdMagic, sizeof rec.magic) == 0) {
                    if (rec.nonce == nonce) { memcpy(id.bytes, rec.id, sizeof rec.id); got = true; }
                    else foreign = true;                               // another run's file: rank 0 of this run will replace it
                }
                fclose(f);
            }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (!got) {
            LOG_ERROR(foreign ? "The communicator id file belongs to another run (nonce mismatch): " : "Timed out waiting for the communicator id in ", idFile);
            return false;
        }
    }
    // ncclCommInitRank has no timeout of its own
    std::mutex mu;
    std::condition_variable cv;
    bool joined = false;
    std::thread watchdog([&] {
        std::unique_lock<std::mutex> lock(mu);
        if (!cv.wait_for(lock, std::chrono::seconds(joinTimeoutSeconds), [&] { return joined; })) {
            fprintf(stderr, "lfg: rank %d of %d still waits for the other ranks after %d s (stale or foreign communicator id in %s?): giving up\n",
                    rank, ranks, joinTimeoutSeconds, idFile.c_str());
            fflush(stderr);
            _exit(3);
        }
    });
    const int rc = lfg_comm_init(m_ctx, ranks, rank, &id);
    { std::lock_guard<std::mutex> lock(mu); joined = true; }
    cv.notify_one();
    watchdog.join();
    if (rank == 0) (void)remove(idFile.c_str());                       // every rank has read it: nothing to go stale
    if (rc != LFG_OK) {
        LOG_ERROR("Failed to join the communicator: ", lfg_last_error(m_ctx));
        return false;
    }
    LOG_INFO("Rank ", rank, " of ", ranks, " joined the communicator");
    return true;
}
