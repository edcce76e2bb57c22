// FrameSource -- where ProcessFrame gets its input.  The reference captures an X11 window
// (WindowCapture::CaptureFrame, src/window_capture.cpp:232-470, out of scope on a headless GPU box);
// the only part of it on the hot path is the hand-off of a tightly packed 4-byte-per-pixel host
// image to a device Frame (CopyToStagingBuffer, :472-568), which is what CaptureFrame() does here
// through a pinned-host ring instead of a staging buffer allocated and freed per frame.
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "frame_manager.hpp"

class FrameSource {
public:
    virtual ~FrameSource() = default;
    virtual bool Initialize(uint32_t width, uint32_t height) = 0;
    // Fill `host` (width*height*4 bytes, RGBA8, tightly packed) with the next frame.
    virtual bool NextFrame(uint8_t* host) = 0;
    virtual bool GetSize(uint32_t& width, uint32_t& height) const = 0;   // WindowCapture::GetWindowSize
};

// Synthetic content, bit-identical to linux-fg_amd/synth.py: frame 0 is gradient + hashed noise,
// frame k is frame k-1 translated by `shift` with the exposed border filled from a second noise
// stream (seed + k).  Counter-based "lowbias32" hash, so no generator state has to match.
class SyntheticCapture : public FrameSource {
public:
    static constexpr uint32_t kBaseSeed = 0x5EED0000u;
    explicit SyntheticCapture(uint32_t stream = 0, int shiftX = 3, int shiftY = -2)
        : m_seed(kBaseSeed + stream), m_shiftX(shiftX), m_shiftY(shiftY) {}

    bool Initialize(uint32_t width, uint32_t height) override;
    bool NextFrame(uint8_t* host) override;
    bool GetSize(uint32_t& width, uint32_t& height) const override { width = m_width; height = m_height; return m_width != 0; }

    static uint32_t LowBias32(uint32_t x);
    static void NoiseBytes(uint8_t* out, uint32_t width, uint32_t height, uint32_t seed);
    static void MakePrev(uint8_t* out, uint32_t width, uint32_t height, uint32_t seed);
    static void Translate(const uint8_t* prev, uint8_t* out, uint32_t width, uint32_t height, int tx, int ty, uint32_t seed);

private:
    uint32_t m_seed, m_width = 0, m_height = 0, m_index = 0;
    int m_shiftX, m_shiftY;
    std::vector<uint8_t> m_last;
};

// A source played back from memory: the first `frames` frames of another source are produced ONCE, in Initialize, and handed
// out in rotation afterwards (one memcpy into the pinned staging slot per frame -- what CopyToStagingBuffer does with a
// captured image, src/window_capture.cpp:472-568).  For measurements in which the host's frame synthesis (5 ms per 1080p
// frame on one core) must not be what bounds the loop: bench.py's PCIe-inclusive figure.
class ReplayCapture : public FrameSource {
public:
    ReplayCapture(std::unique_ptr<FrameSource> inner, uint32_t frames) : m_inner(std::move(inner)), m_count(frames ? frames : 1) {}

    bool Initialize(uint32_t width, uint32_t height) override;
    bool NextFrame(uint8_t* host) override;
    bool GetSize(uint32_t& width, uint32_t& height) const override { return m_inner->GetSize(width, height); }

private:
    std::unique_ptr<FrameSource> m_inner;
    uint32_t m_count, m_next = 0;
    size_t m_bytes = 0;
    std::vector<uint8_t> m_frames;
};

// Raw RGBA8 frames (tightly packed, back to back) from a file or a pipe ("-" = stdin): the headless stand-in
// for a capture device.  NextFrame fails at end of input.
class RawFileCapture : public FrameSource {
public:
    explicit RawFileCapture(std::string path) : m_path(std::move(path)) {}
    ~RawFileCapture() override { if (m_file && m_file != stdin) fclose(m_file); }

    bool Initialize(uint32_t width, uint32_t height) override;
    bool NextFrame(uint8_t* host) override;
    bool GetSize(uint32_t& width, uint32_t& height) const override { width = m_width; height = m_height; return m_width != 0; }

private:
    std::string m_path;
    FILE* m_file = nullptr;
    uint32_t m_width = 0, m_height = 0;
};
