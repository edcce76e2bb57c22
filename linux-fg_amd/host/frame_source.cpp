#include "frame_source.hpp"

#include <cstring>

uint32_t SyntheticCapture::LowBias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

void SyntheticCapture::NoiseBytes(uint8_t* out, uint32_t width, uint32_t height, uint32_t seed) {
    const uint32_t base = LowBias32(seed);
    const uint32_t n = width * height;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t h = LowBias32(i + base);
        out[4 * i + 0] = (uint8_t)(h & 0xFF); out[4 * i + 1] = (uint8_t)((h >> 8) & 0xFF);
        out[4 * i + 2] = (uint8_t)((h >> 16) & 0xFF); out[4 * i + 3] = (uint8_t)(h >> 24);
    }
}

void SyntheticCapture::MakePrev(uint8_t* out, uint32_t width, uint32_t height, uint32_t seed) {
    NoiseBytes(out, width, height, seed);
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x)
            for (uint32_t c = 0; c < 4; ++c) {
                uint8_t& b = out[(size_t)(y * width + x) * 4 + c];
                b = (uint8_t)((((x * (c + 1) + 2 * y) >> 3) + (b & 0x7Fu)) & 0xFFu);
            }
}

void SyntheticCapture::Translate(const uint8_t* prev, uint8_t* out, uint32_t width, uint32_t height,
                                 int tx, int ty, uint32_t seed) {
    NoiseBytes(out, width, height, seed ^ 0xA5A5A5A5u);
    const int W = (int)width, H = (int)height;
    for (int y = 0; y < H; ++y) {
        const int sy = y - ty;
        if (sy < 0 || sy >= H) continue;
        for (int x = 0; x < W; ++x) {
            const int sx = x - tx;
            if (sx < 0 || sx >= W) continue;
            std::memcpy(out + ((size_t)y * W + x) * 4, prev + ((size_t)sy * W + sx) * 4, 4);
        }
    }
}

bool SyntheticCapture::Initialize(uint32_t width, uint32_t height) {
    if (width == 0 || height == 0) {
        LOG_ERROR("SyntheticCapture: zero frame size");
        return false;
    }
    m_width = width; m_height = height; m_index = 0;
    m_last.assign((size_t)width * height * 4, 0);
    return true;
}

bool SyntheticCapture::NextFrame(uint8_t* host) {
    if (m_last.empty()) return false;
    if (m_index == 0) MakePrev(host, m_width, m_height, m_seed);
    else Translate(m_last.data(), host, m_width, m_height, m_shiftX, m_shiftY, m_seed + m_index);
    std::memcpy(m_last.data(), host, m_last.size());
    ++m_index;
    return true;
}

bool RawFileCapture::Initialize(uint32_t width, uint32_t height) {
    if (width == 0 || height == 0) {
        LOG_ERROR("RawFileCapture: zero frame size");
        return false;
    }
    m_width = width; m_height = height;
    if (!m_file) m_file = m_path == "-" ? stdin : fopen(m_path.c_str(), "rb");
    if (!m_file) {
        LOG_ERROR("RawFileCapture: cannot open ", m_path);
        return false;
    }
    return true;
}

bool RawFileCapture::NextFrame(uint8_t* host) {
    const size_t need = (size_t)m_width * m_height * 4;
    size_t got = 0;
    while (m_file && got < need) {                       // pipes deliver short reads
        const size_t n = fread(host + got, 1, need - got, m_file);
        if (n == 0) break;
        got += n;
    }
    if (got != need) {
        LOG_ERROR("RawFileCapture: end of input (", got, " of ", need, " bytes of the next frame)");
        return false;
    }
    return true;
}

bool ReplayCapture::Initialize(uint32_t width, uint32_t height) {
    if (!m_inner || !m_inner->Initialize(width, height)) return false;
    m_bytes = (size_t)width * height * 4;
    m_frames.resize(m_bytes * m_count);
    for (uint32_t i = 0; i < m_count; ++i)
        if (!m_inner->NextFrame(m_frames.data() + (size_t)i * m_bytes)) { LOG_ERROR("ReplayCapture: the source ended after ", i, " frames"); return false; }
    m_next = 0;
    return true;
}

bool ReplayCapture::NextFrame(uint8_t* host) {
    if (m_frames.empty()) return false;
    std::memcpy(host, m_frames.data() + (size_t)m_next * m_bytes, m_bytes);
    m_next = (m_next + 1) % m_count;
    return true;
}
