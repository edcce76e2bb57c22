// Test helper (CPU only): writes N consecutive SyntheticCapture frames to stdout as raw RGBA8 so that
// tests/test_host_logic.py can compare them with linux-fg_amd/synth.py.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "frame_source.hpp"

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const uint32_t w = (uint32_t)atoi(argv[1]), h = (uint32_t)atoi(argv[2]), stream = (uint32_t)atoi(argv[3]);
    const int n = atoi(argv[4]);
    SyntheticCapture cap(stream);
    if (!cap.Initialize(w, h)) return 1;
    std::vector<uint8_t> buf((size_t)w * h * 4);
    for (int i = 0; i < n; ++i) {
        if (!cap.NextFrame(buf.data())) return 1;
        fwrite(buf.data(), 1, buf.size(), stdout);
    }
    return 0;
}
