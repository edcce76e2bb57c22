// lfg_host -- headless counterpart of the reference's CLI (src/main.cpp:21-144): same option names
// and the same derivation of the output size; the X11 window id is replaced by a synthetic stream
// index and the paced SDL loop by a fixed number of frames run flat out.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "scaler.hpp"

static void PrintUsage() {
    std::cout << "Usage: lfg_host [options] [stream-index]\n"
              << "Options:\n"
              << "  --help                   Show this help message\n"
              << "  --input-width WIDTH      Input width (default: 1920)\n"
              << "  --input-height HEIGHT    Input height (default: 1080)\n"
              << "  --output-width WIDTH     Output width\n"
              << "  --output-height HEIGHT   Output height\n"
              << "  --target-fps FPS         Accepted for compatibility; frames are not paced\n"
              << "  --no-interpolation       Disable frame interpolation\n"
              << "  --interpolation-factor F Interpolation blend factor (0.0-1.0, default: 0.5)\n"
              << "  --factors A,B,...        Several generated frames per pair, in presentation order (e.g. 0.25,0.5,0.75\n"
              << "                           for 60 -> 240 fps); motion runs once per pair\n"
              << "  --in-flight N            Frames in flight on the GPU, 1..3 (default 2): a frame's upscale and first motion units\n"
              << "                           run while the previous frame's last long units finish; same frames, same order\n"
              << "  --ranks N --rank R --comm-file FILE\n"
              << "                           One process per GPU (BASELINE config 4): the batch shares its previous frame\n"
              << "                           (synthetic stream 0, captured on rank 0, broadcast over RCCL per frame); each\n"
              << "                           rank's current frames are its own stream.  FILE carries the communicator id.\n"
              << "  --comm-nonce N           A number the launcher gives every rank of ONE run (default: LFG_COMM_NONCE, else 0; more than one rank needs a non-zero one): an\n"
              << "                           id file left behind by another run is then never joined\n"
              << "  --frames N               Number of input frames to process (default: 10)\n"
              << "  --device N               HIP device ordinal (default: 0)\n"
              << "  --dump-dir DIR           Write every presented frame to DIR as raw RGBA8\n"
              << "  --input-raw FILE         Read input frames (raw RGBA8, tightly packed, back to back) from FILE, '-' = stdin\n"
              << "  --output-raw FILE        Append every presented frame (raw RGBA8) to FILE, '-' = stdout\n"
              << "  --replay N               Produce the source's first N frames once, before the clock starts, and play them back in rotation\n"
              << "                           (the host's frame synthesis, 5 ms per 1080p frame, then does not bound the loop)\n"
              << "  --present-null           The presenter looks at no pixel (default: a strided checksum over every presented frame)\n"
              << "  --sync-present           Wait for each call's own frames (reference behaviour) instead of pipelining\n"
              << "  --quiet                  Only warnings and errors\n";
}

int main(int argc, char* argv[]) {
    // (HIP's default of four hardware queues is one short of three lanes + a copy stream + a communication stream: streams that
    //  share a queue run in turn.  Before the first HIP call; an explicit setting wins.  INTEGRATION.md, "Hardware queues".)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    ScalerConfig config;
    config.enableInterpolation = true;
    config.interpolationFactor = 0.5f;
    config.targetFps = 60;
    uint32_t stream = 0;
    int frames = 10, device = 0;
    std::string dumpDir, inputRaw, outputRaw, commFile;
    int ranks = 0, rank = 0, inFlight = 2;
    unsigned long long commNonce = getenv("LFG_COMM_NONCE") ? strtoull(getenv("LFG_COMM_NONCE"), nullptr, 0) : 0ull;
    std::vector<float> factors;
    bool syncPresent = false, presentNull = false;
    int replay = 0;

    for (int i = 1; i < argc; i++) {
        if (strcmp(argv[i], "--help") == 0) { PrintUsage(); return 0; }
        else if (strcmp(argv[i], "--input-width") == 0 && i + 1 < argc) config.inputWidth = std::atoi(argv[++i]);
        else if (strcmp(argv[i], "--input-height") == 0 && i + 1 < argc) config.inputHeight = std::atoi(argv[++i]);
        else if (strcmp(argv[i], "--output-width") == 0 && i + 1 < argc) config.outputWidth = std::atoi(argv[++i]);
        else if (strcmp(argv[i], "--output-height") == 0 && i + 1 < argc) config.outputHeight = std::atoi(argv[++i]);
        else if (strcmp(argv[i], "--target-fps") == 0 && i + 1 < argc) config.targetFps = std::atoi(argv[++i]);
        else if (strcmp(argv[i], "--no-interpolation") == 0) config.enableInterpolation = false;
        else if (strcmp(argv[i], "--interpolation-factor") == 0 && i + 1 < argc) config.interpolationFactor = (float)std::atof(argv[++i]);
        else if (strcmp(argv[i], "--factors") == 0 && i + 1 < argc) {
            factors.clear();
            for (const char* p = argv[++i]; *p;) {
                char* end = nullptr;
                factors.push_back(std::strtof(p, &end));
                if (end == p) { LOG_ERROR("Invalid --factors list"); return 1; }
                p = *end == ',' ? end + 1 : end;
            }
        }
        else if (strcmp(argv[i], "--in-flight") == 0 && i + 1 < argc) inFlight = std::atoi(argv[++i]);
        else if (strcmp(argv[i], "--ranks") == 0 && i + 1 < argc) ranks = std::atoi(argv[++i]);
        else if (strcmp(argv[i], "--rank") == 0 && i + 1 < argc) rank = std::atoi(argv[++i]);
        else if (strcmp(argv[i], "--comm-file") == 0 && i + 1 < argc) commFile = argv[++i];
        else if (strcmp(argv[i], "--comm-nonce") == 0 && i + 1 < argc) commNonce = strtoull(argv[++i], nullptr, 0);
        else if (strcmp(argv[i], "--frames") == 0 && i + 1 < argc) frames = std::atoi(argv[++i]);
        else if (strcmp(argv[i], "--device") == 0 && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (strcmp(argv[i], "--dump-dir") == 0 && i + 1 < argc) dumpDir = argv[++i];
        else if (strcmp(argv[i], "--input-raw") == 0 && i + 1 < argc) inputRaw = argv[++i];
        else if (strcmp(argv[i], "--output-raw") == 0 && i + 1 < argc) outputRaw = argv[++i];
        else if (strcmp(argv[i], "--sync-present") == 0) syncPresent = true;
        else if (strcmp(argv[i], "--replay") == 0 && i + 1 < argc) replay = std::atoi(argv[++i]);
        else if (strcmp(argv[i], "--present-null") == 0) presentNull = true;
        else if (strcmp(argv[i], "--quiet") == 0) Logger::Get().SetMinLevel(Logger::Level::WARNING);
        else {
            char* endPtr;
            stream = (uint32_t)std::strtoul(argv[i], &endPtr, 0);
            if (*endPtr != '\0') { LOG_ERROR("Invalid stream index"); return 1; }
        }
    }
    if (config.inputWidth == 0) config.inputWidth = 1920;
    if (config.inputHeight == 0) config.inputHeight = 1080;
    // Output size: as src/main.cpp:76-90 (keep the aspect ratio when only one side is given).
    if (config.outputWidth == 0 || config.outputHeight == 0) {
        if (config.outputHeight != 0) {
            float scale = (float)config.outputHeight / config.inputHeight;
            config.outputWidth = static_cast<uint32_t>(config.inputWidth * scale);
        } else if (config.outputWidth != 0) {
            float scale = (float)config.outputWidth / config.inputWidth;
            config.outputHeight = static_cast<uint32_t>(config.inputHeight * scale);
        } else {
            config.outputWidth = config.inputWidth;
            config.outputHeight = config.inputHeight;
        }
    }

    if (!HipContext::Get().Initialize(device)) { LOG_ERROR("Failed to initialize HIP"); return 1; }
    if (!FrameManager::Get().Initialize(config.outputWidth, config.outputHeight)) {
        LOG_ERROR("Failed to initialize frame manager");
        HipContext::Get().Cleanup();
        return 1;
    }
    if (ranks > 0) {
        if (commFile.empty() || rank < 0 || rank >= ranks) { LOG_ERROR("--ranks needs --rank in range and --comm-file"); return 1; }
        if (!HipContext::Get().InitComm(ranks, rank, commFile, (uint64_t)commNonce)) { HipContext::Get().Cleanup(); return 1; }
        Scaler::Get().SetSharedPreviousSource(std::make_unique<SyntheticCapture>(0));   // the batch's previous frames: stream 0
        if (stream == 0) stream = (uint32_t)rank + 1;                                   // a rank's own current frames
    }
    {
        std::unique_ptr<FrameSource> source;
        if (!inputRaw.empty()) source = std::make_unique<RawFileCapture>(inputRaw);
        else source = std::make_unique<SyntheticCapture>(stream);
        if (replay > 0) source = std::make_unique<ReplayCapture>(std::move(source), (uint32_t)replay);
        Scaler::Get().SetFrameSource(std::move(source));
    }
    Scaler::Get().SetPipelined(!syncPresent);
    Scaler::Get().SetFramesInFlight(inFlight);
    if (!factors.empty()) Scaler::Get().SetInterpolationFactors(factors);
    FILE* rawOut = nullptr;
    if (!outputRaw.empty()) {
        rawOut = outputRaw == "-" ? stdout : fopen(outputRaw.c_str(), "wb");
        if (!rawOut) { LOG_ERROR("Cannot open ", outputRaw); return 1; }
    }
    FILE* report = rawOut == stdout ? stderr : stdout;          // keep the pixel stream clean
    uint64_t checksum = 0, presented = 0, generated = 0;
    Scaler::Get().SetPresenter([&](const uint8_t* rgba, uint32_t w, uint32_t h, bool interpolated) {
        const size_t n = (size_t)w * h * 4;
        uint64_t s = 0;
        if (!presentNull) for (size_t i = 0; i < n; i += 64) s += rgba[i];
        checksum = checksum * 1315423911ull + s;
        if (!dumpDir.empty()) {
            char name[512];
            snprintf(name, sizeof name, "%s/frame_%04llu_%s_%ux%u.rgba", dumpDir.c_str(), (unsigned long long)presented,
                     interpolated ? "interp" : "real", w, h);
            if (FILE* f = fopen(name, "wb")) { fwrite(rgba, 1, n, f); fclose(f); }
        }
        if (rawOut) fwrite(rgba, 1, n, rawOut);
        ++presented;
        generated += interpolated ? 1 : 0;
    });
    if (!Scaler::Get().Initialize(config)) {
        LOG_ERROR("Failed to initialize scaler");
        FrameManager::Get().Cleanup();
        HipContext::Get().Cleanup();
        return 1;
    }

    const auto t0 = std::chrono::steady_clock::now();
    bool ok = true;
    for (int i = 0; i < frames && ok; ++i) ok = Scaler::Get().ProcessFrame();
    ok = ok && Scaler::Get().Flush();                           // the last call's frames
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    // Teardown order as src/main.cpp:138-141.
    Scaler::Get().Cleanup();
    FrameManager::Get().Cleanup();
    HipContext::Get().Cleanup();
    if (!ok) { LOG_ERROR("ProcessFrame failed: ", Logger::Get().GetLastError()); return 1; }
    if (rawOut && rawOut != stdout) fclose(rawOut);
    fprintf(report, "{\"input_frames\": %d, \"presented\": %llu, \"interpolated\": %llu, \"seconds\": %.4f, "
           "\"presented_fps\": %.2f, \"checksum\": %llu, \"pipelined\": %s, \"replay\": %d, \"present_null\": %s, \"in_flight\": %d, "
           "\"note\": \"includes %s, PCIe upload and readback\"}\n",
           frames, (unsigned long long)presented, (unsigned long long)generated, sec, presented / sec,
           (unsigned long long)checksum, syncPresent ? "false" : "true", replay, presentNull ? "true" : "false", inFlight,
           replay > 0 ? "one memcpy per input frame into the staging slot" : "host frame synthesis");
    return 0;
}
