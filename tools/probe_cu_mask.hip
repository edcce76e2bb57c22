// Which CUs does a stream created with hipExtStreamCreateWithCUMask use on an MI355X (8 XCDs x 32 CUs)?
// Every workgroup records (XCC_ID, HW_ID.se_id, HW_ID.cu_id); the host prints the set a mask leaves.
//   hipcc --offload-arch=gfx950 -O2 -o build_variants/probe_cu_mask tools/probe_cu_mask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#include <tuple>
#include <map>

__global__ void where_kernel(uint32_t *out, int spin) {
    const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID[3:0]
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
}

static void run(const char *name, const std::vector<uint32_t> &mask, bool masked) {
    hipStream_t s;
    hipError_t e = masked ? hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data())
                          : hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
    const int groups = 4096;
    uint32_t *d; hipMalloc(&d, groups * 8); hipMemsetAsync(d, 0xff, groups * 8, s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s);
    where_kernel<<<groups, 256, 0, s>>>(d, 2000);                    // 2000 ticks of 100 MHz = 20 us per workgroup
    hipEventRecord(b, s);
    e = hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<uint32_t> h(groups * 2); hipMemcpy(h.data(), d, groups * 8, hipMemcpyDeviceToHost);
    std::set<std::tuple<int, int, int>> cus; std::map<int, std::set<std::pair<int, int>>> perx;
    for (int i = 0; i < groups; ++i) {
        const uint32_t hw = h[i * 2]; const int xcc = h[i * 2 + 1] & 15;
        const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus.insert({xcc, se * 2 + sh, cu}); perx[xcc].insert({se * 2 + sh, cu});
    }
    printf("%-28s %s  %.3f ms  distinct CUs %zu :", name, hipGetErrorString(e), ms, cus.size());
    for (auto &p : perx) printf(" x%d=%zu", p.first, p.second.size());
    printf("\n");
    if (cus.size() <= 24 || (masked && cus.size() >= 200 && cus.size() < 256)) {
        // print the CUs used (few) or the CUs missing (many used)
        std::set<std::tuple<int, int, int>> all;
        static std::set<std::tuple<int, int, int>> full;
        if (!masked) full = cus;
        if (cus.size() <= 24) { printf("   used:"); for (auto &c : cus) printf(" (x%d se%d cu%d)", std::get<0>(c), std::get<1>(c), std::get<2>(c)); printf("\n"); }
        else { printf("   missing:"); for (auto &c : full) if (!cus.count(c)) printf(" (x%d se%d cu%d)", std::get<0>(c), std::get<1>(c), std::get<2>(c)); printf("\n"); }
    }
    if (!masked) { static bool once = false; if (!once) { once = true; } }
    hipFree(d); hipStreamDestroy(s);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    const int words = (p.multiProcessorCount + 31) / 32;
    std::vector<uint32_t> full(words, 0xffffffffu);
    run("unmasked stream", full, false);
    run("mask: all bits", full, true);
    { auto m = full; m[0] &= ~0xffu; run("mask: bits 0-7 cleared", m, true); }
    { auto m = full; m[0] &= ~0xffffu; run("mask: bits 0-15 cleared", m, true); }
    { auto m = full; m[words - 1] &= ~0xff000000u; run("mask: top 8 bits cleared", m, true); }
    { std::vector<uint32_t> m(words, 0); m[0] = 0xff; run("mask: only bits 0-7", m, true); }
    { std::vector<uint32_t> m(words, 0); m[0] = 0xff00; run("mask: only bits 8-15", m, true); }
    { std::vector<uint32_t> m(words, 0); m[0] = 0x1; run("mask: only bit 0", m, true); }
    { std::vector<uint32_t> m(words, 0); m[0] = 0x2; run("mask: only bit 1", m, true); }
    { std::vector<uint32_t> m(words, 0); m[1] = 0xff; run("mask: only bits 32-39", m, true); }
    { std::vector<uint32_t> m(1, 0xffffff00u); run("mask: one word, 0-7 cleared", m, true); }
    return 0;
}
