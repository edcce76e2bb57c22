// comm_probe.hip -- a stand-in for RCCL's device kernel where no second rank exists (lfg_comm_probe, lfg_comm.cpp).
//
// What a broadcast needs from the chip is decided by the kernel RCCL launches for it.  From the gfx950 code objects inside the
// image's two librccl.so (tools/rccl_kernel_footprint.py, profiles/r05_rccl_kernel_footprint.txt): PyTorch's RCCL 2.26.6,
// rcclGenericKernel<1|2|4, false|true>: 256 threads per workgroup, 261 - 280 vector registers per lane (17 - 32 of them accumulation
// registers), 19,744 bytes of LDS, one workgroup per channel; ROCm 7.2's RCCL 2.27.7, ncclDevKernel_Generic_1|2|4: 512 threads, 248 - 256
// registers (two waves per SIMD: 496 - 512 of its 512), 37,664 bytes -- a CU to itself per channel in both.  A SIMD has 512
// registers per lane: a wave of that kernel cannot share a SIMD with a wave of the persistent prefilter kernel (256 registers), and a
// prefilter workgroup puts a wave on every SIMD of its CU -- the broadcast needs CUs that hold NO prefilter workgroup
// (the library's streams leave such CUs free while a communicator exists: lfg_own_stream_create).
// This kernel asks for the same: 256 threads, registers up to v243 and a16 (261 with the allocation granule), 19,744 bytes of LDS;
// it stays for the given time and leaves.  On a communicator of one rank ncclBroadcast launches nothing at all, so this is what
// tests/test_gpu_comm.py puts behind a full persistent grid to see whether a broadcast would have found a CU.
#include <hip/hip_runtime.h>

#include "lfg_internal.hpp"

namespace lfg {

__global__ __launch_bounds__(256) void comm_probe_kernel(int ticks, uint32_t *sink) {
    __shared__ uint32_t lds[19744 / 4];
    lds[threadIdx.x] = threadIdx.x;
    asm volatile("v_mov_b32 v243, 0\n\tv_accvgpr_write_b32 a16, 0" ::: "v243", "a16");
    const long long t0 = wall_clock64();                              // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (sink && lds[(threadIdx.x + 1) & 255] == 0xffffffffu) *sink = 1;           // (keeps the LDS array)
}

hipError_t launch_comm_probe(hipStream_t s, int workgroups, int microseconds) {
    if (workgroups < 1 || workgroups > 64 || microseconds < 0 || microseconds > 100000) return hipErrorInvalidValue;
    hipLaunchKernelGGL(comm_probe_kernel, dim3((unsigned)workgroups), dim3(256), 0, s, microseconds * 100, (uint32_t *)nullptr);
    return hipGetLastError();
}

}  // namespace lfg
