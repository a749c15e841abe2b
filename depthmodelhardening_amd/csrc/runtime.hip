// Library-wide state: version string and the per-thread error message.
#include "common.hpp"

namespace dmh {
thread_local char g_err[512] = "";
}

using namespace dmh;

namespace {
// A stand-in with the launch geometry of a ring collective's device kernel: `channels` persistent workgroups of 256 threads,
// each streaming its contiguous share of the buffer (read + write, 16 bytes per lane) -- what occupies CUs beside the
// convolution kernels while a gradient all-reduce is in flight.  Diagnostics only (tools/rccl_overlap.py): with ONE rank RCCL
// launches no device kernel at all, so its scheduling against the persistent K10 workgroups cannot be observed on one GPU.
__global__ __launch_bounds__(256) void channel_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4,
                                                           int rounds) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    for (int r = 0; r < rounds; ++r)
        for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
            float4 v = src[i];
            v.x += 0.f;
            dst[i] = v;
        }
}
}  // namespace

extern "C" {
const char* dmh_version(void) { return "dmh_hip 0.1 (gfx950)"; }
const char* dmh_last_error(void) { return dmh::g_err; }

int dmh_debug_channel_copy(const float* src, float* dst, int64_t n, int channels, int rounds, void* stream) {
    DMH_REQUIRE(src && dst && n > 0 && n % 4 == 0 && channels > 0 && channels <= 1024 && rounds > 0, "bad arguments");
    DMH_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "pointers must be 16-byte aligned");
    hipLaunchKernelGGL(channel_copy_kernel, dim3((unsigned)channels), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), (size_t)(n / 4), rounds);
    return check_launch("dmh_debug_channel_copy");
}
}
