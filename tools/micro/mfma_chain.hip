// Micro-benchmark (development tool): does v_mfma_f32_32x32x2_f32 lose issue slots when the same accumulator comes back after
// D - 1 other MFMAs?  One wave per SIMD, 64 MFMAs per iteration over D accumulators in rotation (K10: D = 2 within a position
// pair; K18: 4; 16 = no dependency in sight).   hipcc --offload-arch=gfx950 -O3 -o var/mfma_chain tools/micro/mfma_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(float* out, int n) {
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[p][v] = 0.f;
    const float x = threadIdx.x * 0.001f, y = 1.0001f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            // groups of D accumulators, 8 MFMAs on each accumulator of a group before the next group (K10's slot order for D = 2)
            const int grp = (m / (8 * D)) * D, a = (grp + (m % D)) & 15;
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int v = 0; v < 16; ++v) s += acc[p][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int D>
void run(float* d) {
    const int n = 2048;
    const size_t smem = 100 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0, best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<D>), dim3(256), dim3(256), smem, 0, d, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("same accumulator every %2d MFMAs: %.2f cycles per MFMA at a nominal 2.4 GHz\n", D, best * 1e-3 * 2.4e9 / (n * 64.0));
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 256 * sizeof(float));
    run<1>(d); run<2>(d); run<4>(d); run<8>(d); run<16>(d); run<2>(d); run<16>(d);
    return 0;
}
