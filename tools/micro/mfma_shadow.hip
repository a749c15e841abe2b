// Micro-benchmark: how much other work issues in the shadow of v_mfma_f32_32x32x2_f32 on gfx950 (development tool).
// One wave per SIMD (256 threads/CU, LDS sized so that one workgroup fills the CU); per MFMA F filler instructions of
// one kind are placed between it and the next MFMA (4 independent accumulators).  Reports cycles per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int F>
__global__ __launch_bounds__(256) void k(float* out, int n) {
    extern __shared__ f32x4 lds[];
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = threadIdx.x * 0.001f, y = 1.0001f;
    float f0 = x, f1 = x + 1, f2 = x + 2, f3 = x + 3;
    f32x4 r = {0, 0, 0, 0};
    const int li = threadIdx.x;
    for (int i = 0; i < n; ++i) {
#define FILL()                                                                                         \
    _Pragma("unroll") for (int q = 0; q < F; ++q) {                                                    \
        if (KIND == 0) { if (q & 1) f0 = __builtin_fmaf(f0, y, x); else f1 = __builtin_fmaf(f1, y, x); } \
        if (KIND == 1) { f32x4 t = lds[li + 256 * q]; r += t; }                                         \
        if (KIND == 2) { lds[li + 256 * q] = r; }                                                       \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); FILL()
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0); FILL()
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0); FILL()
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0); FILL()
    }
    f32x16 s = a0 + a1 + a2 + a3;
    float acc = f0 + f1 + f2 + f3 + r.x + r.y + r.z + r.w;
    for (int v = 0; v < 16; ++v) acc += s[v];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND, int F>
void run(float* d, const char* name) {
    const int n = 2048;
    const size_t smem = 100 * 1024;   // > half the LDS: one workgroup per CU = one wave per SIMD
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND, F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL((k<KIND, F>), dim3(256), dim3(256), smem, 0, d, n); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-14s x%-2d per MFMA: %.1f cycles/MFMA @2.4GHz\n", name, F, ms * 1e-3 * 2.4e9 / (n * 4.0));
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * sizeof(float));
    run<0, 0>(d, "none");
    run<0, 4>(d, "v_fma_f32"); run<0, 8>(d, "v_fma_f32"); run<0, 12>(d, "v_fma_f32"); run<0, 16>(d, "v_fma_f32"); run<0, 24>(d, "v_fma_f32");
    run<1, 1>(d, "ds_read_b128"); run<1, 2>(d, "ds_read_b128"); run<1, 4>(d, "ds_read_b128"); run<1, 8>(d, "ds_read_b128");
    run<2, 1>(d, "ds_write_b128"); run<2, 2>(d, "ds_write_b128"); run<2, 4>(d, "ds_write_b128");
    return 0;
}
