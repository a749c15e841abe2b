// Micro-benchmark (development tool): does a SECOND wave on the same SIMD issue its vector / LDS / memory instructions
// while the first wave's v_mfma_f32_32x32x2_f32 occupies the matrix pipe?  (mfma_shadow.hip answered the one-wave case:
// nothing hides behind the fp32 MFMA of the SAME wave.)
//
// W waves per SIMD (W = 1: 256 threads per CU, W = 2: 512), LDS sized so that one workgroup fills the CU.  Every wave runs
// n x 4 x (MFMA + F filler instructions of one kind).  Reported: SIMD cycles per MFMA = elapsed cycles / (n * 4 * W).
//   no overlap between waves  -> the W = 2 column equals the W = 1 column (the SIMD serialises everything)
//   full overlap              -> the W = 2 column stays at the bare MFMA cost until the fillers alone exceed it
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shadow2 tools/micro/mfma_shadow2.hip && /tmp/mfma_shadow2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int F, int NT>
__global__ __launch_bounds__(NT) void k(float* out, const float* __restrict__ src, int n) {
    extern __shared__ f32x4 lds[];
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = threadIdx.x * 0.001f, y = 1.0001f;
    float f0 = x, f1 = x + 1, f2 = x + 2, f3 = x + 3;
    f32x4 r = {0, 0, 0, 0};
    const int li = threadIdx.x;
    const float* gp = src + (blockIdx.x * NT + threadIdx.x);
    for (int i = 0; i < n; ++i) {
#define FILL()                                                                                           \
    _Pragma("unroll") for (int q = 0; q < F; ++q) {                                                      \
        if (KIND == 0) { if (q & 1) f0 = __builtin_fmaf(f0, y, x); else f1 = __builtin_fmaf(f1, y, x); }  \
        if (KIND == 1) { f32x4 t = lds[li + NT * (q & 3)]; r += t; }                                      \
        if (KIND == 2) { lds[li + NT * (q & 3)] = r; }                                                    \
        if (KIND == 3) { f2 += __builtin_nontemporal_load(gp + ((i * 4 + q) & 1023) * 65536); }           \
    }                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); FILL()
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0); FILL()
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0); FILL()
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0); FILL()
    }
    f32x16 s = a0 + a1 + a2 + a3;
    float acc = f0 + f1 + f2 + f3 + r.x + r.y + r.z + r.w;
    for (int v = 0; v < 16; ++v) acc += s[v];
    out[blockIdx.x * NT + threadIdx.x] = acc;
}

// the same fillers with NO MFMA: what the filler stream costs by itself
template <int KIND, int F, int NT>
__global__ __launch_bounds__(NT) void kf(float* out, const float* __restrict__ src, int n) {
    extern __shared__ f32x4 lds[];
    float x = threadIdx.x * 0.001f, y = 1.0001f;
    float f0 = x, f1 = x + 1, f2 = x + 2;
    f32x4 r = {0, 0, 0, 0};
    const int li = threadIdx.x;
    const float* gp = src + (blockIdx.x * NT + threadIdx.x);
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            FILL()
        }
    }
    out[blockIdx.x * NT + threadIdx.x] = f0 + f1 + f2 + r.x + r.y + r.z + r.w;
}

template <int KIND, int F, int NT>
float run1(float* d, const float* src, bool filler_only) {
    const int n = 1024;
    const size_t smem = 100 * 1024;   // > half the LDS: one workgroup per CU
    auto fn = filler_only ? kf<KIND, F, NT> : k<KIND, F, NT>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(fn, dim3(256), dim3(NT), smem, 0, d, src, n); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    return ms * 1e-3f * 2.4e9f / (n * 4.0f);      // cycles per (MFMA + F fillers) group of ONE wave
}

template <int KIND, int F>
void run(float* d, const float* src, const char* name) {
    const float w1 = run1<KIND, F, 256>(d, src, false), w2 = run1<KIND, F, 512>(d, src, false);
    const float f1 = F ? run1<KIND, F, 256>(d, src, true) : 0.f, f2 = F ? run1<KIND, F, 512>(d, src, true) : 0.f;
    printf("%-16s x%-2d | 1 wave/SIMD: %6.1f cyc per MFMA (fillers alone %6.1f) | 2 waves/SIMD: %6.1f SIMD cyc per MFMA (fillers alone %6.1f)\n",
           name, F, w1, f1, w2 / 2.f, f2 / 2.f);
}

int main() {
    float *d, *src;
    hipMalloc(&d, 256 * 512 * sizeof(float));
    hipMalloc(&src, (size_t)1024 * 65536 * sizeof(float) + 256 * 512 * sizeof(float));
    hipMemset(src, 0, (size_t)1024 * 65536 * sizeof(float) + 256 * 512 * sizeof(float));
    run<0, 0>(d, src, "none");
    run<0, 4>(d, src, "v_fma_f32"); run<0, 8>(d, src, "v_fma_f32"); run<0, 16>(d, src, "v_fma_f32");
    run<1, 1>(d, src, "ds_read_b128"); run<1, 2>(d, src, "ds_read_b128"); run<1, 4>(d, src, "ds_read_b128"); run<1, 8>(d, src, "ds_read_b128");
    run<2, 1>(d, src, "ds_write_b128"); run<2, 2>(d, src, "ds_write_b128"); run<2, 4>(d, src, "ds_write_b128");
    run<3, 1>(d, src, "global_load_b32"); run<3, 2>(d, src, "global_load_b32"); run<3, 4>(d, src, "global_load_b32");
    return 0;
}
