// Micro-benchmark: 16-byte loads at 16-byte-aligned vs 4-byte-aligned addresses (development tool; the K7 backward
// kernels read the padded gradient at a +1 float column shift).  hipcc --offload-arch=gfx950 -O3 unaligned_load.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
struct __attribute__((packed, aligned(4))) quad_u { float x, y, z, w; };
template <int SHIFT>
__global__ void k(const float* __restrict__ in, float* __restrict__ out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v;
    if (SHIFT == 0) {
        v = reinterpret_cast<const float4*>(in)[i];
    } else {
        const quad_u q = *reinterpret_cast<const quad_u*>(in + 4 * i + SHIFT);
        v = make_float4(q.x, q.y, q.z, q.w);
    }
    reinterpret_cast<float4*>(out)[i] = v;
}
// aligned loads + neighbour exchange: (a.y, a.z, a.w, next lane's a.x); lane 63 reads its extra float itself
__global__ void k_shuffle(const float* __restrict__ in, float* __restrict__ out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 a = reinterpret_cast<const float4*>(in)[i];
    float nx = __shfl_down(a.x, 1, 64);
    if ((threadIdx.x & 63) == 63) nx = in[4 * i + 4];
    reinterpret_cast<float4*>(out)[i] = make_float4(a.y, a.z, a.w, nx);
}
int main() {
    const size_t n4 = (size_t)1 << 25;    // 512 MB in, 512 MB out
    float *a, *b;
    hipMalloc(&a, n4 * 16 + 64); hipMalloc(&b, n4 * 16);
    hipMemset(a, 0, n4 * 16 + 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 g((unsigned)((n4 + 255) / 256)), t(256);
    for (int rep = 0; rep < 2; ++rep) {
        float ms[4];
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<0>, g, t, 0, 0, a, b, n4); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[0], e0, e1);
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<1>, g, t, 0, 0, a, b, n4); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[1], e0, e1);
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<2>, g, t, 0, 0, a, b, n4); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[2], e0, e1);
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_shuffle, g, t, 0, 0, a, b, n4); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[3], e0, e1);
        if (rep) for (int j = 0; j < 4; ++j)
            printf("%s: %.3f ms per 1 GiB moved -> %.2f TB/s\n", j == 0 ? "aligned float4       " : j == 1 ? "+1 float (4 B aligned)" : j == 2 ? "+2 floats (8 B aligned)" : "aligned + lane shift  ",
                   ms[j] / 5, 2.0 * n4 * 16 / (ms[j] / 5 * 1e-3) / 1e12);
    }
    return 0;
}
