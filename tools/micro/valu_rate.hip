// Micro-benchmark: issue rate of scalar vs packed fp32 FMA on gfx950 at 1 / 2 / 4 / 8 waves per SIMD on all 256 CUs, by HIP events
// (cycles at the nominal 2.4 GHz).  The roof of a kernel made of scalar (non-packed) fp32 vector instructions.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// (-fno-slp-vectorize: otherwise hipcc packs the eight independent scalar chains into v_pk_fma_f32 and both columns read alike)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ void k_scalar(float* out, int n, unsigned long long* clk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0001f, c = 0.5f;
    for (int i = 0; i < n; ++i) {
        a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (clk && threadIdx.x == 0) clk[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}
__global__ void k_packed(float* out, int n) {
    float2v a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float2v m = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
    for (int i = 0; i < n; ++i) {
        a0 = __builtin_elementwise_fma(a0, m, c); a1 = __builtin_elementwise_fma(a1, m, c); a2 = __builtin_elementwise_fma(a2, m, c); a3 = __builtin_elementwise_fma(a3, m, c);
        a4 = __builtin_elementwise_fma(a4, m, c); a5 = __builtin_elementwise_fma(a5, m, c); a6 = __builtin_elementwise_fma(a6, m, c); a7 = __builtin_elementwise_fma(a7, m, c);
    }
    float2v s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
int main() {
    float* d; hipMalloc(&d, 256 * 32 * 256 * sizeof(float));
    unsigned long long* clk; hipMalloc(&clk, 256 * 8 * sizeof(unsigned long long));
    static unsigned long long hclk[256 * 8];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 32768;
    for (int waves = 1; waves <= 8; waves += (waves < 4 ? 1 : 2)) {
        dim3 grid(256 * waves), block(256);  // `waves` workgroups of 4 waves per CU -> `waves` waves per SIMD
        for (int rep = 0; rep < 2; ++rep) {
            float ms_s, ms_p;
            hipEventRecord(e0); hipLaunchKernelGGL(k_scalar, grid, block, 0, 0, d, n, clk); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_s, e0, e1);
            hipEventRecord(e0); hipLaunchKernelGGL(k_packed, grid, block, 0, 0, d, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_p, e0, e1);
            if (rep) {
                const double inst = (double)n * 8 * waves;  // wave-instructions per SIMD
                printf("waves/SIMD %d: scalar fma %.3f ms -> %.2f cycles/inst/SIMD @2.4GHz (%.1f TFLOP/s) | packed %.3f ms -> %.2f cycles/inst (%.1f TFLOP/s)\n",
                       waves, ms_s, ms_s * 1e-3 * 2.4e9 / inst, 256.0 * 4 * inst * 64 * 2 / (ms_s * 1e-3) / 1e12,
                       ms_p, ms_p * 1e-3 * 2.4e9 / inst, 256.0 * 4 * inst * 64 * 4 / (ms_p * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
