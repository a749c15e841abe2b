// Micro-benchmark (development tool): what a vector-memory instruction costs a wave that runs v_mfma_f32_32x32x2_f32 back to
// back (one wave per SIMD, as K10 / K17 / K18 run), and whether it matters where the instructions stand: N 16-byte buffer loads
// per 48 MFMAs either SPREAD (one after each of the first N MFMAs: K10's slots) or GROUPED (all N after the first MFMA).  The
// loads hit a 64 KB buffer (L2) and are consumed one iteration later, so that their latency is not part of the number.  Also
// N `buffer_load ... lds` (LDS-DMA, K10's filter path).
//   hipcc --offload-arch=gfx950 -O3 -o var/mfma_vmem tools/micro/mfma_vmem.hip && var/mfma_vmem
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}

constexpr int NM = 48;      // MFMAs per iteration

template <int N, int MODE>      // MODE 0 spread, 1 grouped, 2 spread LDS-DMA, 3 grouped LDS-DMA
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(const float* buf, float* out, int n) {
    extern __shared__ f32x4 lds[];
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[p][v] = 0.f;
    const rsrc_t rs = make_rsrc(buf, 64 * 1024);
    const unsigned off = threadIdx.x * 16u;
    const float x = threadIdx.x * 0.001f, y = 1.0001f;
    f32x4 ld[N > 0 ? N : 1], sum = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < (N > 0 ? N : 1); ++q) ld[q] = f32x4{0, 0, 0, 0};
    const unsigned ldst = (unsigned)(unsigned long long)(lds) + (threadIdx.x >> 6) * 1024u;
    for (int i = 0; i < n; ++i) {
        if (MODE < 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < N; ++q) sum += ld[q];
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            acc[m & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[m & 15], 0, 0, 0);
            const int lo = (MODE & 1) ? (m == 0 ? 0 : N) : (m < N ? m : N), hi = (MODE & 1) ? (m == 0 ? N : N) : (m < N ? m + 1 : N);
#pragma unroll
            for (int q = lo; q < hi; ++q) {
                if (MODE < 2) {
                    ld[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, (unsigned)(q * 4096), 0));
                } else {
                    const unsigned m0v = __builtin_amdgcn_readfirstlane(ldst + (unsigned)q * 4096u);
                    const unsigned so = (unsigned)(q * 4096);
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(off), "s"(rs), "s"(m0v), "s"(so) : "memory");
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = sum.x + sum.y + sum.z + sum.w + reinterpret_cast<float*>(lds)[threadIdx.x];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int v = 0; v < 16; ++v) s += acc[p][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int N, int MODE>
double run(const float* buf, float* d) {
    const int n = 1024;
    const size_t smem = 100 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<N, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0, best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<N, MODE>), dim3(256), dim3(256), smem, 0, buf, d, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best * 1e-3 * 2.4e9 / n;     // cycles per iteration at a nominal 2.4 GHz
}

int main() {
    float *buf, *d;
    hipMalloc(&buf, 64 * 1024);
    hipMemset(buf, 0, 64 * 1024);
    hipMalloc(&d, 256 * 256 * sizeof(float));
    const double base = run<0, 0>(buf, d);
    printf("48 MFMAs alone: %.0f cycles @2.4 GHz (%.1f per MFMA)\n", base, base / NM);
#define ROW(N) { const double a = run<N, 0>(buf, d), b = run<N, 1>(buf, d), c = run<N, 2>(buf, d), e = run<N, 3>(buf, d);           \
        printf("N = %2d   buffer_load_dwordx4: spread %+7.0f (%.0f each)  grouped %+7.0f (%.0f each)   |   ... lds: spread %+7.0f (%.0f each)  " \
               "grouped %+7.0f (%.0f each)\n", N, a - base, (a - base) / N, b - base, (b - base) / N, c - base, (c - base) / N, e - base, (e - base) / N); }
    ROW(1) ROW(2) ROW(4) ROW(8) ROW(13)
    return 0;
}
