// Micro-benchmark (report only; nothing here is on the product path): what the ONE arithmetic that could move K10 would
// sustain.  K10 multiplies exact-fp32 operands on v_mfma_f32_32x32x2_f32 (64 cycles per MFMA, the vector pipe blocked
// meanwhile: DESIGN.md section 9).  The alternative costed here: every fp32 operand split ONCE, when it is staged, into three
// bf16 terms (x = hi + mid + lo, 3 x 8 = 24 mantissa bits), a product as the six terms of order <= 2
//     a b ~= ah bh + ah bm + am bh + ah bl + al bh + am bm
// on v_mfma_f32_32x32x16_bf16 (32 cycles per MFMA, of which 8 hold the vector issue port), accumulated in fp32.
//
// Two synthetic loops with K10's wave tile (32 x 32 outputs x 16 Winograd positions = 256 accumulator registers, one wave per
// SIMD, operands read from LDS by ds_read_b128) and K10's per-chunk instruction census beside the MFMAs:
//   MODE 0  fp32 MFMA, per 8 input channels:   64 MFMA, 32 ds_read_b128, 76 VALU (input transform), 16 ds_read2_b32, 16 ds_write_b64
//   MODE 1  bf16 x 3,  per 16 input channels:  96 MFMA, 96 ds_read_b128 (1.5 x the operand bytes), the transform of 4 channels
//           (128 VALU) + the split of their 64 values per lane (v_cvt_pk_bf16_f32 / shift / mask / subtract: 6 per value),
//           32 ds_read2_b32, 48 ds_write_b64 (three bf16 images)
//   MODE 2  as 1 without the split and transform VALU (MFMA + operand reads only): what the matrix pipe + LDS allow
// Output: cycles per 16 input channels of one wave's tile, and the direct-convolution-equivalent TFLOP/s of 256 CUs at the
// clock the chip held (measured from the loop's wall time at a nominal 2.4 GHz AND reported as wall time per chunk).
//
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o var/bf16x3 tools/micro/bf16x3.hip && var/bf16x3
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {        // two floats -> packed bf16 (RNE): one instruction
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 r = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, r);
}

// (x0, x1) -> the three packed bf16 words (hi, mid, lo): 12 instructions per pair
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = cvt_pk(x0, x1);
    const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
    const float r0 = x0 - h0, r1 = x1 - h1;
    mid = cvt_pk(r0, r1);
    const float m0 = __builtin_bit_cast(float, mid << 16), m1 = __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = cvt_pk(r0 - m0, r1 - m1);
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void loop_kernel(float* out, int n) {
    extern __shared__ f32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // operand images (contents are irrelevant for the timing; filled once so that no lane reads NaN patterns)
    for (int i = tid; i < 6 * 1024; i += 256) lds[i] = f32x4{1.f + i * 1e-6f, 0.5f, 0.25f, 0.125f};
    __syncthreads();
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[p][v] = 0.f;
    const f32x4* A = lds + lane + wv * 64;          // lane-linear 16-byte words: conflict-free ds_read_b128
    float* raw = reinterpret_cast<float*>(lds + 6 * 1024) + tid * 2;
    float* img = reinterpret_cast<float*>(lds + 7 * 1024) + tid * 2;
    float carry = 0.f;
    for (int it = 0; it < n; ++it) {
        if (MODE == 0) {
            // ---- one chunk of 8 input channels on the fp32 MFMA (K10's loop): run twice per "16 channels" by the caller's n.
            //      The operands of position p + 1 are requested before the MFMAs of position p (K10's software pipeline).
            f32x4 ua = A[0], vb = A[2048];
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const f32x4 ua_n = A[((p + 1) & 7) * 256], vb_n = A[((p + 1) & 7) * 256 + 2048];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[ks], vb[ks], acc[p], 0, 0, 0);
                // the input transform of 2 channels, 1/16 of it per position: a raw read, 5 adds, half a float2 image write
                const float2 d = *reinterpret_cast<const float2*>(raw + (p & 3) * 512);
                float t0 = d.x - d.y + carry, t1 = d.x + d.y, t2 = t1 - t0, t3 = t0 - d.y;
                carry = t2 + t3;
                *reinterpret_cast<float2*>(img + (p & 3) * 512) = make_float2(t0, t3);
                ua = ua_n;
                vb = vb_n;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // ---- one chunk of 16 input channels, three bf16 terms per operand, six MFMAs per position
            f32x4 a[3], b[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                a[t] = A[(t & 7) * 256];
                b[t] = A[(t & 7) * 256 + 2048];
            }
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                f32x4 an[3], bn[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    an[t] = A[(((p + 1) * 3 + t) & 7) * 256];
                    bn[t] = A[(((p + 1) * 3 + t) & 7) * 256 + 2048];
                }
#define BF(X) __builtin_bit_cast(bf16x8, X)
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[0]), BF(b[0]), acc[p], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[0]), BF(b[1]), acc[p], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[1]), BF(b[0]), acc[p], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[0]), BF(b[2]), acc[p], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[2]), BF(b[0]), acc[p], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[1]), BF(b[1]), acc[p], 0, 0, 0);
                if (MODE == 1) {
                    // 1/16 of the staging work of the wave's 4 channels: 2 raw reads, 8 transform adds, the split of 4 values
                    // (two pairs), three 8-byte writes (one word of each bf16 image)
                    const float2 d0 = *reinterpret_cast<const float2*>(raw + (p & 3) * 512);
                    const float2 d1 = *reinterpret_cast<const float2*>(raw + (p & 3) * 512 + 2048);
                    const float t0 = d0.x - d1.x + carry, t1 = d0.y + d1.y, t2 = d1.x - d0.y, t3 = d0.y - d1.y;
                    const float v0 = t0 - t2, v1 = t1 + t2, v2 = t2 - t1, v3 = t1 - t3;
                    carry = v3 * 1e-30f;
                    unsigned h0, m0, l0, h1, m1, l1;
                    split2(v0, v1, h0, m0, l0);
                    split2(v2, v3, h1, m1, l1);
                    *reinterpret_cast<uint2*>(img + (p & 3) * 512) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(img + (p & 3) * 512 + 2048) = make_uint2(m0, m1);
                    *reinterpret_cast<uint2*>(img + (p & 3) * 512 + 4096) = make_uint2(l0, l1);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    a[t] = an[t];
                    b[t] = bn[t];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = carry;
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int v = 0; v < 16; ++v) s += acc[p][v];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
double run(float* d, int n, int grid) {
    const size_t smem = 140 * 1024;     // one workgroup per CU = one wave per SIMD, as K10 runs
    hipFuncSetAttribute(reinterpret_cast<const void*>(loop_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0.f, best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop_kernel<MODE>), dim3(grid), dim3(256), smem, 0, d, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    return best * 1e-3;
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* d;
    hipMalloc(&d, (size_t)cus * 256 * sizeof(float));
    const int n16 = 4096;               // chunks of 16 input channels per wave
    // direct-convolution-equivalent flops of one wave's 16-channel chunk: 32 x 32 outputs x 16 positions are 32 channels x
    // 32 tiles x 4 pixels; 2 x 9 x 16 flops per output pixel and output channel
    const double flops16 = 2.0 * 9 * 16 * 32 * 32 * 4;
    const double t0 = run<0>(d, 2 * n16, cus), t1 = run<1>(d, n16, cus), t2 = run<2>(d, n16, cus);
    const double per0 = t0 / n16, per1 = t1 / n16, per2 = t2 / n16;
    printf("per 16 input channels of a 32 x 32 x 16-position wave tile, %d CUs x 4 waves, wall time:\n", cus);
    printf("  fp32 MFMA loop (2 chunks of 8: 128 MFMA + K10's staging census)   %.3f us  = %6.0f cycles @2.4 GHz  %6.1f TFLOP/s direct-equivalent\n",
           per0 * 1e6, per0 * 2.4e9, flops16 * 4 * cus / per0 / 1e12);
    printf("  bf16 x 3 loop (96 MFMA + 96 operand reads + transform + split)      %.3f us  = %6.0f cycles @2.4 GHz  %6.1f TFLOP/s direct-equivalent\n",
           per1 * 1e6, per1 * 2.4e9, flops16 * 4 * cus / per1 / 1e12);
    printf("  bf16 x 3, MFMA + operand reads only                                 %.3f us  = %6.0f cycles @2.4 GHz  %6.1f TFLOP/s direct-equivalent\n",
           per2 * 1e6, per2 * 2.4e9, flops16 * 4 * cus / per2 / 1e12);
    printf("  ratio fp32 / bf16x3 (with staging): %.2f\n", per0 / per1);
    hipFree(d);
    return 0;
}
