// Shared host/device helpers for libdmh_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "dmh_hip.h"

namespace dmh {

constexpr int WAVE = 64;

// ---- host-side error reporting ---------------------------------------------------------
extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, const char* a = "", long long b = 0, long long c = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}

#define DMH_REQUIRE(cond, msg)                                                     \
    do {                                                                            \
        if (!(cond)) return dmh::fail(DMH_EINVAL, "%s: requirement failed: " msg, __func__); \
    } while (0)

inline int check_launch(const char* fn) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", fn, hipGetErrorString(e));
        return DMH_ELAUNCH;
    }
    return DMH_OK;
}

// ---- device helpers ----------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
    return v;
}

// Sum over a block of NT threads (NT multiple of 64, <= 1024); result valid in thread 0.
// `red` is LDS scratch of at least NT/64 floats.  Fixed order -> bitwise reproducible.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NT / WAVE; ++i) t += red[i];
    }
    return t;
}

__device__ __forceinline__ int reflect_idx(int p, int n) {
    // ReflectionPad2d(1) index map, then clamped so that far-out-of-range halo slots stay in bounds
    p = p < 0 ? -p : p;
    p = p >= n ? 2 * (n - 1) - p : p;
    return min(max(p, 0), n - 1);
}

// ~0.5-ulp reciprocal: v_rcp_f32 (1 ulp) + one Newton step.  5 instructions instead of the ~12 of an IEEE divide.
__device__ __forceinline__ float fast_rcp(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    return r * (2.0f - x * r);
}

// ---- buffer-resource loads and DPP lane shifts (K1, K13, K15, K16) -----------------------------------------------
// One 128-bit descriptor in SGPRs per tensor, 32-bit byte offsets in VGPRs, a wave-uniform offset in an SGPR: no 64-bit
// address arithmetic per load, and offsets >= the descriptor's size read 0 instead of faulting (zero padding for free).
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}
// Packed fp32 additions (v_pk_add_f32: two IEEE additions per instruction) for the Winograd transforms of K10 / K17 / K18, which
// sit beside fp32 MFMAs that shadow no vector instruction.  As asm: hipcc splits <2 x float> additions into v_add_f32 pairs once
// the operands come out of ds_read2_b32, and -fno-slp-vectorize (build.py) keeps it from forming them by itself.  The operand
// selects were checked on the device (tools/micro/pk_opsel.hip).  (x, y) below = (low, high) half of a register pair.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define DMH_PK_FORM(NAME, MODS)                                                  \
    __device__ __forceinline__ f32x2 NAME(const f32x2 a, const f32x2 b) {        \
        f32x2 r;                                                                 \
        asm("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(r) : "v"(a), "v"(b));         \
        return r;                                                                \
    }
DMH_PK_FORM(pk_add, "")                                                              // (a.x + b.x, a.y + b.y)
DMH_PK_FORM(pk_sub, "neg_lo:[0,1] neg_hi:[0,1]")                                     // (a.x - b.x, a.y - b.y)
DMH_PK_FORM(pk_bfly, "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]")                    // (a.x + b.y, a.x - b.y)
DMH_PK_FORM(pk_bfly_neg, "op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[1,1] neg_hi:[1,0]")   // (-a.x - b.y, -a.x + b.y)
DMH_PK_FORM(pk_col01, "op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1]")                   // (a.x - b.x, a.y + b.x)
DMH_PK_FORM(pk_col23, "op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]")      // (-a.y + b.x, a.y - b.y)
#define DMH_PK_MOV(NAME, MODS)                                                   \
    __device__ __forceinline__ f32x2 NAME(const f32x2 a, const f32x2 b) {        \
        f32x2 r;                                                                 \
        asm("v_pk_mov_b32 %0, %1, %2 " MODS : "=v"(r) : "v"(a), "v"(b));         \
        return r;                                                                \
    }
DMH_PK_MOV(pk_lo_lo, "op_sel:[0,0]")     // (a.x, b.x): one instruction regroups two results of packed additions
DMH_PK_MOV(pk_hi_hi, "op_sel:[1,1]")     // (a.y, b.y)

__device__ __forceinline__ float ldb(rsrc_t rs, unsigned byte_off, unsigned s_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, s_off, 0));
}
// wave_shr:1 / wave_shl:1 DPP controls (GFX9): lane i reads lane i-1 / i+1; the edge lane reads 0.  The compiler folds
// them into the consuming v_add_f32 / v_fma_f32.
__device__ __forceinline__ float lane_prev(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// Philox4x32-R counter-based generator (Salmon et al., SC'11); 7 rounds pass BigCrush.
template <int ROUNDS = 7>
struct Philox {
    uint32_t k0, k1;
    __device__ __forceinline__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    __device__ __forceinline__ uint4 operator()(uint64_t ctr_lo, uint64_t ctr_hi) const {
        uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi,
                 c3 = (uint32_t)(ctr_hi >> 32);
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
            const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
            const uint32_t n0 = hi1 ^ c1 ^ a, n1 = lo1, n2 = hi0 ^ c3 ^ b, n3 = lo0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};

// two uniform 32-bit words -> one N(0,1) sample (Box-Muller)
__device__ __forceinline__ float normal_from_bits(uint32_t u0, uint32_t u1) {
    const float a = ((float)u0 + 1.0f) * 2.3283064365386963e-10f;  // (0,1]
    const float b = (float)u1 * 2.3283064365386963e-10f;           // [0,1)
    return sqrtf(-2.0f * __logf(a)) * __cosf(6.283185307179586f * b);
}

// two uniform 32-bit words -> two independent N(0,1) samples (both Box-Muller branches)
__device__ __forceinline__ float2 normal_pair_from_bits(uint32_t u0, uint32_t u1) {
    const float a = ((float)u0 + 1.0f) * 2.3283064365386963e-10f;  // (0,1]
    const float b = (float)u1 * 2.3283064365386963e-10f;           // [0,1)
    const float rad = sqrtf(-2.0f * __logf(a));
    float sn, cs;
    __sincosf(6.283185307179586f * b, &sn, &cs);
    return make_float2(rad * cs, rad * sn);
}

// ---- stream-K decomposition of the Winograd convolution kernels (K10 wino_conv.hip, K17 wino32_conv.hip) -------------------
// first unit of workgroup w's range (w = 0 .. grid): the first units % grid workgroups get one unit more (no division: this runs
// in every workgroup of the main kernel and several times per workgroup of the fix-up kernel).  A boundary inside an item is
// kept at least 3 chunks away from both of the item's ends (the staging pipeline is 3 chunks deep); items of fewer than 6
// chunks are never cut.  per = units / grid, rem = units % grid (host).
__host__ __device__ inline int sk_boundary(int per, int rem, int nch, int w) {
    const int b = w * per + (w < rem ? w : rem);
    const int item = b / nch;
    int c = b - item * nch;
    if (c != 0) {
        if (nch >= 6) c = c < 3 ? 3 : (c > nch - 3 ? nch - 3 : c);
        else c = (2 * c < nch) ? 0 : nch;
    }
    return item * nch + c;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a process that drives a second GPU (or
// launches from two threads) must not inherit a per-process "already configured" flag.  One bit per device ordinal, set
// with release order after the attribute call succeeded; devices beyond 63 configure on every launch (cheap, correct).
template <typename Kernel>
inline hipError_t configure_dynamic_lds(Kernel kernel, size_t bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && bit) done.fetch_or(bit, std::memory_order_release);
    return e;
}

}  // namespace dmh
