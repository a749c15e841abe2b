// K11 -- 3x3 stride-1 convolution with FEW channels (<= 4, 16 or 32 in, <= 32 out) at full image resolution: the last
// decoder stage, MD2/networks/depth_decoder.py:38-41 upconv(0,0) 32->16 @160x512, upconv(0,1) 16->16 @320x1024 and the
// disparity heads (MD2/layers.py:127-141 Conv3x3), forward and backward-data.
//
// Why a second convolution kernel: the Winograd kernel (K10) amortises its input transform over 64 output channels;
// with 16 it would spend its time transforming.  MIOpen runs these shapes at 32-42 TFLOP/s (tools/wino_bench.py).
// Here the convolution is a direct implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32):
//     D[kout 16][pixel 16] += W[kout 16][4 channels] * X[4 channels][pixel 16]       per tap and channel quad,
// i.e. 9 * C/4 MFMAs per 16 output pixels and 16-channel output block.  The whole filter lives in registers
// (9 * C/4 VGPRs per output block, loaded once per wave); the input tile (C x (TH+2) x 66) is staged in LDS once per
// workgroup and every MFMA's B operand is one ds_read_b32 straight out of it (lanes = 16 consecutive pixels x 4
// channels; the channel stride is padded to 16 mod 32 banks: conflict-free).  ~100 VGPRs: several workgroups share a
// CU and hide each other's staging.  Zero padding (0, 1 or 2) is applied while staging.
// Backward-data = the same kernel reading the filter flipped and with the channel roles swapped, pad' = 2 - pad.
#include "common.hpp"

using namespace dmh;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef DMH_SMALL_ABLATE      // timing experiments (tools/small_ablate.sh): 1 no global loads, 2 no MFMA loop, 4 no stores
#define DMH_SMALL_ABLATE 0
#endif
constexpr int NT = 256;
constexpr int TW = 64;           // output columns per workgroup: 4 waves x 16 pixels

struct SArgs {
    const float* x;
    const float* w;              // forward filter [Kw][Cw][3][3]
    const float* bias;           // [n_out] or null
    float* y;
    int B, n_in, n_out, Kw, Cw, H, W, Ho, Wo, pad, backward;
    int gx, gy;
};

// NQ = n_in / 4 (4 or 8); NKB = 16-channel output blocks (1 or 2); TH output rows per workgroup
template <int NQ, int NKB, int TH>
__global__ __launch_bounds__(NT) void small_conv_kernel(SArgs a) {
    constexpr int C = 4 * NQ;
    constexpr int RH = TH + 2, RW = TW + 2;
    constexpr int NWR = (RW + 6) / 4;                        // aligned 16-byte words per staged row (18)
    constexpr int RWA = 4 * NWR;                             // LDS row pitch in floats (72)
    constexpr int CS = ((RH * RWA + 15) / 32) * 32 + 16;     // channel stride: == 16 (mod 32) floats, >= RH*RWA
    static_assert(CS >= RH * RWA, "channel stride");
    extern __shared__ float tile[];                          // [C][CS]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int bid = blockIdx.x;
    const int gxi = bid % a.gx;  bid /= a.gx;
    const int gyi = bid % a.gy;
    const int b = bid / a.gy;
    const int oy0 = gyi * TH, ox0 = gxi * TW;
    const size_t HW = (size_t)a.H * a.W;
    const int coff = (4 - (a.pad & 3)) & 3;                  // columns of a row's first word in front of the tile

    // ---- stage the input tile: rows oy0-pad .. +RH, cols ox0-pad .. +RW of every channel (zero outside the image).
    //      Wave w stages channels [NQ w, NQ w + NQ) as aligned 16-BYTE WORDS, lanes along the rows (18 words cover the 66
    //      columns of a row from its first column rounded down to a multiple of 4): 12 buffer loads per lane instead of one
    //      dword per row and lane (40) -- a vector-memory instruction costs the wave ~64 cycles on the pipe the MFMAs of the
    //      other workgroups of the CU want -- and an out-of-range offset (reads 0) IS the zero padding.  A word is inside or
    //      outside the image as a whole when W is a multiple of 4 (every layer here); otherwise the word that straddles the
    //      right edge is masked before it is written (`partial`).  All loads are issued before the first LDS write.
    {
        constexpr int NROW = NQ * RH;                              // rows staged per wave
        constexpr int PER_L = (NROW * NWR + 63) / 64;
        const int wvu = __builtin_amdgcn_readfirstlane(wv);
        const rsrc_t xrs = make_rsrc(a.x, (unsigned)((size_t)a.B * a.n_in * HW * 4));
        const bool partial = a.pad > 0 && (a.W & 3) != 0;
        const int ixa = ox0 - a.pad - coff;                        // multiple of 4 (ox0 is a multiple of 64)
        f32x4 stage[PER_L];
#pragma unroll
        for (int k = 0; k < PER_L; ++k) {
            const int item = lane + 64 * k, rl = item / NWR, f = item - rl * NWR;
            const int c = NQ * wvu + rl / RH, iy = oy0 - a.pad + (rl % RH), cx = ixa + 4 * f;
            const bool ok = item < NROW * NWR && c < a.n_in && iy >= 0 && iy < a.H && cx >= 0 && cx < a.W;
            const unsigned vo = ok ? (unsigned)(((b * a.n_in + c) * a.H + iy) * a.W + cx) * 4u : 0xFFFFFFFFu;
            f32x4 v = (DMH_SMALL_ABLATE & 1) ? f32x4{1.f, 1.f, 1.f, 1.f}
                                             : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, 0, 0));
            if (partial) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (cx + j < a.W) ? v[j] : 0.f;
            }
            stage[k] = v;
        }
#pragma unroll
        for (int k = 0; k < PER_L; ++k) {
            const int item = lane + 64 * k, rl = item / NWR, f = item - rl * NWR;
            if (item < NROW * NWR)
                *reinterpret_cast<f32x4*>(tile + (NQ * wvu + rl / RH) * CS + (rl % RH) * RWA + 4 * f) = stage[k];
        }
    }

    // ---- the filter of this lane: A operand of mfma 16x16x4 = W[kout = lane & 15][channel = 4q + (lane >> 4)]; its nine
    //      taps are contiguous in memory: two 16-byte loads + one dword (out of range = 0 for channels the layer lacks)
    const rsrc_t wrs = make_rsrc(a.w, (unsigned)((size_t)a.Kw * a.Cw * 9 * 4));
    float wreg[NKB][NQ][9];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int ko = kb * 16 + (lane & 15);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int ci = 4 * q + (lane >> 4);
            const bool ok = ko < a.n_out && ci < a.n_in;
            // forward: w[ko][ci][t]; backward-data: w[ci][ko][8 - t] (flipped, channel roles swapped)
            // (out-of-range base 0x80000000: adding the immediate offsets cannot wrap back into the filter)
            const unsigned wo = ok ? (unsigned)((a.backward ? ci * a.Cw + ko : ko * a.Cw + ci) * 9) * 4u : 0x80000000u;
            const f32x4 w0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wo, 0, 0));
            const f32x4 w1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wo + 16u, 0, 0));
            const float w8 = ldb(wrs, wo + 32u, 0u);
            const float taps[9] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w8};
#pragma unroll
            for (int t = 0; t < 9; ++t) wreg[kb][q][t] = a.backward ? taps[8 - t] : taps[t];
        }
    }
    __syncthreads();

    // ---- wave wv owns the 16-pixel column block wv of every row of the tile
    const float* src = tile + (lane >> 4) * CS + 16 * wv + (lane & 15) + coff;
    const int ox = ox0 + 16 * wv + (lane & 15);
    float bs[NKB][4];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ko = kb * 16 + 4 * (lane >> 4) + r;
            bs[kb][r] = (a.bias && ko < a.n_out) ? a.bias[ko] : 0.f;
        }
    float* yb = a.y + (size_t)b * a.n_out * a.Ho * a.Wo;
#pragma unroll 1
    for (int row = 0; row < TH; ++row) {
        f32x4 acc[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) acc[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < ((DMH_SMALL_ABLATE & 2) ? 0 : NQ); ++q)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float xv = src[4 * q * CS + (row + t / 3) * RWA + (t % 3)];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
                    acc[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[kb][q][t], xv, acc[kb], 0, 0, 0);
            }
        const int oy = oy0 + row;
        if (oy < a.Ho && ox < a.Wo && !((DMH_SMALL_ABLATE & 4) && acc[0][0] != 123.f)) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ko = kb * 16 + 4 * (lane >> 4) + r;     // D row = (lane >> 4) * 4 + reg, col = lane & 15
                    if (ko < a.n_out) yb[((size_t)ko * a.Ho + oy) * a.Wo + ox] = acc[kb][r] + bs[kb][r];
                }
        }
    }
}

template <int NQ, int NKB, int TH>
int launch(SArgs& a, hipStream_t st) {
    constexpr int RH = TH + 2, RWA = 4 * ((TW + 2 + 6) / 4);
    constexpr int CS = ((RH * RWA + 15) / 32) * 32 + 16;
    constexpr size_t smem = (size_t)4 * NQ * CS * sizeof(float);
    static std::atomic<uint64_t> configured{0};     // per device, see configure_dynamic_lds
    if (configure_dynamic_lds(small_conv_kernel<NQ, NKB, TH>, smem, configured) != hipSuccess)
        return fail(DMH_ELAUNCH, "%s: cannot raise the dynamic LDS limit", "dmh_conv3x3_small");
    a.gx = (a.Wo + TW - 1) / TW;
    a.gy = (a.Ho + TH - 1) / TH;
    const long long blocks = (long long)a.B * a.gx * a.gy;
    if (blocks >= (1ll << 31)) return fail(DMH_EINVAL, "%s: grid too large", "dmh_conv3x3_small");
    hipLaunchKernelGGL((small_conv_kernel<NQ, NKB, TH>), dim3((unsigned)blocks), dim3(NT), smem, st, a);
    return check_launch("dmh_conv3x3_small");
}

}  // namespace

extern "C" {

int dmh_conv3x3_small(const float* x, const float* w, const float* bias, int B, int Kw, int Cw, int H, int W, int pad,
                      int backward, float* y, void* stream) {
    DMH_REQUIRE(x && w && y, "null pointer");
    DMH_REQUIRE(B > 0 && Kw > 0 && Cw > 0 && H > 0 && W > 0, "bad sizes");
    DMH_REQUIRE(pad >= 0 && pad <= 2, "pad must be 0, 1 or 2");
    SArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    a.B = B; a.Kw = Kw; a.Cw = Cw; a.H = H; a.W = W; a.pad = pad; a.backward = backward ? 1 : 0;
    a.n_in = backward ? Kw : Cw;
    a.n_out = backward ? Cw : Kw;
    a.Ho = H + 2 * pad - 2; a.Wo = W + 2 * pad - 2;
    DMH_REQUIRE(a.Ho >= 1 && a.Wo >= 1, "image smaller than the filter");
    DMH_REQUIRE((int64_t)a.n_in * H * W < ((int64_t)1 << 31) && (int64_t)a.n_out * a.Ho * a.Wo < ((int64_t)1 << 31),
                "image too large");
    DMH_REQUIRE((int64_t)B * a.n_in * H * W < ((int64_t)1 << 30), "input larger than 4 GB (32-bit byte offsets of the buffer loads)");
    hipStream_t st = (hipStream_t)stream;
    if (a.n_in <= 4 && a.n_out <= 16) return launch<1, 1, 8>(a, st);      // disparity-head backward: 1 -> 16
    if (a.n_in <= 4 && a.n_out <= 32) return launch<1, 2, 8>(a, st);
    if (a.n_in == 16 && a.n_out <= 16) return launch<4, 1, 8>(a, st);
    if (a.n_in == 16 && a.n_out <= 32) return launch<4, 2, 8>(a, st);
    if (a.n_in == 32 && a.n_out <= 16) return launch<8, 1, 4>(a, st);
    return fail(DMH_EINVAL, "%s: supported channel counts are <=4 -> <=32, 16 -> <=32 and 32 -> <=16", "dmh_conv3x3_small");
}

}  // extern "C"
