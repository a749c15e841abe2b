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
    constexpr int CS = ((RH * RW + 15) / 32) * 32 + 16;     // channel stride: == 16 (mod 32) floats, >= RH*RW
    static_assert(CS >= RH * RW, "channel stride");
    extern __shared__ float tile[];                          // [C][CS]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int bid = blockIdx.x;
    const int gxi = bid % a.gx;  bid /= a.gx;
    const int gyi = bid % a.gy;
    const int b = bid / a.gy;
    const int oy0 = gyi * TH, ox0 = gxi * TW;
    const size_t HW = (size_t)a.H * a.W;
    const float* xb = a.x + (size_t)b * a.n_in * HW;      // n_in <= C: missing channels are staged as zeros

    // ---- stage the input tile: rows oy0-pad .. +RH, cols ox0-pad .. +RW of every channel (zero outside the image).
    //      Wave w stages channels [NQ w, NQ w + NQ): one 64-column wave-load per (channel, row) through a buffer resource
    //      -- the row's byte offset is SALU arithmetic in an SGPR, the per-lane column offset a loop-invariant register
    //      whose out-of-range value (0xFFFFFFFF: reads 0) IS the zero padding -- and an LDS write at an immediate
    //      offset from one per-lane base.  One vector instruction per row (the select between the column offset and the
    //      out-of-range offset for rows outside the image); round 2's flat element index cost ~35 per ELEMENT (index
    //      decomposition, clamps, 64-bit addresses, masks): 1,900 vector instructions per thread ahead of 288 MFMAs that
    //      share the same pipe.  The two halo columns 64, 65 of every row: a flat pass of <= 3 elements per thread.
    //      All loads are issued before the first LDS write, so their latencies overlap.
    {
        constexpr int RPW = NQ * RH;                               // rows staged per wave
        constexpr int TAILS = (2 * C * RH + NT - 1) / NT;
        const int wvu = __builtin_amdgcn_readfirstlane(wv);
        const rsrc_t xrs = make_rsrc(a.x, (unsigned)((size_t)a.B * a.n_in * HW * 4));
        const int ix = ox0 - a.pad + lane;
        const unsigned col_off = (ix >= 0 && ix < a.W) ? (unsigned)ix * 4u : 0xFFFFFFFFu;
        float stage[RPW], tail[TAILS];
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int c = NQ * wvu + k / RH, iy = oy0 - a.pad + (k % RH);          // uniform
            const bool ok = c < a.n_in && iy >= 0 && iy < a.H;
            // (clamped, not selected, and pinned to an SGPR: a scalar offset in a VGPR makes the load a waterfall loop)
            unsigned rowb = (unsigned)(((b * a.n_in + min(c, a.n_in - 1)) * a.H + min(max(iy, 0), a.H - 1)) * a.W) * 4u;
            asm volatile("" : "+s"(rowb));
            stage[k] = (DMH_SMALL_ABLATE & 1) ? 1.f : ldb(xrs, ok ? col_off : 0xFFFFFFFFu, rowb);
        }
#pragma unroll
        for (int k = 0; k < TAILS; ++k) {
            const int e = tid + NT * k, rr = e >> 1, c = rr / RH, r = rr - c * RH;
            const int iy = oy0 - a.pad + r, ix2 = ox0 - a.pad + 64 + (e & 1);
            const bool ok = rr < C * RH && c < a.n_in && iy >= 0 && iy < a.H && ix2 >= 0 && ix2 < a.W;
            tail[k] = ldb(xrs, ok ? (unsigned)(((b * a.n_in + c) * a.H + iy) * a.W + ix2) * 4u : 0xFFFFFFFFu, 0u);
        }
        float* const dst = tile + NQ * wvu * CS + lane;
#pragma unroll
        for (int k = 0; k < RPW; ++k) dst[(k / RH) * CS + (k % RH) * RW] = stage[k];
#pragma unroll
        for (int k = 0; k < TAILS; ++k) {
            const int e = tid + NT * k, rr = e >> 1, c = rr / RH, r = rr - c * RH;
            if (rr < C * RH) tile[c * CS + r * RW + 64 + (e & 1)] = tail[k];
        }
    }

    // ---- the filter of this lane: A operand of mfma 16x16x4 = W[kout = lane & 15][channel = 4q + (lane >> 4)]; nine taps
    //      at immediate offsets from one per-(block, quad) offset, out of range (= 0) for channels the layer does not have
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
            // (out-of-range base 0x80000000: adding the tap's immediate offset cannot wrap back into the filter)
            const unsigned wo = ok ? (unsigned)((a.backward ? ci * a.Cw + ko : ko * a.Cw + ci) * 9) * 4u : 0x80000000u;
#pragma unroll
            for (int t = 0; t < 9; ++t) wreg[kb][q][t] = ldb(wrs, wo + 4u * (unsigned)(a.backward ? 8 - t : t), 0u);
        }
    }
    __syncthreads();

    // ---- wave wv owns the 16-pixel column block wv of every row of the tile
    const float* src = tile + (lane >> 4) * CS + 16 * wv + (lane & 15);
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
                const float xv = src[4 * q * CS + (row + t / 3) * RW + (t % 3)];
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
    constexpr int RH = TH + 2, RW = TW + 2;
    constexpr int CS = ((RH * RW + 15) / 32) * 32 + 16;
    constexpr size_t smem = (size_t)4 * NQ * CS * sizeof(float);
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(small_conv_kernel<NQ, NKB, TH>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return fail(DMH_ELAUNCH, "%s: cannot raise the dynamic LDS limit", "dmh_conv3x3_small");
        configured = true;
    }
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
