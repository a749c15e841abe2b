// K13 -- the disparity heads: 3x3 stride-1 convolution to ONE output channel (MD2/networks/depth_decoder.py:43-44
// dispconv, MD2/layers.py:127-141 Conv3x3 on the reflection-padded decoder feature), forward.
//
// With a single output channel an MFMA formulation wastes 15 of its 16 rows (K11 takes 224 us for 16 -> 1 at 320x1024,
// MIOpen 465 us) although the layer is a 254 MB streaming read.  Here it is plain vector FMAs: the input tile of 16
// channels x 10 x 66 is staged in LDS (zero padding while staging), a thread owns two horizontally adjacent output
// pixels and per channel reads 3 rows x 4 columns as six 8-byte LDS reads for 18 FMAs, with the 9 filter taps of the
// channel fed as SGPR operands from scalar loads.  Channels are processed 16 at a time (any multiple of 16).
#include "common.hpp"

using namespace dmh;

namespace {

constexpr int NT = 256;
constexpr int TH = 8, TW = 64;          // output tile: 8 rows x 64 columns = 256 threads x 2 pixels
constexpr int KC = 16;                  // channels per LDS stage
constexpr int RH = TH + 2, RW = TW + 2;

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3))) void head_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, int C, int H, int W, int Ho, int Wo,
                                                       int pad, int gx, int gy, float* __restrict__ y) {
    __shared__ float tile[KC * RH * RW];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int gxi = bid % gx;  bid /= gx;
    const int gyi = bid % gy;
    const int b = bid / gy;
    const int oy0 = gyi * TH, ox0 = gxi * TW;
    const size_t HW = (size_t)H * W;
    const float* xb = x + (size_t)b * C * HW;
    const int r = tid >> 5, c2 = tid & 31;                  // output row in the tile, column pair
    float acc0 = 0.f, acc1 = 0.f;
    constexpr int PER_T = (KC * RH * RW + NT - 1) / NT;
    for (int c0 = 0; c0 < C; c0 += KC) {
        float stage[PER_T];
#pragma unroll
        for (int k = 0; k < PER_T; ++k) {
            const int e = tid + NT * k;
            const int c = e / (RH * RW), rem = e - c * (RH * RW), rr = rem / RW, xx = rem - rr * RW;
            const int iy = oy0 - pad + rr, ix = ox0 - pad + xx;
            const bool ok = e < KC * RH * RW && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float v = xb[(size_t)(c0 + (e < KC * RH * RW ? c : 0)) * HW + (size_t)min(max(iy, 0), H - 1) * W +
                               min(max(ix, 0), W - 1)];
            stage[k] = ok ? v : 0.f;
        }
        __syncthreads();                                    // the previous stage has been consumed
#pragma unroll
        for (int k = 0; k < PER_T; ++k) {
            const int e = tid + NT * k;
            if (e < KC * RH * RW) tile[e] = stage[k];
        }
        __syncthreads();
        const float* tp = tile + r * RW + 2 * c2;
        const float* wc = w + (size_t)c0 * 9;               // uniform: scalar loads
#pragma unroll 4
        for (int c = 0; c < KC; ++c) {     // 4 channels in flight: fully unrolled it holds 192 registers of LDS data
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float2 lo = *reinterpret_cast<const float2*>(tp + c * (RH * RW) + ky * RW);
                const float2 hi = *reinterpret_cast<const float2*>(tp + c * (RH * RW) + ky * RW + 2);
                const float w0 = wc[c * 9 + ky * 3], w1 = wc[c * 9 + ky * 3 + 1], w2 = wc[c * 9 + ky * 3 + 2];
                acc0 = fmaf(lo.x, w0, acc0); acc0 = fmaf(lo.y, w1, acc0); acc0 = fmaf(hi.x, w2, acc0);
                acc1 = fmaf(lo.y, w0, acc1); acc1 = fmaf(hi.x, w1, acc1); acc1 = fmaf(hi.y, w2, acc1);
            }
        }
    }
    const int oy = oy0 + r, ox = ox0 + 2 * c2;
    if (oy < Ho && ox < Wo) {
        const float bs = bias ? bias[0] : 0.f;
        float* yp = y + ((size_t)b * Ho + oy) * Wo + ox;
        if (ox + 1 < Wo && (Wo & 1) == 0) {
            *reinterpret_cast<float2*>(yp) = make_float2(acc0 + bs, acc1 + bs);
        } else {
            yp[0] = acc0 + bs;
            if (ox + 1 < Wo) yp[1] = acc1 + bs;
        }
    }
}

}  // namespace

extern "C" {

int dmh_conv3x3_head(const float* x, const float* w, const float* bias, int B, int C, int H, int W, int pad, float* y,
                     void* stream) {
    DMH_REQUIRE(x && w && y, "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && C % KC == 0, "input channels must be a multiple of 16");
    DMH_REQUIRE(pad >= 0 && pad <= 2, "pad must be 0, 1 or 2");
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    DMH_REQUIRE(Ho >= 1 && Wo >= 1, "image smaller than the filter");
    DMH_REQUIRE((int64_t)C * H * W < ((int64_t)1 << 31), "image too large");
    const int gx = (Wo + TW - 1) / TW, gy = (Ho + TH - 1) / TH;
    const long long blocks = (long long)B * gx * gy;
    DMH_REQUIRE(blocks < (1ll << 31), "grid too large");
    hipLaunchKernelGGL(head_conv_kernel, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, x, w, bias, C, H, W, Ho,
                       Wo, pad, gx, gy, y);
    return check_launch("dmh_conv3x3_head");
}

}  // extern "C"
