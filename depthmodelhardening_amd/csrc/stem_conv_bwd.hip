// K12 -- gradient w.r.t. the INPUT IMAGE of the encoder's first convolution (7x7, stride 2, pad 3, 3 -> 64 channels,
// torchvision ResNet.conv1 under MD2/networks/resnet_encoder.py:88).  Every attack step needs it (the patch gradient
// flows through the pasted scene), and MIOpen serves it with a per-image GEMM + col2im + layout transposes
// (profiles/r01_bench_timed_region.csv: 120 + 120 launches, ~8.4 ms of a step).
//
// Direct gather form, no atomics.  One thread owns a 2x2 block of image pixels (the four stride-2 parity classes) and
// all CIN image channels.  For gradient channel k the block reads a 4x4 neighbourhood of g_y[k] (shared by the four
// pixels) and applies the 49 taps split by parity: 9 + 12 + 12 + 16 taps x CIN FMAs,
//     g_x[c][2Y+py][2X+px] = sum_k sum_{dy,dx} g_y[k][Y-1+dy][X-1+dx] * w[k][c][py+5-2dy][px+5-2dx]   (taps in 0..6).
// The filter is uniform across the wave: it is read through scalar loads and fed to the FMAs as SGPR operands; g_y
// tiles are staged in LDS 8 channels at a time (zero outside the image).  VALU-bound: 147 FMAs + 16 LDS reads per
// (thread, k).
#include "common.hpp"

using namespace dmh;

namespace {

constexpr int NT = 256;
constexpr int BY = 8, BX = 32;              // 2x2 blocks per workgroup: 16 x 64 image pixels
constexpr int KC = 8;                       // gradient channels per LDS stage
constexpr int RW = BX + 3;                  // g_y region: rows Y0-1 .. Y0+BY+1 (BY + 3 of them), cols X0-1 .. X0+BX+1

// Window form (K19, the attack's patch gradient: only the image gradient under the pasted object is read): the workgroups
// cover the wh x ww block window (2x2-pixel blocks = pixels of g_y's frame) at the per-sample, even image-pixel origin win_org, and
// g_y is a compact [B, K, sh, sw] window of its Ho x Wo frame at per-sample origin gy_org -- it must hold everything the
// image window reads (rows / columns -1 .. +2 around it); outside the frame g_y is zero as before.
struct StemWin {
    const int* win_org;     // [B,2] image pixels (even), or NULL: the whole image
    const int* gy_org;      // [B,2] origin of the compact g_y window (NULL: g_y is the whole frame)
    int wh, ww;             // window size in blocks
    int sh, sw;             // plane size of g_y
};

// KS = 4 (the window form: a twelfth of the pixels, ~500 workgroups of a 64-channel serial loop -- latency-bound at 93 us): the
// workgroup covers BY / 4 rows of blocks and each of its four waves takes a quarter of the gradient channels of every LDS
// stage; the four partial sums meet in LDS and are added in wave order (fixed: no atomics).
template <int CIN, int KS>
__global__ __launch_bounds__(NT) void stem_conv_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                           int K, int Ho, int Wo, int gx, int gyb, const StemWin sw_,
                                                           float* __restrict__ gxo) {
    constexpr int BY = ::BY / KS, RH = BY + 3;      // rows of blocks per workgroup, g_y rows staged for them
    constexpr int NG = NT / KS;                     // threads per channel group (one wave when KS = 4)
    __shared__ float tile[KC * RH * RW];
    __shared__ float red[KS > 1 ? (KS - 1) * NG * 4 * CIN : 1];
    const int tid = threadIdx.x;
    const int grp = __builtin_amdgcn_readfirstlane(tid / NG), lt = tid - grp * NG;
    int bid = blockIdx.x;
    const int bxi = bid % gx;  bid /= gx;
    const int byi = bid % gyb;
    const int b = bid / gyb;
    const int wy0 = sw_.win_org ? sw_.win_org[2 * b] >> 1 : 0, wx0 = sw_.win_org ? sw_.win_org[2 * b + 1] >> 1 : 0;
    const int gy0 = sw_.gy_org ? sw_.gy_org[2 * b] : 0, gx0 = sw_.gy_org ? sw_.gy_org[2 * b + 1] : 0;
    const int Y0 = wy0 + byi * BY, X0 = wx0 + bxi * BX;
    const int Yend = wy0 + sw_.wh, Xend = wx0 + sw_.ww;
    const int ty = lt / BX, tx = lt - ty * BX;
    const size_t HWo = (size_t)sw_.sh * sw_.sw;
    const float* gb = gy + (size_t)b * K * HWo;

    float acc[2][2][CIN];
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int c = 0; c < CIN; ++c) acc[py][px][c] = 0.f;

    constexpr int PER_T = (KC * RH * RW + NT - 1) / NT;
    for (int k0 = 0; k0 < K; k0 += KC) {
        float stage[PER_T];
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int e = tid + NT * i;
            const int kk = e / (RH * RW), rem = e - kk * (RH * RW), r = rem / RW, xx = rem - r * RW;
            const int oy = Y0 - 1 + r - gy0, ox = X0 - 1 + xx - gx0;       // inside the (compact) g_y plane
            const bool ok = e < KC * RH * RW && oy >= 0 && oy < sw_.sh && ox >= 0 && ox < sw_.sw;
            const float v = gb[(size_t)(k0 + (ok ? kk : 0)) * HWo + (size_t)min(max(oy, 0), sw_.sh - 1) * sw_.sw +
                               min(max(ox, 0), sw_.sw - 1)];
            stage[i] = ok ? v : 0.f;
        }
        __syncthreads();                    // the previous stage has been consumed
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int e = tid + NT * i;
            if (e < KC * RH * RW) tile[e] = stage[i];
        }
        __syncthreads();
#pragma unroll 1
        for (int kk = grp * (KC / KS); kk < (grp + 1) * (KC / KS); ++kk) {
            const float* tp = tile + kk * (RH * RW) + ty * RW + tx;
            const float* wk = w + (size_t)(k0 + kk) * CIN * 49;          // uniform: scalar loads
            float g[4][4];
#pragma unroll
            for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) g[dy][dx] = tp[dy * RW + dx];
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int dy = 0; dy < 4; ++dy) {
                        const int ky = py + 5 - 2 * dy;
                        if (ky < 0 || ky > 6) continue;
#pragma unroll
                        for (int px = 0; px < 2; ++px)
#pragma unroll
                            for (int dx = 0; dx < 4; ++dx) {
                                const int kx = px + 5 - 2 * dx;
                                if (kx < 0 || kx > 6) continue;
                                acc[py][px][c] = fmaf(g[dy][dx], wk[c * 49 + ky * 7 + kx], acc[py][px][c]);
                            }
                    }
        }
    }
    if (KS > 1) {       // partial sums of waves 1 .. KS-1 to LDS, added by wave 0 in wave order
        if (grp > 0) {
            float* rp = red + ((grp - 1) * NG + lt) * 4 * CIN;
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px)
#pragma unroll
                    for (int c = 0; c < CIN; ++c) rp[(py * 2 + px) * CIN + c] = acc[py][px][c];
        }
        __syncthreads();
        if (grp > 0) return;
        for (int g2 = 0; g2 < KS - 1; ++g2) {
            const float* rp = red + (g2 * NG + lt) * 4 * CIN;
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px)
#pragma unroll
                    for (int c = 0; c < CIN; ++c) acc[py][px][c] += rp[(py * 2 + px) * CIN + c];
        }
    }
    const int Y = Y0 + ty, X = X0 + tx;
    if (Y < Yend && X < Xend) {
        const int H = 2 * Ho, W = 2 * Wo;
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            float* op = gxo + (((size_t)b * CIN + c) * H + 2 * Y) * W + 2 * X;
            *reinterpret_cast<float2*>(op) = make_float2(acc[0][0][c], acc[0][1][c]);
            *reinterpret_cast<float2*>(op + W) = make_float2(acc[1][0][c], acc[1][1][c]);
        }
    }
}

}  // namespace

extern "C" {

static int launch_stem_bwd(const float* g_y, const float* w, int B, int K, int Cin, int H, int W, const StemWin& sw_,
                           float* g_x, void* stream, const char* fn) {
    const int Ho = H / 2, Wo = W / 2;
    // few workgroups (a window): four channel groups per workgroup, a quarter of the rows
    const int gx = (sw_.ww + BX - 1) / BX;
    const bool split = (long long)B * gx * ((sw_.wh + BY - 1) / BY) < 2048;
    const int by = split ? BY / 4 : BY, gyb = (sw_.wh + by - 1) / by;
    const long long blocks = (long long)B * gx * gyb;
    if (blocks >= (1ll << 31)) return fail(DMH_EINVAL, "%s: grid too large", fn);
    hipStream_t st = (hipStream_t)stream;
#define DMH_LAUNCH(CIN)                                                                                                          \
    do {                                                                                                                         \
        if (split)                                                                                                               \
            hipLaunchKernelGGL((stem_conv_bwd_kernel<CIN, 4>), dim3((unsigned)blocks), dim3(NT), 0, st, g_y, w, K, Ho, Wo, gx,   \
                               gyb, sw_, g_x);                                                                                   \
        else                                                                                                                     \
            hipLaunchKernelGGL((stem_conv_bwd_kernel<CIN, 1>), dim3((unsigned)blocks), dim3(NT), 0, st, g_y, w, K, Ho, Wo, gx,   \
                               gyb, sw_, g_x);                                                                                   \
    } while (0)
    if (Cin == 1) DMH_LAUNCH(1);
    else if (Cin == 2) DMH_LAUNCH(2);
    else if (Cin == 3) DMH_LAUNCH(3);
    else DMH_LAUNCH(4);
#undef DMH_LAUNCH
    return check_launch(fn);
}

int dmh_conv7x7s2_bwd_data(const float* g_y, const float* w, int B, int K, int Cin, int H, int W, float* g_x,
                           void* stream) {
    DMH_REQUIRE(g_y && w && g_x, "null pointer");
    DMH_REQUIRE(B > 0 && K > 0 && K % KC == 0, "gradient channel count must be a multiple of 8");
    DMH_REQUIRE(Cin >= 1 && Cin <= 4, "1 to 4 image channels");
    DMH_REQUIRE(H >= 2 && W >= 2 && (H & 1) == 0 && (W & 1) == 0, "image height and width must be even");
    const int Ho = H / 2, Wo = W / 2;
    DMH_REQUIRE((int64_t)K * Ho * Wo < ((int64_t)1 << 31), "image too large");
    StemWin sw_;
    sw_.win_org = nullptr; sw_.gy_org = nullptr; sw_.wh = Ho; sw_.ww = Wo; sw_.sh = Ho; sw_.sw = Wo;
    return launch_stem_bwd(g_y, w, B, K, Cin, H, W, sw_, g_x, stream, "dmh_conv7x7s2_bwd_data");
}

int dmh_conv7x7s2_bwd_data_win(const float* g_y, const float* w, const int* img_org, const int* gy_org, int B, int K, int Cin,
                               int H, int W, int hwin, int wwin, int sh, int sw, float* g_x, void* stream) {
    DMH_REQUIRE(g_y && w && g_x && img_org && gy_org, "null pointer");
    DMH_REQUIRE(B > 0 && K > 0 && K % KC == 0, "gradient channel count must be a multiple of 8");
    DMH_REQUIRE(Cin >= 1 && Cin <= 4, "1 to 4 image channels");
    DMH_REQUIRE(H >= 2 && W >= 2 && (H & 1) == 0 && (W & 1) == 0, "image height and width must be even");
    DMH_REQUIRE(hwin >= 2 && wwin >= 2 && (hwin & 1) == 0 && (wwin & 1) == 0 && hwin <= H && wwin <= W,
                "the image window must be even-sized and inside the image");
    DMH_REQUIRE(sh >= 1 && sw >= 1 && sh <= H / 2 && sw <= W / 2 && (int64_t)K * sh * sw < ((int64_t)1 << 31),
                "the g_y window must lie inside its frame");
    StemWin sw_;
    sw_.win_org = img_org; sw_.gy_org = gy_org; sw_.wh = hwin / 2; sw_.ww = wwin / 2; sw_.sh = sh; sw_.sw = sw;
    return launch_stem_bwd(g_y, w, B, K, Cin, H, W, sw_, g_x, stream, "dmh_conv7x7s2_bwd_data_win");
}

}  // extern "C"
