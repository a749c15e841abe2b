// K14 -- the encoder's first layer with its input normalisation fused, on the fp32 matrix cores:
//     y[b,k,oy,ox] = sum_{c,ky,kx} w[k,c,ky,kx] * xn[b,c,2oy-3+ky,2ox-3+kx],   xn = (x - mean) * inv_std inside the image,
//                                                                             0 in the padding
// i.e. MD2/networks/resnet_encoder.py:89-90  `x = (input_image - 0.45) / 0.225; x = self.encoder.conv1(x)`
// (torchvision ResNet.conv1: 7x7, stride 2, padding 3, 3 -> 64 channels, no bias) -- SURVEY.md section 8f rank 4, "fuse
// (x - 0.45)/0.225 + first 7x7/2 conv ... (first MFMA use)".  The image it reads is K3's output in every attack step.
// Replaces two element-wise passes over the image, MIOpen's NHWC implicit GEMM and its layout transposes.
//
// GEMM view: M = 64 output channels, N = pixels, K = 3*7*7 = 147 -> 74 pairs for v_mfma_f32_32x32x2_f32 (exact fp32).
//   * a workgroup (4 waves) owns 4 output rows x 32 output columns; the normalised input region (13 rows x 69 columns
//     x 3 channels, zero outside the image) is staged once in LDS; wave w computes output row w: 148 MFMAs
//     (74 k-pairs x 2 channel halves) for 32 pixels x 64 channels.
//   * B operand (im2col) = ONE ds_read_b32 per k-pair with an immediate offset: the 147 taps are ordered so that the two
//     taps of a pair differ by one of three fixed strides (next column: 63 pairs; next row at kx = 6: 9 pairs; next
//     channel at (6,6): 1 pair; the last tap pairs with a zero weight), and each lane keeps three base addresses
//     (its half-wave's second tap already added).  No address arithmetic in the loop.
//   * A operand = the whole filter in registers: 148 values per lane, loaded once; workgroups are persistent over tiles.
//   * output: a 32x32 MFMA result gives each lane 16 channels of one pixel: stores are 128-byte row segments.
// MFMA-bound: 18.5 GFLOP at the attack shape (12 x 3 x 320 x 1024) = 118 us at the 157 TFLOP/s fp32 MFMA peak; measured
// 241 us = 77 TFLOP/s (ATen's normalisation + MIOpen's convolution: 447 us) -- tools/stem_bench.py.
#include "common.hpp"

using namespace dmh;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;
constexpr int TR = 4, TC = 32;                 // output rows (one per wave) x columns per workgroup tile
constexpr int RH = 2 * TR + 5;                 // 13 input rows
constexpr int RW = 2 * TC + 6;                 // 70 input columns (69 used + one finite pad column for the dummy tap)
constexpr int PLANE = RH * RW;                 // 910 floats per channel
constexpr int NPAIR = 74;

// tap of pair j, first element: (c, ky, kx) and the stride class of the second element
struct PairDesc { int off; int cls; int k0; int k1; };   // off: LDS offset of the first tap; k0/k1: filter tap index (c*49+ky*7+kx) or -1
__host__ __device__ constexpr PairDesc pair_desc(int j) {
    // 0..62: (c, ky, kx even pairs)   63..71: kx = 6, (ky, ky+1) pairs   72: (c=0,c=1) at (6,6)   73: (c=2, 6, 6) + dummy
    if (j < 63) {
        const int c = j / 21, r = j % 21, ky = r / 3, kx = 2 * (r % 3);
        return {c * PLANE + ky * RW + kx, 0, c * 49 + ky * 7 + kx, c * 49 + ky * 7 + kx + 1};
    }
    if (j < 72) {
        const int q = j - 63, c = q / 3, ky = 2 * (q % 3);
        return {c * PLANE + ky * RW + 6, 1, c * 49 + ky * 7 + 6, c * 49 + (ky + 1) * 7 + 6};
    }
    if (j == 72) return {6 * RW + 6, 2, 0 * 49 + 48, 1 * 49 + 48};
    return {2 * PLANE + 6 * RW + 6, 0, 2 * 49 + 48, -1};
}

struct SArgs {
    const float* x;
    const float* w;
    float* y;
    int B, H, W, Ho, Wo, gx, gy, ntiles;
    float mean, inv_std;
    // window form (K19): only the hw x ww window of the Ho x Wo output at the per-sample origin org [B,2] is computed, into
    // a compact [B,64,hw,ww] tensor; org == NULL: the whole output
    const int* org;
    int hw, ww;
};

__global__ __launch_bounds__(NT, 2) void stem_conv_fwd_kernel(const SArgs a) {
    __shared__ float tile[3 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, h = lane >> 5;

    // ---- the filter, once: lane supplies A[i = n (+32)][k = h] of every pair
    float wA[NPAIR], wB[NPAIR];
#pragma unroll
    for (int j = 0; j < NPAIR; ++j) {
        const PairDesc d = pair_desc(j);
        const int kk = h ? d.k1 : d.k0;
        wA[j] = kk >= 0 ? a.w[(size_t)n * 147 + kk] : 0.f;
        wB[j] = kk >= 0 ? a.w[(size_t)(n + 32) * 147 + kk] : 0.f;
    }
    // per-lane LDS bases (floats): pixel n of output row wv sits at input row 2*wv, column 2*n of the region
    const int base = (2 * wv) * RW + 2 * n;
    const float* b0 = tile + base + (h ? 1 : 0);
    const float* b1 = tile + base + (h ? RW : 0);
    const float* b2 = tile + base + (h ? PLANE : 0);

    const size_t HW = (size_t)a.H * a.W, HWo = (size_t)a.hw * a.ww;
    for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        int q = t;
        const int bxi = q % a.gx;  q /= a.gx;
        const int byi = q % a.gy;
        const int b = q / a.gy;
        const int wy0 = a.org ? a.org[2 * b] : 0, wx0 = a.org ? a.org[2 * b + 1] : 0;
        const int oy0 = wy0 + byi * TR, ox0 = wx0 + bxi * TC;
        const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
        const float* xb = a.x + (size_t)b * 3 * HW;
        __syncthreads();                              // the previous tile's reads are done
        // (a register-staged double buffer was measured slower: 148 filter registers + 32 accumulators leave no room for
        //  the 11 staging registers -- scratch spills, 344 us instead of 241 us; two workgroups per CU overlap instead)
        for (int e = tid; e < 3 * PLANE; e += NT) {
            const int c = e / PLANE, rem = e - c * PLANE, r = rem / RW, cc = rem - r * RW;
            const int iy = iy0 + r, ix = ix0 + cc;
            const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const float v = xb[(size_t)c * HW + (size_t)min(max(iy, 0), a.H - 1) * a.W + min(max(ix, 0), a.W - 1)];
            tile[e] = ok ? (v - a.mean) * a.inv_std : 0.f;
        }
        __syncthreads();
        f32x16 accA, accB;
#pragma unroll
        for (int v = 0; v < 16; ++v) accA[v] = accB[v] = 0.f;
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) {
            const PairDesc d = pair_desc(j);
            const float bv = (d.cls == 0 ? b0 : d.cls == 1 ? b1 : b2)[d.off];
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[j], bv, accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(wB[j], bv, accB, 0, 0, 0);
        }
        // D[i][n]: lane holds column n, rows i = 8*(v/4) + 4*h + v%4
        const int oy = oy0 + wv - wy0, ox = ox0 + n - wx0;        // inside the (compact) output plane
        if (oy < a.hw && ox < a.ww) {
            float* yb = a.y + (size_t)b * 64 * HWo + (size_t)oy * a.ww + ox;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = 8 * (v >> 2) + 4 * h + (v & 3);
                yb[(size_t)i * HWo] = accA[v];
                yb[(size_t)(i + 32) * HWo] = accB[v];
            }
        }
    }
}

}  // namespace

extern "C" {

static int launch_stem_fwd(const float* x, const float* w, const int* org, int B, int H, int W, int hw, int ww, float mean,
                           float std, float* y, void* stream, const char* fn) {
    SArgs a;
    a.x = x;
    a.w = w;
    a.y = y;
    a.B = B;
    a.H = H;
    a.W = W;
    a.Ho = H / 2;
    a.Wo = W / 2;
    a.org = org;
    a.hw = hw;
    a.ww = ww;
    a.gx = (ww + TC - 1) / TC;
    a.gy = (hw + TR - 1) / TR;
    const long long tiles = (long long)B * a.gx * a.gy;
    if (tiles >= (1ll << 31)) return fail(DMH_EINVAL, "%s: grid too large", fn);
    a.ntiles = (int)tiles;
    a.mean = mean;
    a.inv_std = 1.0f / std;
    const int blocks = (int)(tiles < 512 ? tiles : 512);        // 2 workgroups per CU, persistent over the tiles
    hipLaunchKernelGGL(stem_conv_fwd_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, a);
    return check_launch(fn);
}

int dmh_stem_conv_norm_fwd(const float* x, const float* w, int B, int H, int W, float mean, float std, float* y,
                           void* stream) {
    DMH_REQUIRE(x && w && y, "null pointer");
    DMH_REQUIRE(B > 0 && H >= 2 && W >= 2 && (H & 1) == 0 && (W & 1) == 0, "image height and width must be even");
    DMH_REQUIRE(std > 0.f, "std must be positive");
    DMH_REQUIRE((int64_t)B * 64 * (H / 2) * (W / 2) < ((int64_t)1 << 40), "tensor too large");
    return launch_stem_fwd(x, w, nullptr, B, H, W, H / 2, W / 2, mean, std, y, stream, "dmh_stem_conv_norm_fwd");
}

int dmh_stem_conv_norm_fwd_win(const float* x, const float* w, const int* org, int B, int H, int W, int hw, int ww, float mean,
                               float std, float* y, void* stream) {
    DMH_REQUIRE(x && w && y && org, "null pointer");
    DMH_REQUIRE(B > 0 && H >= 2 && W >= 2 && (H & 1) == 0 && (W & 1) == 0, "image height and width must be even");
    DMH_REQUIRE(std > 0.f, "std must be positive");
    DMH_REQUIRE(hw >= 1 && ww >= 1 && hw <= H / 2 && ww <= W / 2, "the window must lie inside the H/2 x W/2 output");
    return launch_stem_fwd(x, w, org, B, H, W, hw, ww, mean, std, y, stream, "dmh_stem_conv_norm_fwd_win");
}

}  // extern "C"
