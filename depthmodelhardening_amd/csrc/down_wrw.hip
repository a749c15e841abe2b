// K20 -- weight gradients of the convolutions that open a down-sampling ResNet block: conv1 (3x3, stride 2, pad 1) and the
// shortcut's 1x1 stride-2 convolution of the SAME input (torchvision BasicBlock of layer2.0 / layer3.0 / layer4.0 under
// MD2/networks/resnet_encoder.py:94-98) -- the train-pass counterpart of K15, one launch for both filters:
//     dW3[k][c][ky][kx] = sum_{b,oy,ox} g3[b][k][oy][ox] * x[b][c][2 oy - 1 + ky][2 ox - 1 + kx]        (0 outside the image)
//     dWd[k][c]         = sum_{b,oy,ox} gd[b][k][oy][ox] * x[b][c][2 oy][2 ox]
// MIOpen served these with NHWC implicit GEMMs that accumulate with float atomics (the last reason a trained step was not
// reproducible in its low bits) between layout transposes.  Here, as in K16 / K18, the PIXEL axis is the reduction dimension of
// the fp32 MFMA and every sum has a fixed order:
//   * a workgroup owns a 64 x 64 (output x input channel) block of all ten filter taps (nine of the 3x3 + the 1x1, whose
//     pixel IS the 3x3's centre tap: one more MFMA on an operand already read) and walks a contiguous slice of the row tiles
//     (image, output row, 32 output pixels); wave (kq, cq) holds its 32 x 32 quadrant of the ten blocks in 160 accumulator
//     registers for the whole slice;
//   * per pixel pair: D_t[k][c] += G[k][2 px] X_t[2 px][c] on v_mfma_f32_32x32x2_f32 -- two gradient reads, nine input reads,
//     ten MFMAs; tiles of g3 / gd (64 x 32) and x (64 channels x 3 rows x 68 columns, zero outside the image) in LDS, the
//     next tile prefetched into registers while the current one is multiplied;
//   * one partial block set per workgroup; down_wrw_reduce_kernel adds the slices in order.  No atomics.
#include "common.hpp"

using namespace dmh;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;
constexpr int TP = 32;                      // output pixels per row tile
constexpr int RW = 2 * TP + 4;              // staged input columns: aligned 16-byte words from column 2 ox0 - 4
constexpr int XP = 3 * RW + 1;              // input plane pitch (odd: the 32 channels of an operand hit different banks)
constexpr int GP = TP + 1;                  // gradient row pitch
constexpr int XW = 64 * 3 * (RW / 4);       // 16-byte words of an input tile
constexpr int GW = 64 * (TP / 4);           // 16-byte words of one gradient tile
constexpr int XPT = (XW + NT - 1) / NT;     // words per thread
constexpr int GPT = GW / NT;

struct DArgs {
    const float* x;
    const float* g3;
    const float* gd;        // may be null: 3x3 only
    float* part;            // [pairs][S][10][64][64]
    int B, C, K, H, W, Ho, Wo;
    int cbn;                // C / 64
    int S;                  // pixel slices per (k-block, c-block) pair
    int tiles_x, ntiles;    // row tiles per output row, in all
};

__global__ __launch_bounds__(NT) void down_wrw_kernel(const DArgs a) {
    extern __shared__ float smem[];              // 69 KB: input tile, then the two gradient tiles
    float* xl = smem;
    float* g3l = smem + 64 * XP;
    float* gdl = g3l + 64 * GP;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pair = blockIdx.x / a.S, s = blockIdx.x - pair * a.S;
    const int kb = pair / a.cbn, cb = pair - kb * a.cbn;
    const int t_begin = (int)((long long)a.ntiles * s / a.S), t_end = (int)((long long)a.ntiles * (s + 1) / a.S);
    const int kq = wv >> 1, cq = wv & 1, li = lane & 31, lh = lane >> 5;
    const size_t HW = (size_t)a.H * a.W, HWo = (size_t)a.Ho * a.Wo;
    const bool with_d = a.gd != nullptr;

    f32x16 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

    float4 xr[XPT], gr[GPT], hr[GPT];
    auto fetch = [&](int tile) {
        int q = tile;
        const int oxb = q % a.tiles_x;  q /= a.tiles_x;
        const int oy = q % a.Ho, b = q / a.Ho;
        const int ox0 = oxb * TP;
        const float* xb = a.x + ((size_t)b * a.C + (size_t)cb * 64) * HW;
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int e = tid + NT * k;
            const int c = e / (3 * (RW / 4)), rem = e - c * (3 * (RW / 4)), r = rem / (RW / 4), wq = rem - r * (RW / 4);
            const int iy = 2 * oy - 1 + r, ix = 2 * ox0 - 4 + 4 * wq;          // a word lies inside or outside the row as a whole
            const bool ok = e < XW && iy >= 0 && iy < a.H && ix >= 0 && ix + 3 < a.W;   // (W is a multiple of 4)
            xr[k] = ok ? *reinterpret_cast<const float4*>(xb + (size_t)c * HW + (size_t)iy * a.W + ix) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const size_t gbase = ((size_t)b * a.K + (size_t)kb * 64) * HWo + (size_t)oy * a.Wo + ox0;
#pragma unroll
        for (int k = 0; k < GPT; ++k) {
            const int e = tid + NT * k;
            const int kk = e / (TP / 4), wq = e - kk * (TP / 4);
            const bool ok = ox0 + 4 * wq + 3 < a.Wo;                            // (Wo is a multiple of 4)
            const size_t o = gbase + (size_t)kk * HWo + 4 * wq;
            gr[k] = ok ? *reinterpret_cast<const float4*>(a.g3 + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            hr[k] = (ok && with_d) ? *reinterpret_cast<const float4*>(a.gd + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int e = tid + NT * k;
            if (e < XW) {
                const int c = e / (3 * (RW / 4)), rem = e - c * (3 * (RW / 4));       // rem = r * (RW/4) + wq: row-major inside the plane
                float* d = xl + c * XP + 4 * rem;
                d[0] = xr[k].x; d[1] = xr[k].y; d[2] = xr[k].z; d[3] = xr[k].w;
            }
        }
#pragma unroll
        for (int k = 0; k < GPT; ++k) {
            const int e = tid + NT * k;
            const int kk = e / (TP / 4), wq = e - kk * (TP / 4);
            float* d = g3l + kk * GP + 4 * wq;
            d[0] = gr[k].x; d[1] = gr[k].y; d[2] = gr[k].z; d[3] = gr[k].w;
            float* f = gdl + kk * GP + 4 * wq;
            f[0] = hr[k].x; f[1] = hr[k].y; f[2] = hr[k].z; f[3] = hr[k].w;
        }
    };

    if (t_begin < t_end) fetch(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();                        // the previous tile's operands have been read
        stage();
        __syncthreads();
        if (tile + 1 < t_end) fetch(tile + 1);  // in flight while this tile is multiplied
        // lane = (row / column li of the 32 x 32 block, k-step lh = which pixel of the pair)
        const float* ga = g3l + (32 * kq + li) * GP + lh;
        const float* da = gdl + (32 * kq + li) * GP + lh;
        // input column of output pixel p for tap kx: 2 p - 1 + kx, staged from column 2 ox0 - 4: index 2 p + kx + 3
        const float* xa = xl + (32 * cq + li) * XP + 2 * lh + 3;
#pragma unroll 4
        for (int pp = 0; pp < TP / 2; ++pp) {
            const float A3 = ga[2 * pp], Ad = da[2 * pp];
            const float* xp = xa + 4 * pp;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float Bv = xp[ky * RW + kx];
                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(A3, Bv, acc[ky * 3 + kx], 0, 0, 0);
                    if (ky == 1 && kx == 1) acc[9] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ad, Bv, acc[9], 0, 0, 0);
                }
        }
    }
    // D[i][n]: lane holds column n = li, rows i = 8 (v / 4) + 4 lh + v % 4
    float* pb = a.part + (size_t)blockIdx.x * 10 * 64 * 64;
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int i = 8 * (v >> 2) + 4 * lh + (v & 3);
            pb[((size_t)t * 64 + 32 * kq + i) * 64 + 32 * cq + li] = acc[t][v];
        }
}

// dW3[k][c][t] / dWd[k][c] = sum over the S slices, in order.  Threads run over (tap, k, c) with c fastest: the 84 MB of
// partials are read in full lines, the (small) filters are written with a stride of nine floats
__global__ __launch_bounds__(NT) void down_wrw_reduce_kernel(const float* __restrict__ part, int K, int C, int S,
                                                            float* __restrict__ dw3, float* __restrict__ dwd) {
    const int e = blockIdx.x * NT + threadIdx.x;
    const int KC = K * C;
    if (e >= KC * 10) return;
    const int t = e / KC, kc = e - t * KC, k = kc / C, c = kc - k * C;
    const int pair = (k >> 6) * (C >> 6) + (c >> 6);
    const float* p = part + ((size_t)pair * S * 10 + t) * 4096 + (size_t)(k & 63) * 64 + (c & 63);
    float sum = 0.f;
    for (int s = 0; s < S; ++s) sum += p[(size_t)s * 10 * 4096];
    if (t < 9) dw3[(size_t)kc * 9 + t] = sum;
    else if (dwd) dwd[kc] = sum;
}

int slices_for(int pairs, int ntiles) {
    int S = 512 / pairs;
    if (S < 1) S = 1;
    if (S > ntiles) S = ntiles;
    return S;
}

}  // namespace

extern "C" {

int64_t dmh_down_wrw_workspace_size(int B, int C, int K, int H, int W) {
    if (B <= 0 || C <= 0 || K <= 0 || C % 64 || K % 64 || H < 2 || W < 8 || (H & 1) || (W & 7)) return -1;
    const int pairs = (K / 64) * (C / 64), ntiles = B * (H / 2) * ((W / 2 + TP - 1) / TP);
    return (int64_t)pairs * slices_for(pairs, ntiles) * 10 * 64 * 64;
}

int dmh_down_wrw(const float* x, const float* g3, const float* gd, int B, int C, int K, int H, int W, float* workspace, float* dw3,
                 float* dwd, void* stream) {
    DMH_REQUIRE(x && g3 && workspace && dw3 && (dwd || !gd), "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && K > 0 && C % 64 == 0 && K % 64 == 0, "both channel counts must be multiples of 64");
    DMH_REQUIRE(H >= 2 && W >= 8 && (H & 1) == 0 && (W & 7) == 0, "input height must be even, width a multiple of 8");
    DMH_REQUIRE((int64_t)B * C * H * W < ((int64_t)1 << 40) && (int64_t)K * C * 10 < ((int64_t)1 << 31), "tensor too large");
    DArgs a;
    a.x = x; a.g3 = g3; a.gd = gd; a.part = workspace;
    a.B = B; a.C = C; a.K = K; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2;
    a.cbn = C / 64;
    a.tiles_x = (a.Wo + TP - 1) / TP;
    const int64_t ntiles = (int64_t)B * a.Ho * a.tiles_x;
    DMH_REQUIRE(ntiles < ((int64_t)1 << 30), "too many row tiles");
    a.ntiles = (int)ntiles;
    const int pairs = (K / 64) * (C / 64);
    a.S = slices_for(pairs, a.ntiles);
    constexpr size_t smem = (size_t)(64 * XP + 2 * 64 * GP) * sizeof(float);
    static std::atomic<uint64_t> configured{0};     // per device, see configure_dynamic_lds
    if (configure_dynamic_lds(down_wrw_kernel, smem, configured) != hipSuccess)
        return fail(DMH_ELAUNCH, "%s: cannot raise the dynamic LDS limit", "dmh_down_wrw");
    hipLaunchKernelGGL(down_wrw_kernel, dim3((unsigned)(pairs * a.S)), dim3(NT), smem, (hipStream_t)stream, a);
    hipLaunchKernelGGL(down_wrw_reduce_kernel, dim3((unsigned)(((int64_t)K * C * 10 + NT - 1) / NT)), dim3(NT), 0,
                       (hipStream_t)stream, workspace, K, C, a.S, dw3, gd ? dwd : nullptr);
    return check_launch("dmh_down_wrw");
}

}  // extern "C"
