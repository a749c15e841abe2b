// K21 -- weight gradient of the encoder's first convolution on the normalised image:
//     dW[k][c][ky][kx] = sum_{b,oy,ox} g[b][k][oy][ox] * xn[b][c][2 oy - 3 + ky][2 ox - 3 + kx],
//     xn = (x - mean) / std inside the image, 0 in the padding
// (nn.Conv2d(3, 64, 7, stride 2, padding 3) after `(input_image - 0.45) / 0.225`, MD2/networks/resnet_encoder.py:89-90; the
// train pass only).  MIOpen's kernel for it accumulates 2.6 M pixels per tap with float atomics (1.26 ms at batch 32, not
// reproducible in the low bits; its deterministic mode takes 396 ms).  Same scheme as K20 / K16: the PIXEL axis is the
// reduction dimension of v_mfma_f32_32x32x2_f32,
//     D[k 64][tap 160] += G[k][2 px] * X[2 px][tap]        (147 taps, padded to five blocks of 32)
//   * a workgroup owns a slice of the row tiles (image, 4 output rows, 32 output pixels); wave w takes output row w of every
//     tile and accumulates ALL ten 32 x 32 blocks (2 channel halves x 5 tap blocks = 160 accumulator registers) over its
//     pixels -- four independent partial sums per workgroup;
//   * the normalised image tile (3 channels x 13 rows x 72 columns, zero outside the image: K14's staging) and the gradient
//     tile (64 channels x 4 rows x 32 pixels) sit in LDS; an X operand is one ds_read_b32 at (pixel base) + (the lane's tap
//     offset, five per lane, fixed for the whole kernel); a pixel pair = two gradient reads, five image reads, ten MFMAs;
//   * stem_wrw_reduce_kernel adds the (workgroup, wave) partials in order.  No atomics: bitwise reproducible.
#include "common.hpp"

using namespace dmh;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;
constexpr int TR = 4, TC = 32;              // output rows (one per wave) x pixels per tile
constexpr int RH = 2 * TR + 5;              // 13 input rows
constexpr int RW = 2 * TC + 8;              // 72 staged columns: aligned 16-byte words from column 2 ox0 - 4 (69 used)
constexpr int PLANE = RH * RW;              // 936 floats per channel
constexpr int XW = 3 * RH * (RW / 4);       // 702 words of an image tile
constexpr int GP = TR * TC + 1;             // gradient row pitch (129)
constexpr int GW = 64 * TR * (TC / 4);      // 2048 words of a gradient tile
constexpr int XPT = (XW + NT - 1) / NT, GPT = GW / NT;
constexpr int NTAP = 147, NBLK = 5;

struct TArgs {
    const float* x;
    const float* g;
    float* part;            // [workgroups][4 waves][10 blocks][32][32]
    int B, H, W, Ho, Wo, tiles_x, tiles_y, ntiles;
    float mean, inv_std;
};

__global__ __launch_bounds__(NT) void stem_wrw_kernel(const TArgs a) {
    __shared__ float xl[3 * PLANE];
    __shared__ float gl[64 * GP];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int t_begin = (int)((long long)a.ntiles * blockIdx.x / gridDim.x);
    const int t_end = (int)((long long)a.ntiles * (blockIdx.x + 1) / gridDim.x);
    const size_t HW = (size_t)a.H * a.W, HWo = (size_t)a.Ho * a.Wo;

    // the lane's tap of each block: (c, ky, kx) = tap 32 blk + li -> offset inside the image tile; input column of output
    // pixel p for tap kx is 2 p - 3 + kx, staged from column 2 ox0 - 4: index 2 p + kx + 1; input row 2 (row in tile) + ky
    int toff[NBLK];
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
        const int tap = 32 * blk + li;
        const int c = tap / 49, r = tap - c * 49, ky = r / 7, kx = r - ky * 7;
        toff[blk] = tap < NTAP ? c * PLANE + ky * RW + kx + 1 : -1;
    }
    f32x16 acc[2 * NBLK];
#pragma unroll
    for (int t = 0; t < 2 * NBLK; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

    float4 xr[XPT], gr[GPT];
    auto fetch = [&](int tile) {
        int q = tile;
        const int bx = q % a.tiles_x;  q /= a.tiles_x;
        const int by = q % a.tiles_y, b = q / a.tiles_y;
        const int oy0 = by * TR, ox0 = bx * TC;
        const float* xb = a.x + (size_t)b * 3 * HW;
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int e = tid + NT * k;
            const int c = e / (RH * (RW / 4)), rem = e - c * (RH * (RW / 4)), r = rem / (RW / 4), wq = rem - r * (RW / 4);
            const int iy = 2 * oy0 - 3 + r, ix = 2 * ox0 - 4 + 4 * wq;
            const bool ok = e < XW && iy >= 0 && iy < a.H && ix >= 0 && ix + 3 < a.W;       // (W is a multiple of 4)
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                v = *reinterpret_cast<const float4*>(xb + (size_t)c * HW + (size_t)iy * a.W + ix);
                v = make_float4((v.x - a.mean) * a.inv_std, (v.y - a.mean) * a.inv_std, (v.z - a.mean) * a.inv_std,
                                (v.w - a.mean) * a.inv_std);
            }
            xr[k] = v;
        }
        const float* gb = a.g + (size_t)b * 64 * HWo;
#pragma unroll
        for (int k = 0; k < GPT; ++k) {
            const int e = tid + NT * k;
            const int kk = e / (TR * (TC / 4)), rem = e - kk * (TR * (TC / 4)), r = rem / (TC / 4), wq = rem - r * (TC / 4);
            const int oy = oy0 + r, ox = ox0 + 4 * wq;
            const bool ok = oy < a.Ho && ox + 3 < a.Wo;                                      // (Wo is a multiple of 4)
            gr[k] = ok ? *reinterpret_cast<const float4*>(gb + (size_t)kk * HWo + (size_t)oy * a.Wo + ox)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int e = tid + NT * k;
            if (e < XW) {
                float* d = xl + 4 * e;                  // word e of [c][r][wq]: planes are contiguous (PLANE = RH * RW)
                d[0] = xr[k].x; d[1] = xr[k].y; d[2] = xr[k].z; d[3] = xr[k].w;
            }
        }
#pragma unroll
        for (int k = 0; k < GPT; ++k) {
            const int e = tid + NT * k;
            const int kk = e / (TR * (TC / 4)), rem = e - kk * (TR * (TC / 4));             // rem = r * (TC/4) + wq
            float* d = gl + kk * GP + 4 * rem;
            d[0] = gr[k].x; d[1] = gr[k].y; d[2] = gr[k].z; d[3] = gr[k].w;
        }
    };

    if (t_begin < t_end) fetch(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();
        stage();
        __syncthreads();
        if (tile + 1 < t_end) fetch(tile + 1);
        // wave wv: output row wv of the tile; pixel pair (2 pp, 2 pp + 1), the lane's k-step lh picks the pixel
        const float* ga = gl + li * GP + wv * TC + lh;
        const float* gb2 = ga + 32 * GP;
        const float* xa = xl + (2 * wv) * RW + 2 * lh;
#pragma unroll 2
        for (int pp = 0; pp < TC / 2; ++pp) {
            const float A0 = ga[2 * pp], A1 = gb2[2 * pp];
            const float* xp = xa + 4 * pp;
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
                const float Bv = toff[blk] >= 0 ? xp[toff[blk]] : 0.f;
                acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, Bv, acc[blk], 0, 0, 0);
                acc[NBLK + blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, Bv, acc[NBLK + blk], 0, 0, 0);
            }
        }
    }
    // D[i][n]: lane holds column n = li (tap), rows i = 8 (v / 4) + 4 lh + v % 4 (channel within the half)
    float* pb = a.part + ((size_t)blockIdx.x * 4 + wv) * (2 * NBLK) * 1024;
#pragma unroll
    for (int t = 0; t < 2 * NBLK; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int i = 8 * (v >> 2) + 4 * lh + (v & 3);
            pb[(size_t)t * 1024 + i * 32 + li] = acc[t][v];
        }
}

// dW[k][tap] = sum over the (workgroup, wave) partials in a fixed order: a workgroup owns 32 consecutive outputs, its eight
// thread rows each add one eighth of the partials, thread row 0 adds the eight sums
__global__ __launch_bounds__(NT) void stem_wrw_reduce_kernel(const float* __restrict__ part, int nparts, float* __restrict__ dw) {
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, ch = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + o;                  // over 64 * 147 = 294 * 32
    const int k = e / NTAP, tap = e - k * NTAP;
    const int t = (k >> 5) * NBLK + (tap >> 5);
    const float* p = part + (size_t)t * 1024 + (k & 31) * 32 + (tap & 31);
    const int s0 = (int)((long long)nparts * ch / 8), s1 = (int)((long long)nparts * (ch + 1) / 8);
    float sum = 0.f;
    for (int s = s0; s < s1; ++s) sum += p[(size_t)s * (2 * NBLK) * 1024];
    red[ch][o] = sum;
    __syncthreads();
    if (ch == 0) {
        float tot = red[0][o];
#pragma unroll
        for (int j = 1; j < 8; ++j) tot += red[j][o];
        dw[e] = tot;
    }
}

int stem_groups(int ntiles) { return ntiles < 512 ? ntiles : 512; }

}  // namespace

extern "C" {

int64_t dmh_stem_wrw_workspace_size(int B, int H, int W) {
    if (B <= 0 || H < 2 || W < 8 || (H & 1) || (W & 7)) return -1;
    const int ntiles = B * ((H / 2 + TR - 1) / TR) * ((W / 2 + TC - 1) / TC);
    return (int64_t)stem_groups(ntiles) * 4 * (2 * NBLK) * 1024;
}

int dmh_stem_wrw(const float* x, const float* g, int B, int H, int W, float mean, float std, float* workspace, float* dw,
                 void* stream) {
    DMH_REQUIRE(x && g && workspace && dw, "null pointer");
    DMH_REQUIRE(B > 0 && H >= 2 && W >= 8 && (H & 1) == 0 && (W & 7) == 0, "image height must be even, width a multiple of 8");
    DMH_REQUIRE(std > 0.f, "std must be positive");
    TArgs a;
    a.x = x; a.g = g; a.part = workspace;
    a.B = B; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2;
    a.tiles_x = (a.Wo + TC - 1) / TC;
    a.tiles_y = (a.Ho + TR - 1) / TR;
    const int64_t ntiles = (int64_t)B * a.tiles_x * a.tiles_y;
    DMH_REQUIRE(ntiles < ((int64_t)1 << 30), "too many tiles");
    a.ntiles = (int)ntiles;
    a.mean = mean;
    a.inv_std = 1.0f / std;
    const int groups = stem_groups(a.ntiles);
    hipLaunchKernelGGL(stem_wrw_kernel, dim3((unsigned)groups), dim3(NT), 0, (hipStream_t)stream, a);
    static_assert(64 * NTAP % 32 == 0, "the reduce kernel's workgroups cover the filter exactly");
    hipLaunchKernelGGL(stem_wrw_reduce_kernel, dim3(64 * NTAP / 32), dim3(NT), 0, (hipStream_t)stream, workspace, groups * 4, dw);
    return check_launch("dmh_stem_wrw");
}

}  // extern "C"
