// Decoder glue -- the element-wise passes between the MIOpen convolutions of the Monodepth2 depth decoder, fused.
//
// Reference: MD2/networks/depth_decoder.py:51-63 runs, per decoder stage, ELU (in place) -> nearest x2 upsample ->
// torch.cat with the skip feature -> ReflectionPad2d(1) (inside Conv3x3, MD2/layers.py:133-136) as four separate
// full-tensor passes, and pads the same ELU output twice where it feeds both the next stage and a disparity head.
// On MI355X these copies are ~30 % of the adversarial-training step (profiles/r01_bench_timed_region.csv:
// reflection_pad2d fwd+bwd alone 12 %).  Here each stage boundary is ONE pass:
//
//   up_cat_pad :  out[B, C1+C2, 2h+2, 2w+2] = pad1_reflect( cat( up2_nearest( ELU(y) ), skip ) )
//   elu_pad    :  out[B, C,  H+2,  W+2]     = pad1_reflect( ELU(z) )          (apply_elu = 0: pad only)
//
// and the convolutions run with padding = 0 on the padded tensors.  Backward kernels are gathers (the adjoint of the
// reflection pad folds the border rows/columns back onto rows 1 and H-2): deterministic, no atomics.
// Pure HBM streaming: one read + one write per element, 64-lane coalesced rows.
#include "common.hpp"

using namespace dmh;

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : expf(x) - 1.f; }
__device__ __forceinline__ float elu_grad(float x) { return x > 0.f ? 1.f : expf(x); }

// sum of the padded-gradient entries that ReflectionPad2d(1) reads from interior position (Y, X)
__device__ __forceinline__ float fold2d(const float* __restrict__ gp, int Y, int X, int H, int W) {
    const int PW = W + 2;
    int rows[3], cols[3], nr = 1, nc = 1;
    rows[0] = Y + 1;
    cols[0] = X + 1;
    if (Y == 1) rows[nr++] = 0;
    if (Y == H - 2) rows[nr++] = H + 1;
    if (X == 1) cols[nc++] = 0;
    if (X == W - 2) cols[nc++] = W + 1;
    float acc = 0.f;
    for (int a = 0; a < nr; ++a)
        for (int b = 0; b < nc; ++b) acc += gp[rows[a] * PW + cols[b]];
    return acc;
}

__global__ __launch_bounds__(NT) void up_cat_pad_fwd_kernel(const float* __restrict__ y, const float* __restrict__ skip,
                                                            int C1, int C2, int h, int w, float* __restrict__ out,
                                                            int64_t total) {
    const int H = 2 * h, W = 2 * w, PH = H + 2, PW = W + 2, C = C1 + C2;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * NT) {
        const int px = (int)(idx % PW);
        const int py = (int)((idx / PW) % PH);
        const int c = (int)((idx / ((int64_t)PW * PH)) % C);
        const int b = (int)(idx / ((int64_t)PW * PH * C));
        const int Y = reflect_idx(py - 1, H), X = reflect_idx(px - 1, W);
        float v;
        if (c < C1)
            v = elu_f(y[(((int64_t)b * C1 + c) * h + (Y >> 1)) * w + (X >> 1)]);
        else
            v = skip[(((int64_t)b * C2 + (c - C1)) * H + Y) * W + X];
        out[idx] = v;
    }
}

__global__ __launch_bounds__(NT) void up_cat_pad_bwd_kernel(const float* __restrict__ y, const float* __restrict__ g_out,
                                                            int C1, int C2, int h, int w, float* __restrict__ g_y,
                                                            float* __restrict__ g_skip, int64_t n_y, int64_t n_skip) {
    const int H = 2 * h, W = 2 * w, C = C1 + C2;
    const int64_t plane = (int64_t)(H + 2) * (W + 2);
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < n_y + n_skip; idx += (int64_t)gridDim.x * NT) {
        if (idx < n_y) {
            const int j = (int)(idx % w);
            const int i = (int)((idx / w) % h);
            const int c = (int)((idx / ((int64_t)w * h)) % C1);
            const int b = (int)(idx / ((int64_t)w * h * C1));
            const float* gp = g_out + ((int64_t)b * C + c) * plane;
            const float acc = fold2d(gp, 2 * i, 2 * j, H, W) + fold2d(gp, 2 * i, 2 * j + 1, H, W) +
                              fold2d(gp, 2 * i + 1, 2 * j, H, W) + fold2d(gp, 2 * i + 1, 2 * j + 1, H, W);
            g_y[idx] = acc * elu_grad(y[idx]);
        } else if (g_skip) {
            const int64_t k = idx - n_y;
            const int X = (int)(k % W);
            const int Y = (int)((k / W) % H);
            const int c = (int)((k / ((int64_t)W * H)) % C2);
            const int b = (int)(k / ((int64_t)W * H * C2));
            g_skip[k] = fold2d(g_out + ((int64_t)b * C + C1 + c) * plane, Y, X, H, W);
        }
    }
}

__global__ __launch_bounds__(NT) void elu_pad_fwd_kernel(const float* __restrict__ z, int C, int H, int W, int apply_elu,
                                                         float* __restrict__ out, int64_t total) {
    const int PH = H + 2, PW = W + 2;
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * NT) {
        const int px = (int)(idx % PW);
        const int py = (int)((idx / PW) % PH);
        const int64_t bc = idx / ((int64_t)PW * PH);
        const float v = z[(bc * H + reflect_idx(py - 1, H)) * W + reflect_idx(px - 1, W)];
        out[idx] = apply_elu ? elu_f(v) : v;
    }
}

__global__ __launch_bounds__(NT) void elu_pad_bwd_kernel(const float* __restrict__ z, const float* __restrict__ g_out,
                                                         int H, int W, int apply_elu, float* __restrict__ g_z,
                                                         int64_t total) {
    const int64_t plane = (int64_t)(H + 2) * (W + 2);
    for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * NT) {
        const int X = (int)(idx % W);
        const int Y = (int)((idx / W) % H);
        const int64_t bc = idx / ((int64_t)W * H);
        const float g = fold2d(g_out + bc * plane, Y, X, H, W);
        g_z[idx] = apply_elu ? g * elu_grad(z[idx]) : g;
    }
}

inline int grid_for(int64_t n) {
    const int64_t b = (n + NT - 1) / NT;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" {

int dmh_dec_up_cat_pad_fwd(const float* y, const float* skip, int B, int C1, int C2, int h, int w, float* out,
                           void* stream) {
    DMH_REQUIRE(y && out && (skip || C2 == 0), "null pointer");
    DMH_REQUIRE(B > 0 && C1 > 0 && C2 >= 0 && h >= 1 && w >= 1, "bad sizes");
    const int64_t total = (int64_t)B * (C1 + C2) * (2 * h + 2) * (2 * w + 2);
    hipLaunchKernelGGL(up_cat_pad_fwd_kernel, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, y, skip, C1, C2, h,
                       w, out, total);
    return check_launch("dmh_dec_up_cat_pad_fwd");
}

int dmh_dec_up_cat_pad_bwd(const float* y, const float* g_out, int B, int C1, int C2, int h, int w, float* g_y,
                           float* g_skip, void* stream) {
    DMH_REQUIRE(y && g_out && g_y, "null pointer");
    DMH_REQUIRE(B > 0 && C1 > 0 && C2 >= 0 && h >= 1 && w >= 1, "bad sizes");
    const int64_t n_y = (int64_t)B * C1 * h * w, n_skip = g_skip ? (int64_t)B * C2 * 4 * h * w : 0;
    hipLaunchKernelGGL(up_cat_pad_bwd_kernel, dim3(grid_for(n_y + n_skip)), dim3(NT), 0, (hipStream_t)stream, y, g_out,
                       C1, C2, h, w, g_y, g_skip, n_y, n_skip);
    return check_launch("dmh_dec_up_cat_pad_bwd");
}

int dmh_elu_pad_fwd(const float* z, int B, int C, int H, int W, int apply_elu, float* out, void* stream) {
    DMH_REQUIRE(z && out, "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && H >= 2 && W >= 2, "bad sizes");
    const int64_t total = (int64_t)B * C * (H + 2) * (W + 2);
    hipLaunchKernelGGL(elu_pad_fwd_kernel, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, z, C, H, W, apply_elu,
                       out, total);
    return check_launch("dmh_elu_pad_fwd");
}

int dmh_elu_pad_bwd(const float* z, const float* g_out, int B, int C, int H, int W, int apply_elu, float* g_z,
                    void* stream) {
    DMH_REQUIRE(z && g_out && g_z, "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && H >= 2 && W >= 2, "bad sizes");
    const int64_t total = (int64_t)B * C * H * W;
    hipLaunchKernelGGL(elu_pad_bwd_kernel, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, z, g_out, H, W,
                       apply_elu, g_z, total);
    return check_launch("dmh_elu_pad_bwd");
}

}  // extern "C"
