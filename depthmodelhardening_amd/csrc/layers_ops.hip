// Stand-alone forms of the two loss layers of MD2/layers.py, for callers that use them outside Trainer.compute_losses
// (which runs the fused K1 + K2 instead): the reference surface `SSIM()(x, y)` and `get_smooth_loss(disp, img)`.
//
//   ssim_map    : out = clamp((1 - SSIM_n / SSIM_d) / 2, 0, 1) per channel, 3x3 means over the ReflectionPad2d(1) image
//                 (MD2/layers.py:223-253).  Backward as in K1: three (x) + two (y) per-pixel coefficient fields whose 3x3
//                 box sums -- with the reflection pad's adjoint folded in -- give the gradient of every input pixel:
//                 deterministic gathers, no atomics.
//   edge_smooth : mean(|d_x disp| exp(-mean_c |d_x img|)) + mean(|d_y disp| exp(-mean_c |d_y img|))   (MD2/layers.py:207-220)
//                 two-stage fixed-order reduction; backward w.r.t. disp is a 4-neighbour gather.
#include "common.hpp"

using namespace dmh;

namespace {

constexpr int NT = 256;
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;

struct Win {
    float sx, sy, sxx, syy, sxy;
};

// 3x3 window sums of the reflection-padded planes around (i, j)
__device__ __forceinline__ Win window(const float* __restrict__ x, const float* __restrict__ y, int i, int j, int H, int W) {
    Win w{0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int di = -1; di <= 1; ++di) {
        const int r = reflect_idx(i + di, H);
#pragma unroll
        for (int dj = -1; dj <= 1; ++dj) {
            const int c = reflect_idx(j + dj, W);
            const float a = x[(size_t)r * W + c], b = y[(size_t)r * W + c];
            w.sx += a; w.sy += b; w.sxx += a * a; w.syy += b * b; w.sxy += a * b;
        }
    }
    return w;
}

__global__ __launch_bounds__(NT) void ssim_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int H, int W,
                                                      float* __restrict__ out) {
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= H * W) return;
    const size_t base = (size_t)blockIdx.y * H * W;
    const int i = t / W, j = t - i * W;
    const Win w = window(x + base, y + base, i, j, H, W);
    const float mx = w.sx / 9.f, my = w.sy / 9.f;
    const float vx = w.sxx / 9.f - mx * mx, vy = w.syy / 9.f - my * my, vxy = w.sxy / 9.f - mx * my;
    const float n = (2.f * mx * my + C1) * (2.f * vxy + C2), d = (mx * mx + my * my + C1) * (vx + vy + C2);
    out[base + t] = fminf(fmaxf((1.f - n / d) * 0.5f, 0.f), 1.f);
}

// pass 1 of the backward: coef[5][plane][H][W] = g * d out / d (mean_x, mean_y, E[xx], E[yy], E[xy]) at every window centre
__global__ __launch_bounds__(NT) void ssim_bwd_coef_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ g, int H, int W, size_t total,
                                                           float* __restrict__ coef) {
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= H * W) return;
    const size_t base = (size_t)blockIdx.y * H * W;
    const int i = t / W, j = t - i * W;
    const Win w = window(x + base, y + base, i, j, H, W);
    const float mx = w.sx / 9.f, my = w.sy / 9.f, exx = w.sxx / 9.f, eyy = w.syy / 9.f, exy = w.sxy / 9.f;
    const float vx = exx - mx * mx, vy = eyy - my * my, vxy = exy - mx * my;
    const float n1 = 2.f * mx * my + C1, n2 = 2.f * vxy + C2, d1 = mx * mx + my * my + C1, d2 = vx + vy + C2;
    const float n = n1 * n2, d = d1 * d2, v = (1.f - n / d) * 0.5f;
    const float go = (v > 0.f && v < 1.f) ? -0.5f * g[base + t] : 0.f;     // d out / d (n / d), clamp-gated
    // n / d as a function of (mx, my, exx, eyy, exy); vxy = exy - mx my, vx = exx - mx^2, vy = eyy - my^2
    const float inv_d = 1.f / d, q = n * inv_d * inv_d;
    const float dn_dmx = 2.f * my * n2 + n1 * (-2.f * my), dn_dmy = 2.f * mx * n2 + n1 * (-2.f * mx), dn_dexy = 2.f * n1;
    const float dd_dmx = 2.f * mx * d2 + d1 * (-2.f * mx), dd_dmy = 2.f * my * d2 + d1 * (-2.f * my), dd_de = d1;
    coef[0 * total + base + t] = go * (dn_dmx * inv_d - q * dd_dmx);
    coef[1 * total + base + t] = go * (dn_dmy * inv_d - q * dd_dmy);
    coef[2 * total + base + t] = go * (-q * dd_de);            // d / d E[xx]
    coef[3 * total + base + t] = go * (-q * dd_de);            // d / d E[yy]
    coef[4 * total + base + t] = go * (dn_dexy * inv_d);       // d / d E[xy]
}

// pass 2: pixel p collects the coefficient fields of every window that reads it.  Window centre q = (i + di, j + dj) reads p
// directly; through the reflection pad p is also read as the mirror image of a padded position, which is the adjoint of
// reflect_idx: centre row r reads padded rows r-1..r+1 -> image rows reflect(r-1..r+1); count how often p's row appears.
__device__ __forceinline__ int mult(int p, int r, int n) {      // times window centre r reads image index p along one axis
    int m = 0;
#pragma unroll
    for (int d = -1; d <= 1; ++d) m += reflect_idx(r + d, n) == p ? 1 : 0;
    return m;
}

__global__ __launch_bounds__(NT) void ssim_bwd_gather_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ coef, int H, int W, size_t total,
                                                             float* __restrict__ gx, float* __restrict__ gy) {
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= H * W) return;
    const size_t base = (size_t)blockIdx.y * H * W;
    const int i = t / W, j = t - i * W;
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = max(i - 2, 0); r <= min(i + 2, H - 1); ++r) {      // a reflected read reaches two rows away at most
        const int mr = mult(i, r, H);
        if (!mr) continue;
        for (int c = max(j - 2, 0); c <= min(j + 2, W - 1); ++c) {
            const int m = mr * mult(j, c, W);
            if (!m) continue;
            const size_t o = base + (size_t)r * W + c;
#pragma unroll
            for (int k = 0; k < 5; ++k) s[k] += (float)m * coef[k * total + o];
        }
    }
    const float a = x[base + t], b = y[base + t];
    if (gx) gx[base + t] = (s[0] + 2.f * a * s[2] + b * s[4]) / 9.f;
    if (gy) gy[base + t] = (s[1] + 2.f * b * s[3] + a * s[4]) / 9.f;
}

// ---- edge-aware smoothness ------------------------------------------------------------------------------------------
__device__ __forceinline__ float edge_w(const float* __restrict__ img, int C, size_t plane, size_t a, size_t b) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += fabsf(img[c * plane + a] - img[c * plane + b]);
    return expf(-s / (float)C);
}

__global__ __launch_bounds__(NT) void edge_smooth_fwd_kernel(const float* __restrict__ disp, const float* __restrict__ img, int C,
                                                             int H, int W, float* __restrict__ partials) {
    __shared__ float red[NT / WAVE];
    const int b = blockIdx.y;
    const size_t plane = (size_t)H * W;
    const float* d = disp + (size_t)b * plane;
    const float* im = img + (size_t)b * C * plane;
    float ax = 0.f, ay = 0.f;
    for (int t = blockIdx.x * NT + threadIdx.x; t < H * W; t += gridDim.x * NT) {
        const int i = t / W, j = t - i * W;
        if (j + 1 < W) ax += fabsf(d[t] - d[t + 1]) * edge_w(im, C, plane, t, t + 1);
        if (i + 1 < H) ay += fabsf(d[t] - d[t + W]) * edge_w(im, C, plane, t, t + W);
    }
    const float sx = block_sum<NT>(ax, red);
    const float sy = block_sum<NT>(ay, red);
    if (threadIdx.x == 0) {
        partials[2 * (blockIdx.y * gridDim.x + blockIdx.x)] = sx;
        partials[2 * (blockIdx.y * gridDim.x + blockIdx.x) + 1] = sy;
    }
}

__global__ __launch_bounds__(NT) void edge_smooth_finalize_kernel(const float* __restrict__ partials, int nblk, double nx, double ny,
                                                                  float* __restrict__ out) {
    __shared__ double rx[NT], ry[NT];
    double ax = 0.0, ay = 0.0;
    for (int i = threadIdx.x; i < nblk; i += NT) {
        ax += (double)partials[2 * i];
        ay += (double)partials[2 * i + 1];
    }
    rx[threadIdx.x] = ax;
    ry[threadIdx.x] = ay;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            rx[threadIdx.x] += rx[threadIdx.x + o];
            ry[threadIdx.x] += ry[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(rx[0] / nx + ry[0] / ny);
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(NT) void edge_smooth_bwd_kernel(const float* __restrict__ disp, const float* __restrict__ img, int C,
                                                             int H, int W, const float* __restrict__ gscale, float inv_nx,
                                                             float inv_ny, float* __restrict__ g_disp) {
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= H * W) return;
    const int b = blockIdx.y;
    const size_t plane = (size_t)H * W;
    const float* d = disp + (size_t)b * plane;
    const float* im = img + (size_t)b * C * plane;
    const int i = t / W, j = t - i * W;
    float acc = 0.f;
    if (j + 1 < W) acc += sgn(d[t] - d[t + 1]) * edge_w(im, C, plane, t, t + 1) * inv_nx;
    if (j > 0) acc -= sgn(d[t - 1] - d[t]) * edge_w(im, C, plane, t - 1, t) * inv_nx;
    if (i + 1 < H) acc += sgn(d[t] - d[t + W]) * edge_w(im, C, plane, t, t + W) * inv_ny;
    if (i > 0) acc -= sgn(d[t - W] - d[t]) * edge_w(im, C, plane, t - W, t) * inv_ny;
    g_disp[(size_t)b * plane + t] = gscale[0] * acc;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + NT - 1) / NT); }
inline int red_blocks(int n) {
    const int b = (n + NT * 4 - 1) / (NT * 4);
    return b < 1 ? 1 : (b > 256 ? 256 : b);
}

// ------------------------------------------------------------------------------------------------ colour pyramid
// inputs[("color", 0, s)], s = 1 .. 3, of the GPU-side sample synthesis (the reference's loader builds them per sample on the
// CPU, MD2/datasets/mono_dataset.py:119-144; here they are block means of the synthesised frame): the three average-pool
// levels in ONE pass over the frame -- a thread owns an 8 x 8 block, reads it once and writes 16 + 4 + 1 means.  Every mean
// is formed as ATen's avg_pool2d forms it (the window's values added row by row in float, then divided by the window size), so
// the result is bit-identical to F.avg_pool2d(x, 2 / 4 / 8).
__global__ __launch_bounds__(NT) void avg_pyramid_kernel(const float* __restrict__ x, int H, int W, float* __restrict__ o1,
                                                         float* __restrict__ o2, float* __restrict__ o3, int64_t nblk) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= nblk) return;
    const int bw = W >> 3, bh = H >> 3;
    const int bx = (int)(i % bw);
    const int by = (int)((i / bw) % bh);
    const int64_t pl = i / ((int64_t)bw * bh);
    const float* src = x + (pl * H + 8 * by) * (int64_t)W + 8 * bx;
    float v[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float4 a = *reinterpret_cast<const float4*>(src + (int64_t)r * W);
        const float4 b = *reinterpret_cast<const float4*>(src + (int64_t)r * W + 4);
        v[r][0] = a.x; v[r][1] = a.y; v[r][2] = a.z; v[r][3] = a.w; v[r][4] = b.x; v[r][5] = b.y; v[r][6] = b.z; v[r][7] = b.w;
    }
    const int W1 = W >> 1, W2 = W >> 2, W3 = W >> 3;
    float* d1 = o1 + (pl * (H >> 1) + 4 * by) * (int64_t)W1 + 4 * bx;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float4 m;
        m.x = (((v[2 * r][0] + v[2 * r][1]) + v[2 * r + 1][0]) + v[2 * r + 1][1]) / 4.f;
        m.y = (((v[2 * r][2] + v[2 * r][3]) + v[2 * r + 1][2]) + v[2 * r + 1][3]) / 4.f;
        m.z = (((v[2 * r][4] + v[2 * r][5]) + v[2 * r + 1][4]) + v[2 * r + 1][5]) / 4.f;
        m.w = (((v[2 * r][6] + v[2 * r][7]) + v[2 * r + 1][6]) + v[2 * r + 1][7]) / 4.f;
        *reinterpret_cast<float4*>(d1 + (int64_t)r * W1) = m;
    }
    float* d2 = o2 + (pl * (H >> 2) + 2 * by) * (int64_t)W2 + 2 * bx;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float m[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) acc += v[4 * r + rr][4 * c + cc];
            m[c] = acc / 16.f;
        }
        *reinterpret_cast<float2*>(d2 + (int64_t)r * W2) = make_float2(m[0], m[1]);
    }
    float acc = 0.f;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr)
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) acc += v[rr][cc];
    o3[(pl * (H >> 3) + by) * (int64_t)W3 + bx] = acc / 64.f;
}

}  // namespace

extern "C" {

int dmh_ssim_map(const float* x, const float* y, int planes, int H, int W, float* out, void* stream) {
    DMH_REQUIRE(x && y && out, "null pointer");
    DMH_REQUIRE(planes > 0 && planes <= 65535 && H >= 2 && W >= 2 && (int64_t)H * W < (1 << 30), "bad sizes (H, W >= 2)");
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3(blocks_for((int64_t)H * W), planes), dim3(NT), 0, (hipStream_t)stream, x, y, H, W, out);
    return check_launch("dmh_ssim_map");
}

int dmh_ssim_map_bwd(const float* x, const float* y, const float* g_out, int planes, int H, int W, float* workspace, float* g_x,
                     float* g_y, void* stream) {
    DMH_REQUIRE(x && y && g_out && workspace && (g_x || g_y), "null pointer");
    DMH_REQUIRE(planes > 0 && planes <= 65535 && H >= 2 && W >= 2 && (int64_t)H * W < (1 << 30), "bad sizes (H, W >= 2)");
    const size_t total = (size_t)planes * H * W;
    hipLaunchKernelGGL(ssim_bwd_coef_kernel, dim3(blocks_for((int64_t)H * W), planes), dim3(NT), 0, (hipStream_t)stream, x, y,
                       g_out, H, W, total, workspace);
    hipLaunchKernelGGL(ssim_bwd_gather_kernel, dim3(blocks_for((int64_t)H * W), planes), dim3(NT), 0, (hipStream_t)stream, x, y,
                       workspace, H, W, total, g_x, g_y);
    return check_launch("dmh_ssim_map_bwd");
}

int64_t dmh_edge_smooth_partials_size(int B, int H, int W) { return 2 * (int64_t)B * red_blocks(H * W); }

int dmh_edge_smooth(const float* disp, const float* img, int B, int C, int H, int W, float* partials, float* out, void* stream) {
    DMH_REQUIRE(disp && img && partials && out, "null pointer");
    DMH_REQUIRE(B > 0 && B <= 65535 && C > 0 && H >= 2 && W >= 2 && (int64_t)H * W < (1 << 30), "bad sizes (H, W >= 2)");
    const int nb = red_blocks(H * W);
    hipLaunchKernelGGL(edge_smooth_fwd_kernel, dim3(nb, B), dim3(NT), 0, (hipStream_t)stream, disp, img, C, H, W, partials);
    hipLaunchKernelGGL(edge_smooth_finalize_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, partials, nb * B,
                       (double)B * H * (W - 1), (double)B * (H - 1) * W, out);
    return check_launch("dmh_edge_smooth");
}

int dmh_edge_smooth_bwd(const float* disp, const float* img, int B, int C, int H, int W, const float* gscale, float* g_disp,
                        void* stream) {
    DMH_REQUIRE(disp && img && gscale && g_disp, "null pointer");
    DMH_REQUIRE(B > 0 && B <= 65535 && C > 0 && H >= 2 && W >= 2 && (int64_t)H * W < (1 << 30), "bad sizes (H, W >= 2)");
    hipLaunchKernelGGL(edge_smooth_bwd_kernel, dim3(blocks_for((int64_t)H * W), B), dim3(NT), 0, (hipStream_t)stream, disp, img, C,
                       H, W, gscale, (float)(1.0 / ((double)B * H * (W - 1))), (float)(1.0 / ((double)B * (H - 1) * W)), g_disp);
    return check_launch("dmh_edge_smooth_bwd");
}

int dmh_avg_pyramid(const float* x, int planes, int H, int W, float* out1, float* out2, float* out3, void* stream) {
    DMH_REQUIRE(x && out1 && out2 && out3, "null pointer");
    DMH_REQUIRE(planes > 0 && H >= 8 && W >= 8 && H % 8 == 0 && W % 8 == 0, "H and W must be multiples of 8");
    DMH_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)out1 & 15) == 0 && ((uintptr_t)out2 & 7) == 0, "pointers must be 16-byte aligned");
    const int64_t nblk = (int64_t)planes * (H / 8) * (W / 8);
    DMH_REQUIRE(nblk < ((int64_t)1 << 31) * NT, "too many blocks");
    hipLaunchKernelGGL(avg_pyramid_kernel, dim3((unsigned)((nblk + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, x, H, W, out1,
                       out2, out3, nblk);
    return check_launch("dmh_avg_pyramid");
}

}  // extern "C"
