// Stand-alone materialised views of generate_images_pred (MD2/trainer.py:481-519) and the adjoint of the bilinear
// disparity up-sampling, for callers that want the reference's intermediate tensors (--materialize_warps, the
// warp_view op).  These keep the reference's exact op order (IEEE divides, normalise / un-normalise round trip of the
// sampling grid, MD2/layers.py:16-25,139-198); the fused loss kernels (photo_loss.hip) never call them.
#include "common.hpp"

using namespace dmh;

namespace {

constexpr int NT = 256;

struct Cam {
    float ik[9];   // inv_K[:3,:3]           MD2/layers.py:164
    float P[12];   // (K @ T)[:3,:]          MD2/layers.py:188
};

__device__ __forceinline__ void load_cam(Cam* cam, const float* __restrict__ K, const float* __restrict__ invK,
                                         const float* __restrict__ T, int b, int t) {
    // t in [0,21): 9 inv_K entries + 12 entries of (K@T)[:3,:]
    if (t < 9) {
        cam->ik[t] = invK[b * 16 + (t / 3) * 4 + (t % 3)];
    } else if (t < 21) {
        const int i = (t - 9) / 4, j = (t - 9) % 4;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += K[b * 16 + i * 4 + k] * T[b * 16 + k * 4 + j];
        cam->P[(t - 9)] = acc;
    }
}

// F.interpolate(disp,[H,W],mode="bilinear",align_corners=False) at one output pixel (MD2/trainer.py:481-482)
__device__ __forceinline__ float disp_at(const float* __restrict__ d, int Hs, int Ws, float rh, float rw, bool same,
                                         int y, int x) {
    if (same) return d[y * Ws + x];
    const float sy = fmaxf(rh * ((float)y + 0.5f) - 0.5f, 0.f);
    const float sx = fmaxf(rw * ((float)x + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float v00 = d[y0 * Ws + x0], v01 = d[y0 * Ws + x1], v10 = d[y1 * Ws + x0], v11 = d[y1 * Ws + x1];
    return hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
}

struct Proj {
    float ix, iy;        // un-clipped sample coordinates in pixels (grid_sample, align_corners=True)
    float px, py, den;   // projected pixel, Z + eps
    float ax, ay, az;    // d(X,Y,Z)/d depth
    float depth;
};

// disp_to_depth -> BackprojectDepth -> Project3D -> grid normalise/un-normalise, op order of the reference.
__device__ __forceinline__ Proj project(const Cam& c, float disp, int x, int y, int H, int W, float min_disp, float dmul) {
    Proj p;
    const float sd = min_disp + dmul * disp;
    p.depth = 1.0f / sd;
    const float fx = (float)x, fy = (float)y;
    const float rx = c.ik[0] * fx + c.ik[1] * fy + c.ik[2];
    const float ry = c.ik[3] * fx + c.ik[4] * fy + c.ik[5];
    const float rz = c.ik[6] * fx + c.ik[7] * fy + c.ik[8];
    const float cx = p.depth * rx, cy = p.depth * ry, cz = p.depth * rz;
    const float X = c.P[0] * cx + c.P[1] * cy + c.P[2] * cz + c.P[3];
    const float Y = c.P[4] * cx + c.P[5] * cy + c.P[6] * cz + c.P[7];
    const float Z = c.P[8] * cx + c.P[9] * cy + c.P[10] * cz + c.P[11];
    p.ax = c.P[0] * rx + c.P[1] * ry + c.P[2] * rz;
    p.ay = c.P[4] * rx + c.P[5] * ry + c.P[6] * rz;
    p.az = c.P[8] * rx + c.P[9] * ry + c.P[10] * rz;
    p.den = Z + 1e-7f;
    p.px = X / p.den;
    p.py = Y / p.den;
    const float gx = (p.px / (float)(W - 1) - 0.5f) * 2.f;  // MD2/layers.py:195-197
    const float gy = (p.py / (float)(H - 1) - 0.5f) * 2.f;
    p.ix = ((gx + 1.f) / 2.f) * (float)(W - 1);             // grid_sampler_unnormalize, align_corners=True
    p.iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    return p;
}

struct Tap {
    unsigned o00, o01, o10, o11;  // unsigned 32-bit offsets: scalar base + 32-bit VGPR offset addressing
    float w00, w01, w10, w11, fx, fy;
};

__device__ __forceinline__ Tap make_tap(float ix, float iy, int H, int W) {
    // padding_mode="border": clip_coordinates, then bilinear corner weights as grid_sampler does
    ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    Tap t;
    t.fx = ix - x0f;
    t.fy = iy - y0f;
    const float gx = (x0f + 1.f) - ix, gy = (y0f + 1.f) - iy;
    t.w00 = gx * gy;
    t.w01 = t.fx * gy;
    t.w10 = gx * t.fy;
    t.w11 = t.fx * t.fy;
    t.o00 = (unsigned)(y0 * W + x0);
    t.o01 = (unsigned)(y0 * W + x1);
    t.o10 = (unsigned)(y1 * W + x0);
    t.o11 = (unsigned)(y1 * W + x1);
    return t;
}

__device__ __forceinline__ float tap_sample(const float* __restrict__ img, const Tap& t) {
    return img[t.o00] * t.w00 + img[t.o01] * t.w01 + img[t.o10] * t.w10 + img[t.o11] * t.w11;
}

// gradient of one warped pixel back to the up-sampled disparity, given d loss / d warped (3 channels)
__device__ __forceinline__ float warp_pixel_bwd(const float* __restrict__ src, const Cam& cam, float d, int x, int y,
                                                int H, int W, float min_disp, float dmul, float g0, float g1,
                                                float g2) {
    const Proj p = project(cam, d, x, y, H, W, min_disp, dmul);
    const Tap t = make_tap(p.ix, p.iy, H, W);
    float gix = 0.f, giy = 0.f;
    const float gc[3] = {g0, g1, g2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* im = src + (size_t)c * H * W;
        const float v00 = im[t.o00], v01 = im[t.o01], v10 = im[t.o10], v11 = im[t.o11];
        gix += gc[c] * ((v01 - v00) * (1.f - t.fy) + (v11 - v10) * t.fy);
        giy += gc[c] * ((v10 - v00) * (1.f - t.fx) + (v11 - v01) * t.fx);
    }
    // clip_coordinates_set_grad: zero outside the open interval (0, size-1)
    if (!(p.ix > 0.f && p.ix < (float)(W - 1))) gix = 0.f;
    if (!(p.iy > 0.f && p.iy < (float)(H - 1))) giy = 0.f;
    const float g_depth = (gix * (p.ax - p.px * p.az) + giy * (p.ay - p.py * p.az)) * (1.0f / p.den);
    return g_depth * (-(p.depth * p.depth)) * dmul;
}

// ------------------------------------------------------------------------------------------------ upsample adjoint
// g_disp[b,j,i] (+)= sum over the full-resolution pixels whose bilinear footprint touches (j,i).
// Gather form: deterministic, no atomics.  One thread per low-resolution texel.
__global__ __launch_bounds__(NT) void upsample_adjoint_kernel(const float* __restrict__ g_up, float* __restrict__ g_disp,
                                                              int B, int H, int W, int Hs, int Ws, int accumulate) {
    const int idx = blockIdx.x * NT + threadIdx.x;
    if (idx >= B * Hs * Ws) return;
    const int i = idx % Ws, j = (idx / Ws) % Hs, b = idx / (Ws * Hs);
    const float rh = (float)Hs / (float)H, rw = (float)Ws / (float)W;
    const int fy = (H + Hs - 1) / Hs, fx = (W + Ws - 1) / Ws;  // integer upsampling factors (>= true ratio)
    const int ylo = max(0, fy * j - fy), yhi = min(H - 1, fy * j + 2 * fy);
    const int xlo = max(0, fx * i - fx), xhi = min(W - 1, fx * i + 2 * fx);
    const float* g = g_up + (size_t)b * H * W;
    float acc = 0.f;
    for (int y = ylo; y <= yhi; ++y) {
        const float sy = fmaxf(rh * ((float)y + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)sy, y1 = y0 + (y0 < Hs - 1 ? 1 : 0);
        const float ly = sy - (float)y0;
        const float wy = (y0 == j ? 1.f - ly : 0.f) + (y1 == j ? ly : 0.f);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int x = xlo; x <= xhi; ++x) {
            const float sx = fmaxf(rw * ((float)x + 0.5f) - 0.5f, 0.f);
            const int x0 = (int)sx, x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
            const float lx = sx - (float)x0;
            const float wx = (x0 == i ? 1.f - lx : 0.f) + (x1 == i ? lx : 0.f);
            row += wx * g[y * W + x];
        }
        acc += wy * row;
    }
    if (accumulate) acc += g_disp[idx];
    g_disp[idx] = acc;
}

// ------------------------------------------------------------------------------------------------ materialised views
__global__ __launch_bounds__(NT) void warp_view_fwd_kernel(const float* __restrict__ source,
                                                           const float* __restrict__ disp_all,
                                                           const float* __restrict__ K, const float* __restrict__ invK,
                                                           const float* __restrict__ T, int H, int W, int Hs, int Ws,
                                                           float min_disp, float dmul, float* __restrict__ depth,
                                                           float* __restrict__ sample, float* __restrict__ color) {
    __shared__ Cam s_cam;
    const int b = blockIdx.y;
    if (threadIdx.x < 21) load_cam(&s_cam, K, invK, T, b, threadIdx.x);
    __syncthreads();
    const int idx = blockIdx.x * NT + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W;
    const bool same = (Hs == H && Ws == W);
    const float d = disp_at(disp_all + (size_t)b * Hs * Ws, Hs, Ws, (float)Hs / (float)H, (float)Ws / (float)W, same,
                            y, x);
    const Proj p = project(s_cam, d, x, y, H, W, min_disp, dmul);
    const size_t pix = (size_t)b * H * W + idx;
    if (depth) depth[pix] = p.depth;
    if (sample) {
        sample[pix * 2 + 0] = (p.px / (float)(W - 1) - 0.5f) * 2.f;
        sample[pix * 2 + 1] = (p.py / (float)(H - 1) - 0.5f) * 2.f;
    }
    if (color) {
        const Tap t = make_tap(p.ix, p.iy, H, W);
        const float* src = source + (size_t)b * 3 * H * W;
        color[((size_t)b * 3 + 0) * H * W + idx] = tap_sample(src, t);
        color[((size_t)b * 3 + 1) * H * W + idx] = tap_sample(src + H * W, t);
        color[((size_t)b * 3 + 2) * H * W + idx] = tap_sample(src + 2 * H * W, t);
    }
}

__global__ __launch_bounds__(NT) void warp_view_bwd_kernel(const float* __restrict__ source,
                                                           const float* __restrict__ disp_all,
                                                           const float* __restrict__ K, const float* __restrict__ invK,
                                                           const float* __restrict__ T, int H, int W, int Hs, int Ws,
                                                           float min_disp, float dmul,
                                                           const float* __restrict__ grad_color,
                                                           const float* __restrict__ grad_depth,
                                                           float* __restrict__ g_up) {
    __shared__ Cam s_cam;
    const int b = blockIdx.y;
    if (threadIdx.x < 21) load_cam(&s_cam, K, invK, T, b, threadIdx.x);
    __syncthreads();
    const int idx = blockIdx.x * NT + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W;
    const bool same = (Hs == H && Ws == W);
    const float d = disp_at(disp_all + (size_t)b * Hs * Ws, Hs, Ws, (float)Hs / (float)H, (float)Ws / (float)W, same,
                            y, x);
    const size_t hw = (size_t)H * W;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (grad_color) {
        g0 = grad_color[((size_t)b * 3 + 0) * hw + idx];
        g1 = grad_color[((size_t)b * 3 + 1) * hw + idx];
        g2 = grad_color[((size_t)b * 3 + 2) * hw + idx];
    }
    float g = warp_pixel_bwd(source + (size_t)b * 3 * hw, s_cam, d, x, y, H, W, min_disp, dmul, g0, g1, g2);
    if (grad_depth) {
        const float sd = min_disp + dmul * d;
        g += grad_depth[(size_t)b * hw + idx] * (-1.0f / (sd * sd)) * dmul;
    }
    g_up[(size_t)b * hw + idx] = g;
}

}  // namespace

extern "C" {

int dmh_upsample_bilinear_adjoint(const float* g_up, float* g_disp, int B, int H, int W, int Hs, int Ws,
                                  int accumulate, void* stream) {
    DMH_REQUIRE(g_up && g_disp, "null pointer");
    DMH_REQUIRE(B > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && Hs <= H && Ws <= W, "bad sizes");
    const int n = B * Hs * Ws;
    hipLaunchKernelGGL(upsample_adjoint_kernel, dim3((n + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, g_up,
                       g_disp, B, H, W, Hs, Ws, accumulate);
    return check_launch("dmh_upsample_bilinear_adjoint");
}

int dmh_warp_view_fwd(const float* source, const float* disp, const float* K, const float* inv_K, const float* T,
                      int B, int H, int W, int Hs, int Ws, float min_depth, float max_depth, float* depth,
                      float* sample, float* color, void* stream) {
    DMH_REQUIRE(source && disp && K && inv_K && T, "null input");
    DMH_REQUIRE(B > 0 && H >= 2 && W >= 2 && Hs > 0 && Ws > 0 && Hs <= H && Ws <= W, "bad sizes");
    DMH_REQUIRE(min_depth > 0.f && max_depth > min_depth, "bad depth range");
    const double mn = 1.0 / (double)max_depth, mx = 1.0 / (double)min_depth;
    hipLaunchKernelGGL(warp_view_fwd_kernel, dim3((H * W + NT - 1) / NT, B), dim3(NT), 0, (hipStream_t)stream, source,
                       disp, K, inv_K, T, H, W, Hs, Ws, (float)mn, (float)(mx - mn), depth, sample, color);
    return check_launch("dmh_warp_view_fwd");
}

int dmh_warp_view_bwd(const float* source, const float* disp, const float* K, const float* inv_K, const float* T,
                      int B, int H, int W, int Hs, int Ws, float min_depth, float max_depth,
                      const float* grad_color, const float* grad_depth, float* g_up, void* stream) {
    DMH_REQUIRE(source && disp && K && inv_K && T && g_up, "null input");
    DMH_REQUIRE(B > 0 && H >= 2 && W >= 2 && Hs > 0 && Ws > 0 && Hs <= H && Ws <= W, "bad sizes");
    DMH_REQUIRE(min_depth > 0.f && max_depth > min_depth, "bad depth range");
    const double mn = 1.0 / (double)max_depth, mx = 1.0 / (double)min_depth;
    hipLaunchKernelGGL(warp_view_bwd_kernel, dim3((H * W + NT - 1) / NT, B), dim3(NT), 0, (hipStream_t)stream, source,
                       disp, K, inv_K, T, H, W, Hs, Ws, (float)mn, (float)(mx - mn), grad_color, grad_depth, g_up);
    return check_launch("dmh_warp_view_bwd");
}

}  // extern "C"
