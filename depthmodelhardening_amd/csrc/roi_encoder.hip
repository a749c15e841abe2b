// K19, encoder side -- the attack's backward pass through the encoder head evaluated only where the patch gradient needs it.
//
// d cost / d patch reads d cost / d image under the pasted object only (physicalTrans.py:156-165: adv = scene (1 - m) +
// patch m; torchattacks/attacks/phy_obj_atk.py:96 differentiates w.r.t. the patch), so the backward of the encoder's first
// stages -- conv1 / bn1 / relu / maxpool and layer1 of torchvision's ResNet under MD2/networks/resnet_encoder.py:85-98 -- is
// needed on one window per scene: the object's box, widened by what each stage reaches.  The convolutions stay the K10
// launches (zero padding at the window's edge: every stage spoils one more ring of the window, and the plan makes the
// window that much larger than what is read of it); this file holds the passes around them:
//
//   roi_crop      out[b, c, i, j] = src[b, c, org_b + (i, j)]  (* [gate > 0])     compact windows of whole-frame tensors
//   roi_mask      out = g * [gate[b, c, org_b + (i, j)] > 0]                      ReLU mask of a compact gradient
//   roi_paste     dst[b, c, win_b + (i, j)] = src[...]                            a window written back into a whole-frame tensor
//   stem_bwd_win  g_z = scale[c] [feat > 0] (g_feat + maxpool3x3/2 adjoint of g_pool)   on a window of the 1/2-resolution map,
//                 g_pool given as a compact window of the 1/4-resolution map      (encoder_glue.hip: stem_bwd_kernel)
#include "common.hpp"

using namespace dmh;

namespace {

constexpr int NT = 256;

// mode 0: crop src; 1: crop src and mask by gate's window; 2: mask the compact g by gate's window; +4: gate is itself a
// compact window (same origin and size)
// (round 6: CPT consecutive channel planes of one sample per thread -- they share the window's geometry, and a block that moved
//  one plane of a 56 x 78 window lived longer on its argument / origin loads than on its data: see roi_glue_bwd2_kernel)
template <int CPT>
__global__ __launch_bounds__(NT) void roi_crop_kernel(const float* __restrict__ src, const float* __restrict__ gate,
                                                      const float* __restrict__ g, const int* __restrict__ org, int C, int H,
                                                      int W, int hc, int wc, int mode, float* __restrict__ out) {
    const int t = (blockIdx.x * NT + threadIdx.x) * 2;         // wc is even
    if (t >= hc * wc) return;
    const int groups = C / CPT;
    const int b = (int)blockIdx.y / groups, plane0 = b * C + ((int)blockIdx.y - b * groups) * CPT;
    const int i = t / wc, j = t - i * wc;
    size_t so = ((size_t)plane0 * H + org[2 * b] + i) * W + org[2 * b + 1] + j;      // even: 8-byte aligned
    size_t co = (size_t)plane0 * hc * wc + t;
#pragma unroll
    for (int k = 0; k < CPT; ++k, so += (size_t)H * W, co += (size_t)hc * wc) {
        float2 v;
        if ((mode & 3) == 2) v = *reinterpret_cast<const float2*>(g + co);
        else v = *reinterpret_cast<const float2*>(src + so);
        if ((mode & 3) != 0) {
            const float2 q = *reinterpret_cast<const float2*>(gate + ((mode & 4) ? co : so));
            v.x = q.x > 0.f ? v.x : 0.f;
            v.y = q.y > 0.f ? v.y : 0.f;
        }
        *reinterpret_cast<float2*>(out + co) = v;
    }
}

// dst[b, c, win_b + (i, j)] = src[...] for the h x w window at frame position win_org: src is a compact [B,C,sh,sw] window at
// frame origin src_org, or (src_org == NULL) a whole-frame tensor like dst
template <int CPT>
__global__ __launch_bounds__(NT) void roi_paste_kernel(const float* __restrict__ src, const int* __restrict__ src_org, int sh,
                                                       int sw, const int* __restrict__ win_org, int C, int H, int W, int h,
                                                       int w, float* __restrict__ dst) {
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= h * w) return;
    const int groups = C / CPT;
    const int b = (int)blockIdx.y / groups, plane0 = b * C + ((int)blockIdx.y - b * groups) * CPT;
    const int i = t / w, j = t - i * w;
    const int Y = win_org[2 * b] + i, X = win_org[2 * b + 1] + j;
    const int sy = Y - (src_org ? src_org[2 * b] : 0), sx = X - (src_org ? src_org[2 * b + 1] : 0);
    float* d = dst + ((size_t)plane0 * H + Y) * W + X;
    const float* sp = src + ((size_t)plane0 * sh + sy) * sw + sx;
#pragma unroll
    for (int k = 0; k < CPT; ++k, d += (size_t)H * W, sp += (size_t)sh * sw) *d = *sp;
}

// one thread = one 2 x 2 quad of the window of the H x W map (quad (i, j) = rows 2i, 2i+1: aligned with pooling cell (i, j))
__global__ __launch_bounds__(NT) void stem_bwd_win_kernel(const float* __restrict__ feat, const unsigned char* __restrict__ argmax,
                                                          const float* __restrict__ g_feat, const float* __restrict__ g_pool,
                                                          const float* __restrict__ scale, const int* __restrict__ org,
                                                          const int* __restrict__ pool_org, int C, int H, int W, int hs, int ws,
                                                          int hq, int wq, float* __restrict__ g_z) {
    const int PH = H >> 1, PW = W >> 1, qh = hs >> 1, qw = ws >> 1;
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= qh * qw) return;
    const int plane = blockIdx.y, b = plane / C, c = plane - b * C;
    const int oy = org[2 * b], ox = org[2 * b + 1];               // even
    const int py0 = pool_org[2 * b], px0 = pool_org[2 * b + 1];
    const int qi = t / qw, qj = t - qi * qw;
    const int i = (oy >> 1) + qi, j = (ox >> 1) + qj;             // pooling-cell coordinates in the frame
    const size_t base = (size_t)plane * H * W, pbase = (size_t)plane * PH * PW;
    const float* gq = g_pool + (size_t)plane * hq * wq;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int di = 0; di < 2; ++di) {
        const int oi = i + di;
        if (oi >= PH) continue;
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            const int oj = j + dj;
            if (oj >= PW) continue;
            const int a = argmax[pbase + (size_t)oi * PW + oj];
            const int ky = a / 3, kx = a - ky * 3;
            const int ry = 2 * di - 1 + ky, rx = 2 * dj - 1 + kx;       // position relative to the quad origin
            if (ry >= 0 && ry < 2 && rx >= 0 && rx < 2) {
                const int wi = oi - py0, wj = oj - px0;
                const float gp = (wi >= 0 && wi < hq && wj >= 0 && wj < wq) ? gq[wi * wq + wj] : 0.f;
                if (ry == 0 && rx == 0) acc[0][0] += gp;
                if (ry == 0 && rx == 1) acc[0][1] += gp;
                if (ry == 1 && rx == 0) acc[1][0] += gp;
                if (ry == 1 && rx == 1) acc[1][1] += gp;
            }
        }
    }
    const float s = scale[c];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const size_t o = base + (size_t)(2 * i + r) * W + 2 * j;
        const float2 f = *reinterpret_cast<const float2*>(feat + o);
        float g0 = acc[r][0], g1 = acc[r][1];
        if (g_feat) {
            const float2 gf = *reinterpret_cast<const float2*>(g_feat + o);
            g0 += gf.x;
            g1 += gf.y;
        }
        *reinterpret_cast<float2*>(g_z + ((size_t)plane * hs + 2 * qi + r) * ws + 2 * qj) =
            make_float2(f.x > 0.f ? g0 * s : 0.f, f.y > 0.f ? g1 * s : 0.f);
    }
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + NT - 1) / NT); }

}  // namespace

extern "C" {

int dmh_roi_paste(const float* src, const int* src_org, int sh, int sw, const int* win_org, int B, int C, int H, int W, int h,
                  int w, float* dst, void* stream) {
    DMH_REQUIRE(src && win_org && dst, "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && (int64_t)B * C <= 65535 && h >= 1 && w >= 1 && h <= H && w <= W && (int64_t)H * W < (1 << 30),
                "bad sizes (window inside the frame)");
    DMH_REQUIRE(src_org ? (h <= sh && w <= sw) : (sh == H && sw == W), "the source must hold the window");
    if (C % 8 == 0)
        hipLaunchKernelGGL(roi_paste_kernel<8>, dim3(blocks_for((int64_t)h * w), B * (C / 8)), dim3(NT), 0, (hipStream_t)stream, src,
                           src_org, sh, sw, win_org, C, H, W, h, w, dst);
    else
        hipLaunchKernelGGL(roi_paste_kernel<1>, dim3(blocks_for((int64_t)h * w), B * C), dim3(NT), 0, (hipStream_t)stream, src,
                           src_org, sh, sw, win_org, C, H, W, h, w, dst);
    return check_launch("dmh_roi_paste");
}

int dmh_roi_crop(const float* src, const float* gate, const float* g, const int* org, int B, int C, int H, int W, int hc,
                 int wc, int gate_compact, float* out, void* stream) {
    DMH_REQUIRE(org && out && (src || (g && gate)), "null pointer");
    DMH_REQUIRE(!(src && g), "give either a whole-frame src or a compact g");
    DMH_REQUIRE(B > 0 && C > 0 && (int64_t)B * C <= 65535 && hc >= 1 && wc >= 2 && (wc & 1) == 0 && (W & 1) == 0 && hc <= H &&
                    wc <= W && (int64_t)H * W < (1 << 30),
                "bad sizes (window inside the frame, even widths)");
    const int mode = (g ? 2 : (gate ? 1 : 0)) | ((gate && gate_compact) ? 4 : 0);
    if (C % 8 == 0)
        hipLaunchKernelGGL(roi_crop_kernel<8>, dim3(blocks_for((int64_t)hc * wc / 2), B * (C / 8)), dim3(NT), 0, (hipStream_t)stream,
                           src, gate, g, org, C, H, W, hc, wc, mode, out);
    else
        hipLaunchKernelGGL(roi_crop_kernel<1>, dim3(blocks_for((int64_t)hc * wc / 2), B * C), dim3(NT), 0, (hipStream_t)stream, src,
                           gate, g, org, C, H, W, hc, wc, mode, out);
    return check_launch("dmh_roi_crop");
}

int dmh_stem_bn_relu_pool_bwd_win(const float* feat, const unsigned char* argmax, const float* g_feat, const float* g_pool,
                                  const float* scale, const int* org, const int* pool_org, int B, int C, int H, int W, int hs,
                                  int ws, int hq, int wq, float* g_z, void* stream) {
    DMH_REQUIRE(feat && argmax && g_pool && scale && org && pool_org && g_z, "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && H >= 2 && W >= 2 && (H & 1) == 0 && (W & 1) == 0, "H and W must be even and >= 2");
    DMH_REQUIRE(hs >= 2 && ws >= 2 && (hs & 1) == 0 && (ws & 1) == 0 && hs <= H && ws <= W && hq >= 1 && wq >= 1 &&
                    hq <= H / 2 && wq <= W / 2,
                "windows must be even-sized and inside their frames");
    DMH_REQUIRE((int64_t)B * C <= 65535 && (int64_t)H * W < (1 << 30), "tensor too large");
    hipLaunchKernelGGL(stem_bwd_win_kernel, dim3(blocks_for((int64_t)(hs / 2) * (ws / 2)), B * C), dim3(NT), 0,
                       (hipStream_t)stream, feat, argmax, g_feat, g_pool, scale, org, pool_org, C, H, W, hs, ws, hq, wq, g_z);
    return check_launch("dmh_stem_bn_relu_pool_bwd_win");
}

}  // extern "C"
