// K3 -- fused EOT paste for gfx950: zero-pad the object patch to the scene frame, perspective-warp
// patch and mask with the per-sample homography, composite over the scene, and bilinear-resize the
// result to the network resolution -- one pass, nothing materialised at 375x1242.
//
// Reference: physicalTrans.py:107-123 (padding_img), :130-166 (project: torchvision perspective of
// image and mask per sample), torchattacks/attacks/phy_obj_atk.py:87-90 (composite + Resize of scene
// and mask).  Third-party arithmetic (torchvision 0.8.2, absent from the reference tree):
//   perspective  = grid_sample(bilinear, zeros, align_corners=False) on the grid of
//                  functional_tensor._perspective_grid (pixel centres x+0.5, theta/(0.5*size), -1)
//   Resize       = F.interpolate(bilinear, align_corners=False), no antialias
//
// HBM-bound: the scene (5.6 MB/sample) is read once, the 1.2 MB patch+mask stay L2-resident, the
// output (5.2 MB/sample) is written once with 64-lane coalesced rows.  The backward is a gather per patch
// texel over the inverse homography: deterministic, no atomics.
#include <stdlib.h>

#include "common.hpp"

using namespace dmh;

namespace {

constexpr int NT = 256;

struct Homog {
    float t00, t01, t02, t10, t11, t12, g, h;
    float SWf, SHf;
};

__device__ __forceinline__ Homog load_homog(const float* __restrict__ c, int SW, int SH) {
    Homog m;
    const float sx = 0.5f * (float)SW, sy = 0.5f * (float)SH;
    m.t00 = c[0] / sx;
    m.t01 = c[1] / sx;
    m.t02 = c[2] / sx;
    m.t10 = c[3] / sy;
    m.t11 = c[4] / sy;
    m.t12 = c[5] / sy;
    m.g = c[6];
    m.h = c[7];
    m.SWf = (float)SW;
    m.SHf = (float)SH;
    return m;
}

struct PTap {       // bilinear footprint of one composite pixel inside the patch
    int x0, y0;     // patch coordinates of the top-left texel (may be out of range)
    float fx, fy;
    bool any;       // footprint overlaps the patch rectangle
};

__device__ __forceinline__ PTap patch_tap(const Homog& m, int X, int Y, int l_pad, int t_pad, int PW, int PH) {
    const float xn = (float)X + 0.5f, yn = (float)Y + 0.5f;
    const float den = xn * m.g + yn * m.h + 1.0f;
    const float gx = (xn * m.t00 + yn * m.t01 + m.t02) / den - 1.0f;
    const float gy = (xn * m.t10 + yn * m.t11 + m.t12) / den - 1.0f;
    const float ix = ((gx + 1.f) * m.SWf - 1.f) / 2.f;  // grid_sampler_unnormalize, align_corners=False
    const float iy = ((gy + 1.f) * m.SHf - 1.f) / 2.f;
    PTap t;
    // keep the float -> int conversion in range for wild coordinates (far outside => no overlap)
    const float cx = fminf(fmaxf(ix, -4.f), m.SWf + 4.f), cy = fminf(fmaxf(iy, -4.f), m.SHf + 4.f);
    const float x0f = floorf(cx), y0f = floorf(cy);
    t.fx = cx - x0f;
    t.fy = cy - y0f;
    t.x0 = (int)x0f - l_pad;
    t.y0 = (int)y0f - t_pad;
    t.any = (ix == cx) && (iy == cy) && t.x0 >= -1 && t.x0 < PW && t.y0 >= -1 && t.y0 < PH;
    return t;
}

// zeros-padded bilinear sample of one PHxPW plane
__device__ __forceinline__ float patch_sample(const float* __restrict__ p, const PTap& t, int PW, int PH) {
    const bool xa = t.x0 >= 0, xb = t.x0 + 1 < PW, ya = t.y0 >= 0, yb = t.y0 + 1 < PH;
    const float* r0 = p + t.y0 * PW + t.x0;
    const float v00 = (xa && ya) ? r0[0] : 0.f, v01 = (xb && ya) ? r0[1] : 0.f;
    const float v10 = (xa && yb) ? r0[PW] : 0.f, v11 = (xb && yb) ? r0[PW + 1] : 0.f;
    const float gx = 1.f - t.fx, gy = 1.f - t.fy;
    return v00 * (gx * gy) + v01 * (t.fx * gy) + v10 * (gx * t.fy) + v11 * (t.fx * t.fy);
}

struct RTap {  // F.interpolate(bilinear, align_corners=False) footprint of one output pixel
    int y0, y1, x0, x1;
    float ly, lx;
};

__device__ __forceinline__ RTap resize_tap(int oy, int ox, int SH, int SW, int OH, int OW) {
    RTap r;
    const float rh = (float)SH / (float)OH, rw = (float)SW / (float)OW;
    const float sy = fmaxf(rh * ((float)oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(rw * ((float)ox + 0.5f) - 0.5f, 0.f);
    r.y0 = (int)sy;
    r.x0 = (int)sx;
    r.y1 = r.y0 + (r.y0 < SH - 1 ? 1 : 0);
    r.x1 = r.x0 + (r.x0 < SW - 1 ? 1 : 0);
    r.ly = sy - (float)r.y0;
    r.lx = sx - (float)r.x0;
    return r;
}

// Object bounding box in the scene: image of the patch rectangle (one texel of margin) under the inverse homography.
// ok = all four corners finite and in front of the horizon; composite pixels whose centre lies outside the box have
// an empty patch footprint (mask 0: the composite is the scene).
struct ObjBox { float x0, x1, y0, y1; bool ok; };
__device__ __forceinline__ ObjBox object_box(const float* __restrict__ c, const dmh_paste_args& a) {
    ObjBox b;
    b.x0 = 3.0e38f;
    b.x1 = -3.0e38f;
    b.y0 = 3.0e38f;
    b.y1 = -3.0e38f;
    b.ok = true;
    const float i00 = c[4] - c[5] * c[7], i01 = c[2] * c[7] - c[1], i02 = c[1] * c[5] - c[2] * c[4];
    const float i10 = c[5] * c[6] - c[3], i11 = c[0] - c[2] * c[6], i12 = c[2] * c[3] - c[0] * c[5];
    const float i20 = c[3] * c[7] - c[4] * c[6], i21 = c[1] * c[6] - c[0] * c[7], i22 = c[0] * c[4] - c[1] * c[3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float px = (float)a.l_pad + ((q & 1) ? (float)a.PW + 1.f : -1.f);
        const float py = (float)a.t_pad + ((q & 2) ? (float)a.PH + 1.f : -1.f);
        const float w = i20 * px + i21 * py + i22;
        const float xs = (i00 * px + i01 * py + i02) / w, ys = (i10 * px + i11 * py + i12) / w;
        b.ok = b.ok && (xs == xs) && (ys == ys) && (c[6] * xs + c[7] * ys + 1.0f > 0.f);
        b.x0 = fminf(b.x0, xs);
        b.x1 = fmaxf(b.x1, xs);
        b.y0 = fminf(b.y0, ys);
        b.y1 = fmaxf(b.y1, ys);
    }
    return b;
}


__global__ __launch_bounds__(NT) void paste_fwd_kernel(const dmh_paste_args a, float* __restrict__ adv,
                                                       float* __restrict__ mask_out) {
    const int n = blockIdx.y;
    const int idx = blockIdx.x * NT + threadIdx.x;
    if (idx >= a.OH * a.OW) return;
    const int oy = idx / a.OW;
    int ox = idx - oy * a.OW;
    if (a.flip && a.flip[n]) ox = a.OW - 1 - ox;   // compute the mirrored source pixel, store at idx
    const Homog m = load_homog(a.coeffs + n * 8, a.SW, a.SH);
    const RTap r = resize_tap(oy, ox, a.SH, a.SW, a.OH, a.OW);
    const size_t shw = (size_t)a.SH * a.SW, phw = (size_t)a.PH * a.PW;
    const bool warp_only = a.mode == DMH_PASTE_WARP_ONLY;
    const float* sc = warp_only ? nullptr : a.scene + (size_t)(a.scene_index ? a.scene_index[n] : n) * a.scene_bstride;
    const int Ys[2] = {r.y0, r.y1}, Xs[2] = {r.x0, r.x1};
    float comp[2][2][3], mm[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int Y = Ys[j], X = Xs[i];
            const size_t so = (size_t)Y * a.SW + X;
            const PTap t = patch_tap(m, X, Y, a.l_pad, a.t_pad, a.PW, a.PH);
            float mk = 0.f;
            if (t.any) mk = patch_sample(a.pmask, t, a.PW, a.PH);
            mm[j][i] = mk;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float o = 0.f;
                if (t.any) o = patch_sample(a.patch + c * phw, t, a.PW, a.PH);
                comp[j][i][c] = warp_only ? o : sc[c * shw + so] * (1.f - mk) + o * mk;  // phy_obj_atk.py:88
            }
        }
    const float hy = 1.f - r.ly, hx = 1.f - r.lx;
    const size_t ohw = (size_t)a.OH * a.OW;
    if (adv) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            adv[((size_t)n * 3 + c) * ohw + idx] = hy * (hx * comp[0][0][c] + r.lx * comp[0][1][c]) +
                                                  r.ly * (hx * comp[1][0][c] + r.lx * comp[1][1][c]);
    }
    if (mask_out)
        mask_out[(size_t)n * ohw + idx] = hy * (hx * mm[0][0] + r.lx * mm[0][1]) + r.ly * (hx * mm[1][0] + r.lx * mm[1][1]);
}

// Four output pixels of one row per thread (OW % 4 == 0): the row taps are formed once, groups whose source
// footprint lies outside the object's bounding box skip the homography and the patch altogether (87 % of a frame at
// 5-10 m) and reduce to the plain bilinear resize of the scene, results leave as 16-byte stores.  With `flip` the
// thread computes the mirrored group and stores it reversed.
__global__ __launch_bounds__(NT) void paste_fwd4_kernel(const dmh_paste_args a, float* __restrict__ adv,
                                                        float* __restrict__ mask_out) {
    const int n = blockIdx.y;
    const int gpr = a.OW >> 2;                       // groups per row
    const int gid = blockIdx.x * NT + threadIdx.x;
    if (gid >= a.OH * gpr) return;
    const int oy = gid / gpr, gx = gid - oy * gpr;
    const bool flip = a.flip && a.flip[n];
    const int ox0 = (flip ? gpr - 1 - gx : gx) << 2;   // first source column of the group that is computed
    const float* c = a.coeffs + n * 8;
    const Homog m = load_homog(c, a.SW, a.SH);
    const float rh = (float)a.SH / (float)a.OH, rw = (float)a.SW / (float)a.OW;
    const float sy = fmaxf(rh * ((float)oy + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, y1 = y0 + (y0 < a.SH - 1 ? 1 : 0);
    const float ly = sy - (float)y0, hy = 1.f - ly;
    const ObjBox bb = object_box(c, a);
    const float bx0 = bb.x0, bx1 = bb.x1, by0 = bb.y0, by1 = bb.y1;
    const bool box_ok = bb.ok;
    const size_t shw = (size_t)a.SH * a.SW, phw = (size_t)a.PH * a.PW, ohw = (size_t)a.OH * a.OW;
    const bool warp_only = a.mode == DMH_PASTE_WARP_ONLY;
    const float* sc = warp_only ? nullptr : a.scene + (size_t)(a.scene_index ? a.scene_index[n] : n) * a.scene_bstride;
    // source columns of the group
    int x0[4], x1[4];
    float lx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float sx = fmaxf(rw * ((float)(ox0 + i) + 0.5f) - 0.5f, 0.f);
        x0[i] = (int)sx;
        x1[i] = x0[i] + (x0[i] < a.SW - 1 ? 1 : 0);
        lx[i] = sx - (float)x0[i];
    }
    // pixel centres (X + 0.5, Y + 0.5) of the footprint against the box
    const bool outside = box_ok && ((float)x1[3] + 0.5f < bx0 || (float)x0[0] + 0.5f > bx1 || (float)y1 + 0.5f < by0 ||
                                    (float)y0 + 0.5f > by1);
    float res[3][4], mres[4];
    if (outside) {
#pragma unroll
        for (int i = 0; i < 4; ++i) mres[i] = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float* r0 = sc + ch * shw + (size_t)y0 * a.SW;
            const float* r1 = sc + ch * shw + (size_t)y1 * a.SW;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float hx = 1.f - lx[i];
                res[ch][i] = warp_only ? 0.f : hy * (hx * r0[x0[i]] + lx[i] * r0[x1[i]]) + ly * (hx * r1[x0[i]] + lx[i] * r1[x1[i]]);
            }
        }
    } else {
        const int Ys[2] = {y0, y1};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int Xs[2] = {x0[i], x1[i]};
            float comp[2][2][3], mm[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int Y = Ys[j], X = Xs[q];
                    const size_t so = (size_t)Y * a.SW + X;
                    const PTap t = patch_tap(m, X, Y, a.l_pad, a.t_pad, a.PW, a.PH);
                    float mk = 0.f;
                    if (t.any) mk = patch_sample(a.pmask, t, a.PW, a.PH);
                    mm[j][q] = mk;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        float o = 0.f;
                        if (t.any) o = patch_sample(a.patch + ch * phw, t, a.PW, a.PH);
                        comp[j][q][ch] = warp_only ? o : sc[ch * shw + so] * (1.f - mk) + o * mk;  // phy_obj_atk.py:88
                    }
                }
            const float hx = 1.f - lx[i];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
                res[ch][i] = hy * (hx * comp[0][0][ch] + lx[i] * comp[0][1][ch]) + ly * (hx * comp[1][0][ch] + lx[i] * comp[1][1][ch]);
            mres[i] = hy * (hx * mm[0][0] + lx[i] * mm[0][1]) + ly * (hx * mm[1][0] + lx[i] * mm[1][1]);
        }
    }
    const size_t o = (size_t)oy * a.OW + ((size_t)gx << 2);
    if (adv) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float4 v = flip ? make_float4(res[ch][3], res[ch][2], res[ch][1], res[ch][0])
                                  : make_float4(res[ch][0], res[ch][1], res[ch][2], res[ch][3]);
            *reinterpret_cast<float4*>(adv + ((size_t)n * 3 + ch) * ohw + o) = v;
        }
    }
    if (mask_out)
        *reinterpret_cast<float4*>(mask_out + (size_t)n * ohw + o) =
            flip ? make_float4(mres[3], mres[2], mres[1], mres[0]) : make_float4(mres[0], mres[1], mres[2], mres[3]);
}

// Tiled forward (OW % 4 == 0, resize ratios up to ~1.24 x 1.43: the attack's 375x1242 -> 320x1024).  A workgroup owns
// TH x TW output pixels: (1a) the scene rectangle their bilinear taps cover is copied once into LDS with coalesced
// row loads, (1b) if the object's bounding box meets that rectangle the composite is formed in place, ONCE per scene
// pixel (paste_fwd4 re-formed it for every output tap: 2.8x), (2) every thread resizes four output pixels out of LDS
// and stores them as 16-byte words.  Same per-pixel arithmetic as paste_fwd4_kernel.
constexpr int TH = 8, TW = 128;          // output tile (TH * TW / 4 == NT)
constexpr int RMAX = 12, CMAX = 160;     // scene rectangle held in LDS: 4 planes x 12 x 160 floats = 30 KB
constexpr int NIT = (RMAX * CMAX + NT - 1) / NT;

__device__ __forceinline__ int resize_src0(float ratio, int o) { return (int)fmaxf(ratio * ((float)o + 0.5f) - 0.5f, 0.f); }

__global__ __launch_bounds__(NT) void paste_fwd_tile_kernel(const dmh_paste_args a, float* __restrict__ adv,
                                                            float* __restrict__ mask_out) {
    __shared__ float tile[4][RMAX * CMAX];
    const int n = blockIdx.z, tid = threadIdx.x;
    const bool flip = a.flip && a.flip[n];
    const float* c = a.coeffs + n * 8;
    const float rh = (float)a.SH / (float)a.OH, rw = (float)a.SW / (float)a.OW;
    const int oyA = blockIdx.y * TH, oyB = min(oyA + TH, a.OH) - 1;
    const int oxA = blockIdx.x * TW, oxB = min(oxA + TW, a.OW) - 1;      // computed (un-mirrored) columns
    const int ry0 = resize_src0(rh, oyA), ry1 = min(resize_src0(rh, oyB) + 1, a.SH - 1);
    const int rx0 = resize_src0(rw, oxA), rx1 = min(resize_src0(rw, oxB) + 1, a.SW - 1);
    const int nr = min(ry1 - ry0 + 1, RMAX), nc = min(rx1 - rx0 + 1, CMAX);   // the host checked that nothing is clipped
    const int ne = nr * nc;
    const size_t shw = (size_t)a.SH * a.SW, phw = (size_t)a.PH * a.PW, ohw = (size_t)a.OH * a.OW;
    const bool warp_only = a.mode == DMH_PASTE_WARP_ONLY;
    const float* sc = warp_only ? nullptr : a.scene + (size_t)(a.scene_index ? a.scene_index[n] : n) * a.scene_bstride;

    // ---- 1a: scene rectangle -> LDS (mask plane 0)
    {
        const float inv_nc = 1.f / (float)nc;
        float v[NIT][3];
        int at[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int e = min(tid + i * NT, ne - 1);
            const int r = (int)(((float)e + 0.5f) * inv_nc), cc = e - r * nc;     // exact: e < 2^11
            at[i] = r * CMAX + cc;
            const size_t so = (size_t)(ry0 + r) * a.SW + (rx0 + cc);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) v[i][ch] = warp_only ? 0.f : sc[ch * shw + so];
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (tid + i * NT < ne) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) tile[ch][at[i]] = v[i][ch];
                tile[3][at[i]] = 0.f;
            }
        }
    }
    // ---- 1b: composite inside (bounding box  ∩  rectangle); wave-uniform decision
    {
        const ObjBox bb = object_box(c, a);
        int X0 = rx0, X1 = rx1, Y0 = ry0, Y1 = ry1;
        if (bb.ok) {     // pixel centres X + 0.5 inside [x0, x1] (clamped before the float -> int conversion)
            X0 = max(X0, (int)ceilf(fminf(fmaxf(bb.x0 - 0.5f, -1.f), (float)a.SW)));
            X1 = min(X1, (int)floorf(fminf(fmaxf(bb.x1 - 0.5f, -1.f), (float)a.SW)));
            Y0 = max(Y0, (int)ceilf(fminf(fmaxf(bb.y0 - 0.5f, -1.f), (float)a.SH)));
            Y1 = min(Y1, (int)floorf(fminf(fmaxf(bb.y1 - 0.5f, -1.f), (float)a.SH)));
        }
        const int bw = X1 - X0 + 1, bh = Y1 - Y0 + 1;
        if (bw > 0 && bh > 0) {
            const Homog m = load_homog(c, a.SW, a.SH);
            __syncthreads();                        // 1a's plain copies are in place
            const float inv_bw = 1.f / (float)bw;
#pragma unroll 1
            for (int e = tid; e < bw * bh; e += NT) {
                const int r = (int)(((float)e + 0.5f) * inv_bw), cc = e - r * bw;
                const int Y = Y0 + r, X = X0 + cc;
                const PTap t = patch_tap(m, X, Y, a.l_pad, a.t_pad, a.PW, a.PH);
                if (!t.any) continue;
                const int li = (Y - ry0) * CMAX + (X - rx0);
                const float mk = patch_sample(a.pmask, t, a.PW, a.PH);
                tile[3][li] = mk;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float o = patch_sample(a.patch + ch * phw, t, a.PW, a.PH);
                    tile[ch][li] = warp_only ? o : tile[ch][li] * (1.f - mk) + o * mk;      // phy_obj_atk.py:88
                }
            }
        }
    }
    __syncthreads();
    // ---- 2: bilinear resize of four output pixels per thread out of LDS
    const int oy = oyA + (tid >> 5), ox0 = oxA + ((tid & 31) << 2);
    if (oy > oyB || ox0 > oxB) return;
    const float sy = fmaxf(rh * ((float)oy + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, y1 = y0 + (y0 < a.SH - 1 ? 1 : 0);
    const float ly = sy - (float)y0, hy = 1.f - ly;
    const int j0 = (y0 - ry0) * CMAX - rx0, j1 = (y1 - ry0) * CMAX - rx0;
    float res[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float sx = fmaxf(rw * ((float)(ox0 + i) + 0.5f) - 0.5f, 0.f);
        const int x0 = (int)sx, x1 = x0 + (x0 < a.SW - 1 ? 1 : 0);
        const float lx = sx - (float)x0, hx = 1.f - lx;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
            res[ch][i] = hy * (hx * tile[ch][j0 + x0] + lx * tile[ch][j0 + x1]) + ly * (hx * tile[ch][j1 + x0] + lx * tile[ch][j1 + x1]);
    }
    const size_t o = (size_t)oy * a.OW + (flip ? a.OW - 4 - ox0 : ox0);
    if (adv) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            *reinterpret_cast<float4*>(adv + ((size_t)n * 3 + ch) * ohw + o) =
                flip ? make_float4(res[ch][3], res[ch][2], res[ch][1], res[ch][0])
                     : make_float4(res[ch][0], res[ch][1], res[ch][2], res[ch][3]);
    }
    if (mask_out)
        *reinterpret_cast<float4*>(mask_out + (size_t)n * ohw + o) =
            flip ? make_float4(res[3][3], res[3][2], res[3][1], res[3][0]) : make_float4(res[3][0], res[3][1], res[3][2], res[3][3]);
}

// ---------------------------------------------------------------------------------------------- backward
// Deterministic gather: a patch texel (v, u) collects, sample by sample and pixel by pixel in a fixed order, what the
// forward pass spread over it -- no float atomics, no zero-initialised output, run-to-run bitwise identical (an attack
// is a chain of sign() steps: bit-reproducible gradients make it replayable).  LPT lanes share a texel: lane q takes
// the samples n = q, q + LPT, ... (one each at the attack's 12 scenes, so 78,000 texels fill the chip instead of 1.2
// workgroups per CU walking 12 samples in turn) and a fixed xor-butterfly adds the LPT partial sums.
//   forward:  adv[n,c,oy,ox] = sum_{Y in {y0,y1}, X in {x0,x1}} wy wx [ scene (1 - mk) + o mk ](Y, X),
//             o(Y,X) = bilinear(patch, S_n(X, Y)),  mk = bilinear(mask, S_n(X, Y))          (S_n: the homography)
//   backward: g_patch[c,v,u] = sum_n sum_{(X,Y): texel (v,u) is a tap of S_n(X,Y)} tent * mk(Y,X) * G_n,c(Y,X),
//             G_n,c(Y,X)     = sum_{(oy,ox): (Y,X) is a resize tap of (oy,ox)} wy wx g_adv[n,c,oy,ox]
// The composite pixels (X, Y) that can reach a texel are those inside the image, under the inverse homography, of
// the 2x2-texel square around it: a handful (the object is 0.4-0.9 scene pixels per texel at 5-10 m).
constexpr int LPT = 16;

__global__ __launch_bounds__(NT) void paste_bwd_kernel(const dmh_paste_args a, const float* __restrict__ g_adv,
                                                       float* __restrict__ g_patch) {
    const int idx = blockIdx.x * (NT / LPT) + (threadIdx.x / LPT), sub = threadIdx.x % LPT;
    if (idx >= a.PH * a.PW) return;             // whole LPT-lane groups leave together
    const int v = idx / a.PW, u = idx - v * a.PW;
    const float Uc = (float)(u + a.l_pad) + 0.5f, Vc = (float)(v + a.t_pad) + 0.5f;   // texel centre, padded frame
    const size_t ohw = (size_t)a.OH * a.OW, phw = (size_t)a.PH * a.PW;
    const float rh = (float)a.SH / (float)a.OH, rw = (float)a.SW / (float)a.OW;
    const float irh = (float)a.OH / (float)a.SH, irw = (float)a.OW / (float)a.SW;
    const bool warp_only = a.mode == DMH_PASTE_WARP_ONLY;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
    for (int n = sub; n < a.N; n += LPT) {
        const float* c = a.coeffs + n * 8;
        const Homog m = load_homog(c, a.SW, a.SH);
        // adjugate of M = [[c0,c1,c2],[c3,c4,c5],[c6,c7,1]] (source = M * scene, pixel-centre coordinates)
        const float i00 = c[4] - c[5] * c[7], i01 = c[2] * c[7] - c[1], i02 = c[1] * c[5] - c[2] * c[4];
        const float i10 = c[5] * c[6] - c[3], i11 = c[0] - c[2] * c[6], i12 = c[2] * c[3] - c[0] * c[5];
        const float i20 = c[3] * c[7] - c[4] * c[6], i21 = c[1] * c[6] - c[0] * c[7], i22 = c[0] * c[4] - c[1] * c[3];
        float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float px = Uc + ((q & 1) ? 1.f : -1.f), py = Vc + ((q & 2) ? 1.f : -1.f);
            const float w = i20 * px + i21 * py + i22;   // adjugate: the common factor det(M) cancels in the ratios
            const float xs = (i00 * px + i01 * py + i02) / w, ys = (i10 * px + i11 * py + i12) / w;
            ok = ok && (xs == xs) && (ys == ys) && (c[6] * xs + c[7] * ys + 1.0f > 0.f);   // finite, in front of the horizon
            xmin = fminf(xmin, xs);
            xmax = fmaxf(xmax, xs);
            ymin = fminf(ymin, ys);
            ymax = fmaxf(ymax, ys);
        }
        if (!ok) continue;
        // candidates: pixel centres X + 0.5 inside the box (weights vanish at its border, so rounding there is harmless)
        const int Xlo = max(0, (int)ceilf(xmin - 0.5f - 1e-3f)), Xhi = min(a.SW - 1, (int)floorf(xmax - 0.5f + 1e-3f));
        const int Ylo = max(0, (int)ceilf(ymin - 0.5f - 1e-3f)), Yhi = min(a.SH - 1, (int)floorf(ymax - 0.5f + 1e-3f));
        if (Xhi - Xlo > 64 || Yhi - Ylo > 64) continue;            // degenerate projection (object filling the frame)
        const bool flip = a.flip && a.flip[n];
        const float* g0 = g_adv + (size_t)n * 3 * ohw;
        for (int Y = Ylo; Y <= Yhi; ++Y)
            for (int X = Xlo; X <= Xhi; ++X) {
                const PTap t = patch_tap(m, X, Y, a.l_pad, a.t_pad, a.PW, a.PH);
                if (!t.any) continue;
                const float wu = (u == t.x0) ? 1.f - t.fx : (u == t.x0 + 1 ? t.fx : 0.f);
                const float wv = (v == t.y0) ? 1.f - t.fy : (v == t.y0 + 1 ? t.fy : 0.f);
                float w = wu * wv;
                if (w == 0.f) continue;
                if (!warp_only) w *= patch_sample(a.pmask, t, a.PW, a.PH);
                if (w == 0.f) continue;
                // adjoint of the resize: output pixels whose bilinear taps include (Y, X)
                const int oylo = max(0, (int)floorf(((float)Y - 0.5f) * irh - 0.5f)), oyhi = min(a.OH - 1, (int)ceilf(((float)Y + 1.5f) * irh - 0.5f));
                const int oxlo = max(0, (int)floorf(((float)X - 0.5f) * irw - 0.5f)), oxhi = min(a.OW - 1, (int)ceilf(((float)X + 1.5f) * irw - 0.5f));
                float G0 = 0.f, G1 = 0.f, G2 = 0.f;
                for (int oy = oylo; oy <= oyhi; ++oy) {
                    const float sy = fmaxf(rh * ((float)oy + 0.5f) - 0.5f, 0.f);
                    const int y0 = (int)sy, y1 = y0 + (y0 < a.SH - 1 ? 1 : 0);
                    const float ly = sy - (float)y0;
                    const float wyy = (y0 == Y ? 1.f - ly : 0.f) + (y1 == Y ? ly : 0.f);
                    if (wyy == 0.f) continue;
                    for (int ox = oxlo; ox <= oxhi; ++ox) {
                        const float sx = fmaxf(rw * ((float)ox + 0.5f) - 0.5f, 0.f);
                        const int x0 = (int)sx, x1 = x0 + (x0 < a.SW - 1 ? 1 : 0);
                        const float lx = sx - (float)x0;
                        const float wxx = (x0 == X ? 1.f - lx : 0.f) + (x1 == X ? lx : 0.f);
                        if (wxx == 0.f) continue;
                        const size_t o = (size_t)oy * a.OW + (flip ? a.OW - 1 - ox : ox);
                        const float ww = wyy * wxx;
                        G0 = fmaf(ww, g0[o], G0);
                        G1 = fmaf(ww, g0[ohw + o], G1);
                        G2 = fmaf(ww, g0[2 * ohw + o], G2);
                    }
                }
                acc0 = fmaf(w, G0, acc0);
                acc1 = fmaf(w, G1, acc1);
                acc2 = fmaf(w, G2, acc2);
            }
    }
#pragma unroll
    for (int d = LPT / 2; d >= 1; d >>= 1) {    // a + b is commutative: every lane of the group ends with the same bits
        acc0 += __shfl_xor(acc0, d, LPT);
        acc1 += __shfl_xor(acc1, d, LPT);
        acc2 += __shfl_xor(acc2, d, LPT);
    }
    if (sub == 0) {
        g_patch[idx] = acc0;
        g_patch[phw + idx] = acc1;
        g_patch[2 * phw + idx] = acc2;
    }
}

// DMH_PASTE_FWD4=1 keeps the untiled four-pixel kernel (timing comparisons, tools/prof_k3.py)
inline bool force_paste4() {
    static const bool f = [] { const char* e = getenv("DMH_PASTE_FWD4"); return e && e[0] == '1'; }();
    return f;
}

int check_paste(const dmh_paste_args* a) {
    DMH_REQUIRE(a != nullptr, "args is null");
    DMH_REQUIRE(a->mode == DMH_PASTE_COMPOSITE || a->mode == DMH_PASTE_WARP_ONLY, "bad mode");
    DMH_REQUIRE((a->scene || a->mode == DMH_PASTE_WARP_ONLY) && a->patch && a->pmask && a->coeffs, "null input");
    DMH_REQUIRE(a->N > 0 && a->SH >= 2 && a->SW >= 2 && a->OH > 0 && a->OW > 0, "bad sizes");
    DMH_REQUIRE(a->PH > 0 && a->PW > 0 && a->l_pad >= 0 && a->t_pad >= 0, "bad patch geometry");
    DMH_REQUIRE(a->l_pad + a->PW <= a->SW && a->t_pad + a->PH <= a->SH, "patch does not fit the padded frame");
    DMH_REQUIRE(a->N <= 65535, "N too large for grid.y");
    return DMH_OK;
}

}  // namespace

extern "C" {

int dmh_eot_paste_fwd(const dmh_paste_args* a, float* adv, float* mask_out, void* stream) {
    if (int rc = check_paste(a)) return rc;
    DMH_REQUIRE(adv || mask_out, "no output requested");
    const bool vec4 = (a->OW % 4) == 0 && ((uintptr_t)adv % 16) == 0 && ((uintptr_t)mask_out % 16) == 0 &&
                      (a->mode == DMH_PASTE_WARP_ONLY || a->scene != nullptr);
    // the tiled kernel's LDS rectangle must hold the taps of a TH x TW output tile
    const float rh = (float)a->SH / (float)a->OH, rw = (float)a->SW / (float)a->OW;
    const bool tiled = vec4 && (int)floorf(rh * (TH - 1)) + 3 <= RMAX && (int)floorf(rw * (TW - 1)) + 3 <= CMAX && !force_paste4();
    if (tiled)
        hipLaunchKernelGGL(paste_fwd_tile_kernel, dim3((a->OW + TW - 1) / TW, (a->OH + TH - 1) / TH, a->N), dim3(NT), 0,
                           (hipStream_t)stream, *a, adv, mask_out);
    else if (vec4)
        hipLaunchKernelGGL(paste_fwd4_kernel, dim3((a->OH * (a->OW / 4) + NT - 1) / NT, a->N), dim3(NT), 0, (hipStream_t)stream,
                           *a, adv, mask_out);
    else
        hipLaunchKernelGGL(paste_fwd_kernel, dim3((a->OH * a->OW + NT - 1) / NT, a->N), dim3(NT), 0, (hipStream_t)stream,
                           *a, adv, mask_out);
    return check_launch("dmh_eot_paste_fwd");
}

int dmh_eot_paste_bwd(const dmh_paste_args* a, const float* g_adv, float* g_patch, void* stream) {
    if (int rc = check_paste(a)) return rc;
    DMH_REQUIRE(g_adv && g_patch, "null gradient buffers");
    const int per_block = NT / LPT;
    hipLaunchKernelGGL(paste_bwd_kernel, dim3((a->PH * a->PW + per_block - 1) / per_block), dim3(NT), 0, (hipStream_t)stream,
                       *a, g_adv, g_patch);
    return check_launch("dmh_eot_paste_bwd");
}

}  // extern "C"
