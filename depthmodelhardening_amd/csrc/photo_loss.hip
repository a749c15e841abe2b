// K1 -- fused photometric-reprojection loss for gfx950 (MI355X).
//
// One launch covers every scale: a 64x16 output tile keeps the target tile (+halo) resident in
// LDS, evaluates the identity term once, then for each scale warps the source view into an LDS
// tile (bilinear-upsampled disparity -> depth -> back-project -> project -> border-clamped
// bilinear gather), evaluates SSIM(3x3, reflect)+L1 out of LDS, takes the per-pixel min/argmin
// and accumulates block partial sums.  The backward kernel recomputes the warp on a halo-2 tile,
// turns d loss / d warped into three 3x3 box sums of per-pixel SSIM coefficient fields, and chains
// through the bilinear weights, the projective divide and 1/(a+b*disp).
//
// Reference semantics: MD2/trainer.py:472-537,589-660; MD2/layers.py:16-25,139-198,223-253;
// DH/trainer.py:557-590,638-708.  (file:line relative to /root/reference/DepthNetworks/...)
//
// HBM-bound stencil+gather: no MFMA.  Rows are read as 64-lane coalesced 256-B segments; the
// gather rides L1/L2 (stereo flow is horizontal, so neighbouring lanes hit neighbouring texels).
#include "common.hpp"

using namespace dmh;

namespace {

constexpr int TW = 64, TH = 16, NT = 256;
constexpr int PXT = TH / (NT / TW);  // pixels per thread (vertical run) = 4
static_assert(PXT == 4, "thread owns a vertical run of 4 pixels");

// forward: halo-1 tile
constexpr int F_HW = TW + 2, F_HH = TH + 2, F_LD = F_HW + 1, F_PLANE = F_HH * F_LD;
// backward: halo-2 tile
constexpr int B_HW = TW + 4, B_HH = TH + 4, B_LD = B_HW + 1, B_PLANE = B_HH * B_LD;

constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;

// min waves per SIMD requested from the register allocator (tuning knobs)
#ifndef DMH_FWD_WAVES
#define DMH_FWD_WAVES 2
#endif
#ifndef DMH_BWD_WAVES
#define DMH_BWD_WAVES 2
#endif


struct Cam {
    float ik[9];   // inv_K[:3,:3]           MD2/layers.py:164
    float P[12];   // (K @ T)[:3,:]          MD2/layers.py:188
    float A[9];    // P[:, :3] @ inv_K[:3,:3]: d(X,Y,Z)/d depth is affine in the pixel coordinates (fast path)
};

// second stage of load_cam (after a barrier): A = P[:, :3] @ ik
__device__ __forceinline__ void compose_cam(Cam* cam, int t) {
    if (t < 9) {
        const int i = t / 3, j = t % 3;
        cam->A[t] = cam->P[i * 4 + 0] * cam->ik[0 * 3 + j] + cam->P[i * 4 + 1] * cam->ik[1 * 3 + j] +
                    cam->P[i * 4 + 2] * cam->ik[2 * 3 + j];
    }
}

struct KArgs {
    dmh_photo_args a;
    float* sel[DMH_MAX_SCALES];
    float* to_opt[DMH_MAX_SCALES];
    float* partials;
    const float* csel[DMH_MAX_SCALES];
    const float* gvec;
    const float* fin;
    float* g_up[DMH_MAX_SCALES];
    int tiles_x, tiles_y, nblk;
    float min_disp, dmul;  // scaled_disp = min_disp + dmul*disp   MD2/layers.py:21-23
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
// FAST (the fused loss kernels): reciprocals by rcp + one Newton step instead of IEEE divides, and the sampling
// coordinate is the projected pixel itself instead of the reference's x/(W-1) -> (.-0.5)*2 -> ((.+1)/2)*(W-1)
// round trip (an identity up to ~1.5 ulp of the coordinate, 6e-5 px at x~500, which both sides carry anyway).
template <bool FAST>
__device__ __forceinline__ Proj project(const Cam& c, float disp, int x, int y, int H, int W, float min_disp,
                                        float dmul) {
    Proj p;
    const float sd = min_disp + dmul * disp;
    p.depth = FAST ? fast_rcp(sd) : 1.0f / sd;
    const float fx = (float)x, fy = (float)y;
    if (FAST) {
        p.ax = c.A[0] * fx + c.A[1] * fy + c.A[2];
        p.ay = c.A[3] * fx + c.A[4] * fy + c.A[5];
        p.az = c.A[6] * fx + c.A[7] * fy + c.A[8];
        p.den = (p.depth * p.az + c.P[11]) + 1e-7f;
        const float rden = fast_rcp(p.den);
        p.px = (p.depth * p.ax + c.P[3]) * rden;
        p.py = (p.depth * p.ay + c.P[7]) * rden;
        p.ix = p.px;
        p.iy = p.py;
        return p;
    }
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

// Gather through a buffer resource: one 128-bit descriptor in SGPRs per source image, 32-bit byte offsets in
// VGPRs and the colour-plane offset in an SGPR -- no 64-bit address arithmetic per load (guide T8).
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const float* p, unsigned bytes) {
    // the pointer is the same in every lane (derived from blockIdx); tell the compiler so
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}

__device__ __forceinline__ float tap_sample_buf(rsrc_t rs, unsigned plane_bytes, const Tap& t) {
#ifdef DMH_ABLATE_GATHER
    return (float)t.o00 * t.w00 + (float)t.o01 * t.w01 + (float)t.o10 * t.w10 + (float)t.o11 * t.w11;
#else
    const float v00 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, t.o00 * 4u, plane_bytes, 0));
    const float v01 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, t.o01 * 4u, plane_bytes, 0));
    const float v10 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, t.o10 * 4u, plane_bytes, 0));
    const float v11 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, t.o11 * 4u, plane_bytes, 0));
    return v00 * t.w00 + v01 * t.w01 + v10 * t.w10 + v11 * t.w11;
#endif
}

__device__ __forceinline__ float ld_buf(rsrc_t rs, unsigned elem) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, elem * 4u, 0, 0));
}

// disp_at through a buffer resource (same arithmetic)
__device__ __forceinline__ float disp_at_buf(rsrc_t rd, int Hs, int Ws, float rh, float rw, bool same, int y, int x) {
    if (same) return ld_buf(rd, (unsigned)(y * Ws + x));
    const float sy = fmaxf(rh * ((float)y + 0.5f) - 0.5f, 0.f);
    const float sx = fmaxf(rw * ((float)x + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float v00 = ld_buf(rd, (unsigned)(y0 * Ws + x0)), v01 = ld_buf(rd, (unsigned)(y0 * Ws + x1));
    const float v10 = ld_buf(rd, (unsigned)(y1 * Ws + x0)), v11 = ld_buf(rd, (unsigned)(y1 * Ws + x1));
    return hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
}

__device__ __forceinline__ float tap_sample(const float* __restrict__ img, const Tap& t) {
#ifdef DMH_ABLATE_GATHER  // timing-only build: no gather loads (results are wrong)
    return (float)t.o00 * t.w00 + (float)t.o01 * t.w01 + (float)t.o10 * t.w10 + (float)t.o11 * t.w11;
#else
    return img[t.o00] * t.w00 + img[t.o01] * t.w01 + img[t.o10] * t.w10 + img[t.o11] * t.w11;
#endif
}

// SSIM window statistics with every factor scaled by 81 (mu = s/9): SSIM_n/SSIM_d is unchanged,
// the five divisions by 9 disappear and one division is left (MD2/layers.py:243-253).
struct SsimTerms {
    float A1, A2, B1, B2;  // 81*(2 mu_x mu_y + C1), 81*(2 sigma_xy + C2), 81*(mu_x^2+mu_y^2+C1), 81*(sigma_x+sigma_y+C2)
};

__device__ __forceinline__ SsimTerms ssim_terms(float sx, float sy, float sxx, float syy, float sxy) {
    const float sxsy = sx * sy, sx2 = sx * sx, sy2 = sy * sy;
    SsimTerms t;
    t.A1 = 2.f * sxsy + 81.f * C1;
    t.A2 = 2.f * (9.f * sxy - sxsy) + 81.f * C2;
    t.B1 = sx2 + sy2 + 81.f * C1;
    t.B2 = (9.f * (sxx + syy) - sx2 - sy2) + 81.f * C2;
    return t;
}

__device__ __forceinline__ float ssim_val(float sx, float sy, float sxx, float syy, float sxy) {
    const SsimTerms t = ssim_terms(sx, sy, sxx, syy, sxy);
    const float r = (t.A1 * t.A2) * __builtin_amdgcn_rcpf(t.B1 * t.B2);  // 1-ulp reciprocal: << the SSIM conditioning noise
    return fminf(fmaxf((1.f - r) * 0.5f, 0.f), 1.f);
}

// compute_reprojection_loss (MD2/trainer.py:525-537) for the thread's 4 vertically adjacent pixels.
// sp/st: LDS planes [3][rows][LD] of pred and target; (row0,col0) = LDS coords of the first window's corner.
template <int LD, int PLANE>
__device__ __forceinline__ void reproj4(const float* sp, const float* st, int row0, int col0, bool no_ssim,
                                        float out[PXT]) {
    float ss[PXT] = {0.f, 0.f, 0.f, 0.f}, l1[PXT] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* P = sp + c * PLANE + row0 * LD + col0;
        const float* Q = st + c * PLANE + row0 * LD + col0;
        float hx[PXT + 2], hy[PXT + 2], hxx[PXT + 2], hyy[PXT + 2], hxy[PXT + 2];
#pragma unroll
        for (int r = 0; r < PXT + 2; ++r) {
            const float x0 = P[r * LD], x1 = P[r * LD + 1], x2 = P[r * LD + 2];
            const float y0 = Q[r * LD], y1 = Q[r * LD + 1], y2 = Q[r * LD + 2];
            hx[r] = x0 + x1 + x2;
            hy[r] = y0 + y1 + y2;
            hxx[r] = x0 * x0 + x1 * x1 + x2 * x2;
            hyy[r] = y0 * y0 + y1 * y1 + y2 * y2;
            hxy[r] = x0 * y0 + x1 * y1 + x2 * y2;
            if (r >= 1 && r <= PXT) l1[r - 1] += fabsf(y1 - x1);
        }
        if (!no_ssim) {
#pragma unroll
            for (int k = 0; k < PXT; ++k)
                ss[k] += ssim_val(hx[k] + hx[k + 1] + hx[k + 2], hy[k] + hy[k + 1] + hy[k + 2],
                                  hxx[k] + hxx[k + 1] + hxx[k + 2], hyy[k] + hyy[k + 1] + hyy[k + 2],
                                  hxy[k] + hxy[k + 1] + hxy[k + 2]);
        }
    }
#pragma unroll
    for (int k = 0; k < PXT; ++k)
        out[k] = no_ssim ? l1[k] * (1.f / 3.f) : (0.85f / 3.f) * ss[k] + (0.15f / 3.f) * l1[k];
}

// Copy an image tile with halo HALO into LDS planes [3][HH][LD]; out-of-image slots use the
// ReflectionPad2d(1) index map (MD2/layers.py:234,240-241).
template <int HALO, int HW_, int HH_, int LD, int PLANE>
__device__ __forceinline__ void load_tile(float* s, const float* __restrict__ img, int H, int W, int x0, int y0) {
    for (int i = threadIdx.x; i < 3 * HH_ * HW_; i += NT) {
        const int c = i / (HH_ * HW_);
        const int rem = i - c * (HH_ * HW_);
        const int r = rem / HW_, col = rem - r * HW_;
        const int gy = reflect_idx(y0 - HALO + r, H), gx = reflect_idx(x0 - HALO + col, W);
        s[c * PLANE + r * LD + col] = img[(c * H + gy) * W + gx];
    }
}

// Warp the source view into an LDS tile with halo HALO (the (scale, frame) body of generate_images_pred).
template <int HALO, int HW_, int HH_, int LD, int PLANE>
__device__ __forceinline__ void warp_tile(float* s, const float* __restrict__ src, const float* __restrict__ disp,
                                          const Cam& cam, int H, int W, int Hs, int Ws, int x0, int y0, float min_disp,
                                          float dmul) {
    const bool same = (Hs == H && Ws == W);
    const float rh = (float)Hs / (float)H, rw = (float)Ws / (float)W;
    const unsigned plane = (unsigned)(H * W) * 4u;
    const rsrc_t rs = make_rsrc(src, 3u * plane);
    const rsrc_t rd = make_rsrc(disp, (unsigned)(Hs * Ws) * 4u);
    for (int i = threadIdx.x; i < HH_ * HW_; i += NT) {
        const int r = i / HW_, col = i - r * HW_;
        const int gy = reflect_idx(y0 - HALO + r, H), gx = reflect_idx(x0 - HALO + col, W);
        const float d = disp_at_buf(rd, Hs, Ws, rh, rw, same, gy, gx);
        const Proj p = project<true>(cam, d, gx, gy, H, W, min_disp, dmul);
        const Tap t = make_tap(p.ix, p.iy, H, W);
        s[0 * PLANE + r * LD + col] = tap_sample_buf(rs, 0u, t);
        s[1 * PLANE + r * LD + col] = tap_sample_buf(rs, plane, t);
        s[2 * PLANE + r * LD + col] = tap_sample_buf(rs, 2u * plane, t);
    }
}

__device__ __forceinline__ void decode_block(const KArgs& k, int& tx0, int& ty0, int& b, int& blk) {
    // XCD-aware mapping: consecutive workgroup ids round-robin over the 8 XCDs, so give each XCD a
    // contiguous run of tiles (neighbouring tiles share halo rows and gather lines in that XCD's L2).
    int id = blockIdx.x;
    const int n = k.nblk;
    if ((n & 7) == 0) id = (id & 7) * (n >> 3) + (id >> 3);
    blk = id;
    const int per_img = k.tiles_x * k.tiles_y;
    b = id / per_img;
    const int t = id - b * per_img;
    ty0 = (t / k.tiles_x) * TH;
    tx0 = (t - (t / k.tiles_x) * k.tiles_x) * TW;
}

// ------------------------------------------------------------------------------------------------ forward
template <int NF>
__global__ __launch_bounds__(NT, DMH_FWD_WAVES) void photo_fwd_kernel(const KArgs k) {
    __shared__ float s_tgt[3 * F_PLANE];
    __shared__ float s_buf[3 * F_PLANE];
    __shared__ Cam s_cam[DMH_MAX_FRAMES];
    __shared__ float s_red[NT / WAVE];

    const dmh_photo_args& a = k.a;
    const int H = a.H, W = a.W;
    constexpr int F = NF;
    int x0, y0, b, blk;
    decode_block(k, x0, y0, b, blk);
    const int tid = threadIdx.x, tx = tid & (TW - 1), tg = tid / TW;
    const size_t img_off = (size_t)b * 3 * H * W;

    if (tid < 21 * F) load_cam(&s_cam[tid / 21], a.K, a.inv_K, a.T[tid / 21], b, tid % 21);
    load_tile<1, F_HW, F_HH, F_LD, F_PLANE>(s_tgt, a.target + img_off, H, W, x0, y0);
    __syncthreads();
    if (tid < 9 * F) compose_cam(&s_cam[tid / 9], tid % 9);

    const int qx = x0 + tx;
    const int qy0 = y0 + tg * PXT;
    bool valid[PXT];
#pragma unroll
    for (int i = 0; i < PXT; ++i) valid[i] = (qx < W) && (qy0 + i < H);

    // identity terms (same for every scale; the reference recomputes them 4x, MD2/trainer.py:608-621)
    float ident[NF][PXT];
    if (a.automask) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            __syncthreads();
            load_tile<1, F_HW, F_HH, F_LD, F_PLANE>(s_buf, a.source[f] + img_off, H, W, x0, y0);
            __syncthreads();
            reproj4<F_LD, F_PLANE>(s_buf, s_tgt, tg * PXT, tx, a.no_ssim != 0, ident[f]);
        }
    }

    const Philox<7> rng(a.seed);
    float acc1[DMH_MAX_SCALES], acc2[DMH_MAX_SCALES];
#pragma unroll
    for (int s = 0; s < DMH_MAX_SCALES; ++s) {
        if (s >= a.num_scales) break;
        float best[PXT];
        int bestf[PXT];
#pragma unroll
        for (int i = 0; i < PXT; ++i) {
            best[i] = 3.0e38f;
            bestf[i] = 0;
        }
        const float* disp = a.disp[s] + (size_t)b * a.Hs[s] * a.Ws[s];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            __syncthreads();
            warp_tile<1, F_HW, F_HH, F_LD, F_PLANE>(s_buf, a.source[f] + img_off, disp, s_cam[f], H, W, a.Hs[s],
                                                    a.Ws[s], x0, y0, k.min_disp, k.dmul);
            __syncthreads();
            float v[PXT];
            reproj4<F_LD, F_PLANE>(s_buf, s_tgt, tg * PXT, tx, a.no_ssim != 0, v);
#pragma unroll
            for (int i = 0; i < PXT; ++i)
                if (v[i] < best[i]) {
                    best[i] = v[i];
                    bestf[i] = f;
                }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < PXT; ++i) {
            if (!valid[i]) continue;
            const int qy = qy0 + i;
            const size_t pix = ((size_t)b * H + qy) * W + qx;
            bool chosen = true;
            float val = best[i];
            if (a.automask) {
                float idn = 3.0e38f;
                const int nf = (a.variant == DMH_VARIANT_MD2) ? F : 1;
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    float nz = 0.f;
                    const int fi = (a.variant == DMH_VARIANT_MD2) ? f : 0;
                    if (a.noise_mode == DMH_NOISE_TENSOR) {
                        nz = a.noise[s][(((size_t)b * nf + fi) * H + qy) * W + qx];
                    } else if (a.noise_mode == DMH_NOISE_PHILOX) {
                        // one Philox call serves the thread's 4 pixels: two Box-Muller pairs (cos, sin) each
                        const uint64_t ctr = a.offset + ((((uint64_t)s * a.B + b) * nf + fi) * H + qy0) * W + qx;
                        const uint4 r = rng(ctr, 0x646d68ull);
                        const float2 n01 = normal_pair_from_bits(r.x, r.y), n23 = normal_pair_from_bits(r.z, r.w);
                        nz = (i == 0 ? n01.x : i == 1 ? n01.y : i == 2 ? n23.x : n23.y) * 0.00001f;
                    }
                    // MD2: noise per identity channel, then min over channels (trainer.py:642-654);
                    // DH : min over frames first, one noise plane (DH/trainer.py:671,687-690)
                    const float cand = ident[f][i] + ((a.variant == DMH_VARIANT_MD2) ? nz : 0.f);
                    idn = fminf(idn, cand);
                    if (a.variant != DMH_VARIANT_MD2 && f == F - 1) idn += nz;
                }
                if (a.variant == DMH_VARIANT_MD2) {
                    chosen = best[i] < idn;  // torch.min keeps the first (identity) on ties
                    val = chosen ? best[i] : idn;
                } else {
                    chosen = best[i] <= idn;  // argmin over [reprojection, identity]: first wins ties
                    val = chosen ? best[i] : 0.f;
                }
            }
            k.sel[s][pix] = chosen ? (float)(1 + bestf[i]) : 0.f;
            if (k.to_opt[s]) k.to_opt[s][pix] = val;
            s1 += val;
            s2 += chosen ? 1.f : 0.f;
        }
        acc1[s] = s1;
        acc2[s] = s2;
    }
    // one reduction for all scales at the end (keeps the scale loop free of extra barriers)
#pragma unroll
    for (int s = 0; s < DMH_MAX_SCALES; ++s) {
        if (s >= a.num_scales) break;
        const float t1 = block_sum<NT>(acc1[s], s_red);
        const float t2 = block_sum<NT>(acc2[s], s_red);
        if (tid == 0) {
            k.partials[((size_t)s * k.nblk + blk) * 2 + 0] = t1;
            k.partials[((size_t)s * k.nblk + blk) * 2 + 1] = t2;
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// gradient of one warped pixel back to the up-sampled disparity, given d loss / d warped (3 channels)
template <bool FAST>
__device__ __forceinline__ float warp_pixel_bwd(const float* __restrict__ src, const Cam& cam, float d, int x, int y,
                                                int H, int W, float min_disp, float dmul, float g0, float g1,
                                                float g2) {
    const Proj p = project<FAST>(cam, d, x, y, H, W, min_disp, dmul);
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
    const float g_depth = (gix * (p.ax - p.px * p.az) + giy * (p.ay - p.py * p.az)) * (FAST ? fast_rcp(p.den) : 1.0f / p.den);
    return g_depth * (-(p.depth * p.depth)) * dmul;
}

// fused-kernel version of warp_pixel_bwd: buffer-resource gathers, fast projection
__device__ __forceinline__ float warp_pixel_bwd_buf(rsrc_t rs, unsigned plane, const Cam& cam, float d, int x, int y,
                                                    int H, int W, float min_disp, float dmul, float g0, float g1,
                                                    float g2) {
    const Proj p = project<true>(cam, d, x, y, H, W, min_disp, dmul);
    const Tap t = make_tap(p.ix, p.iy, H, W);
    float gix = 0.f, giy = 0.f;
    const float gc[3] = {g0, g1, g2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned po = plane * (unsigned)c;
        const float v00 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, t.o00 * 4u, po, 0));
        const float v01 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, t.o01 * 4u, po, 0));
        const float v10 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, t.o10 * 4u, po, 0));
        const float v11 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, t.o11 * 4u, po, 0));
        gix += gc[c] * ((v01 - v00) * (1.f - t.fy) + (v11 - v10) * t.fy);
        giy += gc[c] * ((v10 - v00) * (1.f - t.fx) + (v11 - v01) * t.fx);
    }
    if (!(p.ix > 0.f && p.ix < (float)(W - 1))) gix = 0.f;
    if (!(p.iy > 0.f && p.iy < (float)(H - 1))) giy = 0.f;
    const float g_depth = (gix * (p.ax - p.px * p.az) + giy * (p.ay - p.py * p.az)) * fast_rcp(p.den);
    return g_depth * (-(p.depth * p.depth)) * dmul;
}

template <int NF>
__global__ __launch_bounds__(NT, DMH_BWD_WAVES) void photo_bwd_kernel(const KArgs k) {
    __shared__ float s_tgt[3 * B_PLANE];
    __shared__ float s_wrp[3 * B_PLANE];
    __shared__ float s_cf[3 * F_PLANE];  // a0, ax, ay coefficient fields on the halo-1 tile
    __shared__ Cam s_cam[DMH_MAX_FRAMES];

    const dmh_photo_args& a = k.a;
    const int H = a.H, W = a.W;
    constexpr int F = NF;
    int x0, y0, b, blk;
    decode_block(k, x0, y0, b, blk);
    const int tid = threadIdx.x, tx = tid & (TW - 1), tg = tid / TW;
    const size_t img_off = (size_t)b * 3 * H * W;

    if (tid < 21 * F) load_cam(&s_cam[tid / 21], a.K, a.inv_K, a.T[tid / 21], b, tid % 21);
    load_tile<2, B_HW, B_HH, B_LD, B_PLANE>(s_tgt, a.target + img_off, H, W, x0, y0);
    __syncthreads();
    if (tid < 9 * F) compose_cam(&s_cam[tid / 9], tid % 9);

    const int qx = x0 + tx, qy0 = y0 + tg * PXT;
    const float mxl = (qx == 1) ? 2.f : 1.f, mxr = (qx == W - 2) ? 2.f : 1.f;  // reflection-pad adjoint

#pragma unroll
    for (int s = 0; s < DMH_MAX_SCALES; ++s) {
        if (s >= a.num_scales) break;
        float up = k.gvec[DMH_FIN_LOSS] / (float)a.num_scales + k.gvec[DMH_FIN_LOSS_S + s] +
                   k.gvec[DMH_FIN_REPROJ_S + s];
        up *= (a.variant == DMH_VARIANT_MD2) ? 1.0f / ((float)a.B * (float)H * (float)W)
                                             : 1.0f / (k.fin[DMH_FIN_COUNT_S + s] + 1e-7f);
        const float* disp = a.disp[s] + (size_t)b * a.Hs[s] * a.Ws[s];
        const float* sel = k.csel[s] + (size_t)b * H * W;
        const bool same = (a.Hs[s] == H && a.Ws[s] == W);
        const float rh = (float)a.Hs[s] / (float)H, rw = (float)a.Ws[s] / (float)W;
        float acc[PXT] = {0.f, 0.f, 0.f, 0.f};

        for (int f = 0; f < F; ++f) {
            const float* src = a.source[f] + img_off;
            __syncthreads();
            warp_tile<2, B_HW, B_HH, B_LD, B_PLANE>(s_wrp, src, disp, s_cam[f], H, W, a.Hs[s], a.Ws[s], x0, y0,
                                                    k.min_disp, k.dmul);
            __syncthreads();
            float gw[3][PXT];
            const float fsel = (float)(1 + f);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // (1) coefficient fields on the halo-1 tile: d v(px)/d x_q = a0 + ay*y_q + ax*x_q
                for (int i = tid; i < F_HH * F_HW; i += NT) {
                    const int r = i / F_HW, col = i - r * F_HW;
                    const int py = y0 - 1 + r, px = x0 - 1 + col;
                    float a0 = 0.f, cax = 0.f, cay = 0.f;
                    if (!a.no_ssim && py >= 0 && py < H && px >= 0 && px < W && sel[py * W + px] == fsel) {
                        const float* P = s_wrp + c * B_PLANE + r * B_LD + col;
                        const float* Q = s_tgt + c * B_PLANE + r * B_LD + col;
                        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
                        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                            for (int cc = 0; cc < 3; ++cc) {
                                const float xv = P[rr * B_LD + cc], yv = Q[rr * B_LD + cc];
                                sx += xv;
                                sy += yv;
                                sxx += xv * xv;
                                syy += yv * yv;
                                sxy += xv * yv;
                            }
                        // d v/d x_q = a0 + ay*y_q + ax*x_q with v = clamp((1 - n/d)/2); in the 81-scaled terms
                        //   a0 = -G (s_y (A2-A1) - r s_x (B2-B1)) / d',  ay = -9 G A1 / d',  ax = 9 G r B1 / d'
                        const SsimTerms t = ssim_terms(sx, sy, sxx, syy, sxy);
                        const float invd = fast_rcp(t.B1 * t.B2);
                        const float rr_ = (t.A1 * t.A2) * invd;
                        const float v = (1.f - rr_) * 0.5f;
                        if (v >= 0.f && v <= 1.f) {  // clamp passes gradient on the closed interval
                            const float gs = up * (0.85f / 3.f) * invd;
                            a0 = -gs * (sy * (t.A2 - t.A1) - rr_ * sx * (t.B2 - t.B1));
                            cay = -9.f * gs * t.A1;
                            cax = 9.f * gs * rr_ * t.B1;
                        }
                    }
                    s_cf[0 * F_PLANE + r * F_LD + col] = a0;
                    s_cf[1 * F_PLANE + r * F_LD + col] = cax;
                    s_cf[2 * F_PLANE + r * F_LD + col] = cay;
                }
                __syncthreads();
                // (2) 3x3 box sums (with the reflection fold) for the thread's 4 pixels
                float h0[PXT + 2], h1[PXT + 2], h2[PXT + 2];
#pragma unroll
                for (int r = 0; r < PXT + 2; ++r) {
                    const float* c0 = s_cf + (tg * PXT + r) * F_LD + tx;
                    h0[r] = mxl * c0[0] + c0[1] + mxr * c0[2];
                    h1[r] = mxl * c0[F_PLANE] + c0[F_PLANE + 1] + mxr * c0[F_PLANE + 2];
                    h2[r] = mxl * c0[2 * F_PLANE] + c0[2 * F_PLANE + 1] + mxr * c0[2 * F_PLANE + 2];
                }
#pragma unroll
                for (int i = 0; i < PXT; ++i) {
                    const int qy = qy0 + i;
                    const float myt = (qy == 1) ? 2.f : 1.f, myb = (qy == H - 2) ? 2.f : 1.f;
                    const float S0 = myt * h0[i] + h0[i + 1] + myb * h0[i + 2];
                    const float Sx = myt * h1[i] + h1[i + 1] + myb * h1[i + 2];
                    const float Sy = myt * h2[i] + h2[i + 1] + myb * h2[i + 2];
                    const float xq = s_wrp[c * B_PLANE + (tg * PXT + i + 2) * B_LD + tx + 2];
                    const float yq = s_tgt[c * B_PLANE + (tg * PXT + i + 2) * B_LD + tx + 2];
                    float g = S0 + yq * Sy + xq * Sx;
                    if (qx < W && qy < H && sel[qy * W + qx] == fsel) {
                        const float l1w = a.no_ssim ? (1.f / 3.f) : (0.15f / 3.f);
                        const float df = xq - yq;
                        g += up * l1w * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
                    }
                    gw[c][i] = g;
                }
                __syncthreads();
            }
            // (3) chain through the bilinear gather, the projective divide and disp_to_depth
            const unsigned plane = (unsigned)(H * W) * 4u;
            const rsrc_t rs = make_rsrc(src, 3u * plane);
            const rsrc_t rd = make_rsrc(disp, (unsigned)(a.Hs[s] * a.Ws[s]) * 4u);
#pragma unroll
            for (int i = 0; i < PXT; ++i) {
                const int qy = qy0 + i;
                if (qx < W && qy < H) {
                    const float d = disp_at_buf(rd, a.Hs[s], a.Ws[s], rh, rw, same, qy, qx);
                    acc[i] += warp_pixel_bwd_buf(rs, plane, s_cam[f], d, qx, qy, H, W, k.min_disp, k.dmul, gw[0][i],
                                                 gw[1][i], gw[2][i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PXT; ++i) {
            const int qy = qy0 + i;
            if (qx < W && qy < H) k.g_up[s][((size_t)b * H + qy) * W + qx] = acc[i];
        }
    }
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
    const Proj p = project<false>(s_cam, d, x, y, H, W, min_disp, dmul);
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
    float g = warp_pixel_bwd<false>(source + (size_t)b * 3 * hw, s_cam, d, x, y, H, W, min_disp, dmul, g0, g1, g2);
    if (grad_depth) {
        const float sd = min_disp + dmul * d;
        g += grad_depth[(size_t)b * hw + idx] * (-1.0f / (sd * sd)) * dmul;
    }
    g_up[(size_t)b * hw + idx] = g;
}

int check_photo(const dmh_photo_args* a) {
    DMH_REQUIRE(a != nullptr, "args is null");
    DMH_REQUIRE(a->B > 0 && a->H >= 3 && a->W >= 3, "need B>0, H>=3, W>=3");
    DMH_REQUIRE(a->num_frames >= 1 && a->num_frames <= DMH_MAX_FRAMES, "num_frames out of range");
    DMH_REQUIRE(a->num_scales >= 1 && a->num_scales <= DMH_MAX_SCALES, "num_scales out of range");
    DMH_REQUIRE(a->target && a->K && a->inv_K, "null target/K/inv_K");
    DMH_REQUIRE(a->min_depth > 0.f && a->max_depth > a->min_depth, "bad depth range");
    DMH_REQUIRE(a->variant == DMH_VARIANT_MD2 || a->variant == DMH_VARIANT_DH, "unknown variant");
    DMH_REQUIRE((int64_t)a->B * 3 * a->H * a->W < (int64_t)1 << 40, "tensor too large");
    for (int f = 0; f < a->num_frames; ++f) DMH_REQUIRE(a->source[f] && a->T[f], "null source/T");
    for (int s = 0; s < a->num_scales; ++s) {
        DMH_REQUIRE(a->disp[s] != nullptr, "null disp");
        DMH_REQUIRE(a->Hs[s] >= 1 && a->Ws[s] >= 1 && a->Hs[s] <= a->H && a->Ws[s] <= a->W, "bad disp size");
        if (a->noise_mode == DMH_NOISE_TENSOR) DMH_REQUIRE(a->noise[s] != nullptr, "null noise tensor");
    }
    DMH_REQUIRE(a->noise_mode >= DMH_NOISE_NONE && a->noise_mode <= DMH_NOISE_PHILOX, "bad noise_mode");
    return DMH_OK;
}

void fill_kargs(KArgs& k, const dmh_photo_args* a) {
    memset(&k, 0, sizeof(k));
    k.a = *a;
    k.tiles_x = (a->W + TW - 1) / TW;
    k.tiles_y = (a->H + TH - 1) / TH;
    k.nblk = k.tiles_x * k.tiles_y * a->B;
    const double min_disp = 1.0 / (double)a->max_depth, max_disp = 1.0 / (double)a->min_depth;
    k.min_disp = (float)min_disp;
    k.dmul = (float)(max_disp - min_disp);
}

}  // namespace

extern "C" {

int64_t dmh_photo_partials_size(int B, int H, int W, int num_scales) {
    const int64_t tiles = (int64_t)((W + TW - 1) / TW) * ((H + TH - 1) / TH) * B;
    return tiles * 2 * num_scales;
}

int dmh_photo_loss_fwd(const dmh_photo_args* a, float* const sel[DMH_MAX_SCALES],
                       float* const to_opt[DMH_MAX_SCALES], float* partials, void* stream) {
    if (int rc = check_photo(a)) return rc;
    DMH_REQUIRE(sel != nullptr && partials != nullptr, "null outputs");
    KArgs k;
    fill_kargs(k, a);
    for (int s = 0; s < a->num_scales; ++s) {
        DMH_REQUIRE(sel[s] != nullptr, "null sel[s]");
        k.sel[s] = sel[s];
        k.to_opt[s] = to_opt ? to_opt[s] : nullptr;
    }
    k.partials = partials;
    switch (a->num_frames) {
        case 1: hipLaunchKernelGGL(photo_fwd_kernel<1>, dim3(k.nblk), dim3(NT), 0, (hipStream_t)stream, k); break;
        case 2: hipLaunchKernelGGL(photo_fwd_kernel<2>, dim3(k.nblk), dim3(NT), 0, (hipStream_t)stream, k); break;
        case 3: hipLaunchKernelGGL(photo_fwd_kernel<3>, dim3(k.nblk), dim3(NT), 0, (hipStream_t)stream, k); break;
        default: hipLaunchKernelGGL(photo_fwd_kernel<4>, dim3(k.nblk), dim3(NT), 0, (hipStream_t)stream, k); break;
    }
    return check_launch("dmh_photo_loss_fwd");
}

int dmh_photo_loss_bwd(const dmh_photo_args* a, const float* const sel[DMH_MAX_SCALES], const float* gvec,
                       const float* fin, float* const g_up[DMH_MAX_SCALES], void* stream) {
    if (int rc = check_photo(a)) return rc;
    DMH_REQUIRE(sel && gvec && fin && g_up, "null argument");
    KArgs k;
    fill_kargs(k, a);
    for (int s = 0; s < a->num_scales; ++s) {
        DMH_REQUIRE(sel[s] != nullptr && g_up[s] != nullptr, "null sel[s]/g_up[s]");
        k.csel[s] = sel[s];
        k.g_up[s] = g_up[s];
    }
    k.gvec = gvec;
    k.fin = fin;
    switch (a->num_frames) {
        case 1: hipLaunchKernelGGL(photo_bwd_kernel<1>, dim3(k.nblk), dim3(NT), 0, (hipStream_t)stream, k); break;
        case 2: hipLaunchKernelGGL(photo_bwd_kernel<2>, dim3(k.nblk), dim3(NT), 0, (hipStream_t)stream, k); break;
        case 3: hipLaunchKernelGGL(photo_bwd_kernel<3>, dim3(k.nblk), dim3(NT), 0, (hipStream_t)stream, k); break;
        default: hipLaunchKernelGGL(photo_bwd_kernel<4>, dim3(k.nblk), dim3(NT), 0, (hipStream_t)stream, k); break;
    }
    return check_launch("dmh_photo_loss_bwd");
}

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
