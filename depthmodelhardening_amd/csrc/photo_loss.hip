// K1 -- fused photometric-reprojection loss for gfx950 (MI355X), wave-strip formulation.
//
// One wave owns a vertical strip of the image: lane l holds column X0 - halo + l and the wave walks down the rows.
// Everything a 3x3 SSIM window needs from the neighbouring columns comes from the neighbouring LANES (DPP wave
// shifts fused into v_add_f32), everything it needs from the neighbouring rows from a rolling pair of row-sum
// records in registers -- no LDS tile, no barrier, every pixel's warp is evaluated exactly once per strip.
//
//   forward : per row, the target row sums once, the identity term once (the reference recomputes it per scale,
//             MD2/trainer.py:608-621), then for every scale bilinear-upsampled disparity -> projection -> border-
//             clamped bilinear gather of the source (buffer loads, L1/L2 resident) -> row sums -> SSIM + L1 for
//             the row above -> per-pixel min / argmin -> one selection byte per pixel (2 bits per scale) and
//             fixed-order partial sums per wave (bitwise reproducible, no atomics).
//   backward: one wave per (strip, scale); recomputes the warp, turns d loss / d warped into three SSIM
//             coefficient fields whose 3x3 box sums (reflection-pad adjoint folded in as x2 multipliers) give the
//             gradient of every warped pixel two rows later, chains through the bilinear taps and the projection,
//             and applies the ADJOINT OF THE DISPARITY UP-SAMPLING in the same wave: horizontal part through a
//             128-float LDS row, vertical part in two accumulators.  Coarse-scale results go to a per-strip
//             staging block (strips overlap by one low-resolution texel); a small gather kernel adds the <= 4
//             overlapping blocks in a fixed order.  No full-resolution gradient is ever written.
//
// Numerics.  The sample coordinate is carried as pixel + delta:  with A = (K T)[:3,:3] inv_K[:3,:3] and D = A - I
// (evaluated in float64 per wave, so that the 1e-7-sized entries of D survive),
//     delta_x = (e_x + sd * (P03 - x m)) / (a_z + sd m),   e_x = D0.(x,y,1) - x D2.(x,y,1),  a_z = 1 + D2.(x,y,1),
//     sd = min_disp + (max_disp - min_disp) disp = 1 / depth,  m = P23 + 1e-7          (MD2/layers.py:16-25,163-198)
// which is the reference's projection divided through by depth.  floor() and the bilinear fraction are taken from
// delta alone, so the coordinate noise is ~1e-7 * |delta| instead of ~1e-7 * |x| of the reference's own fp32 chain:
// far fewer bilinear-cell flips against exact arithmetic than the fp32 reference itself has.  SSIM sums are taken
// of (value - 0.5), which removes most of the E[x^2] - mu^2 cancellation (sigma is shift invariant).
//
// Reference semantics: MD2/trainer.py:472-537,589-660; MD2/layers.py:16-25,139-198,223-253;
// DH/trainer.py:557-590,638-708.  (file:line relative to /root/reference/DepthNetworks/...)
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

using namespace dmh;

namespace {

// min waves per SIMD requested from the register allocator (tuning knobs)
#ifndef DMH_FWD_WAVES
#define DMH_FWD_WAVES 1
#endif
#ifndef DMH_BWD_WAVES
#define DMH_BWD_WAVES 2
#endif

constexpr int NT = 256;            // 4 independent waves per workgroup, one strip tile each; no barriers
constexpr int WPB = NT / WAVE;
constexpr int FW_OUT = WAVE - 2;   // forward strips: halo of 1 column each side
constexpr int BW_OUT = WAVE - 4;   // backward strips: halo of 2 columns each side
constexpr float SHIFT = 0.5f;      // SSIM statistics are taken of (value - SHIFT)
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
constexpr float K81C1 = 81.f * C1, K81C2 = 81.f * C2, K9S = 9.f * SHIFT;

// ---------------------------------------------------------------------------------------------- cross-lane
// lane_prev / lane_next (wave_shr:1 / wave_shl:1 DPP) and the buffer-resource loads make_rsrc / ldb: common.hpp
__device__ __forceinline__ float hsum3(float v) { return (lane_prev(v) + v) + lane_next(v); }

__device__ __forceinline__ float uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---------------------------------------------------------------------------------------------- projection
struct CamW {  // wave-uniform constants of one (image, frame): D = A - I, translation column, m = P23 + eps
    float d00, d01, d02, d10, d11, d12, d20, d21, d22, p03, p13, m;
};

__device__ __forceinline__ CamW load_cam_w(const float* __restrict__ K, const float* __restrict__ invK,
                                           const float* __restrict__ T, int b) {
    const float* k = K + (size_t)b * 16;
    const float* ik = invK + (size_t)b * 16;
    const float* t = T + (size_t)b * 16;
    double P[3][4], A[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += (double)k[i * 4 + q] * (double)t[q * 4 + j];   // (K @ T)[:3,:]  layers.py:188
            P[i][j] = acc;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 3; ++q) acc += P[i][q] * (double)ik[q * 4 + j];                 // inv_K[:3,:3]  layers.py:164
            A[i][j] = acc;
        }
    CamW c;
    c.d00 = uni((float)(A[0][0] - 1.0));
    c.d01 = uni((float)A[0][1]);
    c.d02 = uni((float)A[0][2]);
    c.d10 = uni((float)A[1][0]);
    c.d11 = uni((float)(A[1][1] - 1.0));
    c.d12 = uni((float)A[1][2]);
    c.d20 = uni((float)A[2][0]);
    c.d21 = uni((float)A[2][1]);
    c.d22 = uni((float)(A[2][2] - 1.0));
    c.p03 = uni((float)P[0][3]);
    c.p13 = uni((float)P[1][3]);
    c.m = uni((float)(P[2][3] + 1e-7));                                                       // layers.py:191 eps
    return c;
}

struct LaneProj { float fx, qx, cx0, cy0, nx; };    // column-dependent part (constant down the strip)
struct RowProj { float az, ex, ey, ny; };           // completed for one row

__device__ __forceinline__ LaneProj lane_proj(const CamW& c, int xr) {
    LaneProj p;
    p.fx = (float)xr;
    p.qx = fmaf(c.d20, p.fx, c.d22);
    p.cx0 = fmaf(c.d00, p.fx, c.d02);
    p.cy0 = fmaf(c.d10, p.fx, c.d12);
    p.nx = fmaf(-p.fx, c.m, c.p03);
    return p;
}
__device__ __forceinline__ RowProj row_proj(const CamW& c, const LaneProj& p, int yr) {
    RowProj r;
    const float fy = (float)yr;
    const float q = fmaf(c.d21, fy, p.qx);
    r.az = 1.f + q;
    r.ex = fmaf(-p.fx, q, fmaf(c.d01, fy, p.cx0));
    r.ey = fmaf(-fy, q, fmaf(c.d11, fy, p.cy0));
    r.ny = fmaf(-fy, c.m, c.p13);
    return r;
}

// sample coordinate p + d along one axis of n texels: border clamp of grid_sample(padding_mode="border",
// align_corners=True) and clip_coordinates_set_grad (gradient only strictly inside (0, n-1)).
__device__ __forceinline__ void split_coord(float d, int p, int n, int& p0, int& p1, float& t, bool& grad_ok) {
    d = fminf(fmaxf(d, -(float)(p + 2)), (float)(n - p + 1));   // keeps the int conversion in range; beyond: clamped anyway
    const float fl = floorf(d);
    p0 = p + (int)fl;
    t = d - fl;
    grad_ok = true;
    if (p0 < 0) {
        p0 = 0;
        t = 0.f;
        grad_ok = false;
    } else if (p0 >= n - 1) {
        p0 = n - 1;
        t = 0.f;
        grad_ok = false;
    }
    if (p0 == 0 && t == 0.f) grad_ok = false;
    p1 = min(p0 + 1, n - 1);
}

struct Tap {
    unsigned o00, o01, o10, o11;   // byte offsets inside one colour plane
    float tx, ty;
    bool gx_ok, gy_ok;
};

__device__ __forceinline__ Tap make_tap(float dx, float dy, int x, int y, int W, int H) {
    Tap t;
    int x0, x1, y0, y1;
    split_coord(dx, x, W, x0, x1, t.tx, t.gx_ok);
    split_coord(dy, y, H, y0, y1, t.ty, t.gy_ok);
    t.o00 = (unsigned)(y0 * W + x0) * 4u;
    t.o01 = (unsigned)(y0 * W + x1) * 4u;
    t.o10 = (unsigned)(y1 * W + x0) * 4u;
    t.o11 = (unsigned)(y1 * W + x1) * 4u;
    return t;
}

// ---------------------------------------------------------------------------------------------- disparity rows
// F.interpolate(disp, [H,W], mode="bilinear", align_corners=False) (MD2/trainer.py:481-482) along a strip: the
// x geometry is a per-lane constant.
struct DispGeo {
    int Hs, Ws, f;        // f = H / Hs (exact power of two, checked on the host); 1 = same size
    float rh;
    int x0, x1;           // per lane
    float lx;
};

__device__ __forceinline__ DispGeo disp_geo(int Hs, int Ws, int H, int W, int xr) {
    DispGeo g;
    g.Hs = Hs;
    g.Ws = Ws;
    g.f = H / Hs;
    g.rh = (float)Hs / (float)H;
    const float rw = (float)Ws / (float)W;
    const float sx = fmaxf(rw * ((float)xr + 0.5f) - 0.5f, 0.f);
    g.x0 = (int)sx;
    g.x1 = g.x0 + (g.x0 < Ws - 1 ? 1 : 0);
    g.lx = sx - (float)g.x0;
    if (g.f == 1) {
        g.x0 = g.x1 = xr;
        g.lx = 0.f;
    }
    return g;
}

struct RowLerp { int y0, y1; float ly; };
__device__ __forceinline__ RowLerp row_lerp(const DispGeo& g, int yr) {
    RowLerp r;
    const float sy = fmaxf(g.rh * ((float)yr + 0.5f) - 0.5f, 0.f);
    r.y0 = uni((int)sy);
    r.y1 = r.y0 + (r.y0 < g.Hs - 1 ? 1 : 0);
    r.ly = sy - (float)r.y0;
    return r;
}

// Raw disparity texels of one image row, fetched one row ahead of their use (the projection -> gather chain of the
// next row then starts from registers instead of from a memory round trip).
struct DispPre { float v00, v01, v10, v11; };
__device__ __forceinline__ DispPre disp_fetch(rsrc_t rd, const DispGeo& g, int yr) {
    DispPre p;
    if (g.f == 1) {
        p.v00 = ldb(rd, (unsigned)(yr * g.Ws + g.x0) * 4u, 0u);
        p.v01 = p.v10 = p.v11 = 0.f;
        return p;
    }
    const RowLerp r = row_lerp(g, yr);
    p.v00 = ldb(rd, (unsigned)(r.y0 * g.Ws + g.x0) * 4u, 0u);
    p.v01 = ldb(rd, (unsigned)(r.y0 * g.Ws + g.x1) * 4u, 0u);
    p.v10 = ldb(rd, (unsigned)(r.y1 * g.Ws + g.x0) * 4u, 0u);
    p.v11 = ldb(rd, (unsigned)(r.y1 * g.Ws + g.x1) * 4u, 0u);
    return p;
}
// same association as ATen's upsample_bilinear2d
__device__ __forceinline__ float disp_finish(const DispPre& p, const DispGeo& g, int yr) {
    if (g.f == 1) return p.v00;
    const RowLerp r = row_lerp(g, yr);
    const float hx = 1.f - g.lx;
    return (1.f - r.ly) * (hx * p.v00 + g.lx * p.v01) + r.ly * (hx * p.v10 + g.lx * p.v11);
}

// Forward kernel: the x-interpolated source row pair is cached until the pair changes (a uniform branch).
struct DispRow { float dA, dB; int y0; };
__device__ __forceinline__ float disp_value_cached(rsrc_t rd, const DispGeo& g, DispRow& c, int yr) {
    if (g.f == 1) return ldb(rd, (unsigned)(yr * g.Ws + g.x0) * 4u, 0u);
    const RowLerp r = row_lerp(g, yr);
    if (r.y0 != c.y0) {
        c.y0 = r.y0;
        const float v00 = ldb(rd, (unsigned)(r.y0 * g.Ws + g.x0) * 4u, 0u), v01 = ldb(rd, (unsigned)(r.y0 * g.Ws + g.x1) * 4u, 0u);
        const float v10 = ldb(rd, (unsigned)(r.y1 * g.Ws + g.x0) * 4u, 0u), v11 = ldb(rd, (unsigned)(r.y1 * g.Ws + g.x1) * 4u, 0u);
        const float hx = 1.f - g.lx;
        c.dA = hx * v00 + g.lx * v01;
        c.dB = hx * v10 + g.lx * v11;
    }
    return (1.f - r.ly) * c.dA + r.ly * c.dB;
}

// Branch-free: the two source rows are re-read every image row (they sit in L1; a cached row pair would cost a
// branch and register copies in the hot loop).  Same association as ATen's upsample_bilinear2d.
__device__ __forceinline__ float disp_value(rsrc_t rd, const DispGeo& g, int yr) {
    if (g.f == 1) return ldb(rd, (unsigned)(yr * g.Ws + g.x0) * 4u, 0u);
    const RowLerp r = row_lerp(g, yr);
    const float v00 = ldb(rd, (unsigned)(r.y0 * g.Ws + g.x0) * 4u, 0u), v01 = ldb(rd, (unsigned)(r.y0 * g.Ws + g.x1) * 4u, 0u);
    const float v10 = ldb(rd, (unsigned)(r.y1 * g.Ws + g.x0) * 4u, 0u), v11 = ldb(rd, (unsigned)(r.y1 * g.Ws + g.x1) * 4u, 0u);
    const float hx = 1.f - g.lx;
    return (1.f - r.ly) * (hx * v00 + g.lx * v01) + r.ly * (hx * v10 + g.lx * v11);
}

// ---------------------------------------------------------------------------------------------- SSIM
// Window statistics with every factor scaled by 81 (mu = S/9) and sums taken of (value - SHIFT):
//   n = (2 mu_x mu_y + C1)(2 sigma_xy + C2),  d = (mu_x^2 + mu_y^2 + C1)(sigma_x + sigma_y + C2)   MD2/layers.py:243-253
struct TgtWin { float sy, my, b1y, b2y; };   // target-side terms of one channel, shared by every stream

__device__ __forceinline__ TgtWin tgt_win(float Sy, float Syy) {
    TgtWin t;
    t.sy = Sy;
    t.my = K9S + Sy;
    t.b1y = fmaf(t.my, t.my, K81C1);
    t.b2y = fmaf(-Sy, Sy, fmaf(9.f, Syy, K81C2));
    return t;
}

struct SsimTerms { float A1, A2, B1, B2, mx; };
__device__ __forceinline__ SsimTerms ssim_terms(float Sx, float Sxx, float Sxy, const TgtWin& t) {
    SsimTerms s;
    s.mx = K9S + Sx;
    s.A1 = fmaf(s.mx + s.mx, t.my, K81C1);
    s.A2 = fmaf(2.f, fmaf(9.f, Sxy, -Sx * t.sy), K81C2);
    s.B1 = fmaf(s.mx, s.mx, t.b1y);
    s.B2 = fmaf(-Sx, Sx, fmaf(9.f, Sxx, t.b2y));
    return s;
}
__device__ __forceinline__ float ssim_val(float Sx, float Sxx, float Sxy, const TgtWin& t) {
    const SsimTerms s = ssim_terms(Sx, Sxx, Sxy, t);
    const float r = (s.A1 * s.A2) * __builtin_amdgcn_rcpf(s.B1 * s.B2);   // 1-ulp reciprocal: << the SSIM conditioning noise
    return fminf(fmaxf(fmaf(-0.5f, r, 0.5f), 0.f), 1.f);
}

// ---------------------------------------------------------------------------------------------- launch geometry
struct KArgs {
    dmh_photo_args a;
    uint8_t* sel;
    const uint8_t* csel;
    float* to_opt[DMH_MAX_SCALES];
    float* partials;
    const float* gvec;
    const float* fin;
    float* g_disp[DMH_MAX_SCALES];
    float* stage[DMH_MAX_SCALES];              // per-strip partial low-resolution gradients of the coarse scales
    float* pose_part;                          // POSE: [NS][strip][NF][12] partial sums of d loss / d (K T)[:3,:]
    int sy_slots[DMH_MAX_SCALES], sx_slots[DMH_MAX_SCALES];
    int tiles_x, tiles_y, ntiles, R;
    float min_disp, dmul;                      // scaled_disp = min_disp + dmul * disp   MD2/layers.py:21-23
};

// Rows per strip: tall strips amortise the halo rows, but the launch should still be several waves per SIMD deep.
__host__ inline int pick_rows(int B, int H, int W, int out_cols) {
    const int tx = (W + out_cols - 1) / out_cols;
    const int64_t want = out_cols == FW_OUT ? 8192 : 4096;   // forward strips are 1 wave each, backward strips 1 per scale
    if (const char* e = getenv("DMH_K1_ROWS")) {   // tuning knob (tools/kbench.py)
        const int v = atoi(e);
        if (v >= 2 && v <= 1024) return v;
    }
    if (out_cols == FW_OUT) {
        int R = 32;
        while (R > 8 && (int64_t)B * tx * ((H + R - 1) / R) < want) R >>= 1;
        return R;
    }
    // backward: the row loop runs in threes, so strip heights are multiples of 3 (33 / 18 / 9)
    int R = 33;
    while (R > 9 && (int64_t)B * tx * ((H + R - 1) / R) < want) R = R > 18 ? 18 : 9;
    return R;
}

// XCD-aware order: consecutive workgroup ids round-robin over the 8 XCDs, so hand each XCD a contiguous run of
// strips (horizontally adjacent strips share their halo columns and gather lines in that XCD's L2).
__device__ __forceinline__ int wave_item() {
    int bid = blockIdx.x;
    const int nb = gridDim.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
    return uni(bid * WPB + (int)(threadIdx.x >> 6));
}

template <int NSTR>
struct FRow {   // row sums of one image row: target (y, y^2) and per stream (x, x^2, x y), three channels each
    float ty[3], tyy[3], sx[NSTR][3], sxx[NSTR][3], sxy[NSTR][3];
};

// ------------------------------------------------------------------------------------------------ forward
// NF source frames, SPP + SPL scales per pass.  The rolling row sums of the identity term, of the (optional) hint view and
// of the first SPP scales live in registers (NF * (1 + SPP) streams); those of the next SPL scales live in LDS, private to
// the wave (2 records x 3 x 16 B per lane and stream, lane-linear 16-byte words: conflict-free ds_read/write_b128, no
// barrier).  With one source frame that allows ONE pass of four scales (target sums, identity term and tie-break noise
// formed once instead of twice) -- measured slower than two passes of two scales, see the launch site; kept selectable.
// HINT: DepthHints' extra candidate (DH/trainer.py:510-525,629-636,700-725): the source view warped with the depth
// HINT is one more stream (formed per pass, like the identity term), the per-pixel argmin runs over [reprojection,
// identity, hint], and where the hint wins the proxy log-L1 term is accumulated.
// FL >= 0: the option flags as compile-time constants (bit 0 automask, bit 1 no_ssim, bit 2 MD2 variant, bits 3-4 noise
// mode): the hot configurations get straight-line code without the flag branches and their merge copies; FL = -1 reads the
// flags at run time (every other combination).
template <int NF, int SPP, int SPL, bool HINT, int MINW = DMH_FWD_WAVES, int FL = -1>
__global__ __launch_bounds__(NT, MINW) void photo_fwd_kernel(const KArgs k) {
    constexpr int NSTR = NF * (1 + SPP) + (HINT ? 1 : 0);      // streams with register-resident row sums
    constexpr int ST_HINT = NF * (1 + SPP);
    constexpr int SPT = SPP + SPL;                             // scales per pass
    constexpr int NVAL = NSTR + NF * SPL;                      // candidates per pixel: val[] slots
    constexpr int LREC = 3 * WAVE;                             // float4 words of one LDS record (one stream, one row)
    __shared__ float4 s_rec[SPL > 0 ? WPB * SPL * NF * 2 * LREC : 1];
    const dmh_photo_args& a = k.a;
    const int lane = threadIdx.x & (WAVE - 1);
    float4* const lrec = s_rec + (SPL > 0 ? (threadIdx.x >> 6) * (SPL * NF * 2 * LREC) : 0) + lane;
    const int tile = wave_item();
    if (tile >= k.ntiles) return;
    const int H = a.H, W = a.W, NS = a.num_scales;
    const int per_img = k.tiles_x * k.tiles_y;
    const int b = tile / per_img, tt = tile - b * per_img, ty = tt / k.tiles_x, tx = tt - ty * k.tiles_x;
    const int X0 = tx * FW_OUT, Y0 = ty * k.R;
    const int col = X0 - 1 + lane;
    const int xr = reflect_idx(col, W);
    const bool out_lane = lane >= 1 && lane <= FW_OUT && col < W;
    const int nrows = min(k.R, H - Y0) + 2;

    const unsigned plane = (unsigned)(H * W) * 4u;
    const size_t img_off = (size_t)b * 3 * H * W;
    const rsrc_t rt = make_rsrc(a.target + img_off, 3u * plane);
    rsrc_t rsrc[NF];
    CamW cam[NF];
    LaneProj lp[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        rsrc[f] = make_rsrc(a.source[f] + img_off, 3u * plane);
        cam[f] = load_cam_w(a.K, a.inv_K, a.T[f], b);
        lp[f] = lane_proj(cam[f], xr);
    }
    const bool automask = FL >= 0 ? (FL & 1) != 0 : a.automask != 0;
    const bool no_ssim = FL >= 0 ? (FL & 2) != 0 : a.no_ssim != 0;
    const bool md2 = FL >= 0 ? (FL & 4) != 0 : a.variant == DMH_VARIANT_MD2;
    const int noise_mode = FL >= 0 ? ((FL >> 3) & 3) : a.noise_mode;
    const int nf_noise = md2 ? NF : 1;
    const int npass = (NS + SPT - 1) / SPT;
    const Philox<7> rng(a.seed);
    const rsrc_t rhint = make_rsrc(HINT ? a.depth_hint + (size_t)b * H * W : a.target, (unsigned)(H * W) * 4u);
    const rsrc_t rhmask = make_rsrc(HINT ? a.depth_hint_mask + (size_t)b * H * W : a.target, (unsigned)(H * W) * 4u);
    // grid_sample(align_corners=False) of the hint warp (DH/trainer.py:523-525 omits align_corners): the sample
    // coordinate is px * W/(W-1) - 0.5, i.e. delta' = delta * W/(W-1) + x/(W-1) - 0.5
    const float hsx = (float)W / (float)(W - 1), hsy = (float)H / (float)(H - 1);
    const float hox = (float)xr / (float)(W - 1) - 0.5f;

    for (int s0 = 0, pass = 0; s0 < NS; s0 += SPT, ++pass) {
        DispGeo dg[SPP];
        DispRow dr[SPP];
        rsrc_t rd[SPT];
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int s = min(s0 + j, NS - 1);
            if (j < SPP) {
                dg[j] = disp_geo(a.Hs[s], a.Ws[s], H, W, xr);
                dr[j].y0 = -1;
                dr[j].dA = dr[j].dB = 0.f;
            }
            rd[j] = make_rsrc(a.disp[s] + (size_t)b * a.Hs[s] * a.Ws[s], (unsigned)(a.Hs[s] * a.Ws[s]) * 4u);
        }
        FRow<NSTR> rowA, rowB;
        float l1p[NSTR];       // L1 of the previous row (the centre row of the next window)
        float acc1[SPT], acc2[SPT], acc3[SPT], acc4[SPT];
        float sd_prev[SPT];    // scaled disparity of the previous row (depth of the centre row, for the hint term)
#pragma unroll
        for (int j = 0; j < SPT; ++j) acc1[j] = acc2[j] = acc3[j] = acc4[j] = sd_prev[j] = 0.f;
#pragma unroll
        for (int i = 0; i < NSTR; ++i) l1p[i] = 0.f;

        // One image row: `older` / `newer` are the row sums of the two rows above; the finished centre row is
        // the one in between; the current row's sums replace `older`.  `par` = kk & 1 as a literal: the LDS-resident
        // streams keep `older` in record par and `newer` in record par ^ 1.
        auto step = [&](const int kk, FRow<NSTR>& older, const FRow<NSTR>& newer, const int par) __attribute__((always_inline)) {
            const int r = Y0 - 1 + kk;
            const int yr = reflect_idx(r, H);
            const unsigned rowoff = (unsigned)(yr * W + xr) * 4u;
            const bool emit = kk >= 2;
            const bool olane = out_lane && kk < nrows;   // (a row past the strip's end, see the loop below, emits nothing)
            const int qy = r - 1;
            float tv[3];
            TgtWin tw[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                tv[c] = ldb(rt, rowoff, plane * (unsigned)c) - SHIFT;
                const float hy = hsum3(tv[c]), hyy = hsum3(tv[c] * tv[c]);
                tw[c] = tgt_win(older.ty[c] + newer.ty[c] + hy, older.tyy[c] + newer.tyy[c] + hyy);
                older.ty[c] = hy;
                older.tyy[c] = hyy;
            }
            float val[NVAL];
            // accumulate one stream: xv = this row's (shifted) values of the stream
            auto stream = [&](const int st, const float (&xv)[3]) __attribute__((always_inline)) {
                float ss = 0.f, l1 = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float hx = hsum3(xv[c]), hxx = hsum3(xv[c] * xv[c]), hxy = hsum3(xv[c] * tv[c]);
                    if (emit && !no_ssim)
                        ss += ssim_val(older.sx[st][c] + newer.sx[st][c] + hx, older.sxx[st][c] + newer.sxx[st][c] + hxx,
                                       older.sxy[st][c] + newer.sxy[st][c] + hxy, tw[c]);
                    older.sx[st][c] = hx;
                    older.sxx[st][c] = hxx;
                    older.sxy[st][c] = hxy;
                    l1 += fabsf(tv[c] - xv[c]);
                }
                // compute_reprojection_loss (MD2/trainer.py:525-537) of the centre row: its L1 was formed one row ago
                val[st] = no_ssim ? l1p[st] * (1.f / 3.f) : (0.85f / 3.f) * ss + (0.15f / 3.f) * l1p[st];
                l1p[st] = l1;
            };
            // the same for a stream whose two row-sum records live in LDS (slot ls), candidate slot vi: one 16-byte word per
            // channel (sx, sxx, sxy, l1 of the row) read just before use; same association of the sums as above
            auto stream_lds = [&](const int ls, const int vi, const float (&xv)[3]) __attribute__((always_inline)) {
                float4* const ro = lrec + (ls * 2 + par) * LREC;
                const float4* const rn = lrec + (ls * 2 + (par ^ 1)) * LREC;
                float ss = 0.f, l1 = 0.f, l1prev = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float4 o = ro[c * WAVE], n = rn[c * WAVE];
                    const float hx = hsum3(xv[c]), hxx = hsum3(xv[c] * xv[c]), hxy = hsum3(xv[c] * tv[c]);
                    if (emit && !no_ssim) ss += ssim_val(o.x + n.x + hx, o.y + n.y + hxx, o.z + n.z + hxy, tw[c]);
                    const float l1c = fabsf(tv[c] - xv[c]);
                    l1 += l1c;
                    l1prev += n.w;
                    ro[c * WAVE] = make_float4(hx, hxx, hxy, l1c);
                }
                val[vi] = no_ssim ? l1prev * (1.f / 3.f) : (0.85f / 3.f) * ss + (0.15f / 3.f) * l1prev;
            };
            if (automask) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    float xv[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) xv[c] = ldb(rsrc[f], rowoff, plane * (unsigned)c) - SHIFT;
                    stream(f, xv);
                }
            }
            RowProj rp[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) rp[f] = row_proj(cam[f], lp[f], yr);
            if constexpr (HINT) {
                // the hint view: same projection with depth = hint; a missing hint (depth 0) back-projects to the
                // origin, i.e. the pixel (P03, P13) / (P23 + 1e-7), as in the reference
                const float dh = ldb(rhint, (unsigned)(yr * W + xr) * 4u, 0u);
                const bool have = dh > 0.f;
                const float sdh = have ? fast_rcp(dh) : 0.f;
                const float rden = fast_rcp(fmaf(sdh, cam[0].m, rp[0].az));
                float dx = fmaf(sdh, lp[0].nx, rp[0].ex) * rden, dy = fmaf(sdh, rp[0].ny, rp[0].ey) * rden;
                if (!have) {
                    const float rm = fast_rcp(cam[0].m);
                    dx = cam[0].p03 * rm - (float)xr;
                    dy = cam[0].p13 * rm - (float)yr;
                }
                dx = fmaf(dx, hsx, hox);
                dy = fmaf(dy, hsy, (float)yr / (float)(H - 1) - 0.5f);
                const Tap t = make_tap(dx, dy, xr, yr, W, H);
                const float gx = 1.f - t.tx, gy = 1.f - t.ty;
                const float w00 = gx * gy, w01 = t.tx * gy, w10 = gx * t.ty, w11 = t.tx * t.ty;
                float xv[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const unsigned po = plane * (unsigned)c;
                    xv[c] = (ldb(rsrc[0], t.o00, po) * w00 + ldb(rsrc[0], t.o01, po) * w01 + ldb(rsrc[0], t.o10, po) * w10 +
                             ldb(rsrc[0], t.o11, po) * w11) - SHIFT;
                }
                stream(ST_HINT, xv);
            }
            float sd_centre[SPT];
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                if constexpr (HINT) sd_centre[j] = sd_prev[j];
                if (s0 + j >= NS) break;
                float d;
                if (j < SPP) {
                    d = disp_value_cached(rd[j], dg[j], dr[j], yr);
                } else {    // LDS-resident scale: its up-sampling geometry is re-formed per row instead of held in registers
                    const int s = s0 + j;
                    d = disp_value(rd[j], disp_geo(a.Hs[s], a.Ws[s], H, W, xr), yr);
                }
                const float sd = fmaf(k.dmul, d, k.min_disp);
                if constexpr (HINT) sd_prev[j] = sd;
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const float rden = fast_rcp(fmaf(sd, cam[f].m, rp[f].az));
                    const float dx = fmaf(sd, lp[f].nx, rp[f].ex) * rden, dy = fmaf(sd, rp[f].ny, rp[f].ey) * rden;
                    const Tap t = make_tap(dx, dy, xr, yr, W, H);
                    const float gx = 1.f - t.tx, gy = 1.f - t.ty;
                    const float w00 = gx * gy, w01 = t.tx * gy, w10 = gx * t.ty, w11 = t.tx * t.ty;
                    float xv[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const unsigned po = plane * (unsigned)c;
                        xv[c] = (ldb(rsrc[f], t.o00, po) * w00 + ldb(rsrc[f], t.o01, po) * w01 + ldb(rsrc[f], t.o10, po) * w10 +
                                 ldb(rsrc[f], t.o11, po) * w11) - SHIFT;
                    }
                    if (j < SPP) stream(NF + j * NF + f, xv);
                    else stream_lds((j - SPP) * NF + f, NSTR + (j - SPP) * NF + f, xv);
                }
            }
            if (!emit) return;
            // ---- per-pixel min / argmin for the centre row qy (MD2/trainer.py:640-660, DH/trainer.py:671-708)
            const size_t pix = ((size_t)b * H + qy) * W + col;
            float nz[4] = {0.f, 0.f, 0.f, 0.f};   // tie-break noise: slot j * NF + f of this pass
            if (automask && noise_mode == DMH_NOISE_PHILOX) {
                const uint4 rr = rng(a.offset + (uint64_t)pix * (uint64_t)npass + (uint64_t)pass, 0x646d68ull);
                const float2 n01 = normal_pair_from_bits(rr.x, rr.y), n23 = normal_pair_from_bits(rr.z, rr.w);
                nz[0] = n01.x * 0.00001f;
                nz[1] = n01.y * 0.00001f;
                nz[2] = n23.x * 0.00001f;
                nz[3] = n23.y * 0.00001f;
            }
            float hint_cand = 3.0e38f, hint_d = 0.f, hint_valid = 0.f;
            if constexpr (HINT) {
                const unsigned ho = (unsigned)(min(qy, H - 1) * W + min(max(col, 0), W - 1)) * 4u;
                hint_d = ldb(rhint, ho, 0u);
                hint_valid = ldb(rhmask, ho, 0u);
                hint_cand = val[ST_HINT] + 1000.f * (1.f - hint_valid);          // DH/trainer.py:632-634
            }
            unsigned bits = 0u;
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                if (s0 + j >= NS) break;
                const int s = s0 + j;
                float best = 3.0e38f;
                int bestf = 0;
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const float v = val[j < SPP ? NF + j * NF + f : NSTR + (j - SPP) * NF + f];
                    if (v < best) {
                        best = v;
                        bestf = f;
                    }
                }
                bool chosen = true;
                float v = best;
                if (automask) {
                    float idn = 3.0e38f;
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        float z = 0.f;
                        const int fi = md2 ? f : 0;
                        if (noise_mode == DMH_NOISE_TENSOR) {
                            if (olane) z = a.noise[s][(((size_t)b * nf_noise + fi) * H + qy) * W + col];
                        } else if (noise_mode == DMH_NOISE_PHILOX) {
                            z = nz[(j * NF + fi) & 3];
                        }
                        // MD2: noise per identity channel, then min over channels (trainer.py:642-654);
                        // DH : min over frames first, one noise plane (DH/trainer.py:671,687-690)
                        idn = fminf(idn, val[f] + (md2 ? z : 0.f));
                        if (!md2 && f == NF - 1) idn += z;
                    }
                    if (md2) {
                        chosen = best < idn;   // torch.min keeps the first (identity) on ties
                        v = chosen ? best : idn;
                    } else {
                        // argmin over [reprojection, identity, hint]: first wins ties; the reprojection mask is
                        // (idx != 1), so it also holds where the hint wins (DH/trainer.py:578-584)
                        const bool ident_wins = idn < best && idn <= hint_cand;
                        chosen = !ident_wins;
                        v = chosen ? best : 0.f;
                    }
                }
                bool hint_wins = false;
                float hl = 0.f;
                if constexpr (HINT) {
                    hint_wins = chosen && hint_cand < best;        // idx == 2
                    // proxy supervision log(|hint - depth_s| + 1) * valid where the hint wins (DH/trainer.py:541-555)
                    const float pred = fast_rcp(sd_centre[j]);
                    hl = hint_wins ? __logf(fabsf(hint_d - pred) + 1.f) * hint_valid : 0.f;
                }
                if (olane) {
                    if (k.to_opt[s]) k.to_opt[s][pix] = v;
                    acc1[j] += v;
                    acc2[j] += chosen ? 1.f : 0.f;
                    if constexpr (HINT) {
                        acc3[j] += hl;
                        acc4[j] += hint_wins ? 1.f : 0.f;
                    }
                    bits |= (hint_wins ? 3u : (chosen ? (unsigned)(1 + bestf) : 0u)) << (2 * s);
                }
            }
            if (olane) {
                if (pass > 0) bits |= k.sel[pix];
                k.sel[pix] = (uint8_t)bits;
            }
        };
        // rows in pairs, unconditionally: after two steps the two records are back in their roles, so the loop carries its
        // state in fixed registers (a conditional second step cost ~60 register copies per iteration at the merge); with an
        // odd row count the last step runs on a row past the strip's end and emits nothing
        for (int kk = 0; kk < nrows; kk += 2) {
            step(kk, rowA, rowB, 0);
            step(kk + 1, rowB, rowA, 1);
        }
        // fixed-order wave sums -> one partial pair per (scale, strip)
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            if (s0 + j >= NS) break;
            const float t1 = wave_sum(acc1[j]), t2 = wave_sum(acc2[j]);
            const float t3 = HINT ? wave_sum(acc3[j]) : 0.f, t4 = HINT ? wave_sum(acc4[j]) : 0.f;
            if (lane == 0)
                reinterpret_cast<float4*>(k.partials)[(size_t)(s0 + j) * k.ntiles + tile] = make_float4(t1, t2, t3, t4);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
struct BRow {   // record of one processed row: its row sums, the coefficient row sums of the row above it, and what
                // the chain two rows later needs of it
    float hy[3], hyy[3], hx[3], hxx[3], hxy[3];
    float ch[3][3];          // [channel][a0, ax, ay] horizontal sums (with the reflection fold)
    float xv[3], yv[3], J[3];
    float sdv;               // scaled disparity of this row (depth = 1/sdv, for the depth-hint proxy term)
    unsigned sel;            // selection field of this row for the wave's scale
    // operands of THIS row, requested while the previous row was processed: the warp is evaluated one row ahead, so the
    // twelve gathered source texels are in flight for a whole row of arithmetic before their first use
    float g[3][4];           // source texels at the four bilinear taps, per channel
    float tx, ty, jx, jy;    // bilinear fractions; d ix / d disp, d iy / d disp (0 where the coordinate is clamped)
    float ptv[3];
    unsigned psel;           // selection byte of the row above (the coefficient row)
    float pqa, pqb, ppx, ppy;   // POSE, requested a row ahead: 1/den gated by "x / y not clamped", the sample coordinate
    // POSE, kept for two rows: d warped_c / d ix, / d iy; d ix / d disp, d iy / d disp; the four factors above
    float dvx[3], dvy[3], sjx, sjy, qa, qb, qx, qy;
};

// SAME: the wave's scale has the image resolution (gradient written directly); otherwise the up-sampling adjoint runs.
// POSE: also d loss / d P_f with P_f = (K T_f)[:3,:] (MD2/layers.py:188-191: cam = P [X;1], pix = cam[:2] / (cam[2] + eps)).
// With G = d loss / d pix of a pixel, den = a_z + sd m (so cam[2] + eps = depth den), p = (x, y, 1), X = depth inv_K p:
//     a = G_x / den, b = G_y / den, c = -(a pix_x + b pix_y):   d loss / d P[i][j<3] = sum_k invK[j][k] sum_pix (a,b,c)_i p_k,
//     d loss / d P[i][3] = sum_pix (a,b,c)_i sd   -> twelve sums per (strip, frame), finished on the host side of the C ABI's caller.
// PLAIN: the training configuration (SSIM on, no depth hints) with those two flags as constants; otherwise read at run time.
template <int NF, bool SAME, bool POSE, bool PLAIN>
__device__ __forceinline__ void photo_bwd_strip(const KArgs& k, const int s, const int tile, float* sA, float* sB) {
    const dmh_photo_args& a = k.a;
    const int lane = threadIdx.x & (WAVE - 1);
    const int H = a.H, W = a.W, B = a.B;
    const int per_img = k.tiles_x * k.tiles_y;
    const int b = tile / per_img, tt = tile - b * per_img, ty = tt / k.tiles_x, tx = tt - ty * k.tiles_x;
    const int X0 = tx * BW_OUT, Y0 = ty * k.R;
    const int col = X0 - 2 + lane;
    const int xr = reflect_idx(col, W);
    const bool col_in = col >= 0 && col < W;
    const bool out_lane = lane >= 2 && lane < 2 + BW_OUT && col < W;
    const int nrows = min(k.R, H - Y0) + 4;

    const unsigned plane = (unsigned)(H * W) * 4u;
    const size_t img_off = (size_t)b * 3 * H * W;
    const rsrc_t rt = make_rsrc(a.target + img_off, 3u * plane);
    const rsrc_t rsel = make_rsrc(k.csel + (size_t)b * H * W, (unsigned)(H * W));
    const bool no_ssim = PLAIN ? false : a.no_ssim != 0;
    float up = k.gvec[DMH_FIN_LOSS] / (float)a.num_scales + k.gvec[DMH_FIN_LOSS_S + s] + k.gvec[DMH_FIN_REPROJ_S + s];
    up *= (a.variant == DMH_VARIANT_MD2) ? 1.0f / ((float)B * (float)H * (float)W) : 1.0f / (k.fin[DMH_FIN_COUNT_S + s] + 1e-7f);
    up = uni(up);
    const float l1w = up * (no_ssim ? (1.f / 3.f) : (0.15f / 3.f));
    const float gs0 = no_ssim ? 0.f : up * (0.85f / 3.f);
    const float mxl = (col == 1) ? 2.f : 1.f, mxr = (col == W - 2) ? 2.f : 1.f;   // reflection-pad adjoint

    const int Hs = a.Hs[s], Ws = a.Ws[s];
    const DispGeo dg = disp_geo(Hs, Ws, H, W, xr);
    const rsrc_t rd = make_rsrc(a.disp[s] + (size_t)b * Hs * Ws, (unsigned)(Hs * Ws) * 4u);
    // this strip's block of low-resolution texels (coarse scales)
    const int jlo = row_lerp(dg, Y0).y0;
    const float rw = (float)Ws / (float)W;
    const int ilo = (int)fmaxf(rw * ((float)X0 + 0.5f) - 0.5f, 0.f);
    const int SX = k.sx_slots[s], SY = k.sy_slots[s];
    float* stage = SAME ? nullptr : k.stage[s] + (size_t)tile * SX * SY;
    const unsigned sel_shift = 2u * (unsigned)s;
    // depth hints: where the hint won (selection code 3) the reprojection term of frame 0 applies too, plus the proxy
    // term log(|hint - depth| + 1) * valid / (count + 1e-7)                       (DH/trainer.py:541-555,713-725)
    const bool hints = PLAIN ? false : a.depth_hint != nullptr;
    const rsrc_t rhint = make_rsrc(hints ? a.depth_hint + (size_t)b * H * W : a.target, (unsigned)(H * W) * 4u);
    const rsrc_t rhmask = make_rsrc(hints ? a.depth_hint_mask + (size_t)b * H * W : a.target, (unsigned)(H * W) * 4u);
    float up_h = 0.f;
    if (hints)
        up_h = uni((k.gvec[DMH_FIN_LOSS] / (float)a.num_scales + k.gvec[DMH_FIN_LOSS_S + s] + k.gvec[DMH_FIN_HINT_S + s]) /
                   (k.fin[DMH_FIN_HINTCOUNT_S + s] + 1e-7f));

    for (int f = 0; f < NF; ++f) {
        const rsrc_t rs = make_rsrc(a.source[f] + img_off, 3u * plane);
        const CamW cam = load_cam_w(a.K, a.inv_K, a.T[f], b);
        const LaneProj lp = lane_proj(cam, xr);
        const unsigned fsel = (unsigned)(1 + f);
        BRow recA, recB, recC;
        float accA = 0.f, accB = 0.f;     // vertical up-sampling adjoint: low rows ja and ja + 1
        int ja = jlo;
        float pacc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) pacc[i] = 0.f;

        auto flush = [&](const int j, const float v) __attribute__((always_inline)) {
            const int slot = j - jlo;
            if (slot < SY && lane < SX) {
                float* p = stage + (size_t)slot * SX + lane;
                *p = (f > 0) ? *p + v : v;
            }
        };

        // One image row r = Y0 - 2 + kk.  `cur` receives this row's record; `newer` / `older` are the records of the
        // rows one / two above.  STAGE 0: rows only; 1: + coefficient fields of row r-1; 2: + gradient of row r-2.
        // issue(kk, rec): everything row kk needs from memory.  Its disparity texels were requested one row earlier still
        // (`pdn`), so disparity -> projection -> taps runs on registers and ends in the twelve gathers, which then stay in
        // flight while the PREVIOUS row is processed (round 2 issued them inside the row that consumed them: 28 % of the
        // wave cycles waited on memory).
        DispPre pdn = disp_fetch(rd, dg, reflect_idx(Y0 - 2, H));
        auto issue = [&](const int kk, BRow& rec) __attribute__((always_inline)) {
            const int r = Y0 - 2 + kk;
            const int yr = reflect_idx(r, H);
            const float d = disp_finish(pdn, dg, yr);
            pdn = disp_fetch(rd, dg, reflect_idx(r + 1, H));
            const unsigned rowoff = (unsigned)(yr * W + xr) * 4u;
#pragma unroll
            for (int c = 0; c < 3; ++c) rec.ptv[c] = ldb(rt, rowoff, plane * (unsigned)c);
            const int rc = r - 1;
            const bool p_in = col_in && rc >= 0 && rc < H;
            rec.psel = p_in ? (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rsel, (unsigned)(rc * W + col), 0, 0) : 0u;
            const float sd = fmaf(k.dmul, d, k.min_disp);
            rec.sdv = sd;
            const RowProj rp = row_proj(cam, lp, yr);
            const float rden = fast_rcp(fmaf(sd, cam.m, rp.az));
            const float dx = fmaf(sd, lp.nx, rp.ex) * rden, dy = fmaf(sd, rp.ny, rp.ey) * rden;
            const Tap t = make_tap(dx, dy, xr, yr, W, H);
            const float rd2 = rden * rden * k.dmul;
            rec.jx = t.gx_ok ? fmaf(lp.nx, rp.az, -rp.ex * cam.m) * rd2 : 0.f;   // d ix / d disp
            rec.jy = t.gy_ok ? fmaf(rp.ny, rp.az, -rp.ey * cam.m) * rd2 : 0.f;   // d iy / d disp
            rec.tx = t.tx;
            rec.ty = t.ty;
            if constexpr (POSE) {
                rec.pqa = t.gx_ok ? rden : 0.f;
                rec.pqb = t.gy_ok ? rden : 0.f;
                rec.ppx = (float)xr + dx;
                rec.ppy = (float)yr + dy;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned po = plane * (unsigned)c;
                rec.g[c][0] = ldb(rs, t.o00, po);
                rec.g[c][1] = ldb(rs, t.o01, po);
                rec.g[c][2] = ldb(rs, t.o10, po);
                rec.g[c][3] = ldb(rs, t.o11, po);
            }
        };
        auto step = [&](const int kk, BRow& older, const BRow& newer, BRow& cur, auto stage_tag) __attribute__((always_inline)) {
            constexpr int STAGE = decltype(stage_tag)::value;
            const int r = Y0 - 2 + kk;
            const bool olane = out_lane && kk < nrows;   // (rows past the strip's end, see the loop below, write nothing)
            // (1) this row: warped values and chain factors J_c = d warped_c / d disp from the operands requested a row ago
            const float sd_rq = older.sdv;  // of the row two above (read before `older` becomes the next row's record)
            if constexpr (POSE) {           // this row's request fields -> fields that live for two more rows
                cur.sjx = cur.jx; cur.sjy = cur.jy;
                cur.qa = cur.pqa; cur.qb = cur.pqb; cur.qx = cur.ppx; cur.qy = cur.ppy;
            }
            issue(kk + 1, older);           // only the request fields of `older` are written; its sums are still read below
            const float gx = 1.f - cur.tx, gy = 1.f - cur.ty;
            const float w00 = gx * gy, w01 = cur.tx * gy, w10 = gx * cur.ty, w11 = cur.tx * cur.ty;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v00 = cur.g[c][0], v01 = cur.g[c][1], v10 = cur.g[c][2], v11 = cur.g[c][3];
                cur.xv[c] = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11) - SHIFT;
                cur.yv[c] = cur.ptv[c] - SHIFT;
                const float dvx = (v01 - v00) * gy + (v11 - v10) * cur.ty, dvy = (v10 - v00) * gx + (v11 - v01) * cur.tx;
                cur.J[c] = dvx * cur.jx + dvy * cur.jy;
                if constexpr (POSE) {
                    cur.dvx[c] = dvx;
                    cur.dvy[c] = dvy;
                }
                // (2) row sums
                cur.hx[c] = hsum3(cur.xv[c]);
                cur.hxx[c] = hsum3(cur.xv[c] * cur.xv[c]);
                cur.hxy[c] = hsum3(cur.xv[c] * cur.yv[c]);
                cur.hy[c] = hsum3(cur.yv[c]);
                cur.hyy[c] = hsum3(cur.yv[c] * cur.yv[c]);
            }
            // (3) coefficient fields of the row above (centre row rc): d v(p) / d x_q = a0 + ay*y_q + ax*x_q for every
            //     pixel q of p's window, v = clamp((1 - n/d)/2)
            cur.sel = 0u;
            if constexpr (STAGE >= 1) {
                const unsigned selb = (cur.psel >> sel_shift) & 3u;     // 0 outside the image
                cur.sel = selb;
                const float gate = (selb == fsel || (hints && selb == 3u)) ? gs0 : 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float Sx = older.hx[c] + newer.hx[c] + cur.hx[c], Sxx = older.hxx[c] + newer.hxx[c] + cur.hxx[c];
                    const float Sxy = older.hxy[c] + newer.hxy[c] + cur.hxy[c];
                    const TgtWin tw = tgt_win(older.hy[c] + newer.hy[c] + cur.hy[c], older.hyy[c] + newer.hyy[c] + cur.hyy[c]);
                    const SsimTerms q = ssim_terms(Sx, Sxx, Sxy, tw);
                    const float invd = fast_rcp(q.B1 * q.B2);
                    const float rr = (q.A1 * q.A2) * invd;
                    const float v = fmaf(-0.5f, rr, 0.5f);
                    const float gsd = (v >= 0.f && v <= 1.f) ? gate * invd : 0.f;   // clamp passes the gradient on the closed interval
                    const float a0 = -gsd * (fmaf(tw.my, q.A2, -q.A1 * tw.sy) - rr * fmaf(q.mx, q.B2, -q.B1 * Sx));
                    const float cay = -9.f * gsd * q.A1;
                    const float cax = 9.f * gsd * rr * q.B1;
                    cur.ch[c][0] = fmaf(mxl, lane_prev(a0), a0) + mxr * lane_next(a0);
                    cur.ch[c][1] = fmaf(mxl, lane_prev(cax), cax) + mxr * lane_next(cax);
                    cur.ch[c][2] = fmaf(mxl, lane_prev(cay), cay) + mxr * lane_next(cay);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) cur.ch[c][0] = cur.ch[c][1] = cur.ch[c][2] = 0.f;
            }
            // (4) gradient of the row two above (rq): 3x3 box sums of the coefficient fields, then the chain
            if constexpr (STAGE >= 2) {
                const int rq = r - 2;
                const float myt = (rq == 1) ? 2.f : 1.f, myb = (rq == H - 2) ? 2.f : 1.f;
                const float l1g = (newer.sel == fsel || (hints && newer.sel == 3u)) ? l1w : 0.f;
                float g = 0.f, Gx = 0.f, Gy = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float S0 = fmaf(myt, older.ch[c][0], newer.ch[c][0]) + myb * cur.ch[c][0];
                    const float Sxc = fmaf(myt, older.ch[c][1], newer.ch[c][1]) + myb * cur.ch[c][1];
                    const float Syc = fmaf(myt, older.ch[c][2], newer.ch[c][2]) + myb * cur.ch[c][2];
                    const float df = older.xv[c] - older.yv[c];
                    float gw = fmaf(older.xv[c], Sxc, fmaf(older.yv[c], Syc, S0));
                    gw = fmaf(l1g, df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f), gw);
                    g = fmaf(gw, older.J[c], g);
                    if constexpr (POSE) {
                        Gx = fmaf(gw, older.dvx[c], Gx);
                        Gy = fmaf(gw, older.dvy[c], Gy);
                    }
                }
                if constexpr (POSE) {
                    if (olane) {
                        const float pa = Gx * older.qa, pb = Gy * older.qb, pc = -(pa * older.qx + pb * older.qy);
                        const float fx = (float)col, fy = (float)rq;
                        pacc[0] = fmaf(pa, fx, pacc[0]); pacc[1] = fmaf(pa, fy, pacc[1]); pacc[2] += pa; pacc[3] = fmaf(pa, sd_rq, pacc[3]);
                        pacc[4] = fmaf(pb, fx, pacc[4]); pacc[5] = fmaf(pb, fy, pacc[5]); pacc[6] += pb; pacc[7] = fmaf(pb, sd_rq, pacc[7]);
                        pacc[8] = fmaf(pc, fx, pacc[8]); pacc[9] = fmaf(pc, fy, pacc[9]); pacc[10] += pc; pacc[11] = fmaf(pc, sd_rq, pacc[11]);
                    }
                }
                if (hints && newer.sel == 3u && olane) {
                    const unsigned ho = (unsigned)(rq * W + col) * 4u;
                    const float hd = ldb(rhint, ho, 0u), hv = ldb(rhmask, ho, 0u);
                    const float pred = fast_rcp(sd_rq), df = pred - hd;
                    const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
                    // d log(|hint - pred| + 1) / d pred * d pred / d disp,  pred = 1 / sd,  sd = min_disp + dmul * disp
                    g = fmaf(up_h * hv * sg * fast_rcp(fabsf(df) + 1.f), -(pred * pred) * k.dmul, g);
                }
                if (!olane) g = 0.f;
                if constexpr (SAME) {
                    if (olane) {
                        float* p = k.g_disp[s] + ((size_t)b * H + rq) * W + col;
                        *p = (f > 0) ? *p + g : g;
                    }
                } else {
                    // (5) adjoint of the bilinear up-sampling.  Horizontal: lanes publish g*(1-lx), g*lx; low texel i
                    //     gathers its 2f columns.  Vertical: two running rows, flushed when the row pair advances.
                    __builtin_amdgcn_wave_barrier();
                    sA[lane] = g * (1.f - dg.lx);
                    sB[lane] = g * dg.lx;
                    __builtin_amdgcn_wave_barrier();
                    const int i = ilo + lane;
                    float h = 0.f;
                    if (lane < SX && i < Ws) {
                        const int fz = dg.f;
                        const int cbase = fz * i - (fz >> 1) - X0 + 2;
                        for (int q = 0; q < 2 * fz; ++q) {
                            const int idx = cbase + q;
                            if ((unsigned)idx < (unsigned)WAVE) {
                                const float va = sA[idx], vb = sB[idx];
                                // q < f : columns whose x1 is i (x0 = i - 1); for i = 0 the left-clamped columns (x0 = 0)
                                // q >= f: columns whose x0 is i; at the right edge x1 = x0, so their lx part lands here too
                                h += (q < fz) ? (i == 0 ? va : vb) : (va + (i == Ws - 1 ? vb : 0.f));
                            }
                        }
                    }
                    const RowLerp rl = row_lerp(dg, rq);
                    if (rl.y0 > ja) {
                        flush(ja, accA);
                        accA = accB;
                        accB = 0.f;
                        ja = rl.y0;
                    }
                    accA = fmaf(1.f - rl.ly, h, accA);
                    if (rl.y1 > rl.y0) accB = fmaf(rl.ly, h, accB);
                    else accA = fmaf(rl.ly, h, accA);
                }
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>;
        // record of row kk lives in rec[kk % 3]: no copies between rows
        recB.sdv = recC.sdv = 1.f;
        issue(0, recA);
        step(0, recB, recC, recA, S0());
        step(1, recC, recA, recB, S0());
        step(2, recA, recB, recC, S1());
        step(3, recB, recC, recA, S1());
        // rows in threes, unconditionally (after three steps the records are back in their roles: the loop carries its state
        // in fixed registers, no copies at a merge); up to two steps run on rows past the strip's end and write nothing
        for (int kk = 4; kk < nrows; kk += 3) {
            step(kk, recC, recA, recB, S2());
            step(kk + 1, recA, recB, recC, S2());
            step(kk + 2, recB, recC, recA, S2());
        }
        if constexpr (!SAME) {
            flush(ja, accA);
            if (ja + 1 < Hs) flush(ja + 1, accB);
        }
        if constexpr (POSE) {       // fixed-order wave sums -> one record per (scale, strip, frame)
            float* pp = k.pose_part + (((size_t)s * k.ntiles + tile) * NF + f) * 12;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const float t = wave_sum(pacc[i]);
                if (lane == 0) pp[i] = t;
            }
        }
    }
}

template <int NF, bool POSE = false, bool PLAIN = false>
__global__ __launch_bounds__(NT, POSE ? 1 : DMH_BWD_WAVES) void photo_bwd_kernel(const KArgs k) {
    __shared__ float s_row[WPB][2][WAVE];     // horizontal up-sampling adjoint: g*(1-lx), g*lx of one row
    const int wv = threadIdx.x >> 6;
    const int item = wave_item();
    if (item >= k.ntiles * k.a.num_scales) return;
    // scale fastest: the waves of one workgroup take the scales of ONE strip and share its target / source / selection
    // lines in L1 and L2 (the opposite order re-fetched them per scale: 1.8x the algorithmic traffic, profiles/README.md)
    const int tile = item / k.a.num_scales, s = item - tile * k.a.num_scales;
    if (k.a.Hs[s] == k.a.H) photo_bwd_strip<NF, true, POSE, PLAIN>(k, s, tile, s_row[wv][0], s_row[wv][1]);
    else photo_bwd_strip<NF, false, POSE, PLAIN>(k, s, tile, s_row[wv][0], s_row[wv][1]);
}

// Coarse scales: add the (<= 4) overlapping strip blocks of every low-resolution texel in a fixed order.
struct CArgs {
    const float* stage[DMH_MAX_SCALES];
    float* g_disp[DMH_MAX_SCALES];
    int Hs[DMH_MAX_SCALES], Ws[DMH_MAX_SCALES], sy_slots[DMH_MAX_SCALES], sx_slots[DMH_MAX_SCALES];
    int64_t base[DMH_MAX_SCALES + 1];   // first texel of each coarse scale in the flat index space
    int num_scales, B, H, W, R, tiles_x, tiles_y;
};

__device__ __forceinline__ void low_range(int n_full, int n_low, int lo_px, int hi_px, int& jlo, int& jhi) {
    // low-resolution rows/columns touched by full-resolution pixels [lo_px, hi_px]
    const float r = (float)n_low / (float)n_full;
    const float a = fmaxf(r * ((float)lo_px + 0.5f) - 0.5f, 0.f), b = fmaxf(r * ((float)hi_px + 0.5f) - 0.5f, 0.f);
    jlo = (int)a;
    const int y0 = (int)b;
    jhi = y0 + (y0 < n_low - 1 ? 1 : 0);
}

__global__ __launch_bounds__(NT) void photo_combine_kernel(const CArgs k) {
    const int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (idx >= k.base[k.num_scales]) return;
    int s = 0;
#pragma unroll
    for (int q = 1; q < DMH_MAX_SCALES; ++q)
        if (q < k.num_scales && idx >= k.base[q]) s = q;
    if (k.stage[s] == nullptr) return;
    const int Hs = k.Hs[s], Ws = k.Ws[s], SX = k.sx_slots[s], SY = k.sy_slots[s];
    const int64_t loc = idx - k.base[s];
    const int i = (int)(loc % Ws), j = (int)((loc / Ws) % Hs), b = (int)(loc / ((int64_t)Ws * Hs));
    const int fy = k.H / Hs, fx = k.W / Ws;
    // full-resolution rows / columns that can reach texel (j, i)
    const int ya = max(0, fy * j - (fy >> 1) - 1), yb = min(k.H - 1, fy * j + fy + (fy >> 1));
    const int xa = max(0, fx * i - (fx >> 1) - 1), xb = min(k.W - 1, fx * i + fx + (fx >> 1));
    float acc = 0.f;
    for (int ty = ya / k.R; ty <= yb / k.R; ++ty) {
        int jl, jh;
        low_range(k.H, Hs, ty * k.R, min(ty * k.R + k.R, k.H) - 1, jl, jh);
        if (j < jl || j > jh) continue;
        for (int tx = xa / BW_OUT; tx <= xb / BW_OUT; ++tx) {
            int il, ih;
            low_range(k.W, Ws, tx * BW_OUT, min(tx * BW_OUT + BW_OUT, k.W) - 1, il, ih);
            if (i < il || i > ih) continue;
            const int64_t tile = ((int64_t)b * k.tiles_y + ty) * k.tiles_x + tx;
            acc += k.stage[s][(tile * SY + (j - jl)) * SX + (i - il)];
        }
    }
    k.g_disp[s][loc] = acc;
}

__global__ __launch_bounds__(NT) void unpack_sel_kernel(const uint8_t* __restrict__ sel, int64_t n, int scale,
                                                        float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i < n) out[i] = (float)((sel[i] >> (2 * scale)) & 3u);
}

// ---------------------------------------------------------------------------------------------- host side
int check_photo(const dmh_photo_args* a) {
    DMH_REQUIRE(a != nullptr, "args is null");
    DMH_REQUIRE(a->B > 0 && a->H >= 3 && a->W >= 3, "need B>0, H>=3, W>=3");
    DMH_REQUIRE(a->num_frames >= 1 && a->num_frames <= DMH_MAX_FRAMES, "num_frames out of range (1..3)");
    DMH_REQUIRE(a->num_scales >= 1 && a->num_scales <= DMH_MAX_SCALES, "num_scales out of range");
    DMH_REQUIRE(a->target && a->K && a->inv_K, "null target/K/inv_K");
    DMH_REQUIRE(a->min_depth > 0.f && a->max_depth > a->min_depth, "bad depth range");
    DMH_REQUIRE(a->variant == DMH_VARIANT_MD2 || a->variant == DMH_VARIANT_DH, "unknown variant");
    DMH_REQUIRE((int64_t)a->B * a->H * a->W < (int64_t)1 << 31 && (int64_t)3 * a->H * a->W < (int64_t)1 << 29, "tensor too large");
    for (int f = 0; f < a->num_frames; ++f) DMH_REQUIRE(a->source[f] && a->T[f], "null source/T");
    for (int s = 0; s < a->num_scales; ++s) {
        DMH_REQUIRE(a->disp[s] != nullptr, "null disp");
        const int Hs = a->Hs[s], Ws = a->Ws[s];
        DMH_REQUIRE(Hs >= 1 && Ws >= 1 && Hs <= a->H && Ws <= a->W, "bad disp size");
        const int f = a->H / Hs;
        DMH_REQUIRE(f * Hs == a->H && f * Ws == a->W && (f & (f - 1)) == 0 && f <= 16,
                    "disparity size must be the image size divided by 1, 2, 4, 8 or 16 (MD2/trainer.py:52-53 asserts multiples of 32)");
        if (a->noise_mode == DMH_NOISE_TENSOR) DMH_REQUIRE(a->noise[s] != nullptr, "null noise tensor");
    }
    DMH_REQUIRE(a->noise_mode >= DMH_NOISE_NONE && a->noise_mode <= DMH_NOISE_PHILOX, "bad noise_mode");
    DMH_REQUIRE((a->depth_hint == nullptr) == (a->depth_hint_mask == nullptr), "depth_hint and depth_hint_mask go together");
    if (a->depth_hint)
        DMH_REQUIRE(a->variant == DMH_VARIANT_DH && a->automask && a->num_frames == 1,
                    "depth hints need the DH variant, auto-masking and exactly one (stereo) source frame");
    return DMH_OK;
}

void fill_kargs(KArgs& k, const dmh_photo_args* a, int out_cols) {
    memset(&k, 0, sizeof(k));
    k.a = *a;
    k.R = pick_rows(a->B, a->H, a->W, out_cols);
    k.tiles_x = (a->W + out_cols - 1) / out_cols;
    k.tiles_y = (a->H + k.R - 1) / k.R;
    k.ntiles = k.tiles_x * k.tiles_y * a->B;
    const double min_disp = 1.0 / (double)a->max_depth, max_disp = 1.0 / (double)a->min_depth;
    k.min_disp = (float)min_disp;
    k.dmul = (float)(max_disp - min_disp);
}

// staging block of one backward strip at scale s: low-resolution rows x columns it can touch
void stage_slots(const dmh_photo_args* a, int R, int s, int& sy, int& sx) {
    const int f = a->H / a->Hs[s];
    sy = R / f + 2;
    sx = BW_OUT / f + 2;
    if (R % f) ++sy;
    if (BW_OUT % f) ++sx;
}

}  // namespace

extern "C" {

int64_t dmh_photo_partials_size(int B, int H, int W, int num_scales) {
    const int R = pick_rows(B, H, W, FW_OUT);
    const int64_t tiles = (int64_t)((W + FW_OUT - 1) / FW_OUT) * ((H + R - 1) / R) * B;
    return tiles * 4 * num_scales;
}

int64_t dmh_photo_stage_size(const dmh_photo_args* a) {
    if (!a || check_photo(a) != DMH_OK) return 0;
    KArgs k;
    fill_kargs(k, a, BW_OUT);
    int64_t n = 1;   // never hand out a zero-sized workspace
    for (int s = 0; s < a->num_scales; ++s) {
        if (a->Hs[s] == a->H) continue;
        int sy, sx;
        stage_slots(a, k.R, s, sy, sx);
        n += (int64_t)k.ntiles * sy * sx;
    }
    return n;
}

int dmh_photo_loss_fwd(const dmh_photo_args* a, uint8_t* sel, float* const to_opt[DMH_MAX_SCALES], float* partials,
                       void* stream) {
    if (int rc = check_photo(a)) return rc;
    DMH_REQUIRE(sel != nullptr && partials != nullptr, "null outputs");
    DMH_REQUIRE(a->num_frames <= 3, "the packed selection map holds 2 bits per scale: at most 3 source frames");
    KArgs k;
    fill_kargs(k, a, FW_OUT);
    k.sel = sel;
    for (int s = 0; s < a->num_scales; ++s) k.to_opt[s] = to_opt ? to_opt[s] : nullptr;
    k.partials = partials;
    const dim3 grid((k.ntiles + WPB - 1) / WPB), block(NT);
    switch (a->num_frames) {
        // one source frame: two passes of two scales (3 streams of row sums in registers, 160 VGPRs, 3 waves/SIMD).  Measured
        // alternatives (profiles/README.md): one pass of four scales with all five streams in registers (229 VGPRs, 2
        // waves/SIMD; round 2) +10 %; one pass with the row sums of scales 2-3 in wave-private LDS (round 3, this kernel with
        // SPL = 2: 10 % fewer VALU instructions, but 194 VGPRs = 2 waves/SIMD, or 168 with scratch spills) +20 %:
        // DMH_K1_FWD_VARIANT=2 selects it for timing comparisons.
        case 1: {
            static const int variant = getenv("DMH_K1_FWD_VARIANT") ? atoi(getenv("DMH_K1_FWD_VARIANT")) : 0;
            if (a->depth_hint) hipLaunchKernelGGL((photo_fwd_kernel<1, 2, 0, true>), grid, block, 0, (hipStream_t)stream, k);
            else if (variant == 2) hipLaunchKernelGGL((photo_fwd_kernel<1, 2, 2, false>), grid, block, 0, (hipStream_t)stream, k);
            else if (variant == 1 || !a->automask || a->no_ssim)
                hipLaunchKernelGGL((photo_fwd_kernel<1, 2, 0, false>), grid, block, 0, (hipStream_t)stream, k);
            else {      // the training configurations: flags as constants
                const bool md2 = a->variant == DMH_VARIANT_MD2;
#define DMH_K1_FWD_FL(MD2, NOISE) \
    hipLaunchKernelGGL((photo_fwd_kernel<1, 2, 0, false, 3, 1 | ((MD2) << 2) | ((NOISE) << 3)>), grid, block, 0, (hipStream_t)stream, k)
                if (a->noise_mode == DMH_NOISE_PHILOX) { if (md2) DMH_K1_FWD_FL(1, DMH_NOISE_PHILOX); else DMH_K1_FWD_FL(0, DMH_NOISE_PHILOX); }
                else if (a->noise_mode == DMH_NOISE_TENSOR) { if (md2) DMH_K1_FWD_FL(1, DMH_NOISE_TENSOR); else DMH_K1_FWD_FL(0, DMH_NOISE_TENSOR); }
                else { if (md2) DMH_K1_FWD_FL(1, DMH_NOISE_NONE); else DMH_K1_FWD_FL(0, DMH_NOISE_NONE); }
#undef DMH_K1_FWD_FL
            }
            break;
        }
        case 2: hipLaunchKernelGGL((photo_fwd_kernel<2, 2, 0, false>), grid, block, 0, (hipStream_t)stream, k); break;
        default: hipLaunchKernelGGL((photo_fwd_kernel<3, 1, 0, false>), grid, block, 0, (hipStream_t)stream, k); break;
    }
    return check_launch("dmh_photo_loss_fwd");
}

int64_t dmh_photo_pose_partials_size(const dmh_photo_args* a) {
    if (!a || check_photo(a) != DMH_OK) return 0;
    KArgs k;
    fill_kargs(k, a, BW_OUT);
    return (int64_t)a->num_scales * k.ntiles * a->num_frames * 12;
}

int dmh_photo_loss_bwd(const dmh_photo_args* a, const uint8_t* sel, const float* gvec, const float* fin, float* stage,
                       float* const g_disp[DMH_MAX_SCALES], void* stream) {
    return dmh_photo_loss_bwd_pose(a, sel, gvec, fin, stage, g_disp, nullptr, stream);
}

int dmh_photo_loss_bwd_pose(const dmh_photo_args* a, const uint8_t* sel, const float* gvec, const float* fin, float* stage,
                            float* const g_disp[DMH_MAX_SCALES], float* pose_partials, void* stream) {
    if (int rc = check_photo(a)) return rc;
    DMH_REQUIRE(sel && gvec && fin && stage && g_disp, "null argument");
    DMH_REQUIRE(a->num_frames <= 3, "at most 3 source frames");
    KArgs k;
    fill_kargs(k, a, BW_OUT);
    k.csel = sel;
    k.gvec = gvec;
    k.fin = fin;
    CArgs c;
    memset(&c, 0, sizeof(c));
    int64_t off = 0, texels = 0;
    bool coarse = false;
    for (int s = 0; s < a->num_scales; ++s) {
        DMH_REQUIRE(g_disp[s] != nullptr, "null g_disp[s]");
        k.g_disp[s] = g_disp[s];
        c.g_disp[s] = g_disp[s];
        c.Hs[s] = a->Hs[s];
        c.Ws[s] = a->Ws[s];
        c.base[s] = texels;
        if (a->Hs[s] != a->H) {
            stage_slots(a, k.R, s, k.sy_slots[s], k.sx_slots[s]);
            k.stage[s] = stage + off;
            c.stage[s] = k.stage[s];
            c.sy_slots[s] = k.sy_slots[s];
            c.sx_slots[s] = k.sx_slots[s];
            off += (int64_t)k.ntiles * k.sy_slots[s] * k.sx_slots[s];
            texels += (int64_t)a->B * a->Hs[s] * a->Ws[s];
            coarse = true;
        }
        c.base[s + 1] = texels;
    }
    const int items = k.ntiles * a->num_scales;
    const dim3 grid((items + WPB - 1) / WPB), block(NT);
    k.pose_part = pose_partials;
    if (pose_partials) {
        switch (a->num_frames) {
            case 1: hipLaunchKernelGGL((photo_bwd_kernel<1, true>), grid, block, 0, (hipStream_t)stream, k); break;
            case 2: hipLaunchKernelGGL((photo_bwd_kernel<2, true>), grid, block, 0, (hipStream_t)stream, k); break;
            default: hipLaunchKernelGGL((photo_bwd_kernel<3, true>), grid, block, 0, (hipStream_t)stream, k); break;
        }
    } else {
        switch (a->num_frames) {
            case 1:
                if (!a->no_ssim && !a->depth_hint) hipLaunchKernelGGL((photo_bwd_kernel<1, false, true>), grid, block, 0, (hipStream_t)stream, k);
                else hipLaunchKernelGGL((photo_bwd_kernel<1>), grid, block, 0, (hipStream_t)stream, k);
                break;
            case 2: hipLaunchKernelGGL((photo_bwd_kernel<2>), grid, block, 0, (hipStream_t)stream, k); break;
            default: hipLaunchKernelGGL((photo_bwd_kernel<3>), grid, block, 0, (hipStream_t)stream, k); break;
        }
    }
    if (int rc = check_launch("dmh_photo_loss_bwd")) return rc;
    if (coarse) {
        c.num_scales = a->num_scales;
        c.B = a->B;
        c.H = a->H;
        c.W = a->W;
        c.R = k.R;
        c.tiles_x = k.tiles_x;
        c.tiles_y = k.tiles_y;
        hipLaunchKernelGGL(photo_combine_kernel, dim3((unsigned)((texels + NT - 1) / NT)), block, 0, (hipStream_t)stream, c);
        return check_launch("dmh_photo_loss_bwd (combine)");
    }
    return DMH_OK;
}

int dmh_unpack_selection(const uint8_t* sel, int64_t n, int scale, float* out, void* stream) {
    DMH_REQUIRE(sel && out, "null pointer");
    DMH_REQUIRE(n > 0 && scale >= 0 && scale < DMH_MAX_SCALES, "bad size / scale");
    hipLaunchKernelGGL(unpack_sel_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, sel, n, scale, out);
    return check_launch("dmh_unpack_selection");
}

}  // extern "C"
