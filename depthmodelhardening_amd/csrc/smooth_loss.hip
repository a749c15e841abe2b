// K2 -- edge-aware smoothness on the mean-normalised disparity, every scale in one launch, and
// the deterministic finalisation of the K1 + K2 partial sums.
//
// Reference: MD2/trainer.py:662-668 + MD2/layers.py:207-220.
//   norm = disp / (mean_hw(disp) + 1e-7);  smooth = mean|dx norm| e^{-mean_c|dx I|} + mean|dy norm| e^{-mean_c|dy I|}
// norm is disp times one scalar per image, so one streaming pass over (disp, color) yields both
// sum(disp) and the un-normalised edge-weighted sums; the division happens in the finalise kernel:
//   smooth = sum_b R_b / |mean_b + 1e-7|,  R_b = rawx_b/(B*H*(W-1)) + rawy_b/(B*(H-1)*W).
// The loss is positively homogeneous of degree 1 in norm, which gives the backward in one pass:
//   d smooth / d disp_j = G_j/|den_b| - R_b * sign(den_b) / (den_b^2 * H*W),  den_b = mean_b + 1e-7.
// Pure HBM streaming (16 B per low-resolution pixel); no LDS tiling needed beyond the block reduce.
#include <stdlib.h>

#include "common.hpp"

using namespace dmh;

namespace {

constexpr int NT = 256;
constexpr int ROWS = 8;  // rows per workgroup

struct Layout {
    int nchunk[DMH_MAX_SCALES];
    int blk_base[DMH_MAX_SCALES + 1];  // first block of scale s (blocks = B * nchunk)
};

__host__ __device__ inline Layout make_layout(const dmh_smooth_args& a) {
    Layout l;
    l.blk_base[0] = 0;
    for (int s = 0; s < DMH_MAX_SCALES; ++s) {
        l.nchunk[s] = s < a.num_scales ? (a.Hs[s] + ROWS - 1) / ROWS : 0;
        l.blk_base[s + 1] = l.blk_base[s] + l.nchunk[s] * a.B;
    }
    return l;
}

struct SArgs {
    dmh_smooth_args a;
    Layout l;
    float* partials;             // [blocks][3] = sum disp, rawx, rawy
    const float* gvec;
    const float* sstats;         // [NS][B][2] = mean, R
    float* g_disp[DMH_MAX_SCALES];
    float smooth_wt;
    int accumulate;
};

__device__ __forceinline__ void decode(const SArgs& k, int& s, int& b, int& chunk) {
    const int id = blockIdx.x;
    s = 0;
#pragma unroll
    for (int i = 1; i < DMH_MAX_SCALES; ++i)
        if (i < k.a.num_scales && id >= k.l.blk_base[i]) s = i;
    const int r = id - k.l.blk_base[s];
    b = r / k.l.nchunk[s];
    chunk = r - b * k.l.nchunk[s];
}

__device__ __forceinline__ float edge_w(const float* __restrict__ I, int hw, int p, int q) {
    const float g = fabsf(I[p] - I[q]) + fabsf(I[hw + p] - I[hw + q]) + fabsf(I[2 * hw + p] - I[2 * hw + q]);
    return expf(-(g / 3.f));
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(NT) void smooth_fwd_kernel(const SArgs k) {
    __shared__ float s_red[NT / WAVE];
    int s, b, chunk;
    decode(k, s, b, chunk);
    const int Hs = k.a.Hs[s], Ws = k.a.Ws[s], hw = Hs * Ws;
    const float* d = k.a.disp[s] + (size_t)b * hw;
    const float* I = k.a.color[s] + (size_t)b * 3 * hw;
    const int ylo = chunk * ROWS, yhi = min(Hs, ylo + ROWS);
    float sd = 0.f, rx = 0.f, ry = 0.f;
    for (int y = ylo; y < yhi; ++y)
        for (int x = threadIdx.x; x < Ws; x += NT) {
            const int p = y * Ws + x;
            const float v = d[p];
            sd += v;
            if (x < Ws - 1) rx += fabsf(v - d[p + 1]) * edge_w(I, hw, p, p + 1);
            if (y < Hs - 1) ry += fabsf(v - d[p + Ws]) * edge_w(I, hw, p, p + Ws);
        }
    const float t0 = block_sum<NT>(sd, s_red);
    const float t1 = block_sum<NT>(rx, s_red);
    const float t2 = block_sum<NT>(ry, s_red);
    if (threadIdx.x == 0) {
        float* o = k.partials + (size_t)blockIdx.x * 3;
        o[0] = t0;
        o[1] = t1;
        o[2] = t2;
    }
}

__global__ __launch_bounds__(NT) void smooth_bwd_kernel(const SArgs k) {
    int s, b, chunk;
    decode(k, s, b, chunk);
    const int Hs = k.a.Hs[s], Ws = k.a.Ws[s], hw = Hs * Ws, B = k.a.B;
    const float* d = k.a.disp[s] + (size_t)b * hw;
    const float* I = k.a.color[s] + (size_t)b * 3 * hw;
    float* g = k.g_disp[s] + (size_t)b * hw;
    const float up = (k.gvec[DMH_FIN_LOSS] / (float)k.a.num_scales + k.gvec[DMH_FIN_LOSS_S + s]) * k.smooth_wt /
                         (float)(1 << s) +
                     k.gvec[DMH_FIN_SMOOTH_S + s];
    const float mean = k.sstats[(s * B + b) * 2 + 0], R = k.sstats[(s * B + b) * 2 + 1];
    const float den = mean + 1e-7f;
    const float inv_abs = 1.f / fabsf(den);
    const float shift = R * sgn(den) / (den * den * (float)hw);
    const float cx = 1.f / ((float)B * (float)Hs * (float)(Ws - 1));
    const float cy = 1.f / ((float)B * (float)(Hs - 1) * (float)Ws);
    const int ylo = chunk * ROWS, yhi = min(Hs, ylo + ROWS);
    for (int y = ylo; y < yhi; ++y)
        for (int x = threadIdx.x; x < Ws; x += NT) {
            const int p = y * Ws + x;
            const float v = d[p];
            float G = 0.f;
            if (x < Ws - 1) G += cx * sgn(v - d[p + 1]) * edge_w(I, hw, p, p + 1);
            if (x > 0) G -= cx * sgn(d[p - 1] - v) * edge_w(I, hw, p - 1, p);
            if (y < Hs - 1) G += cy * sgn(v - d[p + Ws]) * edge_w(I, hw, p, p + Ws);
            if (y > 0) G -= cy * sgn(d[p - Ws] - v) * edge_w(I, hw, p - Ws, p);
            float out = up * (G * inv_abs - shift);
            if (k.accumulate) out += g[p];
            g[p] = out;
        }
}

// Four-wide form (every scale width a multiple of 4 and the rows 16-byte aligned): a thread owns four adjacent columns
// and walks RPT rows downward.  Each edge weight exp(-mean_c|dI|) is formed ONCE (the scalar kernel above forms the four
// weights around every pixel: each edge twice), the disparity and colour rows move as 16-byte accesses and each row is
// read once per thread (the row above stays in registers for the vertical edges): 2.25 exponentials and 16 B of loads
// per pixel instead of 4 and 24 loads.  Same per-element expression tree as the scalar kernel.
constexpr int RPT = 8;     // rows walked by one thread

struct Layout4 {
    int tw[DMH_MAX_SCALES];         // threads across one row band (power of two <= 256)
    int cchunks[DMH_MAX_SCALES];    // column chunks of 4 * tw columns
    int rchunks[DMH_MAX_SCALES];    // row chunks of (256 / tw) * RPT rows
    int blk_base[DMH_MAX_SCALES + 1];
};

__host__ __device__ inline Layout4 make_layout4(const dmh_smooth_args& a) {
    Layout4 l;
    l.blk_base[0] = 0;
    for (int s = 0; s < DMH_MAX_SCALES; ++s) {
        l.tw[s] = l.cchunks[s] = l.rchunks[s] = 0;
        if (s < a.num_scales) {
            const int q = (a.Ws[s] + 3) / 4;
            int tw = 1;
            while (tw < q && tw < NT) tw <<= 1;
            l.tw[s] = tw;
            l.cchunks[s] = (q + tw - 1) / tw;
            l.rchunks[s] = (a.Hs[s] + (NT / tw) * RPT - 1) / ((NT / tw) * RPT);
        }
        l.blk_base[s + 1] = l.blk_base[s] + l.cchunks[s] * l.rchunks[s] * a.B;
    }
    return l;
}

struct Row4 {      // one image row under a thread: its four columns and the column to their right
    float4 d;
    float dr;
    float4 I[3];
    float Ir[3];
};

__device__ __forceinline__ Row4 load_row4(const float* __restrict__ d, const float* __restrict__ I, int hw, int Ws, int y,
                                          int x0) {
    Row4 r;
    const int p = y * Ws + x0;
    r.d = *reinterpret_cast<const float4*>(d + p);
    const bool right = x0 + 4 < Ws;
    r.dr = right ? d[p + 4] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        r.I[c] = *reinterpret_cast<const float4*>(I + c * hw + p);
        r.Ir[c] = right ? I[c * hw + p + 4] : 0.f;
    }
    return r;
}

__device__ __forceinline__ float wexp(float a0, float b0, float a1, float b1, float a2, float b2) {
    const float g = fabsf(a0 - b0) + fabsf(a1 - b1) + fabsf(a2 - b2);
    return expf(-(g / 3.f));
}

__global__ __launch_bounds__(NT) void smooth_bwd4_kernel(const SArgs k, const Layout4 l) {
    int s = 0;
#pragma unroll
    for (int i = 1; i < DMH_MAX_SCALES; ++i)
        if (i < k.a.num_scales && (int)blockIdx.x >= l.blk_base[i]) s = i;
    int r = (int)blockIdx.x - l.blk_base[s];
    const int cc = r % l.cchunks[s];
    r /= l.cchunks[s];
    const int rc = r % l.rchunks[s], b = r / l.rchunks[s];
    const int Hs = k.a.Hs[s], Ws = k.a.Ws[s], hw = Hs * Ws, B = k.a.B;
    const int tw = l.tw[s];
    const int tx = threadIdx.x & (tw - 1), tg = threadIdx.x / tw;
    const int x0 = 4 * (cc * tw + tx);
    const int ylo = (rc * (NT / tw) + tg) * RPT, yhi = min(Hs, ylo + RPT);
    if (x0 >= Ws || ylo >= Hs) return;
    const float* d = k.a.disp[s] + (size_t)b * hw;
    const float* I = k.a.color[s] + (size_t)b * 3 * hw;
    float* g = k.g_disp[s] + (size_t)b * hw;
    const float up = (k.gvec[DMH_FIN_LOSS] / (float)k.a.num_scales + k.gvec[DMH_FIN_LOSS_S + s]) * k.smooth_wt /
                         (float)(1 << s) +
                     k.gvec[DMH_FIN_SMOOTH_S + s];
    const float mean = k.sstats[(s * B + b) * 2 + 0], R = k.sstats[(s * B + b) * 2 + 1];
    const float den = mean + 1e-7f;
    const float inv_abs = 1.f / fabsf(den);
    const float shift = R * sgn(den) / (den * den * (float)hw);
    const float cx = 1.f / ((float)B * (float)Hs * (float)(Ws - 1));
    const float cy = 1.f / ((float)B * (float)(Hs - 1) * (float)Ws);

    // vertical edge terms sgn(d(y) - d(y+1)) * w between the row above the band and its first row
    float vprev[4] = {0.f, 0.f, 0.f, 0.f};
    Row4 cur = load_row4(d, I, hw, Ws, ylo, x0);
    if (ylo > 0) {
        const Row4 ur = load_row4(d, I, hw, Ws, ylo - 1, x0);
        const float ud[4] = {ur.d.x, ur.d.y, ur.d.z, ur.d.w}, cd[4] = {cur.d.x, cur.d.y, cur.d.z, cur.d.w};
        const float u0[4] = {ur.I[0].x, ur.I[0].y, ur.I[0].z, ur.I[0].w}, u1[4] = {ur.I[1].x, ur.I[1].y, ur.I[1].z, ur.I[1].w};
        const float u2[4] = {ur.I[2].x, ur.I[2].y, ur.I[2].z, ur.I[2].w};
        const float c0[4] = {cur.I[0].x, cur.I[0].y, cur.I[0].z, cur.I[0].w}, c1[4] = {cur.I[1].x, cur.I[1].y, cur.I[1].z, cur.I[1].w};
        const float c2[4] = {cur.I[2].x, cur.I[2].y, cur.I[2].z, cur.I[2].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) vprev[j] = sgn(ud[j] - cd[j]) * wexp(u0[j], c0[j], u1[j], c1[j], u2[j], c2[j]);
    }
    for (int y = ylo; y < yhi; ++y) {
        const bool below = y < Hs - 1;
        Row4 nxt = cur;
        if (below) nxt = load_row4(d, I, hw, Ws, y + 1, x0);
        const int p = y * Ws + x0;
        const float cd[5] = {cur.d.x, cur.d.y, cur.d.z, cur.d.w, cur.dr};
        const float c0[5] = {cur.I[0].x, cur.I[0].y, cur.I[0].z, cur.I[0].w, cur.Ir[0]};
        const float c1[5] = {cur.I[1].x, cur.I[1].y, cur.I[1].z, cur.I[1].w, cur.Ir[1]};
        const float c2[5] = {cur.I[2].x, cur.I[2].y, cur.I[2].z, cur.I[2].w, cur.Ir[2]};
        // horizontal edge terms of (x0-1, x0) ... (x0+3, x0+4)
        float hl = 0.f;
        if (x0 > 0) hl = sgn(d[p - 1] - cd[0]) * wexp(I[p - 1], c0[0], I[hw + p - 1], c1[0], I[2 * hw + p - 1], c2[0]);
        float hx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            hx[j] = (x0 + j < Ws - 1) ? sgn(cd[j] - cd[j + 1]) * wexp(c0[j], c0[j + 1], c1[j], c1[j + 1], c2[j], c2[j + 1]) : 0.f;
        const float nd[4] = {nxt.d.x, nxt.d.y, nxt.d.z, nxt.d.w};
        const float n0[4] = {nxt.I[0].x, nxt.I[0].y, nxt.I[0].z, nxt.I[0].w};
        const float n1[4] = {nxt.I[1].x, nxt.I[1].y, nxt.I[1].z, nxt.I[1].w};
        const float n2[4] = {nxt.I[2].x, nxt.I[2].y, nxt.I[2].z, nxt.I[2].w};
        float out[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float vy = below ? sgn(cd[j] - nd[j]) * wexp(c0[j], n0[j], c1[j], n1[j], c2[j], n2[j]) : 0.f;
            float G = 0.f;
            if (x0 + j < Ws - 1) G += cx * hx[j];
            if (x0 + j > 0) G -= cx * (j == 0 ? hl : hx[j - 1]);
            if (below) G += cy * vy;
            if (y > 0) G -= cy * vprev[j];
            vprev[j] = vy;
            out[j] = up * (G * inv_abs - shift);
        }
        float4* gp = reinterpret_cast<float4*>(g + p);
        float4 o = make_float4(out[0], out[1], out[2], out[3]);
        if (k.accumulate) {
            const float4 old = *gp;
            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        *gp = o;
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------- finalise
struct FArgs {
    const float* photo;   // [NS][nblk][4] = selected-loss sum, selected count, depth-hint loss sum, depth-hint count
    const float* smooth;  // [blocks][3]
    dmh_smooth_args sm;
    Layout l;
    int B, H, W, nblk, variant;
    float smooth_wt;
    float* fin;
    float* sstats;
};

constexpr int NTF = 1024;  // the finalise kernel is one workgroup: make it as wide as the hardware allows

__device__ __forceinline__ double wave_sum_d(double v) {     // fixed butterfly order: bitwise reproducible
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
    return v;
}

// One workgroup.  Every thread accumulates its slice of the partials in double; the 20 block sums (4 quantities x 4 scales
// of K1, the smoothness term of each scale) are formed together: a wave butterfly each, ONE pass through LDS, then thread q
// adds the 16 wave results of quantity q in wave order (round 2 ran twenty 10-level LDS trees, ~200 barriers, 51 us).
__global__ __launch_bounds__(NTF) void finalize_kernel(const FArgs k) {
    constexpr int NQ = 5 * DMH_MAX_SCALES, NW = NTF / WAVE;
    __shared__ double s_part[NQ][NW];
    __shared__ double s_tot[NQ];
    const int NS = k.sm.num_scales, B = k.B, tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
    double q[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) q[i] = 0.0;
    // photometric partials: one strided pass over [NS][nblk] float4, all scales' loads independent and in flight
    const float4* ph = reinterpret_cast<const float4*>(k.photo);
    for (int i = tid; i < k.nblk; i += NTF) {
#pragma unroll
        for (int s = 0; s < DMH_MAX_SCALES; ++s) {
            if (s < NS) {
                const float4 v = ph[(size_t)s * k.nblk + i];
                q[4 * s + 0] += (double)v.x;
                q[4 * s + 1] += (double)v.y;
                q[4 * s + 2] += (double)v.z;
                q[4 * s + 3] += (double)v.w;
            }
        }
    }
    // smoothness: one (scale, image) pair per thread, chunks summed in a fixed order
    for (int p = tid; p < NS * B; p += NTF) {
        const int s = p / B, b = p - s * B;
        const int nc = k.l.nchunk[s];
        const float* sp = k.smooth + ((size_t)k.l.blk_base[s] + (size_t)b * nc) * 3;
        double sd = 0.0, rx = 0.0, ry = 0.0;
        for (int c = 0; c < nc; ++c) {
            sd += (double)sp[c * 3 + 0];
            rx += (double)sp[c * 3 + 1];
            ry += (double)sp[c * 3 + 2];
        }
        const int Hs = k.sm.Hs[s], Ws = k.sm.Ws[s];
        const double mean = sd / ((double)Hs * Ws);
        const double R = rx / ((double)B * Hs * (Ws - 1)) + ry / ((double)B * (Hs - 1) * Ws);
        k.sstats[(s * B + b) * 2 + 0] = (float)mean;
        k.sstats[(s * B + b) * 2 + 1] = (float)R;
        const double den = mean + 1e-7;
        const double term = R / (den < 0 ? -den : den);
#pragma unroll
        for (int j = 0; j < DMH_MAX_SCALES; ++j)
            if (j == s) q[4 * DMH_MAX_SCALES + j] += term;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const double w = wave_sum_d(q[i]);
        if (lane == 0) s_part[i][wv] = w;
    }
    __syncthreads();
    if (tid < NQ) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += s_part[tid][w];
        s_tot[tid] = t;
    }
    __syncthreads();
    if (tid == 0) {
        double total = 0.0;
        for (int i = 0; i < DMH_FIN_SIZE; ++i) k.fin[i] = 0.f;
        for (int s = 0; s < NS; ++s) {
            const double S1 = s_tot[4 * s + 0], S2 = s_tot[4 * s + 1], S3 = s_tot[4 * s + 2], S4 = s_tot[4 * s + 3];
            const double reproj = (k.variant == DMH_VARIANT_MD2) ? S1 / ((double)B * k.H * k.W) : S1 / (S2 + 1e-7);
            const double hint = S3 / (S4 + 1e-7);    // depth_hint_loss.sum() / (mask.sum() + 1e-7), DH/trainer.py:721; 0 without hints
            const double smooth = s_tot[4 * DMH_MAX_SCALES + s];
            const double ls = reproj + hint + (double)k.smooth_wt * smooth / (double)(1 << s);
            total += ls;
            k.fin[DMH_FIN_LOSS_S + s] = (float)ls;
            k.fin[DMH_FIN_REPROJ_S + s] = (float)reproj;
            k.fin[DMH_FIN_COUNT_S + s] = (float)S2;
            k.fin[DMH_FIN_SMOOTH_S + s] = (float)smooth;
            k.fin[DMH_FIN_HINT_S + s] = (float)hint;
            k.fin[DMH_FIN_HINTCOUNT_S + s] = (float)S4;
        }
        k.fin[DMH_FIN_LOSS] = (float)(total / NS);
    }
}

int check_smooth(const dmh_smooth_args* a) {
    DMH_REQUIRE(a != nullptr, "args is null");
    DMH_REQUIRE(a->B > 0 && a->num_scales >= 1 && a->num_scales <= DMH_MAX_SCALES, "bad B/num_scales");
    for (int s = 0; s < a->num_scales; ++s) {
        DMH_REQUIRE(a->disp[s] && a->color[s], "null disp/color");
        DMH_REQUIRE(a->Hs[s] >= 2 && a->Ws[s] >= 2, "need Hs,Ws >= 2");
    }
    return DMH_OK;
}

}  // namespace

extern "C" {

int64_t dmh_smooth_partials_size(const dmh_smooth_args* a) {
    // (found by the ASan host build: a num_scales beyond the struct's arrays was indexed here before any check)
    if (!a || a->num_scales < 1 || a->num_scales > DMH_MAX_SCALES || a->B <= 0) return 0;
    const Layout l = make_layout(*a);
    return (int64_t)l.blk_base[a->num_scales] * 3;
}

int dmh_smooth_loss_fwd(const dmh_smooth_args* a, float* partials, void* stream) {
    if (int rc = check_smooth(a)) return rc;
    DMH_REQUIRE(partials != nullptr, "null partials");
    SArgs k;
    memset(&k, 0, sizeof(k));
    k.a = *a;
    k.l = make_layout(*a);
    k.partials = partials;
    hipLaunchKernelGGL(smooth_fwd_kernel, dim3(k.l.blk_base[a->num_scales]), dim3(NT), 0, (hipStream_t)stream, k);
    return check_launch("dmh_smooth_loss_fwd");
}

int dmh_smooth_loss_bwd(const dmh_smooth_args* a, const float* gvec, const float* sstats, float smooth_wt,
                        float* const g_disp[DMH_MAX_SCALES], int accumulate, void* stream) {
    if (int rc = check_smooth(a)) return rc;
    DMH_REQUIRE(gvec && sstats && g_disp, "null argument");
    SArgs k;
    memset(&k, 0, sizeof(k));
    k.a = *a;
    k.l = make_layout(*a);
    k.gvec = gvec;
    k.sstats = sstats;
    k.smooth_wt = smooth_wt;
    k.accumulate = accumulate;
    for (int s = 0; s < a->num_scales; ++s) {
        DMH_REQUIRE(g_disp[s] != nullptr, "null g_disp[s]");
        k.g_disp[s] = g_disp[s];
    }
    // four-wide kernel: every scale's width a multiple of 4 and every row of disp / color / g_disp 16-byte aligned
    bool vec4 = !(getenv("DMH_K2_BWD_SCALAR") && atoi(getenv("DMH_K2_BWD_SCALAR")) != 0);
    for (int s = 0; s < a->num_scales; ++s)
        vec4 = vec4 && a->Ws[s] % 4 == 0 && ((uintptr_t)a->disp[s] % 16) == 0 && ((uintptr_t)a->color[s] % 16) == 0 &&
               ((uintptr_t)g_disp[s] % 16) == 0 && ((size_t)a->Hs[s] * a->Ws[s]) % 4 == 0;
    if (vec4) {
        const Layout4 l4 = make_layout4(*a);
        hipLaunchKernelGGL(smooth_bwd4_kernel, dim3(l4.blk_base[a->num_scales]), dim3(NT), 0, (hipStream_t)stream, k, l4);
    } else {
        hipLaunchKernelGGL(smooth_bwd_kernel, dim3(k.l.blk_base[a->num_scales]), dim3(NT), 0, (hipStream_t)stream, k);
    }
    return check_launch("dmh_smooth_loss_bwd");
}

int dmh_loss_finalize(const float* photo_partials, const float* smooth_partials, int B, int H, int W,
                      const dmh_smooth_args* sm, int variant, float smooth_wt, float* fin, float* sstats,
                      void* stream) {
    if (int rc = check_smooth(sm)) return rc;
    DMH_REQUIRE(photo_partials && smooth_partials && fin && sstats, "null argument");
    DMH_REQUIRE(B == sm->B && H >= 3 && W >= 3, "bad sizes");
    FArgs k;
    memset(&k, 0, sizeof(k));
    k.photo = photo_partials;
    k.smooth = smooth_partials;
    k.sm = *sm;
    k.l = make_layout(*sm);
    k.B = B;
    k.H = H;
    k.W = W;
    k.nblk = (int)(dmh_photo_partials_size(B, H, W, 1) / 4);
    k.variant = variant;
    k.smooth_wt = smooth_wt;
    k.fin = fin;
    k.sstats = sstats;
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(NTF), 0, (hipStream_t)stream, k);
    return check_launch("dmh_loss_finalize");
}

}  // extern "C"
