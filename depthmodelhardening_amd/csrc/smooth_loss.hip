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

__device__ __forceinline__ double block_sum_d(double v, double* red) {
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = NTF / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    return red[0];
}

__global__ __launch_bounds__(NTF) void finalize_kernel(const FArgs k) {
    __shared__ double s_red[NTF];
    __shared__ double s_sm[DMH_MAX_SCALES];
    const int NS = k.sm.num_scales, B = k.B, tid = threadIdx.x;
    if (tid < DMH_MAX_SCALES) s_sm[tid] = 0.0;
    // photometric partials: one strided pass over [NS][nblk] float2, all scales' loads independent and in flight
    double a1[DMH_MAX_SCALES] = {0, 0, 0, 0}, a2[DMH_MAX_SCALES] = {0, 0, 0, 0};
    double a3[DMH_MAX_SCALES] = {0, 0, 0, 0}, a4[DMH_MAX_SCALES] = {0, 0, 0, 0};
    const float4* ph = reinterpret_cast<const float4*>(k.photo);
    for (int i = tid; i < k.nblk; i += NTF) {
#pragma unroll
        for (int s = 0; s < DMH_MAX_SCALES; ++s) {
            if (s < NS) {
                const float4 v = ph[(size_t)s * k.nblk + i];
                a1[s] += (double)v.x;
                a2[s] += (double)v.y;
                a3[s] += (double)v.z;
                a4[s] += (double)v.w;
            }
        }
    }
    double reproj[DMH_MAX_SCALES], count[DMH_MAX_SCALES], hint[DMH_MAX_SCALES], hcount[DMH_MAX_SCALES];
#pragma unroll
    for (int s = 0; s < DMH_MAX_SCALES; ++s) {
        reproj[s] = count[s] = hint[s] = hcount[s] = 0.0;
        if (s < NS) {
            const double S1 = block_sum_d(a1[s], s_red);
            const double S2 = block_sum_d(a2[s], s_red);
            const double S3 = block_sum_d(a3[s], s_red);
            const double S4 = block_sum_d(a4[s], s_red);
            count[s] = S2;
            reproj[s] = (k.variant == DMH_VARIANT_MD2) ? S1 / ((double)B * k.H * k.W) : S1 / (S2 + 1e-7);
            hcount[s] = S4;
            hint[s] = S3 / (S4 + 1e-7);       // depth_hint_loss.sum() / (mask.sum() + 1e-7), DH/trainer.py:721; 0 without hints
        }
    }
    // smoothness: one (scale, image) pair per thread, chunks summed in a fixed order
    double part[DMH_MAX_SCALES] = {0, 0, 0, 0};
    for (int p = tid; p < NS * B; p += NTF) {
        const int s = p / B, b = p - s * B;
        const int nc = k.l.nchunk[s];
        const float* q = k.smooth + ((size_t)k.l.blk_base[s] + (size_t)b * nc) * 3;
        double sd = 0.0, rx = 0.0, ry = 0.0;
        for (int c = 0; c < nc; ++c) {
            sd += (double)q[c * 3 + 0];
            rx += (double)q[c * 3 + 1];
            ry += (double)q[c * 3 + 2];
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
            if (j == s) part[j] += term;
    }
#pragma unroll
    for (int s = 0; s < DMH_MAX_SCALES; ++s) {
        if (s < NS) {
            const double S = block_sum_d(part[s], s_red);
            if (tid == 0) s_sm[s] = S;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double total = 0.0;
        for (int i = 0; i < DMH_FIN_SIZE; ++i) k.fin[i] = 0.f;
        for (int s = 0; s < NS; ++s) {
            const double ls = reproj[s] + hint[s] + (double)k.smooth_wt * s_sm[s] / (double)(1 << s);
            total += ls;
            k.fin[DMH_FIN_LOSS_S + s] = (float)ls;
            k.fin[DMH_FIN_REPROJ_S + s] = (float)reproj[s];
            k.fin[DMH_FIN_COUNT_S + s] = (float)count[s];
            k.fin[DMH_FIN_SMOOTH_S + s] = (float)s_sm[s];
            k.fin[DMH_FIN_HINT_S + s] = (float)hint[s];
            k.fin[DMH_FIN_HINTCOUNT_S + s] = (float)hcount[s];
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
    if (!a) return 0;
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
    hipLaunchKernelGGL(smooth_bwd_kernel, dim3(k.l.blk_base[a->num_scales]), dim3(NT), 0, (hipStream_t)stream, k);
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
