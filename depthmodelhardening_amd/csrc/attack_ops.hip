// K4/K5/K6 -- the element-wise and reduction pieces of the attack inner loops (gfx950).
//
// K4  PGD-L_inf projection step        torchattacks/attacks/phy_obj_atk.py:98-101, pgd_depth.py:76-78
// K5  L0 attack: pattern compose + L0 count (phy_obj_atk_l0.py:94-99,43-52,143-150),
//     tanh mask cost fwd/bwd (:130-132)
// K6  masked squared mean = MSELoss(disp*mask, 0)   (phy_obj_atk.py:94, phy_obj_atk_l0.py:127, pgd_depth.py:68)
//
// All are HBM streaming kernels: 16-byte vector accesses where alignment allows, two-stage
// deterministic reductions (block partials + one finalising block), no host synchronisation.
#include "common.hpp"

using namespace dmh;

namespace {

constexpr int NT = 256;
constexpr int MAX_BLOCKS = 1024;  // partial-sum slots for the reductions

__device__ __forceinline__ float sgnf(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__device__ __forceinline__ float pgd1(float x, float x0, float g, float alpha, float eps) {
    const float xn = x + alpha * sgnf(g);
    const float delta = clampf(xn - x0, -eps, eps);
    return clampf(x0 + delta, 0.f, 1.f);
}

__global__ __launch_bounds__(NT) void pgd_step_kernel(const float* __restrict__ x, const float* __restrict__ x0,
                                                      const float* __restrict__ g, float alpha, float eps,
                                                      float* __restrict__ out, int64_t n, int vec_ok) {
    const int64_t stride = (int64_t)gridDim.x * NT;
    int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (vec_ok) {
        const int64_t n4 = n >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* y4 = reinterpret_cast<const float4*>(x0);
        const float4* g4 = reinterpret_cast<const float4*>(g);
        float4* o4 = reinterpret_cast<float4*>(out);
        for (int64_t j = i; j < n4; j += stride) {
            const float4 a = x4[j], b = y4[j], c = g4[j];
            float4 r;
            r.x = pgd1(a.x, b.x, c.x, alpha, eps);
            r.y = pgd1(a.y, b.y, c.y, alpha, eps);
            r.z = pgd1(a.z, b.z, c.z, alpha, eps);
            r.w = pgd1(a.w, b.w, c.w, alpha, eps);
            o4[j] = r;
        }
        for (int64_t j = (n4 << 2) + i; j < n; j += stride) out[j] = pgd1(x[j], x0[j], g[j], alpha, eps);
    } else {
        for (int64_t j = i; j < n; j += stride) out[j] = pgd1(x[j], x0[j], g[j], alpha, eps);
    }
}

// ---------------------------------------------------------------------------------------------- K5
__global__ __launch_bounds__(NT) void l0_compose_fwd_kernel(const float* __restrict__ obj, const float* __restrict__ pos,
                                                            const float* __restrict__ neg, int C, int HW, float clip,
                                                            int finalize, float* __restrict__ adv,
                                                            int32_t* __restrict__ l0_count) {
    const int i = blockIdx.x * NT + threadIdx.x;
    bool nz = false;
    if (i < HW) {
        float acc = 0.f;
        for (int c = 0; c < C; ++c) {
            const int o = c * HW + i;
            float pp = clampf(pos[o], 0.f, 1.f);
            float pn = -clampf(neg[o], 0.f, 1.f);
            const float tp = pp < clip ? 0.f : pp, tn = pn > -clip ? 0.f : pn;  // cal_l0 thresholding
            acc += fabsf(tp + tn);
            if (finalize) {
                pp = tp;
                pn = tn;
            }
            adv[o] = clampf(obj[o] + (pp + pn), 0.f, 1.f);
        }
        nz = acc != 0.f;
    }
    if (l0_count) {
        const unsigned long long bal = __ballot(nz);
        if ((threadIdx.x & (WAVE - 1)) == 0 && bal) atomicAdd(l0_count, (int32_t)__popcll(bal));
    }
}

__global__ __launch_bounds__(NT) void l0_compose_bwd_kernel(const float* __restrict__ obj, const float* __restrict__ pos,
                                                            const float* __restrict__ neg,
                                                            const float* __restrict__ g_adv, int n,
                                                            float* __restrict__ g_pos, float* __restrict__ g_neg,
                                                            int accumulate) {
    const int o = blockIdx.x * NT + threadIdx.x;
    if (o >= n) return;
    const float p = pos[o], q = neg[o];
    const float v = obj[o] + (clampf(p, 0.f, 1.f) - clampf(q, 0.f, 1.f));
    const float g = (v >= 0.f && v <= 1.f) ? g_adv[o] : 0.f;  // clamp passes gradient on the closed interval
    float gp = (p >= 0.f && p <= 1.f) ? g : 0.f;
    float gn = (q >= 0.f && q <= 1.f) ? -g : 0.f;
    if (accumulate) {
        gp += g_pos[o];
        gn += g_neg[o];
    }
    g_pos[o] = gp;
    g_neg[o] = gn;
}

__device__ __forceinline__ float mask_f(float p) { return tanhf(p / 10.f) / (float)(2.0 - 1e-7) + 0.5f; }

__global__ __launch_bounds__(NT) void l0_mask_fwd_kernel(const float* __restrict__ pos, const float* __restrict__ neg,
                                                         int C, int HW, float* __restrict__ partials) {
    __shared__ float s_red[NT / WAVE];
    float ap = 0.f, an = 0.f;
    for (int i = blockIdx.x * NT + threadIdx.x; i < HW; i += gridDim.x * NT) {
        float mp = -3.0e38f, mn = -3.0e38f;
        for (int c = 0; c < C; ++c) {
            mp = fmaxf(mp, mask_f(pos[c * HW + i]));
            mn = fmaxf(mn, mask_f(neg[c * HW + i]));
        }
        ap += mp;
        an += mn;
    }
    const float t0 = block_sum<NT>(ap, s_red);
    const float t1 = block_sum<NT>(an, s_red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 2 + 0] = t0;
        partials[blockIdx.x * 2 + 1] = t1;
    }
}

__global__ __launch_bounds__(NT) void l0_mask_finalize_kernel(const float* __restrict__ partials, int nblk, int HW,
                                                              float* __restrict__ cost) {
    __shared__ float s_red[NT / WAVE];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nblk; i += NT) {
        a += partials[i * 2 + 0];
        b += partials[i * 2 + 1];
    }
    const float t0 = block_sum<NT>(a, s_red);
    const float t1 = block_sum<NT>(b, s_red);
    if (threadIdx.x == 0) cost[0] = t0 / (float)HW + t1 / (float)HW;
}

__global__ __launch_bounds__(NT) void l0_mask_bwd_kernel(const float* __restrict__ pos, const float* __restrict__ neg,
                                                         int C, int HW, const float* __restrict__ gscale,
                                                         const float* __restrict__ weight, float* __restrict__ g_pos,
                                                         float* __restrict__ g_neg, int accumulate) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= HW) return;
    const float up = gscale[0] * (weight ? weight[0] : 1.f) / (float)HW;
    const float* src[2] = {pos, neg};
    float* dst[2] = {g_pos, g_neg};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int best = 0;
        float bv = -3.0e38f;
        for (int c = 0; c < C; ++c) {  // torch.max(dim) routes the gradient to the first maximal channel
            const float v = mask_f(src[k][c * HW + i]);
            if (v > bv) {
                bv = v;
                best = c;
            }
        }
        for (int c = 0; c < C; ++c) {
            float gv = 0.f;
            if (c == best) {
                const float th = tanhf(src[k][c * HW + i] / 10.f);
                gv = up * (1.f - th * th) / 10.f / (float)(2.0 - 1e-7);
            }
            if (accumulate) gv += dst[k][c * HW + i];
            dst[k][c * HW + i] = gv;
        }
    }
}

// ---------------------------------------------------------------------------------------------- K6
__global__ __launch_bounds__(NT) void sq_mean_fwd_kernel(const float* __restrict__ d, const float* __restrict__ m,
                                                         int64_t n, float* __restrict__ partials) {
    __shared__ float s_red[NT / WAVE];
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
        const float v = m ? d[i] * m[i] : d[i];
        acc += v * v;
    }
    const float t = block_sum<NT>(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

__global__ __launch_bounds__(NT) void sq_mean_finalize_kernel(const float* __restrict__ partials, int nblk, int64_t n,
                                                              float* __restrict__ cost) {
    __shared__ double s_red[NT];
    double a = 0.0;
    for (int i = threadIdx.x; i < nblk; i += NT) a += (double)partials[i];
    s_red[threadIdx.x] = a;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) cost[0] = (float)(s_red[0] / (double)n);
}

__global__ __launch_bounds__(NT) void sq_mean_bwd_kernel(const float* __restrict__ d, const float* __restrict__ m,
                                                         int64_t n, const float* __restrict__ gscale,
                                                         float* __restrict__ g) {
    const float up = gscale[0] * 2.f / (float)n;
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
        const float mk = m ? m[i] : 1.f;
        g[i] = up * d[i] * mk * mk;
    }
}

// ---------------------------------------------------------------------------------------------- K6b
// --gt_depth supervised term (MD2/trainer.py:551-557): with depth(d) = clamp(5.4 / (min_disp + (max_disp - min_disp) d), 1e-3, 80)
// (disp_to_depth, layers.py:16-25; 5.4 = the stereo scale factor) the loss is
//     mean_{b,y,x} ( m objdepth_b + depth(disp_gt) (1 - m)  -  depth(disp) )^2 ,  m = color_objmask[:, 0]
// one streaming pass over disp, disp_gt and channel 0 of the mask (batch stride mask_bstride floats); two-stage fixed-order sum.
__device__ __forceinline__ float sup_depth(float d, float min_disp, float range, bool& inside) {
    const float sd = min_disp + range * d;
    const float z = (1.f / sd) * 5.4f;
    inside = z >= 1e-3f && z <= 80.f;       // torch.clamp passes the gradient on the closed interval
    return fminf(fmaxf(z, 1e-3f), 80.f);
}

__global__ __launch_bounds__(NT) void gt_depth_fwd_kernel(const float* __restrict__ disp, const float* __restrict__ disp_gt,
                                                          const float* __restrict__ m, int64_t mask_bstride,
                                                          const float* __restrict__ objdepth, int B, int64_t HW,
                                                          float min_disp, float range, float* __restrict__ partials) {
    __shared__ float s_red[NT / WAVE];
    float acc = 0.f;
    const int64_t n = (int64_t)B * HW;
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
        const int64_t b = i / HW, r = i - b * HW;
        bool in;
        const float pred = sup_depth(disp[i], min_disp, range, in);
        const float pseudo = sup_depth(disp_gt[i], min_disp, range, in);
        const float mk = m[b * mask_bstride + r];
        const float gt = mk * objdepth[b] + pseudo * (1.f - mk);
        const float v = gt - pred;
        acc += v * v;
    }
    const float t = block_sum<NT>(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

__global__ __launch_bounds__(NT) void gt_depth_bwd_kernel(const float* __restrict__ disp, const float* __restrict__ disp_gt,
                                                          const float* __restrict__ m, int64_t mask_bstride,
                                                          const float* __restrict__ objdepth, int B, int64_t HW,
                                                          float min_disp, float range, const float* __restrict__ gscale,
                                                          float* __restrict__ g) {
    const int64_t n = (int64_t)B * HW;
    const float up = gscale[0] * 2.f / (float)n;
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
        const int64_t b = i / HW, r = i - b * HW;
        bool in, in_gt;
        const float d = disp[i];
        const float pred = sup_depth(d, min_disp, range, in);
        const float pseudo = sup_depth(disp_gt[i], min_disp, range, in_gt);
        const float mk = m[b * mask_bstride + r];
        const float gt = mk * objdepth[b] + pseudo * (1.f - mk);
        // d pred / d disp = -5.4 range / sd^2 inside the clamp, 0 outside
        const float sd = min_disp + range * d;
        const float isd = 1.f / sd;
        g[i] = in ? up * (pred - gt) * (-5.4f * range) * isd * isd : 0.f;
    }
}

inline int grid_for(int64_t n) {
    const int64_t b = (n + NT - 1) / NT;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

inline int red_blocks(int64_t n) {
    const int64_t b = (n + (int64_t)NT * 4 - 1) / ((int64_t)NT * 4);
    return (int)(b < 1 ? 1 : (b > MAX_BLOCKS ? MAX_BLOCKS : b));
}

}  // namespace

extern "C" {

int dmh_pgd_linf_step(const float* x, const float* x0, const float* g, float alpha, float eps, float* out,
                      int64_t n, void* stream) {
    DMH_REQUIRE(x && x0 && g && out && n > 0, "null pointer or n <= 0");
    const uintptr_t al = (uintptr_t)x | (uintptr_t)x0 | (uintptr_t)g | (uintptr_t)out;
    const int vec_ok = (al & 15) == 0;
    const int64_t work = vec_ok ? (n + 3) / 4 : n;
    hipLaunchKernelGGL(pgd_step_kernel, dim3(grid_for(work)), dim3(NT), 0, (hipStream_t)stream, x, x0, g, alpha, eps,
                       out, n, vec_ok);
    return check_launch("dmh_pgd_linf_step");
}

int dmh_l0_compose_fwd(const float* obj, const float* pos, const float* neg, int C, int HW, float l0_clip,
                       int finalize, float* adv, int32_t* l0_count, void* stream) {
    DMH_REQUIRE(obj && pos && neg && adv, "null pointer");
    DMH_REQUIRE(C > 0 && HW > 0, "bad sizes");
    hipLaunchKernelGGL(l0_compose_fwd_kernel, dim3((HW + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, obj, pos, neg,
                       C, HW, l0_clip, finalize, adv, l0_count);
    return check_launch("dmh_l0_compose_fwd");
}

int dmh_l0_compose_bwd(const float* obj, const float* pos, const float* neg, const float* g_adv, int C, int HW,
                       float* g_pos, float* g_neg, int accumulate, void* stream) {
    DMH_REQUIRE(obj && pos && neg && g_adv && g_pos && g_neg, "null pointer");
    DMH_REQUIRE(C > 0 && HW > 0, "bad sizes");
    const int n = C * HW;
    hipLaunchKernelGGL(l0_compose_bwd_kernel, dim3((n + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, obj, pos, neg,
                       g_adv, n, g_pos, g_neg, accumulate);
    return check_launch("dmh_l0_compose_bwd");
}

int64_t dmh_l0_mask_partials_size(int HW) { return 2 * (int64_t)red_blocks(HW); }

int dmh_l0_mask_cost_fwd(const float* pos, const float* neg, int C, int HW, float* partials, float* cost,
                         void* stream) {
    DMH_REQUIRE(pos && neg && partials && cost, "null pointer");
    DMH_REQUIRE(C > 0 && HW > 0, "bad sizes");
    const int nb = red_blocks(HW);
    hipLaunchKernelGGL(l0_mask_fwd_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, pos, neg, C, HW, partials);
    hipLaunchKernelGGL(l0_mask_finalize_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, partials, nb, HW, cost);
    return check_launch("dmh_l0_mask_cost_fwd");
}

int dmh_l0_mask_cost_bwd(const float* pos, const float* neg, int C, int HW, const float* gscale,
                         const float* weight, float* g_pos, float* g_neg, int accumulate, void* stream) {
    DMH_REQUIRE(pos && neg && gscale && g_pos && g_neg, "null pointer");
    DMH_REQUIRE(C > 0 && HW > 0, "bad sizes");
    hipLaunchKernelGGL(l0_mask_bwd_kernel, dim3((HW + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, pos, neg, C, HW,
                       gscale, weight, g_pos, g_neg, accumulate);
    return check_launch("dmh_l0_mask_cost_bwd");
}

int64_t dmh_sq_mean_partials_size(int64_t n) { return red_blocks(n); }

int dmh_masked_sq_mean_fwd(const float* disp, const float* mask, int64_t n, float* partials, float* cost,
                           void* stream) {
    DMH_REQUIRE(disp && partials && cost && n > 0, "null pointer or n <= 0");
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(sq_mean_fwd_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, disp, mask, n, partials);
    hipLaunchKernelGGL(sq_mean_finalize_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, partials, nb, n, cost);
    return check_launch("dmh_masked_sq_mean_fwd");
}

int dmh_masked_sq_mean_bwd(const float* disp, const float* mask, int64_t n, const float* gscale, float* g_disp,
                           void* stream) {
    DMH_REQUIRE(disp && gscale && g_disp && n > 0, "null pointer or n <= 0");
    hipLaunchKernelGGL(sq_mean_bwd_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, disp, mask, n, gscale,
                       g_disp);
    return check_launch("dmh_masked_sq_mean_bwd");
}

int dmh_gt_depth_mse_fwd(const float* disp, const float* disp_gt, const float* objmask, int64_t mask_bstride,
                         const float* objdepth, int B, int64_t HW, float min_depth, float max_depth, float* partials,
                         float* cost, void* stream) {
    DMH_REQUIRE(disp && disp_gt && objmask && objdepth && partials && cost, "null pointer");
    DMH_REQUIRE(B > 0 && HW > 0 && mask_bstride >= HW, "bad sizes (mask batch stride below H*W?)");
    DMH_REQUIRE(min_depth > 0.f && max_depth > min_depth, "need 0 < min_depth < max_depth");
    const int64_t n = (int64_t)B * HW;
    const int nb = red_blocks(n);
    const float min_disp = (float)(1.0 / (double)max_depth), range = (float)(1.0 / (double)min_depth - 1.0 / (double)max_depth);
    hipLaunchKernelGGL(gt_depth_fwd_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, disp, disp_gt, objmask, mask_bstride,
                       objdepth, B, HW, min_disp, range, partials);
    hipLaunchKernelGGL(sq_mean_finalize_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, partials, nb, n, cost);
    return check_launch("dmh_gt_depth_mse_fwd");
}

int dmh_gt_depth_mse_bwd(const float* disp, const float* disp_gt, const float* objmask, int64_t mask_bstride,
                         const float* objdepth, int B, int64_t HW, float min_depth, float max_depth, const float* gscale,
                         float* g_disp, void* stream) {
    DMH_REQUIRE(disp && disp_gt && objmask && objdepth && gscale && g_disp, "null pointer");
    DMH_REQUIRE(B > 0 && HW > 0 && mask_bstride >= HW, "bad sizes (mask batch stride below H*W?)");
    DMH_REQUIRE(min_depth > 0.f && max_depth > min_depth, "need 0 < min_depth < max_depth");
    const float min_disp = (float)(1.0 / (double)max_depth), range = (float)(1.0 / (double)min_depth - 1.0 / (double)max_depth);
    hipLaunchKernelGGL(gt_depth_bwd_kernel, dim3(grid_for((int64_t)B * HW)), dim3(NT), 0, (hipStream_t)stream, disp, disp_gt,
                       objmask, mask_bstride, objdepth, B, HW, min_disp, range, gscale, g_disp);
    return check_launch("dmh_gt_depth_mse_bwd");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------- K8
// Masked depth-error metrics of evaluate_attacks (MD2/evaluate_depth.py:57-99,193-197): disparity -> depth
// (disp_to_depth on |disp|, x5.4 stereo scale, clamp to [1e-3, 80]) and the eight sums, in one pass.
namespace {

constexpr int NQ = 9;  // mask, a1, a2, a3, abs_err, sq_err, log_sq_err, abs_rel, sq_rel

__global__ __launch_bounds__(NT) void depth_err_kernel(const float* __restrict__ dgt, const float* __restrict__ dpr,
                                                       const float* __restrict__ mask, int64_t n, float min_disp,
                                                       float dmul, float scale, float lo, float hi,
                                                       float* __restrict__ partials) {
    __shared__ float s_red[NT / WAVE];
    float acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.f;
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
        const float gt = clampf(scale / (min_disp + dmul * fabsf(dgt[i])), lo, hi);
        const float pr = clampf(scale / (min_disp + dmul * fabsf(dpr[i])), lo, hi);
        const float m = mask ? mask[i] : 1.f;
        const float th = fmaxf(gt / pr, pr / gt), d = gt - pr, ld = logf(gt) - logf(pr);
        acc[0] += m;
        acc[1] += th < 1.25f ? m : 0.f;
        acc[2] += th < 1.25f * 1.25f ? m : 0.f;
        acc[3] += th < 1.25f * 1.25f * 1.25f ? m : 0.f;
        acc[4] += fabsf(d) * m;
        acc[5] += d * d * m;
        acc[6] += ld * ld * m;
        acc[7] += fabsf(d) / gt * m;
        acc[8] += d * d / gt * m;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float t = block_sum<NT>(acc[q], s_red);
        if (threadIdx.x == 0) partials[blockIdx.x * NQ + q] = t;
    }
}

__global__ __launch_bounds__(NT) void depth_err_finalize_kernel(const float* __restrict__ partials, int nblk,
                                                                float* __restrict__ out) {
    __shared__ double s_red[NT];
    double tot[NQ];
    for (int q = 0; q < NQ; ++q) {
        double a = 0.0;
        for (int i = threadIdx.x; i < nblk; i += NT) a += (double)partials[i * NQ + q];
        __syncthreads();
        s_red[threadIdx.x] = a;
        __syncthreads();
        for (int o = NT / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
            __syncthreads();
        }
        tot[q] = s_red[0];
    }
    if (threadIdx.x == 0) {
        const double t = tot[0];
        out[0] = (float)(tot[4] / t);        // abs_err
        out[1] = (float)(tot[7] / t);        // abs_rel
        out[2] = (float)(tot[8] / t);        // sq_rel
        out[3] = (float)sqrt(tot[5] / t);    // rmse
        out[4] = (float)sqrt(tot[6] / t);    // rmse_log
        out[5] = (float)(tot[1] / t);        // a1
        out[6] = (float)(tot[2] / t);        // a2
        out[7] = (float)(tot[3] / t);        // a3
    }
}

}  // namespace

extern "C" {

int64_t dmh_depth_errors_partials_size(int64_t n) { return (int64_t)red_blocks(n) * NQ; }

int dmh_masked_depth_errors(const float* disp_gt, const float* disp_pred, const float* mask, int64_t n,
                            float min_depth, float max_depth, float scale, float clamp_lo, float clamp_hi,
                            float* partials, float* out8, void* stream) {
    DMH_REQUIRE(disp_gt && disp_pred && partials && out8 && n > 0, "null pointer or n <= 0");
    DMH_REQUIRE(min_depth > 0.f && max_depth > min_depth && clamp_lo > 0.f && clamp_hi > clamp_lo, "bad ranges");
    const double mn = 1.0 / (double)max_depth, mx = 1.0 / (double)min_depth;
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(depth_err_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, disp_gt, disp_pred, mask, n,
                       (float)mn, (float)(mx - mn), scale, clamp_lo, clamp_hi, partials);
    hipLaunchKernelGGL(depth_err_finalize_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, partials, nb, out8);
    return check_launch("dmh_masked_depth_errors");
}

}  // extern "C"
