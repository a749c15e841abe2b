// Encoder glue (K9) -- the element-wise passes between the MIOpen convolutions of the ResNet encoder while the
// model is in eval() mode, i.e. inside every PGD / L0 attack step (10 of the 11 forward+backward passes of one
// adversarial-training step; the attack bracket torchattacks/attack.py:165-182 switches the model to eval).
//
// Reference: MD2/networks/resnet_encoder.py:85-98 runs torchvision's ResNet: conv -> BatchNorm2d -> ReLU
// (-> MaxPool2d 3x3/2 in the stem) and, per BasicBlock, conv -> bn -> relu -> conv -> bn -> (+identity) -> relu, each
// as its own full-tensor pass (BN-inference, clamp, add, max-pool + their four backward kernels).  With running
// statistics BatchNorm is the per-channel affine  x * scale[c] + shift[c]
// (scale = weight / sqrt(running_var + eps), shift = bias - running_mean * scale), so each group is ONE pass:
//
//   bn_act        : out = act( x * scale[c] + shift[c] (+ residual) )               act = ReLU or identity
//   stem          : feat = ReLU(x * scale[c] + shift[c]);  pooled = maxpool3x3/2(feat);  argmax (0..8) kept as u8
//
// Backward kernels are gathers (the max-pool adjoint looks its <= 4 covering windows up through the stored argmax):
// deterministic, no atomics.  Pure HBM streaming, 16-byte accesses where the row length allows.
#include "common.hpp"

#include <cmath>

using namespace dmh;

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float relu_f(float v) { return v > 0.f ? v : 0.f; }

template <bool RELU, bool RES>
__global__ __launch_bounds__(NT) void bn_act_fwd_vec(const float4* __restrict__ x, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const float4* __restrict__ res,
                                                     unsigned C, unsigned hw4, unsigned total4, float4* __restrict__ out) {
    const unsigned i = blockIdx.x * NT + threadIdx.x;
    if (i >= total4) return;
    const unsigned c = (i / hw4) % C;
    const float s = scale[c], b = shift[c];
    const float4 v = x[i];
    float4 r = make_float4(fmaf(v.x, s, b), fmaf(v.y, s, b), fmaf(v.z, s, b), fmaf(v.w, s, b));
    if (RES) {
        const float4 q = res[i];
        r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
    }
    if (RELU) r = make_float4(relu_f(r.x), relu_f(r.y), relu_f(r.z), relu_f(r.w));
    out[i] = r;
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(NT) void bn_act_fwd_scalar(const float* __restrict__ x, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ res,
                                                        unsigned C, unsigned hw, unsigned total, float* __restrict__ out) {
    const unsigned i = blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    const unsigned c = (i / hw) % C;
    float r = fmaf(x[i], scale[c], shift[c]);
    if (RES) r += res[i];
    out[i] = RELU ? relu_f(r) : r;
}

// g_res = RELU ? g * [out > 0] : g ;   g_x = g_res * scale[c]
template <bool RELU, bool RES>
__global__ __launch_bounds__(NT) void bn_act_bwd_vec(const float4* __restrict__ out, const float4* __restrict__ g,
                                                     const float* __restrict__ scale, unsigned C, unsigned hw4,
                                                     unsigned total4, float4* __restrict__ g_x, float4* __restrict__ g_res) {
    const unsigned i = blockIdx.x * NT + threadIdx.x;
    if (i >= total4) return;
    const float s = scale[(i / hw4) % C];
    float4 gv = g[i];
    if (RELU) {
        const float4 o = out[i];
        gv = make_float4(o.x > 0.f ? gv.x : 0.f, o.y > 0.f ? gv.y : 0.f, o.z > 0.f ? gv.z : 0.f, o.w > 0.f ? gv.w : 0.f);
    }
    if (RES) g_res[i] = gv;
    g_x[i] = make_float4(gv.x * s, gv.y * s, gv.z * s, gv.w * s);
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(NT) void bn_act_bwd_scalar(const float* __restrict__ out, const float* __restrict__ g,
                                                        const float* __restrict__ scale, unsigned C, unsigned hw,
                                                        unsigned total, float* __restrict__ g_x, float* __restrict__ g_res) {
    const unsigned i = blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    float gv = g[i];
    if (RELU) gv = out[i] > 0.f ? gv : 0.f;
    if (RES) g_res[i] = gv;
    g_x[i] = gv * scale[(i / hw) % C];
}

// Stem.  One thread per pooled output (i, j): it owns the 2x2 quad rows {2i, 2i+1} x cols {2j, 2j+1} of the
// activation and its 3x3 window rows 2i-1..2i+1, cols 2j-1..2j+1 (H, W even, so only the top/left edges clip).
// The argmax follows at::native max_pool_forward_nchw: scan ky, kx ascending, replace on strictly greater.
__global__ __launch_bounds__(NT) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int C, int H, int W,
                                                      float* __restrict__ feat, float* __restrict__ pooled,
                                                      unsigned char* __restrict__ argmax) {
    const int PH = H >> 1, PW = W >> 1;
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= PH * PW) return;
    const int plane = blockIdx.y, c = plane % C;
    const int i = t / PW, j = t - i * PW;
    const float s = scale[c], b = shift[c];
    const float* xp = x + (size_t)plane * H * W;
    float* fp = feat + (size_t)plane * H * W;
    float m = -INFINITY;
    int arg = 4;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = 2 * i - 1 + ky;
        if (yy < 0) continue;
        const float* row = xp + (size_t)yy * W + 2 * j;
        const float2 vc = *reinterpret_cast<const float2*>(row);
        const float a1 = relu_f(fmaf(vc.x, s, b)), a2 = relu_f(fmaf(vc.y, s, b));
        if (j > 0) {
            const float a0 = relu_f(fmaf(row[-1], s, b));
            if (a0 > m) { m = a0; arg = ky * 3; }
        }
        if (a1 > m) { m = a1; arg = ky * 3 + 1; }
        if (a2 > m) { m = a2; arg = ky * 3 + 2; }
        if (ky >= 1) *reinterpret_cast<float2*>(fp + (size_t)yy * W + 2 * j) = make_float2(a1, a2);
    }
    const size_t o = (size_t)plane * PH * PW + t;
    pooled[o] = m;
    argmax[o] = (unsigned char)arg;
}

// g_x = scale[c] * [feat > 0] * ( g_feat + sum over covering windows whose argmax is this pixel of g_pool )
__global__ __launch_bounds__(NT) void stem_bwd_kernel(const float* __restrict__ feat, const unsigned char* __restrict__ argmax,
                                                      const float* __restrict__ g_feat, const float* __restrict__ g_pool,
                                                      const float* __restrict__ scale, int C, int H, int W,
                                                      float* __restrict__ g_x) {
    const int PH = H >> 1, PW = W >> 1;
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= PH * PW) return;
    const int plane = blockIdx.y, c = plane % C;
    const int i = t / PW, j = t - i * PW;
    const size_t base = (size_t)plane * H * W, pbase = (size_t)plane * PH * PW;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (g_pool) {
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
                const int ry = 2 * di - 1 + ky, rx = 2 * dj - 1 + kx;   // position relative to the quad origin
                if (ry >= 0 && ry < 2 && rx >= 0 && rx < 2) {
                    const float gp = g_pool[pbase + (size_t)oi * PW + oj];
                    // compile-time indices keep acc in registers
                    if (ry == 0 && rx == 0) acc[0][0] += gp;
                    if (ry == 0 && rx == 1) acc[0][1] += gp;
                    if (ry == 1 && rx == 0) acc[1][0] += gp;
                    if (ry == 1 && rx == 1) acc[1][1] += gp;
                }
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
        *reinterpret_cast<float2*>(g_x + o) = make_float2(f.x > 0.f ? g0 * s : 0.f, f.y > 0.f ? g1 * s : 0.f);
    }
}

// The same with two horizontally adjacent quads per thread (W a multiple of 4): 16-byte accesses to feat / g_feat / g_x,
// six argmax bytes instead of eight.
__global__ __launch_bounds__(NT) void stem_bwd4_kernel(const float* __restrict__ feat, const unsigned char* __restrict__ argmax,
                                                       const float* __restrict__ g_feat, const float* __restrict__ g_pool,
                                                       const float* __restrict__ scale, int C, int H, int W,
                                                       float* __restrict__ g_x) {
    const int PH = H >> 1, PW = W >> 1, PW2 = PW >> 1;
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= PH * PW2) return;
    const int plane = blockIdx.y, c = plane % C;
    const int i = t / PW2, j = (t - i * PW2) * 2;            // quads (i, j) and (i, j + 1)
    const size_t base = (size_t)plane * H * W, pbase = (size_t)plane * PH * PW;
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // rows 2i, 2i+1; columns 2j .. 2j+3
    if (g_pool) {
#pragma unroll
        for (int di = 0; di < 2; ++di) {
            const int oi = i + di;
            if (oi >= PH) continue;
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {                 // pooling windows j, j + 1, j + 2 touch the four columns
                const int oj = j + dj;
                if (oj >= PW) continue;
                const int a = argmax[pbase + (size_t)oi * PW + oj];
                const int ky = a / 3, kx = a - ky * 3;
                const int ry = 2 * di - 1 + ky, rx = 2 * dj - 1 + kx;   // position relative to the thread's 2 x 4 block
                if (ry >= 0 && ry < 2 && rx >= 0 && rx < 4) {
                    const float gp = g_pool[pbase + (size_t)oi * PW + oj];
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (ry == r && rx == q) acc[r][q] += gp;         // compile-time indices keep acc in registers
                }
            }
        }
    }
    const float s = scale[c];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const size_t o = base + (size_t)(2 * i + r) * W + 2 * j;
        const float4 f = *reinterpret_cast<const float4*>(feat + o);
        float g0 = acc[r][0], g1 = acc[r][1], g2 = acc[r][2], g3 = acc[r][3];
        if (g_feat) {
            const float4 gf = *reinterpret_cast<const float4*>(g_feat + o);
            g0 += gf.x;
            g1 += gf.y;
            g2 += gf.z;
            g3 += gf.w;
        }
        *reinterpret_cast<float4*>(g_x + o) = make_float4(f.x > 0.f ? g0 * s : 0.f, f.y > 0.f ? g1 * s : 0.f,
                                                          f.z > 0.f ? g2 * s : 0.f, f.w > 0.f ? g3 * s : 0.f);
    }
}

// ---- train-mode BatchNorm statistics (torchvision BasicBlock / stem under model.train(), MD2/trainer.py:335-375):
// per-channel mean and biased variance over (B, H, W) in two launches, Welford / Chan in fp32.
//   partial : grid (S, C); block (s, c) reduces every S-th 1024-element slab of channel c -> (count, mean, M2)
//   finalize: one thread per channel combines the S partials, writes the affine of the normalisation
//             scale = weight * invstd, shift = bias - mean * scale, the saved mean / invstd for the backward, and
//             updates the running statistics in place (unbiased variance, torch.nn.BatchNorm2d semantics).
// slabs (4 * NT elements) of one HW-element plane that fall to block s of S: s, s + S, s + 2 S, ...
__device__ __forceinline__ int slabs_of(int HW, int s, int S) {
    const int nslab = (HW + 4 * NT - 1) / (4 * NT);
    return s < nslab ? (nslab - 1 - s) / S + 1 : 0;
}

struct Wf {
    float n, mean, m2;
};
__device__ __forceinline__ Wf wf_merge(Wf a, Wf b) {
    if (b.n == 0.f) return a;
    if (a.n == 0.f) return b;
    const float n = a.n + b.n, d = b.mean - a.mean, f = b.n / n;
    return Wf{n, a.mean + d * f, a.m2 + b.m2 + d * d * a.n * f};
}

__global__ __launch_bounds__(NT) void bn_stats_partial_kernel(const float* __restrict__ x, int B, int C, int HW, int S,
                                                              float* __restrict__ part) {
    const int c = blockIdx.y, s = blockIdx.x;
    // per thread: shifted sums  s1 = sum(v - k), s2 = sum((v - k)^2)  with k = the channel's first element (no division
    // per element, no cancellation); slabs of 4*NT elements of one (b, c) plane, S slabs apart
    const float k = x[(size_t)c * HW];
    float nq[4] = {0.f, 0.f, 0.f, 0.f}, s1q[4] = {0.f, 0.f, 0.f, 0.f}, s2q[4] = {0.f, 0.f, 0.f, 0.f};   // 4 independent chains
    for (int b = 0; b < B; ++b) {
        const float* xp = x + ((size_t)b * C + c) * HW;
        if ((HW & 3) == 0) continue;    // whole 16-byte words: the flattened loop below
        for (int r0 = s * 4 * NT; r0 < HW; r0 += S * 4 * NT) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + q * NT + threadIdx.x;
                v[q] = r < HW ? xp[r] : k;
                nq[q] += r < HW ? 1.f : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float d = v[q] - k;           // padding lanes contribute d = 0
                s1q[q] += d;
                s2q[q] = fmaf(d, d, s2q[q]);
            }
        }
    }
    if ((HW & 3) == 0) {
        // one float4 per thread and slab; the (image, slab) pairs of the block as ONE loop, four loads in flight per thread
        // (an image holds only one or two slabs per block: image after image, every load waited for the one before)
        const int nk = slabs_of(HW, s, S), total = B * nk;
        for (int i0 = 0; i0 < total; i0 += 4) {
            float4 v4[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u, total - 1), b = i / nk, r = (s + (i - b * nk) * S) * 4 * NT + 4 * (int)threadIdx.x;
                ok[u] = i0 + u < total && r < HW;
                v4[u] = ok[u] ? *reinterpret_cast<const float4*>(x + ((size_t)b * C + c) * HW + r) : make_float4(k, k, k, k);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float d = v[q] - k;           // skipped loads contribute d = 0
                    nq[q] += ok[u] ? 1.f : 0.f;
                    s1q[q] += d;
                    s2q[q] = fmaf(d, d, s2q[q]);
                }
            }
        }
    }
    const float n = (nq[0] + nq[1]) + (nq[2] + nq[3]), s1 = (s1q[0] + s1q[1]) + (s1q[2] + s1q[3]),
                s2 = (s2q[0] + s2q[1]) + (s2q[2] + s2q[3]);
    Wf w{n, n > 0.f ? k + s1 / n : 0.f, n > 0.f ? s2 - s1 * s1 / n : 0.f};
    // wave, then block reduction (fixed order: reproducible)
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) {
        Wf other{__shfl_down(w.n, o, WAVE), __shfl_down(w.mean, o, WAVE), __shfl_down(w.m2, o, WAVE)};
        w = wf_merge(w, other);
    }
    __shared__ Wf red[NT / WAVE];
    if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x / WAVE] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        Wf t = red[0];
#pragma unroll
        for (int i = 1; i < NT / WAVE; ++i) t = wf_merge(t, red[i]);
        float* p = part + ((size_t)c * S + s) * 3;
        p[0] = t.n; p[1] = t.mean; p[2] = t.m2;
    }
}

__global__ __launch_bounds__(NT) void bn_stats_finalize_kernel(const float* __restrict__ part, int C, int S,
                                                               const float* __restrict__ weight,
                                                               const float* __restrict__ bias, float momentum, float eps,
                                                               float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, float* __restrict__ scale,
                                                               float* __restrict__ shift, float* __restrict__ save_mean,
                                                               float* __restrict__ save_invstd,
                                                               long long* __restrict__ num_batches_tracked) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= C) return;
    if (c == 0 && num_batches_tracked) num_batches_tracked[0] += 1;     // nn.BatchNorm2d.forward's counter: one thread, one launch
    Wf t{0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        const float* p = part + ((size_t)c * S + s) * 3;
        t = wf_merge(t, Wf{p[0], p[1], p[2]});
    }
    const float var = t.m2 / t.n, invstd = rsqrtf(var + eps);
    const float sc = (weight ? weight[c] : 1.f) * invstd;
    scale[c] = sc;
    shift[c] = (bias ? bias[c] : 0.f) - t.mean * sc;
    save_mean[c] = t.mean;
    save_invstd[c] = invstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * t.mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (t.m2 / fmaxf(t.n - 1.f, 1.f));
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + NT - 1) / NT); }

// ---- train-mode BatchNorm backward with the ReLU mask folded in (round 3; MD2/trainer.py:335-375 train pass through
// torchvision's BasicBlocks).  With g' = g * [out > 0] (or g), xc = x - mean, N = B * HW:
//     db = sum g',   dw = invstd * sum g' xc,   dx = w invstd ( g' - db / N - xc * invstd^2 * (sum g' xc) / N )
// in three launches: per-(slab, channel) partial sums (x, g, out read once), one thread per channel combining them in a fixed
// order (double) into db, dw and the three coefficients of dx, and one element-wise pass (x, g, out read again, dx and --
// for a residual branch -- g' written).  Before: one K9 mask pass + MIOpen's two-pass kernel (8 tensor passes, now 7 / 8
// without / with a residual branch, and no atomics).
template <bool RELU, bool VEC>
__global__ __launch_bounds__(NT) void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ out, const float* __restrict__ mean,
                                                            int B, int C, int HW, int S, float* __restrict__ part) {
    const int c = blockIdx.y, s = blockIdx.x;
    const float m = mean[c];
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};     // 4 independent chains
    for (int b = 0; b < B; ++b) {
        const size_t base = ((size_t)b * C + c) * HW;
        if (VEC) {      // HW % 4 == 0: the flattened loop below
            break;
        } else {
            for (int r0 = s * 4 * NT; r0 < HW; r0 += S * 4 * NT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = r0 + q * NT + (int)threadIdx.x;
                    if (r < HW) {
                        float gv = g[base + r];
                        if (RELU) gv = out[base + r] > 0.f ? gv : 0.f;
                        a1[q] += gv;
                        a2[q] = fmaf(gv, x[base + r] - m, a2[q]);
                    }
                }
            }
        }
    }
    if (VEC) {
        // one float4 per tensor, thread and slab; (image, slab) pairs as one loop, two pairs (4-6 loads) in flight per thread
        const int nk = slabs_of(HW, s, S), total = B * nk;
        for (int i0 = 0; i0 < total; i0 += 2) {
            float4 xv[2], gv[2], ov[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = min(i0 + u, total - 1), b = i / nk, r = (s + (i - b * nk) * S) * 4 * NT + 4 * (int)threadIdx.x;
                const bool ok = i0 + u < total && r < HW;
                const size_t o = ((size_t)b * C + c) * HW + r;
                xv[u] = ok ? *reinterpret_cast<const float4*>(x + o) : make_float4(m, m, m, m);
                gv[u] = ok ? *reinterpret_cast<const float4*>(g + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (RELU) ov[u] = ok ? *reinterpret_cast<const float4*>(out + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float4 gq = gv[u];
                if (RELU)
                    gq = make_float4(ov[u].x > 0.f ? gq.x : 0.f, ov[u].y > 0.f ? gq.y : 0.f, ov[u].z > 0.f ? gq.z : 0.f,
                                     ov[u].w > 0.f ? gq.w : 0.f);
                a1[0] += gq.x; a1[1] += gq.y; a1[2] += gq.z; a1[3] += gq.w;
                a2[0] = fmaf(gq.x, xv[u].x - m, a2[0]); a2[1] = fmaf(gq.y, xv[u].y - m, a2[1]);
                a2[2] = fmaf(gq.z, xv[u].z - m, a2[2]); a2[3] = fmaf(gq.w, xv[u].w - m, a2[3]);
            }
        }
    }
    float s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]), s2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
    // wave, then block reduction (fixed order: reproducible)
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) {
        s1 += __shfl_down(s1, o, WAVE);
        s2 += __shfl_down(s2, o, WAVE);
    }
    __shared__ float red[NT / WAVE][2];
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        red[threadIdx.x / WAVE][0] = s1;
        red[threadIdx.x / WAVE][1] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t1 = red[0][0], t2 = red[0][1];
#pragma unroll
        for (int i = 1; i < NT / WAVE; ++i) {
            t1 += red[i][0];
            t2 += red[i][1];
        }
        part[((size_t)c * S + s) * 2 + 0] = t1;
        part[((size_t)c * S + s) * 2 + 1] = t2;
    }
}

// coef[c] = { w invstd, db / N, invstd^2 sum(g' xc) / N, mean }
__global__ __launch_bounds__(NT) void bn_bwd_finalize_kernel(const float* __restrict__ part, int C, int S, double inv_n,
                                                             const float* __restrict__ weight, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, float* __restrict__ coef,
                                                             float* __restrict__ g_weight, float* __restrict__ g_bias) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < S; ++s) {
        s1 += (double)part[((size_t)c * S + s) * 2 + 0];
        s2 += (double)part[((size_t)c * S + s) * 2 + 1];
    }
    const double is = (double)invstd[c], w = weight ? (double)weight[c] : 1.0;
    if (g_bias) g_bias[c] = (float)s1;
    if (g_weight) g_weight[c] = (float)(s2 * is);
    float4 k;
    k.x = (float)(w * is);
    k.y = (float)(s1 * inv_n);
    k.z = (float)(s2 * is * is * inv_n);
    k.w = mean[c];
    reinterpret_cast<float4*>(coef)[c] = k;
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(NT) void bn_bwd_apply_vec(const float4* __restrict__ x, const float4* __restrict__ g,
                                                       const float4* __restrict__ out, const float4* __restrict__ coef,
                                                       unsigned C, unsigned hw4, unsigned total4, float4* __restrict__ g_x,
                                                       float4* __restrict__ g_res) {
    const unsigned i = blockIdx.x * NT + threadIdx.x;
    if (i >= total4) return;
    const float4 k = coef[(i / hw4) % C];
    const float4 xv = x[i];
    float4 gv = g[i];
    if (RELU) {
        const float4 o = out[i];
        gv = make_float4(o.x > 0.f ? gv.x : 0.f, o.y > 0.f ? gv.y : 0.f, o.z > 0.f ? gv.z : 0.f, o.w > 0.f ? gv.w : 0.f);
    }
    if (RES) g_res[i] = gv;
    g_x[i] = make_float4(k.x * ((gv.x - k.y) - (xv.x - k.w) * k.z), k.x * ((gv.y - k.y) - (xv.y - k.w) * k.z),
                         k.x * ((gv.z - k.y) - (xv.z - k.w) * k.z), k.x * ((gv.w - k.y) - (xv.w - k.w) * k.z));
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(NT) void bn_bwd_apply_scalar(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ out, const float4* __restrict__ coef,
                                                          unsigned C, unsigned hw, unsigned total, float* __restrict__ g_x,
                                                          float* __restrict__ g_res) {
    const unsigned i = blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    const float4 k = coef[(i / hw) % C];
    float gv = g[i];
    if (RELU) gv = out[i] > 0.f ? gv : 0.f;
    if (RES) g_res[i] = gv;
    g_x[i] = k.x * ((gv - k.y) - (x[i] - k.w) * k.z);
}

template <bool RELU, bool RES>
void launch_bn_bwd_apply(const float* x, const float* g, const float* out, const float* coef, int C, int HW, int64_t total,
                         float* g_x, float* g_res, hipStream_t st) {
    if ((HW & 3) == 0)
        hipLaunchKernelGGL((bn_bwd_apply_vec<RELU, RES>), dim3(blocks_for(total / 4)), dim3(NT), 0, st,
                           reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(g),
                           reinterpret_cast<const float4*>(out), reinterpret_cast<const float4*>(coef), (unsigned)C,
                           (unsigned)(HW / 4), (unsigned)(total / 4), reinterpret_cast<float4*>(g_x),
                           reinterpret_cast<float4*>(g_res));
    else
        hipLaunchKernelGGL((bn_bwd_apply_scalar<RELU, RES>), dim3(blocks_for(total)), dim3(NT), 0, st, x, g, out,
                           reinterpret_cast<const float4*>(coef), (unsigned)C, (unsigned)HW, (unsigned)total, g_x, g_res);
}

// ---- per-channel sum of a [B, C, HW] tensor: the bias gradient of a convolution (aten::sum over (0, 2, 3) takes 86 us on
// the decoder's maps; this is one read at streaming rate).  Partial sums per (slab, channel), fixed-order finish in double.
__global__ __launch_bounds__(NT) void channel_sum_partial_kernel(const float* __restrict__ g, int B, int C, int HW, int S,
                                                                 float* __restrict__ part) {
    const int c = blockIdx.y, s = blockIdx.x;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if ((HW & 3) == 0) {    // (image, slab) pairs as one loop, four 16-byte loads in flight per thread
        const int nk = slabs_of(HW, s, S), total = B * nk;
        for (int i0 = 0; i0 < total; i0 += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u, total - 1), b = i / nk, r = (s + (i - b * nk) * S) * 4 * NT + 4 * (int)threadIdx.x;
                v[u] = (i0 + u < total && r < HW) ? *reinterpret_cast<const float4*>(g + ((size_t)b * C + c) * HW + r)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[0] += v[u].x; a[1] += v[u].y; a[2] += v[u].z; a[3] += v[u].w; }
        }
    } else {
        for (int b = 0; b < B; ++b) {
            const float* gp = g + ((size_t)b * C + c) * HW;
            for (int r = s * NT + (int)threadIdx.x; r < HW; r += S * NT) a[0] += gp[r];
        }
    }
    float t = (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) t += __shfl_down(t, o, WAVE);
    __shared__ float red[NT / WAVE];
    if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x / WAVE] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        float u = red[0];
#pragma unroll
        for (int i = 1; i < NT / WAVE; ++i) u += red[i];
        part[(size_t)c * S + s] = u;
    }
}

__global__ __launch_bounds__(NT) void channel_sum_finalize_kernel(const float* __restrict__ part, int C, int S,
                                                                  float* __restrict__ out) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= C) return;
    double t = 0.0;
    for (int s = 0; s < S; ++s) t += (double)part[(size_t)c * S + s];
    out[c] = (float)t;
}

template <bool RELU, bool RES>
void launch_fwd(const float* x, const float* scale, const float* shift, const float* res, int C, int HW, int64_t total,
                float* out, hipStream_t st) {
    if ((HW & 3) == 0)
        hipLaunchKernelGGL((bn_act_fwd_vec<RELU, RES>), dim3(blocks_for(total / 4)), dim3(NT), 0, st,
                           reinterpret_cast<const float4*>(x), scale, shift, reinterpret_cast<const float4*>(res),
                           (unsigned)C, (unsigned)(HW / 4), (unsigned)(total / 4), reinterpret_cast<float4*>(out));
    else
        hipLaunchKernelGGL((bn_act_fwd_scalar<RELU, RES>), dim3(blocks_for(total)), dim3(NT), 0, st, x, scale, shift, res,
                           (unsigned)C, (unsigned)HW, (unsigned)total, out);
}

template <bool RELU, bool RES>
void launch_bwd(const float* out, const float* g, const float* scale, int C, int HW, int64_t total, float* g_x,
                float* g_res, hipStream_t st) {
    if ((HW & 3) == 0)
        hipLaunchKernelGGL((bn_act_bwd_vec<RELU, RES>), dim3(blocks_for(total / 4)), dim3(NT), 0, st,
                           reinterpret_cast<const float4*>(out), reinterpret_cast<const float4*>(g), scale, (unsigned)C,
                           (unsigned)(HW / 4), (unsigned)(total / 4), reinterpret_cast<float4*>(g_x),
                           reinterpret_cast<float4*>(g_res));
    else
        hipLaunchKernelGGL((bn_act_bwd_scalar<RELU, RES>), dim3(blocks_for(total)), dim3(NT), 0, st, out, g, scale,
                           (unsigned)C, (unsigned)HW, (unsigned)total, g_x, g_res);
}

}  // namespace

extern "C" {

int dmh_bn_act_fwd(const float* x, const float* scale, const float* shift, const float* residual, int B, int C, int HW,
                   int relu, float* out, void* stream) {
    DMH_REQUIRE(x && scale && shift && out, "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && HW > 0, "bad sizes");
    const int64_t total = (int64_t)B * C * HW;
    DMH_REQUIRE(total < ((int64_t)1 << 31), "tensor too large");
    hipStream_t st = (hipStream_t)stream;
    if (relu) {
        if (residual) launch_fwd<true, true>(x, scale, shift, residual, C, HW, total, out, st);
        else launch_fwd<true, false>(x, scale, shift, nullptr, C, HW, total, out, st);
    } else {
        if (residual) launch_fwd<false, true>(x, scale, shift, residual, C, HW, total, out, st);
        else launch_fwd<false, false>(x, scale, shift, nullptr, C, HW, total, out, st);
    }
    return check_launch("dmh_bn_act_fwd");
}

int dmh_bn_act_bwd(const float* out, const float* g_out, const float* scale, int B, int C, int HW, int relu, float* g_x,
                   float* g_residual, void* stream) {
    DMH_REQUIRE(g_out && scale && g_x && (out || !relu), "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && HW > 0, "bad sizes");
    const int64_t total = (int64_t)B * C * HW;
    DMH_REQUIRE(total < ((int64_t)1 << 31), "tensor too large");
    hipStream_t st = (hipStream_t)stream;
    if (relu) {
        if (g_residual) launch_bwd<true, true>(out, g_out, scale, C, HW, total, g_x, g_residual, st);
        else launch_bwd<true, false>(out, g_out, scale, C, HW, total, g_x, nullptr, st);
    } else {
        if (g_residual) launch_bwd<false, true>(out, g_out, scale, C, HW, total, g_x, g_residual, st);
        else launch_bwd<false, false>(out, g_out, scale, C, HW, total, g_x, nullptr, st);
    }
    return check_launch("dmh_bn_act_bwd");
}

int64_t dmh_bn_stats_partials_size(int B, int C, int HW) {
    if (B <= 0 || C <= 0 || HW <= 0) return -1;
    int64_t S = ((int64_t)HW + 4 * NT - 1) / (4 * NT);
    if (S > 64) S = 64;
    return (int64_t)C * S * 3;
}

int dmh_bn_train_stats(const float* x, int B, int C, int HW, const float* weight, const float* bias, float momentum,
                       float eps, float* running_mean, float* running_var, float* partials, float* scale, float* shift,
                       float* save_mean, float* save_invstd, void* stream) {
    return dmh_bn_train_stats_tracked(x, B, C, HW, weight, bias, momentum, eps, running_mean, running_var, nullptr, partials,
                                      scale, shift, save_mean, save_invstd, stream);
}

int dmh_bn_train_stats_tracked(const float* x, int B, int C, int HW, const float* weight, const float* bias, float momentum,
                               float eps, float* running_mean, float* running_var, long long* num_batches_tracked,
                               float* partials, float* scale, float* shift, float* save_mean, float* save_invstd,
                               void* stream) {
    DMH_REQUIRE(x && partials && scale && shift && save_mean && save_invstd, "null pointer");
    DMH_REQUIRE((reinterpret_cast<uintptr_t>(num_batches_tracked) & 7) == 0, "num_batches_tracked must be an aligned int64");
    DMH_REQUIRE(B > 0 && C > 0 && HW > 0 && C <= 65535, "bad sizes");
    int64_t S = ((int64_t)HW + 4 * NT - 1) / (4 * NT);     // slabs per plane
    if (S > 64) S = 64;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3((unsigned)S, C), dim3(NT), 0, st, x, B, C, HW, (int)S, partials);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(blocks_for(C)), dim3(NT), 0, st, partials, C, (int)S, weight, bias,
                       momentum, eps, running_mean, running_var, scale, shift, save_mean, save_invstd, num_batches_tracked);
    return check_launch("dmh_bn_train_stats");
}

int64_t dmh_channel_sum_partials_size(int B, int C, int HW) {
    const int64_t S = dmh_bn_stats_partials_size(B, C, HW);
    return S < 0 ? -1 : S / 3;
}

int dmh_channel_sum(const float* g, int B, int C, int HW, float* partials, float* out, void* stream) {
    DMH_REQUIRE(g && partials && out, "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && HW > 0 && C <= 65535, "bad sizes");
    int64_t S = ((int64_t)HW + 4 * NT - 1) / (4 * NT);
    if (S > 64) S = 64;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(channel_sum_partial_kernel, dim3((unsigned)S, C), dim3(NT), 0, st, g, B, C, HW, (int)S, partials);
    hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(blocks_for(C)), dim3(NT), 0, st, partials, C, (int)S, out);
    return check_launch("dmh_channel_sum");
}

int64_t dmh_bn_train_bwd_workspace_size(int B, int C, int HW) {
    const int64_t S = dmh_bn_stats_partials_size(B, C, HW);
    if (S < 0) return -1;
    return S / 3 * 2 + (int64_t)C * 4;      // [C][S][2] partial sums + [C] float4 coefficients
}

int dmh_bn_train_bwd(const float* x, const float* g_out, const float* out, const float* weight, const float* save_mean,
                     const float* save_invstd, int B, int C, int HW, float* workspace, float* g_x, float* g_weight,
                     float* g_bias, float* g_pre, void* stream) {
    DMH_REQUIRE(x && g_out && save_mean && save_invstd && workspace && g_x, "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && HW > 0 && C <= 65535, "bad sizes");
    const int64_t total = (int64_t)B * C * HW;
    DMH_REQUIRE(total < ((int64_t)1 << 31), "tensor too large");
    DMH_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "workspace must be 16-byte aligned");
    int64_t S = ((int64_t)HW + 4 * NT - 1) / (4 * NT);     // slabs per plane, as for the forward statistics
    if (S > 64) S = 64;
    hipStream_t st = (hipStream_t)stream;
    float* coef = workspace;                                // [C] float4 first: keeps its alignment
    float* part = workspace + (size_t)C * 4;
    const dim3 pg((unsigned)S, C);
    const bool vec = (HW & 3) == 0;
    if (out) {
        if (vec) hipLaunchKernelGGL((bn_bwd_partial_kernel<true, true>), pg, dim3(NT), 0, st, x, g_out, out, save_mean, B, C, HW, (int)S, part);
        else hipLaunchKernelGGL((bn_bwd_partial_kernel<true, false>), pg, dim3(NT), 0, st, x, g_out, out, save_mean, B, C, HW, (int)S, part);
    } else {
        if (vec) hipLaunchKernelGGL((bn_bwd_partial_kernel<false, true>), pg, dim3(NT), 0, st, x, g_out, out, save_mean, B, C, HW, (int)S, part);
        else hipLaunchKernelGGL((bn_bwd_partial_kernel<false, false>), pg, dim3(NT), 0, st, x, g_out, out, save_mean, B, C, HW, (int)S, part);
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(blocks_for(C)), dim3(NT), 0, st, part, C, (int)S,
                       1.0 / ((double)B * (double)HW), weight, save_mean, save_invstd, coef, g_weight, g_bias);
    if (out) {
        if (g_pre) launch_bn_bwd_apply<true, true>(x, g_out, out, coef, C, HW, total, g_x, g_pre, st);
        else launch_bn_bwd_apply<true, false>(x, g_out, out, coef, C, HW, total, g_x, nullptr, st);
    } else {
        if (g_pre) launch_bn_bwd_apply<false, true>(x, g_out, out, coef, C, HW, total, g_x, g_pre, st);
        else launch_bn_bwd_apply<false, false>(x, g_out, out, coef, C, HW, total, g_x, nullptr, st);
    }
    return check_launch("dmh_bn_train_bwd");
}

int dmh_stem_bn_relu_pool_fwd(const float* x, const float* scale, const float* shift, int B, int C, int H, int W,
                              float* feat, float* pooled, unsigned char* argmax, void* stream) {
    DMH_REQUIRE(x && scale && shift && feat && pooled && argmax, "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && H >= 2 && W >= 2 && (H & 1) == 0 && (W & 1) == 0, "H and W must be even and >= 2");
    DMH_REQUIRE((int64_t)B * C <= 65535 && (int64_t)H * W < (1 << 30), "tensor too large");
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(blocks_for((int64_t)(H / 2) * (W / 2)), B * C), dim3(NT), 0,
                       (hipStream_t)stream, x, scale, shift, C, H, W, feat, pooled, argmax);
    return check_launch("dmh_stem_bn_relu_pool_fwd");
}

int dmh_stem_bn_relu_pool_bwd(const float* feat, const unsigned char* argmax, const float* g_feat, const float* g_pooled,
                              const float* scale, int B, int C, int H, int W, float* g_x, void* stream) {
    DMH_REQUIRE(feat && argmax && scale && g_x, "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && H >= 2 && W >= 2 && (H & 1) == 0 && (W & 1) == 0, "H and W must be even and >= 2");
    DMH_REQUIRE((int64_t)B * C <= 65535 && (int64_t)H * W < (1 << 30), "tensor too large");
    if ((W & 3) == 0)
        hipLaunchKernelGGL(stem_bwd4_kernel, dim3(blocks_for((int64_t)(H / 2) * (W / 4)), B * C), dim3(NT), 0,
                           (hipStream_t)stream, feat, argmax, g_feat, g_pooled, scale, C, H, W, g_x);
    else
        hipLaunchKernelGGL(stem_bwd_kernel, dim3(blocks_for((int64_t)(H / 2) * (W / 2)), B * C), dim3(NT), 0,
                           (hipStream_t)stream, feat, argmax, g_feat, g_pooled, scale, C, H, W, g_x);
    return check_launch("dmh_stem_bn_relu_pool_bwd");
}

}  // extern "C"
