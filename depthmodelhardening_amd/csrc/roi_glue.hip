// K19 -- windowed decoder glue and the windowed attack cost: the attack computed only where its loss lives.
//
// The attack's cost is -mean((disp * mask)^2) (torchattacks/attacks/phy_obj_atk.py:88-97, phy_obj_atk_l0.py:118-134):
// the disparity is read under the pasted object only, so d cost / d disp is zero outside the object's bounding box, and
// the patch gradient reads d cost / d image under the object only (physicalTrans.py:156-165: adv = scene (1 - m) + patch m).
// The high-resolution tail of the depth decoder (MD2/networks/depth_decoder.py:51-63: upconv(1,0) ... dispconv(0)) has a
// receptive field of a dozen pixels, so inside an attack it is evaluated -- forward and backward, exactly -- on a window
// around each sample's box instead of the whole 320 x 1024 frame.  The convolutions are the unchanged K10 / K11 / K13 / K17
// kernels run with padding 0 on compact [B, C, hc + 2, wc + 2] windows; what changes is the glue between them:
//
//   roi_glue : out[b, :, i, j] = pad1_reflect(cat(up2_nearest(ELU(y)), skip)) at frame position (org_b + (i, j) - 1)
//
// i.e. the up_cat_pad / elu_pad pass of decoder_glue.hip restricted to a per-sample window of the destination frame, with
// sources that are themselves windows (origin table) or whole frames (no table).  The reflection is the frame's, not the
// window's: a window that touches the image border sees exactly the padded values the full pass would produce.
// Backward = a gather over the same index map (deterministic, no atomics); positions no window entry reads get 0.
//
//   roi_cost : cost = sum_b sum_window (sigmoid(d_pre) * mask)^2 / (B H W)        (mask is the full-frame K3 output)
#include <stdlib.h>

#include "common.hpp"

using namespace dmh;

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : expf(x) - 1.f; }
__device__ __forceinline__ float elu_grad(float x) { return x > 0.f ? 1.f : expf(x); }

// row / column pair of a flat pair index by a reciprocal multiplication (magic = 2^32 / w2 rounded up, from the host): the
// 32-bit integer division it replaces is ~25 vector instructions per thread in kernels that move 8-16 bytes per thread
__device__ __forceinline__ void split_rc(int t, int w2, unsigned magic, int& row, int& col2) {
    if (w2 == 1) {
        row = t;
        col2 = 0;
        return;
    }
    int r = (int)__umulhi((unsigned)t, magic);
    if (r * w2 > t) --r;                        // (magic is rounded up: at most one too large)
    row = r;
    col2 = t - r * w2;
}

template <int CPT>
__global__ __launch_bounds__(NT) void roi_glue_fwd_kernel(const dmh_roi_glue_args a, unsigned magic, float* __restrict__ out) {
    // CPT consecutive channel planes of one part (the up-sampled y planes, or the skip planes) of one sample per thread: they
    // share the index map, so it is formed once (see roi_glue_bwd2_kernel)
    const int PH = a.hc + 2, PW = a.wc + 2, C = a.C1 + a.C2;
    const int t = blockIdx.x * NT + threadIdx.x;                // PW is even: a pair never straddles two rows
    if (2 * t >= PH * PW) return;
    const int groups = C / CPT, g1 = a.C1 / CPT;
    const int b = (int)blockIdx.y / groups, cg = (int)blockIdx.y - b * groups;
    int i, j;
    split_rc(t, PW >> 1, magic, i, j);
    j *= 2;
    const int oy = a.dst_org[2 * b], ox = a.dst_org[2 * b + 1];
    const int Y = reflect_idx(oy + i - 1, a.H);
    const int X0 = reflect_idx(ox + j - 1, a.W), X1 = reflect_idx(ox + j, a.W);
    float* op = out + ((size_t)(b * C) + (cg < g1 ? cg * CPT : a.C1 + (cg - g1) * CPT)) * PH * PW + 2 * t;
    if (cg < g1) {
        const int sy0 = a.y_org ? a.y_org[2 * b] : 0, sx0 = a.y_org ? a.y_org[2 * b + 1] : 0;
        const int yy = min(max((a.up ? Y >> 1 : Y) - sy0, 0), a.sh - 1);
        const int x0 = min(max((a.up ? X0 >> 1 : X0) - sx0, 0), a.sw - 1);
        const int x1 = min(max((a.up ? X1 >> 1 : X1) - sx0, 0), a.sw - 1);
        const float* row = a.y + ((size_t)(b * a.C1 + cg * CPT) * a.sh + yy) * a.sw;
        const size_t splane = (size_t)a.sh * a.sw;
#pragma unroll
        for (int k = 0; k < CPT; ++k, row += splane, op += PH * PW) {
            float v0 = row[x0], v1 = row[x1];
            if (a.elu) {
                v0 = elu_f(v0);
                v1 = elu_f(v1);
            }
            *reinterpret_cast<float2*>(op) = make_float2(v0, v1);
        }
    } else {
        const int ky0 = a.skip_org ? a.skip_org[2 * b] : 0, kx0 = a.skip_org ? a.skip_org[2 * b + 1] : 0;
        const int yy = min(max(Y - ky0, 0), a.kh - 1);
        const int x0 = min(max(X0 - kx0, 0), a.kw - 1), x1 = min(max(X1 - kx0, 0), a.kw - 1);
        const float* row = a.skip + ((size_t)(b * a.C2 + (cg - g1) * CPT) * a.kh + yy) * a.kw;
        const size_t kplane = (size_t)a.kh * a.kw;
#pragma unroll
        for (int k = 0; k < CPT; ++k, row += kplane, op += PH * PW) *reinterpret_cast<float2*>(op) = make_float2(row[x0], row[x1]);
    }
}

// sum of the window's padded-gradient entries that read frame position (Y, X): the entry straight above it and, on
// rows / columns 1 and H-2 / W-2, the reflected border entries -- each only if it lies inside this sample's window
__device__ __forceinline__ float gather_pad(const float* __restrict__ gp, int Y, int X, int H, int W, int oy, int ox,
                                            int PH, int PW) {
    int rows[3], cols[3], nr = 0, nc = 0;
    int r = Y + 1 - oy;
    if (r >= 0 && r < PH) rows[nr++] = r;
    if (Y == 1 && oy == 0) rows[nr++] = 0;
    r = H + 1 - oy;
    if (Y == H - 2 && r < PH) rows[nr++] = r;
    int q = X + 1 - ox;
    if (q >= 0 && q < PW) cols[nc++] = q;
    if (X == 1 && ox == 0) cols[nc++] = 0;
    q = W + 1 - ox;
    if (X == W - 2 && q < PW) cols[nc++] = q;
    float acc = 0.f;
    for (int s = 0; s < nr; ++s)
        for (int t = 0; t < nc; ++t) acc += gp[rows[s] * PW + cols[t]];
    return acc;
}

// the same sum for an element whose readers cannot involve the frame's border (1 < Y < H - 2, 1 < X < W - 2): the one entry
// straight above it, if the window holds it
__device__ __forceinline__ float gather_interior(const float* __restrict__ gp, int Y, int X, int oy, int ox, int PH, int PW) {
    const int r = Y + 1 - oy, q = X + 1 - ox;
    return (r >= 0 && r < PH && q >= 0 && q < PW) ? gp[r * PW + q] : 0.f;
}

// grid.y = B*C1 planes of g_y followed by B*C2 planes of g_skip; one thread per source element of the REGION to write: the
// whole plane (0 where no window entry reads the element), or -- for a whole-frame source -- a per-sample rectangle that
// holds everything the window reaches (the caller owns the rest of the plane: pre-zeroed, or never read)
struct Region {
    const int* y_org;       // [B,2] or NULL: whole plane
    const int* skip_org;
    int yh, yw, kh, kw;     // rectangle sizes (plane sizes when the table is NULL)
};

__global__ __launch_bounds__(NT) void roi_glue_bwd_kernel(const dmh_roi_glue_args a, const Region rg,
                                                          const float* __restrict__ g_out, float* __restrict__ g_y,
                                                          float* __restrict__ g_skip) {
    const int PH = a.hc + 2, PW = a.wc + 2, C = a.C1 + a.C2;
    const int t = blockIdx.x * NT + threadIdx.x;
    const int plane = blockIdx.y;
    if (plane < a.B * a.C1) {
        if (t >= rg.yh * rg.yw) return;
        const int b = plane / a.C1, c = plane - b * a.C1;
        const int oy = a.dst_org[2 * b], ox = a.dst_org[2 * b + 1];
        const int sy0 = a.y_org ? a.y_org[2 * b] : 0, sx0 = a.y_org ? a.y_org[2 * b + 1] : 0;
        int yy = t / rg.yw, xx = t - yy * rg.yw;
        if (rg.y_org) {
            yy += rg.y_org[2 * b];
            xx += rg.y_org[2 * b + 1];
        }
        const int Ys = sy0 + yy, Xs = sx0 + xx;
        const float* gp = g_out + (size_t)(b * C + c) * PH * PW;
        float acc;
        if (a.up) {
            const int Y0 = 2 * Ys, X0 = 2 * Xs;
            if (Y0 > 1 && Y0 + 1 < a.H - 2 && X0 > 1 && X0 + 1 < a.W - 2) {
                // away from the frame's border (all but two rows / columns): the readers of the 2 x 2 children are the entries
                // straight above them -- four loads instead of the general reflection gather
                const int r = Y0 + 1 - oy, q = X0 + 1 - ox;
                if (r >= 0 && r + 1 < PH && q >= 0 && q + 1 < PW) {
                    const float* p0 = gp + r * PW + q;
                    acc = p0[0] + p0[1] + p0[PW] + p0[PW + 1];      // (the general path's order of additions)
                } else if (r + 1 < 0 || r >= PH || q + 1 < 0 || q >= PW) {
                    acc = 0.f;
                } else {
                    acc = gather_interior(gp, Y0, X0, oy, ox, PH, PW) + gather_interior(gp, Y0, X0 + 1, oy, ox, PH, PW) +
                          gather_interior(gp, Y0 + 1, X0, oy, ox, PH, PW) + gather_interior(gp, Y0 + 1, X0 + 1, oy, ox, PH, PW);
                }
            } else {
                acc = gather_pad(gp, Y0, X0, a.H, a.W, oy, ox, PH, PW) + gather_pad(gp, Y0, X0 + 1, a.H, a.W, oy, ox, PH, PW) +
                      gather_pad(gp, Y0 + 1, X0, a.H, a.W, oy, ox, PH, PW) +
                      gather_pad(gp, Y0 + 1, X0 + 1, a.H, a.W, oy, ox, PH, PW);
            }
        } else if (Ys > 1 && Ys < a.H - 2 && Xs > 1 && Xs < a.W - 2) {
            acc = gather_interior(gp, Ys, Xs, oy, ox, PH, PW);
        } else {
            acc = gather_pad(gp, Ys, Xs, a.H, a.W, oy, ox, PH, PW);
        }
        const size_t o = ((size_t)plane * a.sh + yy) * a.sw + xx;
        g_y[o] = (a.elu && acc != 0.f) ? acc * elu_grad(a.y[o]) : acc;
    } else {
        if (t >= rg.kh * rg.kw) return;
        const int q = plane - a.B * a.C1, b = q / a.C2, c = q - b * a.C2;
        const int oy = a.dst_org[2 * b], ox = a.dst_org[2 * b + 1];
        const int ky0 = a.skip_org ? a.skip_org[2 * b] : 0, kx0 = a.skip_org ? a.skip_org[2 * b + 1] : 0;
        int yy = t / rg.kw, xx = t - yy * rg.kw;
        if (rg.skip_org) {
            yy += rg.skip_org[2 * b];
            xx += rg.skip_org[2 * b + 1];
        }
        const float* gp = g_out + (size_t)(b * C + a.C1 + c) * PH * PW;
        const int Ys = ky0 + yy, Xs = kx0 + xx;
        g_skip[((size_t)q * a.kh + yy) * a.kw + xx] = (Ys > 1 && Ys < a.H - 2 && Xs > 1 && Xs < a.W - 2)
                                                          ? gather_interior(gp, Ys, Xs, oy, ox, PH, PW)
                                                          : gather_pad(gp, Ys, Xs, a.H, a.W, oy, ox, PH, PW);
    }
}

// One gathered element of g_y / g_skip, exactly as roi_glue_bwd_kernel forms it (the general path of the two-wide kernel).
__device__ __forceinline__ float glue_bwd_y_elem(const dmh_roi_glue_args& a, const float* __restrict__ gp, int Ys, int Xs, int oy,
                                                 int ox, int PH, int PW) {
    if (a.up) {
        const int Y0 = 2 * Ys, X0 = 2 * Xs;
        if (Y0 > 1 && Y0 + 1 < a.H - 2 && X0 > 1 && X0 + 1 < a.W - 2) {
            const int r = Y0 + 1 - oy, q = X0 + 1 - ox;
            if (r >= 0 && r + 1 < PH && q >= 0 && q + 1 < PW) {
                const float* p0 = gp + r * PW + q;
                return p0[0] + p0[1] + p0[PW] + p0[PW + 1];
            }
            if (r + 1 < 0 || r >= PH || q + 1 < 0 || q >= PW) return 0.f;
            return gather_interior(gp, Y0, X0, oy, ox, PH, PW) + gather_interior(gp, Y0, X0 + 1, oy, ox, PH, PW) +
                   gather_interior(gp, Y0 + 1, X0, oy, ox, PH, PW) + gather_interior(gp, Y0 + 1, X0 + 1, oy, ox, PH, PW);
        }
        return gather_pad(gp, Y0, X0, a.H, a.W, oy, ox, PH, PW) + gather_pad(gp, Y0, X0 + 1, a.H, a.W, oy, ox, PH, PW) +
               gather_pad(gp, Y0 + 1, X0, a.H, a.W, oy, ox, PH, PW) + gather_pad(gp, Y0 + 1, X0 + 1, a.H, a.W, oy, ox, PH, PW);
    }
    if (Ys > 1 && Ys < a.H - 2 && Xs > 1 && Xs < a.W - 2) return gather_interior(gp, Ys, Xs, oy, ox, PH, PW);
    return gather_pad(gp, Ys, Xs, a.H, a.W, oy, ox, PH, PW);
}

// Round 6: two adjacent source elements per thread (every region and plane width of the window plan is even, roi.py), the
// row / column of a thread by a reciprocal multiplication instead of an integer division, exact grids per part (grid.z: the
// g_y planes and the g_skip planes have their own block counts) -- the one-element kernel spent as many cycles on index
// arithmetic as on its loads.  Same arithmetic per element (same order of additions): bit-identical results.
struct Region2 {
    Region rg;
    unsigned ymagic, kmagic;    // ceil(2^32 / (width / 2)) of the two parts
    int yblocks, kblocks;       // blocks along x of the two parts
};
template <int CPT>
__global__ __launch_bounds__(NT) void roi_glue_bwd2_kernel(const dmh_roi_glue_args a, const Region2 r2,
                                                           const float* __restrict__ g_out, float* __restrict__ g_y,
                                                           float* __restrict__ g_skip) {
    // CPT consecutive channel planes of one sample per thread: the planes of a sample share their geometry (origins, window
    // offsets, the fast-path test), so the index arithmetic is done once and a block moves CPT times the bytes.  Measured: no
    // gain here (see the launcher: the default is CPT = 1); the forward kernel, roi_crop and roi_paste take 5-15 % from it.
    const Region& rg = r2.rg;
    const int PH = a.hc + 2, PW = a.wc + 2, C = a.C1 + a.C2;
    // flat grid: the blocks of the g_y plane groups, then those of the g_skip plane groups (uniform decode: scalar divisions)
    const int ytotal = r2.yblocks * a.B * (a.C1 / CPT);
    const bool ypart = (int)blockIdx.x < ytotal;
    const int bid = ypart ? (int)blockIdx.x : (int)blockIdx.x - ytotal;
    const int per = ypart ? r2.yblocks : r2.kblocks;
    const int pl = bid / per;
    const int t = (bid - pl * per) * NT + (int)threadIdx.x;
    const size_t gplane = (size_t)PH * PW;
    if (ypart) {
        const int groups = a.C1 / CPT;
        const int b = pl / groups, c0 = (pl - b * groups) * CPT;
        const int w2 = rg.yw >> 1;
        if (t >= rg.yh * w2) return;
        const int oy = a.dst_org[2 * b], ox = a.dst_org[2 * b + 1];
        const int sy0 = a.y_org ? a.y_org[2 * b] : 0, sx0 = a.y_org ? a.y_org[2 * b + 1] : 0;
        int yy, xx;
        split_rc(t, w2, r2.ymagic, yy, xx);
        xx *= 2;
        if (rg.y_org) {
            yy += rg.y_org[2 * b];
            xx += rg.y_org[2 * b + 1];
        }
        const int Ys = sy0 + yy, Xs = sx0 + xx;
        const float* gp = g_out + (size_t)(b * C + c0) * gplane;
        const size_t splane = (size_t)a.sh * a.sw;
        size_t o = ((size_t)(b * a.C1 + c0) * a.sh + yy) * a.sw + xx;
        int fast = 0, off = 0;              // 1: up-sampled, all eight entries inside; 2: plain, both entries inside
        if (a.up) {
            const int Y0 = 2 * Ys, X0 = 2 * Xs, r = Y0 + 1 - oy, q = X0 + 1 - ox;
            if (Y0 > 1 && Y0 + 1 < a.H - 2 && X0 > 1 && X0 + 3 < a.W - 2 && r >= 0 && r + 1 < PH && q >= 0 && q + 3 < PW) {
                fast = 1;
                off = r * PW + q;
            }
        } else {
            const int r = Ys + 1 - oy, q = Xs + 1 - ox;
            if (Ys > 1 && Ys < a.H - 2 && Xs > 1 && Xs + 1 < a.W - 2 && r >= 0 && r < PH && q >= 0 && q + 1 < PW) {
                fast = 2;
                off = r * PW + q;
            }
        }
        // the three cases as three loops (fast is per thread, but the common case must not carry the general path's code in
        // every unrolled iteration: a first version did, and ran at half the speed of the one-plane kernel)
        if (fast == 1) {
#pragma unroll
            for (int j = 0; j < CPT; ++j, gp += gplane, o += splane) {
                const float* p0 = gp + off;
                float v0 = p0[0] + p0[1] + p0[PW] + p0[PW + 1];
                float v1 = p0[2] + p0[3] + p0[PW + 2] + p0[PW + 3];
                if (a.elu) {
                    const float2 yv = *reinterpret_cast<const float2*>(a.y + o);
                    v0 = v0 != 0.f ? v0 * elu_grad(yv.x) : v0;
                    v1 = v1 != 0.f ? v1 * elu_grad(yv.y) : v1;
                }
                *reinterpret_cast<float2*>(g_y + o) = make_float2(v0, v1);
            }
        } else if (fast == 2) {
#pragma unroll
            for (int j = 0; j < CPT; ++j, gp += gplane, o += splane) {
                float v0 = gp[off], v1 = gp[off + 1];
                if (a.elu) {
                    const float2 yv = *reinterpret_cast<const float2*>(a.y + o);
                    v0 = v0 != 0.f ? v0 * elu_grad(yv.x) : v0;
                    v1 = v1 != 0.f ? v1 * elu_grad(yv.y) : v1;
                }
                *reinterpret_cast<float2*>(g_y + o) = make_float2(v0, v1);
            }
        } else {
#pragma unroll 1
            for (int j = 0; j < CPT; ++j, gp += gplane, o += splane) {
                float v0 = glue_bwd_y_elem(a, gp, Ys, Xs, oy, ox, PH, PW);
                float v1 = glue_bwd_y_elem(a, gp, Ys, Xs + 1, oy, ox, PH, PW);
                if (a.elu) {
                    const float2 yv = *reinterpret_cast<const float2*>(a.y + o);
                    v0 = v0 != 0.f ? v0 * elu_grad(yv.x) : v0;
                    v1 = v1 != 0.f ? v1 * elu_grad(yv.y) : v1;
                }
                *reinterpret_cast<float2*>(g_y + o) = make_float2(v0, v1);
            }
        }
    } else {
        const int groups = a.C2 / CPT;
        const int b = pl / groups, c0 = (pl - b * groups) * CPT;
        const int w2 = rg.kw >> 1;
        if (t >= rg.kh * w2) return;
        const int oy = a.dst_org[2 * b], ox = a.dst_org[2 * b + 1];
        const int ky0 = a.skip_org ? a.skip_org[2 * b] : 0, kx0 = a.skip_org ? a.skip_org[2 * b + 1] : 0;
        int yy, xx;
        split_rc(t, w2, r2.kmagic, yy, xx);
        xx *= 2;
        if (rg.skip_org) {
            yy += rg.skip_org[2 * b];
            xx += rg.skip_org[2 * b + 1];
        }
        const float* gp = g_out + (size_t)(b * C + a.C1 + c0) * gplane;
        const int Ys = ky0 + yy, Xs = kx0 + xx;
        const int r = Ys + 1 - oy, qq = Xs + 1 - ox;
        const bool fast = Ys > 1 && Ys < a.H - 2 && Xs > 1 && Xs + 1 < a.W - 2 && r >= 0 && r < PH && qq >= 0 && qq + 1 < PW;
        const bool in0 = Ys > 1 && Ys < a.H - 2 && Xs > 1 && Xs < a.W - 2, in1 = Ys > 1 && Ys < a.H - 2 && Xs + 1 > 1 && Xs + 1 < a.W - 2;
        const size_t kplane = (size_t)a.kh * a.kw;
        float* gs = g_skip + ((size_t)(b * a.C2 + c0) * a.kh + yy) * a.kw + xx;
        if (fast) {
#pragma unroll
            for (int j = 0; j < CPT; ++j, gp += gplane, gs += kplane)
                *reinterpret_cast<float2*>(gs) = make_float2(gp[r * PW + qq], gp[r * PW + qq + 1]);
        } else {
#pragma unroll 1
            for (int j = 0; j < CPT; ++j, gp += gplane, gs += kplane) {
                const float v0 = in0 ? gather_interior(gp, Ys, Xs, oy, ox, PH, PW) : gather_pad(gp, Ys, Xs, a.H, a.W, oy, ox, PH, PW);
                const float v1 = in1 ? gather_interior(gp, Ys, Xs + 1, oy, ox, PH, PW) : gather_pad(gp, Ys, Xs + 1, a.H, a.W, oy, ox, PH, PW);
                *reinterpret_cast<float2*>(gs) = make_float2(v0, v1);
            }
        }
    }
}

// ---- windowed attack cost ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void roi_cost_fwd_kernel(const float* __restrict__ d_pre, const float* __restrict__ mask,
                                                          const int* __restrict__ org, int hd, int wd, int H, int W,
                                                          float* __restrict__ sig, float* __restrict__ partials) {
    __shared__ float s_red[NT / WAVE];
    const int b = blockIdx.y, n = hd * wd;
    const int oy = org[2 * b], ox = org[2 * b + 1];
    float acc = 0.f;
    for (int t = blockIdx.x * NT + threadIdx.x; t < n; t += gridDim.x * NT) {
        const int i = t / wd, j = t - i * wd;
        const float s = 1.f / (1.f + expf(-d_pre[(size_t)b * n + t]));
        sig[(size_t)b * n + t] = s;
        const float v = s * mask[((size_t)b * H + oy + i) * W + ox + j];
        acc += v * v;
    }
    const float r = block_sum<NT>(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = r;
}

__global__ __launch_bounds__(NT) void roi_cost_finalize_kernel(const float* __restrict__ partials, int nblk, double n,
                                                               float* __restrict__ cost) {
    __shared__ double s_red[NT];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += NT) acc += (double)partials[i];
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) cost[0] = (float)(s_red[0] / n);
}

// g_pre = gscale * 2 s m^2 / n * s (1 - s)      (cost -> sigmoid -> pre-activation)
__global__ __launch_bounds__(NT) void roi_cost_bwd_kernel(const float* __restrict__ sig, const float* __restrict__ mask,
                                                          const int* __restrict__ org, int hd, int wd, int H, int W,
                                                          const float* __restrict__ gscale, float inv_n,
                                                          float* __restrict__ g_pre) {
    const int b = blockIdx.y, n = hd * wd;
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= n) return;
    const int oy = org[2 * b], ox = org[2 * b + 1];
    const int i = t / wd, j = t - i * wd;
    const float s = sig[(size_t)b * n + t], m = mask[((size_t)b * H + oy + i) * W + ox + j];
    g_pre[(size_t)b * n + t] = gscale[0] * 2.f * inv_n * s * m * m * s * (1.f - s);
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + NT - 1) / NT); }
inline int cost_blocks(int n) {
    const int b = (n + NT * 4 - 1) / (NT * 4);
    return b < 1 ? 1 : (b > 64 ? 64 : b);
}

int check_glue(const dmh_roi_glue_args* a) {
    DMH_REQUIRE(a && a->y && a->dst_org && (a->skip || a->C2 == 0), "null pointer");
    DMH_REQUIRE(a->B > 0 && a->C1 > 0 && a->C2 >= 0 && a->sh >= 1 && a->sw >= 1 && (a->C2 == 0 || (a->kh >= 1 && a->kw >= 1)),
                "bad sizes");
    DMH_REQUIRE(a->hc >= 2 && a->wc >= 2 && (a->wc & 1) == 0 && a->hc <= a->H && a->wc <= a->W && a->H >= 4 && a->W >= 4,
                "window must be at least 2 x 2, of even width, inside a frame of at least 4 x 4");
    DMH_REQUIRE((int64_t)a->B * (a->C1 + a->C2) <= 65535, "too many planes");
    DMH_REQUIRE((int64_t)a->sh * a->sw < (1 << 30) && (int64_t)a->kh * a->kw < (1 << 30), "source plane too large");
    return DMH_OK;
}

}  // namespace

extern "C" {

int dmh_roi_glue_fwd(const dmh_roi_glue_args* a, float* out, void* stream) {
    if (int rc = check_glue(a)) return rc;
    DMH_REQUIRE(out, "null pointer");
    const unsigned magic = (unsigned)(((uint64_t)1 << 32) / (unsigned)((a->wc + 2) >> 1)) + 1u;
    const int cpt = (a->C1 % 8 == 0 && a->C2 % 8 == 0) ? 8 : ((a->C1 % 4 == 0 && a->C2 % 4 == 0) ? 4 : 1);
    const dim3 grid(blocks_for((int64_t)(a->hc + 2) * (a->wc + 2) / 2), (unsigned)(a->B * ((a->C1 + a->C2) / cpt)));
    if (cpt == 8) hipLaunchKernelGGL(roi_glue_fwd_kernel<8>, grid, dim3(NT), 0, (hipStream_t)stream, *a, magic, out);
    else if (cpt == 4) hipLaunchKernelGGL(roi_glue_fwd_kernel<4>, grid, dim3(NT), 0, (hipStream_t)stream, *a, magic, out);
    else hipLaunchKernelGGL(roi_glue_fwd_kernel<1>, grid, dim3(NT), 0, (hipStream_t)stream, *a, magic, out);
    return check_launch("dmh_roi_glue_fwd");
}

int dmh_roi_glue_bwd(const dmh_roi_glue_args* a, const float* g_out, float* g_y, float* g_skip, const int* y_reg_org,
                     int y_reg_h, int y_reg_w, const int* skip_reg_org, int skip_reg_h, int skip_reg_w, void* stream) {
    if (int rc = check_glue(a)) return rc;
    DMH_REQUIRE(g_out && g_y, "null pointer");
    DMH_REQUIRE(!y_reg_org || (!a->y_org && y_reg_h >= 1 && y_reg_w >= 1 && y_reg_h <= a->sh && y_reg_w <= a->sw),
                "a y region needs a whole-frame y and must fit its plane");
    DMH_REQUIRE(!skip_reg_org || (!a->skip_org && skip_reg_h >= 1 && skip_reg_w >= 1 && skip_reg_h <= a->kh && skip_reg_w <= a->kw),
                "a skip region needs a whole-frame skip and must fit its plane");
    Region rg;
    rg.y_org = y_reg_org;
    rg.skip_org = skip_reg_org;
    rg.yh = y_reg_org ? y_reg_h : a->sh;
    rg.yw = y_reg_org ? y_reg_w : a->sw;
    rg.kh = skip_reg_org ? skip_reg_h : a->kh;
    rg.kw = skip_reg_org ? skip_reg_w : a->kw;
    const bool skip = g_skip && a->C2 > 0;
    const int planes = a->B * a->C1 + (skip ? a->B * a->C2 : 0);
    const int64_t ny = (int64_t)rg.yh * rg.yw, nk = skip ? (int64_t)rg.kh * rg.kw : 0;
    // two elements per thread where every width involved is even and every base 8-byte aligned (the window plan's sizes and
    // origins all are; DMH_ROI_GLUE2=0: the one-element kernel, A/B switch)
    static const bool wide_ok = !(getenv("DMH_ROI_GLUE2") && atoi(getenv("DMH_ROI_GLUE2")) == 0);
    bool wide = wide_ok && !(rg.yw & 1) && !(a->sw & 1) && !(((uintptr_t)g_y | (uintptr_t)a->y) & 7) &&
                (!skip || (!(rg.kw & 1) && !(a->kw & 1) && !((uintptr_t)g_skip & 7)));
    if (wide) {
        Region2 r2;
        r2.rg = rg;
        const unsigned yw2 = (unsigned)(rg.yw >> 1), kw2 = skip ? (unsigned)(rg.kw >> 1) : 1u;
        r2.ymagic = (unsigned)(((uint64_t)1 << 32) / yw2) + 1u;      // >= 2^32 / w2: split_rc corrects the one-off case
        r2.kmagic = (unsigned)(((uint64_t)1 << 32) / kw2) + 1u;
        r2.yblocks = (int)blocks_for(ny / 2);
        r2.kblocks = skip ? (int)blocks_for(nk / 2) : 1;
        const int c2 = skip ? a->C2 : 0;
        // planes per thread: ONE.  The forward kernel, the crops and the pastes gain 5-15 % from eight planes per thread (shared
        // index map); this gather does not -- with eight planes it ran at HALF the speed of one (3.6 against 1.97 ms per step,
        // profiles/README.md round 6), whichever way its three cases were laid out: DMH_ROI_GLUE_CPT=8 / 4 keeps it measurable
        static const int cpt_env = getenv("DMH_ROI_GLUE_CPT") ? atoi(getenv("DMH_ROI_GLUE_CPT")) : 1;
        const int cpt = (cpt_env == 8 && a->C1 % 8 == 0 && c2 % 8 == 0) ? 8 : ((cpt_env >= 4 && a->C1 % 4 == 0 && c2 % 4 == 0) ? 4 : 1);
        const int64_t total = (int64_t)r2.yblocks * a->B * (a->C1 / cpt) + (skip ? (int64_t)r2.kblocks * a->B * (c2 / cpt) : 0);
        DMH_REQUIRE(total < ((int64_t)1 << 31), "too many blocks");
        const dim3 grid((unsigned)total);
        if (cpt == 8) hipLaunchKernelGGL(roi_glue_bwd2_kernel<8>, grid, dim3(NT), 0, (hipStream_t)stream, *a, r2, g_out, g_y, g_skip);
        else if (cpt == 4) hipLaunchKernelGGL(roi_glue_bwd2_kernel<4>, grid, dim3(NT), 0, (hipStream_t)stream, *a, r2, g_out, g_y, g_skip);
        else hipLaunchKernelGGL(roi_glue_bwd2_kernel<1>, grid, dim3(NT), 0, (hipStream_t)stream, *a, r2, g_out, g_y, g_skip);
        return check_launch("dmh_roi_glue_bwd");
    }
    hipLaunchKernelGGL(roi_glue_bwd_kernel, dim3(blocks_for(ny > nk ? ny : nk), planes), dim3(NT), 0, (hipStream_t)stream, *a,
                       rg, g_out, g_y, g_skip);
    return check_launch("dmh_roi_glue_bwd");
}

int64_t dmh_roi_cost_partials_size(int B, int hd, int wd) { return (int64_t)B * cost_blocks(hd * wd); }

int dmh_roi_cost_fwd(const float* d_pre, const float* mask, const int* org, int B, int hd, int wd, int H, int W, float* sig,
                     float* partials, float* cost, void* stream) {
    return dmh_roi_cost_fwd_scaled(d_pre, mask, org, B, hd, wd, H, W, 1.f, sig, partials, cost, stream);
}

int dmh_roi_cost_fwd_scaled(const float* d_pre, const float* mask, const int* org, int B, int hd, int wd, int H, int W,
                            float scale, float* sig, float* partials, float* cost, void* stream) {
    DMH_REQUIRE(d_pre && mask && org && sig && partials && cost, "null pointer");
    DMH_REQUIRE(scale == 1.f || scale == -1.f, "scale must be +1 or -1 (a sign: the result stays bit-identical to -cost)");
    DMH_REQUIRE(B > 0 && B <= 65535 && hd >= 1 && wd >= 1 && hd <= H && wd <= W && (int64_t)hd * wd < (1 << 30), "bad sizes");
    const int nb = cost_blocks(hd * wd);
    hipLaunchKernelGGL(roi_cost_fwd_kernel, dim3(nb, B), dim3(NT), 0, (hipStream_t)stream, d_pre, mask, org, hd, wd, H, W, sig,
                       partials);
    hipLaunchKernelGGL(roi_cost_finalize_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, partials, nb * B,
                       (double)scale * (double)B * (double)H * (double)W, cost);
    return check_launch("dmh_roi_cost_fwd");
}

int dmh_roi_cost_bwd(const float* sig, const float* mask, const int* org, int B, int hd, int wd, int H, int W,
                     const float* gscale, float* g_pre, void* stream) {
    return dmh_roi_cost_bwd_scaled(sig, mask, org, B, hd, wd, H, W, 1.f, gscale, g_pre, stream);
}

int dmh_roi_cost_bwd_scaled(const float* sig, const float* mask, const int* org, int B, int hd, int wd, int H, int W,
                            float scale, const float* gscale, float* g_pre, void* stream) {
    DMH_REQUIRE(sig && mask && org && gscale && g_pre, "null pointer");
    DMH_REQUIRE(scale == 1.f || scale == -1.f, "scale must be +1 or -1");
    DMH_REQUIRE(B > 0 && B <= 65535 && hd >= 1 && wd >= 1 && hd <= H && wd <= W && (int64_t)hd * wd < (1 << 30), "bad sizes");
    hipLaunchKernelGGL(roi_cost_bwd_kernel, dim3(blocks_for((int64_t)hd * wd), B), dim3(NT), 0, (hipStream_t)stream, sig, mask,
                       org, hd, wd, H, W, gscale, (float)((double)scale / ((double)B * (double)H * (double)W)), g_pre);
    return check_launch("dmh_roi_cost_bwd");
}

}  // extern "C"
