// Decoder glue -- the element-wise passes between the MIOpen convolutions of the Monodepth2 depth decoder, fused.
//
// Reference: MD2/networks/depth_decoder.py:51-63 runs, per decoder stage, ELU (in place) -> nearest x2 upsample ->
// torch.cat with the skip feature -> ReflectionPad2d(1) (inside Conv3x3, MD2/layers.py:133-136) as four separate
// full-tensor passes, and pads the same ELU output twice where it feeds both the next stage and a disparity head.
// On MI355X these copies are ~30 % of the adversarial-training step (profiles/r01_bench_timed_region.csv:
// reflection_pad2d fwd+bwd alone 12 %).  Here each stage boundary is ONE pass:
//
//   up_cat_pad :  out[B, C1+C2, 2h+2, 2w+2] = pad1_reflect( cat( up2_nearest( ELU(y) ), skip ) )
//   elu_pad    :  out[B, C,  H+2,  W+2]     = pad1_reflect( ELU(z) )          (apply_elu = 0: pad only)
//
// and the convolutions run with padding = 0 on the padded tensors.  Backward kernels are gathers (the adjoint of the
// reflection pad folds the border rows/columns back onto rows 1 and H-2): deterministic, no atomics.
// Pure HBM streaming: one read + one write per element, 64-lane coalesced rows.
#include <stdlib.h>

#include "common.hpp"

using namespace dmh;

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : expf(x) - 1.f; }
__device__ __forceinline__ float elu_grad(float x) { return x > 0.f ? 1.f : expf(x); }

// sum of the padded-gradient entries that ReflectionPad2d(1) reads from interior position (Y, X)
__device__ __forceinline__ float fold2d(const float* __restrict__ gp, int Y, int X, int H, int W) {
    const int PW = W + 2;
    // only rows/columns 1 and H-2 / W-2 receive reflected border entries: everything else is one load
    if (Y != 1 && Y != H - 2 && X != 1 && X != W - 2) return gp[(Y + 1) * PW + X + 1];
    int rows[3], cols[3], nr = 1, nc = 1;
    rows[0] = Y + 1;
    cols[0] = X + 1;
    if (Y == 1) rows[nr++] = 0;
    if (Y == H - 2) rows[nr++] = H + 1;
    if (X == 1) cols[nc++] = 0;
    if (X == W - 2) cols[nc++] = W + 1;
    float acc = 0.f;
    for (int a = 0; a < nr; ++a)
        for (int b = 0; b < nc; ++b) acc += gp[rows[a] * PW + cols[b]];
    return acc;
}

// Index scheme: blockIdx.y = plane (b * C + c); blockIdx.x * NT + threadIdx.x = PAIR of horizontally adjacent
// elements inside the plane.  Row widths are even (2w, 2w+2, W+2 with W even), so a pair never straddles rows and
// its 8-byte store is aligned: half the store instructions, one 32-bit division per pair.
template <bool VEC2>
__global__ __launch_bounds__(NT) void up_cat_pad_fwd_kernel(const float* __restrict__ y, const float* __restrict__ skip,
                                                            int C1, int C2, int h, int w, float* __restrict__ out) {
    const int H = 2 * h, W = 2 * w, PH = H + 2, PW = W + 2, C = C1 + C2;
    const int idx = (blockIdx.x * NT + threadIdx.x) * 2;   // PW is even
    if (idx >= PH * PW) return;
    const int plane = blockIdx.y, b = plane / C, c = plane - b * C;
    const int py = idx / PW, px = idx - py * PW;
    const int Y = reflect_idx(py - 1, H);
    const int X0 = reflect_idx(px - 1, W), X1 = reflect_idx(px, W);
    float v0, v1;
    if (c < C1) {
        const float* row = y + ((size_t)(b * C1 + c) * h + (Y >> 1)) * w;
        v0 = elu_f(row[X0 >> 1]);
        v1 = elu_f(row[X1 >> 1]);
    } else {
        const float* row = skip + ((size_t)(b * C2 + (c - C1)) * H + Y) * W;
        v0 = row[X0];
        v1 = row[X1];
    }
    *reinterpret_cast<float2*>(out + (size_t)plane * PH * PW + idx) = make_float2(v0, v1);
}

// grid.y = B*C1 planes of g_y followed by B*C2 planes of g_skip; grid.x covers the larger (skip-resolution) plane
template <bool VEC2>
__global__ __launch_bounds__(NT) void up_cat_pad_bwd_kernel(const float* __restrict__ y, const float* __restrict__ g_out,
                                                            int B, int C1, int C2, int h, int w,
                                                            float* __restrict__ g_y, float* __restrict__ g_skip) {
    const int H = 2 * h, W = 2 * w, C = C1 + C2;
    const size_t pplane = (size_t)(H + 2) * (W + 2);
    const int t = blockIdx.x * NT + threadIdx.x;
    const int plane = blockIdx.y;
    if (plane < B * C1) {
        if (t >= h * w) return;
        const int b = plane / C1, c = plane - b * C1;
        const int i = t / w, j = t - i * w;
        const float* gp = g_out + (size_t)(b * C + c) * pplane;
        const float acc = fold2d(gp, 2 * i, 2 * j, H, W) + fold2d(gp, 2 * i, 2 * j + 1, H, W) +
                          fold2d(gp, 2 * i + 1, 2 * j, H, W) + fold2d(gp, 2 * i + 1, 2 * j + 1, H, W);
        const size_t o = (size_t)plane * h * w + t;
        g_y[o] = acc * elu_grad(y[o]);
    } else {
        const int idx = t * 2;   // W = 2w is even
        if (idx >= H * W) return;
        const int q = plane - B * C1, b = q / C2, c = q - b * C2;
        const int Y = idx / W, X = idx - Y * W;
        const float* gp = g_out + (size_t)(b * C + C1 + c) * pplane;
        *reinterpret_cast<float2*>(g_skip + (size_t)q * H * W + idx) =
            make_float2(fold2d(gp, Y, X, H, W), fold2d(gp, Y, X + 1, H, W));
    }
}

template <bool VEC2>
__global__ __launch_bounds__(NT) void elu_pad_fwd_kernel(const float* __restrict__ z, int H, int W, int apply_elu,
                                                         float* __restrict__ out) {
    const int PH = H + 2, PW = W + 2;
    const float* zp = z + (size_t)blockIdx.y * H * W;
    float* op = out + (size_t)blockIdx.y * PH * PW;
    if (VEC2) {
        const int idx = (blockIdx.x * NT + threadIdx.x) * 2;
        if (idx >= PH * PW) return;
        const int py = idx / PW, px = idx - py * PW;
        const float* row = zp + (size_t)reflect_idx(py - 1, H) * W;
        float v0 = row[reflect_idx(px - 1, W)], v1 = row[reflect_idx(px, W)];
        if (apply_elu) {
            v0 = elu_f(v0);
            v1 = elu_f(v1);
        }
        *reinterpret_cast<float2*>(op + idx) = make_float2(v0, v1);
    } else {
        const int idx = blockIdx.x * NT + threadIdx.x;
        if (idx >= PH * PW) return;
        const int py = idx / PW, px = idx - py * PW;
        const float v = zp[(size_t)reflect_idx(py - 1, H) * W + reflect_idx(px - 1, W)];
        op[idx] = apply_elu ? elu_f(v) : v;
    }
}

template <bool VEC2>
__global__ __launch_bounds__(NT) void elu_pad_bwd_kernel(const float* __restrict__ z, const float* __restrict__ g_out,
                                                         int H, int W, int apply_elu, float* __restrict__ g_z) {
    const float* gp = g_out + (size_t)blockIdx.y * (H + 2) * (W + 2);
    const size_t base = (size_t)blockIdx.y * H * W;
    if (VEC2) {
        const int idx = (blockIdx.x * NT + threadIdx.x) * 2;
        if (idx >= H * W) return;
        const int Y = idx / W, X = idx - Y * W;
        float g0 = fold2d(gp, Y, X, H, W), g1 = fold2d(gp, Y, X + 1, H, W);
        if (apply_elu) {
            const float2 zz = *reinterpret_cast<const float2*>(z + base + idx);
            g0 *= elu_grad(zz.x);
            g1 *= elu_grad(zz.y);
        }
        *reinterpret_cast<float2*>(g_z + base + idx) = make_float2(g0, g1);
    } else {
        const int idx = blockIdx.x * NT + threadIdx.x;
        if (idx >= H * W) return;
        const int Y = idx / W, X = idx - Y * W;
        const float g = fold2d(gp, Y, X, H, W);
        g_z[base + idx] = apply_elu ? g * elu_grad(z[base + idx]) : g;
    }
}

// ---- four-wide variants (w even, i.e. W = 2w a multiple of 4) -------------------------------------------------------
// The pair kernels above are instruction-bound next to the memory system (an integer division, two reflections and up
// to four expf per 8 bytes).  Here a thread owns four horizontally adjacent INTERIOR elements: the unpadded side moves
// as aligned 16-byte accesses, the padded side as 4-byte-aligned 16-byte accesses (row pitch W+2 and the +1 column
// shift make them unaligned by construction), the reflected border entries are written / folded by the threads that
// own rows 1, H-2 and the first / last quad of a row, and up_cat_pad evaluates ELU once per source element instead
// of once per upsampled copy.
struct __attribute__((packed, aligned(4))) quad_u { float x, y, z, w; };
__device__ __forceinline__ float4 load4u(const float* p) {
    const quad_u v = *reinterpret_cast<const quad_u*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store4u(float* p, const float4 v) {
    quad_u q;
    q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
    *reinterpret_cast<quad_u*>(p) = q;
}

// writes interior quad (Y, X..X+3) of a padded plane and every border entry that reflects onto it
__device__ __forceinline__ void put_quad(float* __restrict__ op, const float4 v, int Y, int X, int H, int W) {
    const int PW = W + 2;
    const bool first = X == 0, last = X + 4 == W;
    float* r = op + (size_t)(Y + 1) * PW + X;
    store4u(r + 1, v);
    if (first) r[0] = v.y;
    if (last) r[5] = v.z;
    if (Y == 1) {
        float* t = op + X;
        store4u(t + 1, v);
        if (first) t[0] = v.y;
        if (last) t[5] = v.z;
    }
    if (Y == H - 2) {
        float* t = op + (size_t)(H + 1) * PW + X;
        store4u(t + 1, v);
        if (first) t[0] = v.y;
        if (last) t[5] = v.z;
    }
}

// gradient of interior quad (Y, X..X+3): the adjoint of ReflectionPad2d(1) adds padded row 0 / H+1 onto interior row
// 1 / H-2 and padded column 0 / W+1 onto interior column 1 / W-2.  Written as at most three row reads with one extra
// element each at the row ends -- the generic per-element fold2d loops, taken by every wave that holds a first / last
// quad of a row, made the four-wide backward kernels run at 3.7 TB/s.
__device__ __forceinline__ float4 row_quad(const float* __restrict__ gp, int pr, int X, int W) {
    const float* r = gp + (size_t)pr * (W + 2);
    float4 q = load4u(r + X + 1);
    if (X == 0) q.y += r[0];
    if (X + 4 == W) q.z += r[W + 1];
    return q;
}
__device__ __forceinline__ float4 fold_quad(const float* __restrict__ gp, int Y, int X, int H, int W) {
    float4 v = row_quad(gp, Y + 1, X, W);
    if (Y == 1) {
        const float4 t = row_quad(gp, 0, X, W);
        v = make_float4(v.x + t.x, v.y + t.y, v.z + t.z, v.w + t.w);
    }
    if (Y == H - 2) {
        const float4 t = row_quad(gp, H + 1, X, W);
        v = make_float4(v.x + t.x, v.y + t.y, v.z + t.z, v.w + t.w);
    }
    return v;
}

// blocks [0, nA): y planes, a thread = two source elements (i, j0), (i, j0+1) -> two output rows of four;
// blocks [nA, ..): skip planes, a thread = one quad.  bpa / bpb = blocks per plane of each part.
__global__ __launch_bounds__(NT) void up_cat_pad_fwd4_kernel(const float* __restrict__ y, const float* __restrict__ skip,
                                                             int C1, int C2, int h, int w, int nA, int bpa, int bpb,
                                                             float* __restrict__ out) {
    const int H = 2 * h, W = 2 * w, C = C1 + C2;
    const size_t pplane = (size_t)(H + 2) * (W + 2);
    if ((int)blockIdx.x < nA) {
        const int plane = blockIdx.x / bpa, t = (blockIdx.x - plane * bpa) * NT + threadIdx.x;
        const int hw2 = w >> 1;
        if (t >= h * hw2) return;
        const int b = plane / C1, c = plane - b * C1;
        const int i = t / hw2, j0 = (t - i * hw2) * 2;
        const float2 yy = *reinterpret_cast<const float2*>(y + ((size_t)plane * h + i) * w + j0);
        const float e0 = elu_f(yy.x), e1 = elu_f(yy.y);
        const float4 v = make_float4(e0, e0, e1, e1);
        float* op = out + (size_t)(b * C + c) * pplane;
        put_quad(op, v, 2 * i, 2 * j0, H, W);
        put_quad(op, v, 2 * i + 1, 2 * j0, H, W);
    } else {
        const int bi = blockIdx.x - nA, q = bi / bpb, t = (bi - q * bpb) * NT + threadIdx.x;
        const int idx = t * 4;
        if (idx >= H * W) return;
        const int b = q / C2, c = q - b * C2;
        const int Y = idx / W, X = idx - Y * W;
        const float4 v = *reinterpret_cast<const float4*>(skip + (size_t)q * H * W + idx);
        put_quad(out + (size_t)(b * C + C1 + c) * pplane, v, Y, X, H, W);
    }
}

__global__ __launch_bounds__(NT) void up_cat_pad_bwd4_kernel(const float* __restrict__ y, const float* __restrict__ g_out,
                                                             int C1, int C2, int h, int w, int nA, int bpa, int bpb,
                                                             float* __restrict__ g_y, float* __restrict__ g_skip) {
    const int H = 2 * h, W = 2 * w, C = C1 + C2;
    const size_t pplane = (size_t)(H + 2) * (W + 2);
    if ((int)blockIdx.x < nA) {
        const int plane = blockIdx.x / bpa, t = (blockIdx.x - plane * bpa) * NT + threadIdx.x;
        const int hw2 = w >> 1;
        if (t >= h * hw2) return;
        const int b = plane / C1, c = plane - b * C1;
        const int i = t / hw2, j0 = (t - i * hw2) * 2;
        const float* gp = g_out + (size_t)(b * C + c) * pplane;
        const float4 r0 = fold_quad(gp, 2 * i, 2 * j0, H, W), r1 = fold_quad(gp, 2 * i + 1, 2 * j0, H, W);
        const size_t o = ((size_t)plane * h + i) * w + j0;
        const float2 yy = *reinterpret_cast<const float2*>(y + o);
        *reinterpret_cast<float2*>(g_y + o) = make_float2((r0.x + r0.y + r1.x + r1.y) * elu_grad(yy.x),
                                                          (r0.z + r0.w + r1.z + r1.w) * elu_grad(yy.y));
    } else {
        const int bi = blockIdx.x - nA, q = bi / bpb, t = (bi - q * bpb) * NT + threadIdx.x;
        const int idx = t * 4;
        if (idx >= H * W) return;
        const int b = q / C2, c = q - b * C2;
        const int Y = idx / W, X = idx - Y * W;
        *reinterpret_cast<float4*>(g_skip + (size_t)q * H * W + idx) =
            fold_quad(g_out + (size_t)(b * C + C1 + c) * pplane, Y, X, H, W);
    }
}

__global__ __launch_bounds__(NT) void elu_pad_fwd4_kernel(const float* __restrict__ z, int H, int W, int apply_elu,
                                                          float* __restrict__ out) {
    const int idx = (blockIdx.x * NT + threadIdx.x) * 4;
    if (idx >= H * W) return;
    const int Y = idx / W, X = idx - Y * W;
    float4 v = *reinterpret_cast<const float4*>(z + (size_t)blockIdx.y * H * W + idx);
    if (apply_elu) v = make_float4(elu_f(v.x), elu_f(v.y), elu_f(v.z), elu_f(v.w));
    put_quad(out + (size_t)blockIdx.y * (H + 2) * (W + 2), v, Y, X, H, W);
}

__global__ __launch_bounds__(NT) void elu_pad_bwd4_kernel(const float* __restrict__ z, const float* __restrict__ g_out,
                                                          int H, int W, int apply_elu, float* __restrict__ g_z) {
    const int idx = (blockIdx.x * NT + threadIdx.x) * 4;
    if (idx >= H * W) return;
    const int Y = idx / W, X = idx - Y * W;
    const size_t base = (size_t)blockIdx.y * H * W + idx;
    float4 g = fold_quad(g_out + (size_t)blockIdx.y * (H + 2) * (W + 2), Y, X, H, W);
    if (apply_elu) {
        const float4 zz = *reinterpret_cast<const float4*>(z + base);
        g = make_float4(g.x * elu_grad(zz.x), g.y * elu_grad(zz.y), g.z * elu_grad(zz.z), g.w * elu_grad(zz.w));
    }
    *reinterpret_cast<float4*>(g_z + base) = g;
}

inline unsigned blocks_for(int n) { return (unsigned)((n + NT - 1) / NT); }

// DMH_GLUE_PAIRS=1 forces the two-wide kernels (timing comparisons, tools/glue_bench.py)
inline bool use_quads() {
    static const bool q = [] { const char* e = getenv("DMH_GLUE_PAIRS"); return !(e && e[0] == '1'); }();
    return q;
}

}  // namespace

extern "C" {

int dmh_dec_up_cat_pad_fwd(const float* y, const float* skip, int B, int C1, int C2, int h, int w, float* out,
                           void* stream) {
    DMH_REQUIRE(y && out && (skip || C2 == 0), "null pointer");
    DMH_REQUIRE(B > 0 && C1 > 0 && C2 >= 0 && h >= 1 && w >= 1, "bad sizes");
    DMH_REQUIRE((int64_t)B * (C1 + C2) <= 65535 && (int64_t)(2 * h + 2) * (2 * w + 2) < (1 << 30), "tensor too large");
    if ((w & 1) == 0 && use_quads()) {
        const int bpa = (int)blocks_for(h * (w / 2)), bpb = (int)blocks_for(h * w);
        const int64_t nA = (int64_t)B * C1 * bpa, nB = (int64_t)B * C2 * bpb;
        DMH_REQUIRE(nA + nB < ((int64_t)1 << 31), "grid too large");
        hipLaunchKernelGGL(up_cat_pad_fwd4_kernel, dim3((unsigned)(nA + nB)), dim3(NT), 0, (hipStream_t)stream, y, skip, C1,
                           C2, h, w, (int)nA, bpa, bpb, out);
    } else {
        hipLaunchKernelGGL(up_cat_pad_fwd_kernel<true>, dim3(blocks_for((h + 1) * (2 * w + 2)), B * (C1 + C2)), dim3(NT),
                           0, (hipStream_t)stream, y, skip, C1, C2, h, w, out);
    }
    return check_launch("dmh_dec_up_cat_pad_fwd");
}

int dmh_dec_up_cat_pad_bwd(const float* y, const float* g_out, int B, int C1, int C2, int h, int w, float* g_y,
                           float* g_skip, void* stream) {
    DMH_REQUIRE(y && g_out && g_y, "null pointer");
    DMH_REQUIRE(B > 0 && C1 > 0 && C2 >= 0 && h >= 1 && w >= 1, "bad sizes");
    const int planes = B * C1 + (g_skip ? B * C2 : 0);
    DMH_REQUIRE(planes <= 65535 && (int64_t)(2 * h + 2) * (2 * w + 2) < (1 << 30), "tensor too large");
    if ((w & 1) == 0 && use_quads()) {
        const int bpa = (int)blocks_for(h * (w / 2)), bpb = (int)blocks_for(h * w);
        const int64_t nA = (int64_t)B * C1 * bpa, nB = g_skip ? (int64_t)B * C2 * bpb : 0;
        DMH_REQUIRE(nA + nB < ((int64_t)1 << 31), "grid too large");
        hipLaunchKernelGGL(up_cat_pad_bwd4_kernel, dim3((unsigned)(nA + nB)), dim3(NT), 0, (hipStream_t)stream, y, g_out,
                           C1, C2, h, w, (int)nA, bpa, bpb, g_y, g_skip);
    } else {
        const int per_plane = (g_skip && C2 > 0) ? 2 * h * w : h * w;   // threads: one per g_y element / per g_skip pair
        hipLaunchKernelGGL(up_cat_pad_bwd_kernel<true>, dim3(blocks_for(per_plane), planes), dim3(NT), 0,
                           (hipStream_t)stream, y, g_out, B, C1, C2, h, w, g_y, g_skip);
    }
    return check_launch("dmh_dec_up_cat_pad_bwd");
}

int dmh_elu_pad_fwd(const float* z, int B, int C, int H, int W, int apply_elu, float* out, void* stream) {
    DMH_REQUIRE(z && out, "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && H >= 2 && W >= 2, "bad sizes");
    DMH_REQUIRE((int64_t)B * C <= 65535 && (int64_t)(H + 2) * (W + 2) < (1 << 30), "tensor too large");
    if ((W & 3) == 0 && use_quads())
        hipLaunchKernelGGL(elu_pad_fwd4_kernel, dim3(blocks_for(H * W / 4), B * C), dim3(NT), 0, (hipStream_t)stream, z, H,
                           W, apply_elu, out);
    else if ((W & 1) == 0)
        hipLaunchKernelGGL(elu_pad_fwd_kernel<true>, dim3(blocks_for((H + 2) * (W + 2) / 2), B * C), dim3(NT), 0,
                           (hipStream_t)stream, z, H, W, apply_elu, out);
    else
        hipLaunchKernelGGL(elu_pad_fwd_kernel<false>, dim3(blocks_for((H + 2) * (W + 2)), B * C), dim3(NT), 0,
                           (hipStream_t)stream, z, H, W, apply_elu, out);
    return check_launch("dmh_elu_pad_fwd");
}

int dmh_elu_pad_bwd(const float* z, const float* g_out, int B, int C, int H, int W, int apply_elu, float* g_z,
                    void* stream) {
    DMH_REQUIRE(z && g_out && g_z, "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && H >= 2 && W >= 2, "bad sizes");
    DMH_REQUIRE((int64_t)B * C <= 65535 && (int64_t)(H + 2) * (W + 2) < (1 << 30), "tensor too large");
    if ((W & 3) == 0 && use_quads())
        hipLaunchKernelGGL(elu_pad_bwd4_kernel, dim3(blocks_for(H * W / 4), B * C), dim3(NT), 0, (hipStream_t)stream, z,
                           g_out, H, W, apply_elu, g_z);
    else if ((W & 1) == 0)
        hipLaunchKernelGGL(elu_pad_bwd_kernel<true>, dim3(blocks_for(H * W / 2), B * C), dim3(NT), 0, (hipStream_t)stream,
                           z, g_out, H, W, apply_elu, g_z);
    else
        hipLaunchKernelGGL(elu_pad_bwd_kernel<false>, dim3(blocks_for(H * W), B * C), dim3(NT), 0, (hipStream_t)stream,
                           z, g_out, H, W, apply_elu, g_z);
    return check_launch("dmh_elu_pad_bwd");
}

}  // extern "C"
