// K13 -- the disparity heads: 3x3 stride-1 convolution to ONE output channel (MD2/networks/depth_decoder.py:43-44
// dispconv, MD2/layers.py:127-141 Conv3x3 on the reflection-padded decoder feature), forward.
//
// With a single output channel an MFMA formulation wastes 15 of its 16 rows (K11 takes 224 us for 16 -> 1 at 320x1024,
// MIOpen 465 us) although the layer is a 254 MB streaming read.  Here it is plain vector FMAs: the input tile of 16
// channels x 10 x 66 is staged in LDS (zero padding while staging), a thread owns two horizontally adjacent output
// pixels and per channel reads 3 rows x 4 columns as six 8-byte LDS reads for 18 FMAs, with the 9 filter taps of the
// channel fed as SGPR operands from scalar loads.  Channels are processed 16 at a time (any multiple of 16).
//
// Weight gradient (train pass): dW[c][ky][kx] = sum_{b,y,x} g[b][y][x] * x[b][c][y+ky-pad][x+kx-pad] and db = sum g -- a
// reduction over 10 M pixels per channel that MIOpen runs as a 1-output-channel implicit GEMM at 2 TFLOP/s (1.65 ms for
// 16 -> 1 at 320x1024, batch 32; the four heads 3.0 ms per step).  head_wrw_kernel: a wave owns a strip of 62 columns
// x 40 rows of one image with the strip's 40 output gradients in registers and walks the 42 input rows of the strip,
// channel by channel, 14 rows per batch of loads (1 coalesced load, 2 DPP lane shifts and 9 FMAs per row); nine wave sums per channel go to a
// per-strip partial, and head_wrw_reduce_kernel adds the partials in a fixed order (deterministic, no atomics).
// Small maps split their channels over several waves per strip.
#include <stdlib.h>

#include "common.hpp"

using namespace dmh;

namespace {

constexpr int NT = 256;
constexpr int TH = 8, TW = 64;          // output tile: 8 rows x 64 columns = 256 threads x 2 pixels
constexpr int KC = 16;                  // channels per LDS stage
constexpr int RH = TH + 2, RW = TW + 2;

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3))) void head_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, int C, int H, int W, int Ho, int Wo,
                                                       int pad, int gx, int gy, float* __restrict__ y) {
    __shared__ float tile[KC * RH * RW];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int gxi = bid % gx;  bid /= gx;
    const int gyi = bid % gy;
    const int b = bid / gy;
    const int oy0 = gyi * TH, ox0 = gxi * TW;
    const size_t HW = (size_t)H * W;
    const float* xb = x + (size_t)b * C * HW;
    const int r = tid >> 5, c2 = tid & 31;                  // output row in the tile, column pair
    float acc0 = 0.f, acc1 = 0.f;
    constexpr int PER_T = (KC * RH * RW + NT - 1) / NT;
    for (int c0 = 0; c0 < C; c0 += KC) {
        float stage[PER_T];
#pragma unroll
        for (int k = 0; k < PER_T; ++k) {
            const int e = tid + NT * k;
            const int c = e / (RH * RW), rem = e - c * (RH * RW), rr = rem / RW, xx = rem - rr * RW;
            const int iy = oy0 - pad + rr, ix = ox0 - pad + xx;
            const bool ok = e < KC * RH * RW && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float v = xb[(size_t)(c0 + (e < KC * RH * RW ? c : 0)) * HW + (size_t)min(max(iy, 0), H - 1) * W +
                               min(max(ix, 0), W - 1)];
            stage[k] = ok ? v : 0.f;
        }
        __syncthreads();                                    // the previous stage has been consumed
#pragma unroll
        for (int k = 0; k < PER_T; ++k) {
            const int e = tid + NT * k;
            if (e < KC * RH * RW) tile[e] = stage[k];
        }
        __syncthreads();
        const float* tp = tile + r * RW + 2 * c2;
        const float* wc = w + (size_t)c0 * 9;               // uniform: scalar loads
#pragma unroll 4
        for (int c = 0; c < KC; ++c) {     // 4 channels in flight: fully unrolled it holds 192 registers of LDS data
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float2 lo = *reinterpret_cast<const float2*>(tp + c * (RH * RW) + ky * RW);
                const float2 hi = *reinterpret_cast<const float2*>(tp + c * (RH * RW) + ky * RW + 2);
                const float w0 = wc[c * 9 + ky * 3], w1 = wc[c * 9 + ky * 3 + 1], w2 = wc[c * 9 + ky * 3 + 2];
                acc0 = fmaf(lo.x, w0, acc0); acc0 = fmaf(lo.y, w1, acc0); acc0 = fmaf(hi.x, w2, acc0);
                acc1 = fmaf(lo.y, w0, acc1); acc1 = fmaf(hi.x, w1, acc1); acc1 = fmaf(hi.y, w2, acc1);
            }
        }
    }
    const int oy = oy0 + r, ox = ox0 + 2 * c2;
    if (oy < Ho && ox < Wo) {
        const float bs = bias ? bias[0] : 0.f;
        float* yp = y + ((size_t)b * Ho + oy) * Wo + ox;
        if (ox + 1 < Wo && (Wo & 1) == 0) {
            *reinterpret_cast<float2*>(yp) = make_float2(acc0 + bs, acc1 + bs);
        } else {
            yp[0] = acc0 + bs;
            if (ox + 1 < Wo) yp[1] = acc1 + bs;
        }
    }
}

// ---------------------------------------------------------------------------------------------- forward, strips
// pad 0 (the heads run on the reflection-padded decoder features), C a multiple of 4.  The LDS-tile kernel above reaches
// 1.2 TB/s (its staging loop is index arithmetic and a barrier per 16 channels); a head is a 254 MB streaming read.  Here a
// wave owns 62 output columns x FR rows and a quarter of the channels: per channel it walks the strip's FR + 2 input rows
// with ONE coalesced buffer load per row (row offset in an SGPR), takes the two right-hand neighbours by DPP lane shifts
// and does 9 FMAs against taps held in registers into FR row accumulators; the four waves of a workgroup (channel
// quarters) are added through LDS in a fixed order and the bias (and optionally the sigmoid of the disparity head) is
// applied on the way out.

// Sum over the 64 lanes in 11 instructions (wave_sum's six __shfl_down steps are LDS-crossbar permutes of ~100 cycles
// each: nine sums per channel cost more than the channel's arithmetic): quad_perm / row_half_mirror / row_mirror DPP adds
// leave every lane of a 16-lane row with the row's total; four readlanes add the rows.  Fixed order; uniform result.
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_perm<0xB1>(v);         // quad_perm [1,0,3,2]
    v += dpp_perm<0x4E>(v);         // quad_perm [2,3,0,1]
    v += dpp_perm<0x141>(v);        // row_half_mirror
    v += dpp_perm<0x140>(v);        // row_mirror
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

constexpr int FR = 40;                  // output rows per strip
constexpr int FCOLS = 62;               // output columns per strip: lanes 62, 63 only feed their left neighbours
constexpr int FBATCH = 14;              // input rows per batch of loads ((FR + 2) % FBATCH == 0)
static_assert((FR + 2) % FBATCH == 0 && FR % 4 == 0, "strip geometry");

struct HFArgs {
    const float *x, *w, *bias;
    float* y;
    int B, C, H, W, Ho, Wo, sx, sy, sigmoid;
};

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4))) void head_fwd_strip_kernel(const HFArgs a) {
    __shared__ float red[NT / 64][FR][64];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int s = blockIdx.x;
    const int xs = s % a.sx;
    s /= a.sx;
    const int ys = s % a.sy, b = s / a.sy;
    const int ox0 = xs * FCOLS, oy0 = ys * FR;
    const unsigned xo = (unsigned)min(ox0 + lane, a.W - 1) * 4u;
    const rsrc_t rx = make_rsrc(a.x + (size_t)b * a.C * a.H * a.W, (unsigned)a.C * (unsigned)(a.H * a.W) * 4u);
    float acc[FR];
#pragma unroll
    for (int r = 0; r < FR; ++r) acc[r] = 0.f;
    const int cpg = a.C >> 2;
    for (int c = wv * cpg; c < (wv + 1) * cpg; ++c) {
        float wt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = a.w[c * 9 + t];         // wave-uniform
#pragma unroll
        for (int i0 = 0; i0 < FR + 2; i0 += FBATCH) {
            float rows[FBATCH];
#pragma unroll
            for (int j = 0; j < FBATCH; ++j) {
                const unsigned so = (unsigned)((c * a.H + min(oy0 + i0 + j, a.H - 1)) * a.W) * 4u;
                rows[j] = ldb(rx, xo, so);
            }
#pragma unroll
            for (int j = 0; j < FBATCH; ++j) {
                const float x0 = rows[j], x1 = lane_next(x0), x2 = lane_next(x1);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int r = i0 + j - ky;
                    if (r >= 0 && r < FR)
                        acc[r] = fmaf(wt[ky * 3 + 2], x2, fmaf(wt[ky * 3 + 1], x1, fmaf(wt[ky * 3], x0, acc[r])));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int r = 0; r < FR; ++r) red[wv][r][lane] = acc[r];
    __syncthreads();
    const float bs = a.bias ? a.bias[0] : 0.f;
    const int ox = ox0 + lane;
#pragma unroll
    for (int k = 0; k < FR / 4; ++k) {
        const int r = wv * (FR / 4) + k, oy = oy0 + r;
        float v = ((red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane])) + bs;
        if (a.sigmoid) v = 1.f / (1.f + __expf(-v));
        if (lane < FCOLS && ox < a.Wo && oy < a.Ho) a.y[((size_t)b * a.Ho + oy) * a.Wo + ox] = v;
    }
}

// ---------------------------------------------------------------------------------------------- backward-data, strips
// g_x[b][c][iy][ix] = sum_{ky,kx} w[c][ky][kx] * g[b][iy - ky][ix - kx]   (pad 0: g_x is (Ho + 2) x (Wo + 2)), a pure
// streaming write of C planes from one gradient plane.  A wave owns 62 columns x BR rows of g_x and a quarter of the
// channels: the BR + 2 gradient rows of the strip and their two lane-shifted copies stay in registers, every channel is
// 9 FMAs and one coalesced store per row.
constexpr int BR = 20;

struct HBArgs {
    const float *g, *w;
    float* gx;
    int B, C, H, W, Ho, Wo, sx, sy, nsplit, nwaves;     // nsplit: waves that share a strip (channel split), 1 or 4
};

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3))) void head_bwd_strip_kernel(const HBArgs a) {
    const int lane = threadIdx.x & 63;
    const int gw = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (NT / 64) + (threadIdx.x >> 6)));
    if (gw >= a.nwaves) return;
    const int wv = gw % a.nsplit;
    int s = gw / a.nsplit;
    const int xs = s % a.sx;
    s /= a.sx;
    const int ys = s % a.sy, b = s / a.sy;
    const int ix0 = xs * FCOLS, iy0 = ys * BR;
    // lane L holds gradient column ix0 - 2 + L; output column ix0 + L uses lanes L + 2, L + 1, L for kx = 0, 1, 2
    const int gc = ix0 - 2 + lane;
    const bool cok = gc >= 0 && gc < a.Wo;
    const float* gb = a.g + (size_t)b * a.Ho * a.Wo + min(max(gc, 0), a.Wo - 1);
    float g0[BR + 2], g1[BR + 2], g2[BR + 2];     // rows iy0 - 2 .. iy0 + BR - 1 of g; shifted by 0 / 1 / 2 lanes
#pragma unroll
    for (int j = 0; j < BR + 2; ++j) {
        const int gr = iy0 - 2 + j;
        const float v = gb[(size_t)min(max(gr, 0), a.Ho - 1) * a.Wo];
        g0[j] = (cok && gr >= 0 && gr < a.Ho) ? v : 0.f;
    }
#pragma unroll
    for (int j = 0; j < BR + 2; ++j) {
        g1[j] = lane_next(g0[j]);
        g2[j] = lane_next(g1[j]);
    }
    const int ix = ix0 + lane;
    const bool st = lane < FCOLS && ix < a.W;
    const int cpg = a.C / a.nsplit;
    const size_t HW = (size_t)a.H * a.W;
    for (int c = wv * cpg; c < (wv + 1) * cpg; ++c) {
        float wt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = a.w[c * 9 + t];         // wave-uniform
        float* out = a.gx + ((size_t)b * a.C + c) * HW + ix;
#pragma unroll
        for (int r = 0; r < BR; ++r) {
            // output row iy0 + r: gradient rows iy0 + r - ky = register j = r + 2 - ky
            float v = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int j = r + 2 - ky;
                v = fmaf(wt[ky * 3 + 2], g0[j], fmaf(wt[ky * 3 + 1], g1[j], fmaf(wt[ky * 3], g2[j], v)));
            }
            if (st && iy0 + r < a.H) out[(size_t)(iy0 + r) * a.W] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------- weight gradient
constexpr int WRB = 40;                 // rows per strip (gradient values kept in registers)


struct HWArgs {
    const float *x, *g;
    float* part;                        // [strip][C * 9 + 1]
    int B, C, H, W, Ho, Wo, pad, sx, sy, cg, cpg;
};

// PADDED = false (pad 0, the heads on the reflection-padded decoder features): every tap of a valid output pixel lies
// inside the image, no masks
template <bool PADDED>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3))) void head_wrw_kernel(const HWArgs a) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (NT / 64) + (threadIdx.x >> 6)));
    if (wid >= a.B * a.sy * a.sx * a.cg) return;
    const int cgi = wid % a.cg, s = wid / a.cg;
    const int xs = s % a.sx, q = s / a.sx, ys = q % a.sy, b = q / a.sy;
    const int ox = xs * FCOLS + lane, oy0 = ys * WRB;          // 62 output columns per strip: lanes 62, 63 only feed
    const bool colok = lane < FCOLS && ox < a.Wo;              // their left neighbours' taps
    float gr[WRB];
    float sb = 0.f;
#pragma unroll
    for (int r = 0; r < WRB; ++r) {
        const int oy = oy0 + r;
        gr[r] = (colok && oy < a.Ho) ? a.g[((size_t)b * a.Ho + min(oy, a.Ho - 1)) * a.Wo + min(ox, a.Wo - 1)] : 0.f;
        sb += gr[r];
    }
    const int stride = a.C * 9 + 1;
    float* pp = a.part + (size_t)s * stride;
    if (cgi == 0) {
        sb = wave_sum_dpp(sb);
        if (lane == 0) pp[a.C * 9] = sb;
    }
    // lane L holds input column ox - pad of every row; the taps kx = 1, 2 are the next two lanes' values (DPP)
    const int ixl = ox - a.pad;
    const float xm = (ixl >= 0 && ixl < a.W) ? 1.f : 0.f;
    const unsigned xo = (unsigned)min(max(ixl, 0), a.W - 1) * 4u;
    const rsrc_t rx = make_rsrc(a.x + (size_t)b * a.C * a.H * a.W, (unsigned)a.C * (unsigned)(a.H * a.W) * 4u);
    for (int c = cgi * a.cpg; c < (cgi + 1) * a.cpg; ++c) {
        float acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = 0.f;
        // input row i of the strip (i = 0 .. WRB + 1) meets output rows i, i - 1, i - 2 through taps ky = 0, 1, 2.  Rows are
        // fetched RBATCH at a time (all loads of a batch in flight together) and consumed from registers.
        constexpr int RBATCH = 14;
        static_assert((WRB + 2) % RBATCH == 0, "row batches");
#pragma unroll
        for (int i0 = 0; i0 < WRB + 2; i0 += RBATCH) {
            float rows[RBATCH];
#pragma unroll
            for (int j = 0; j < RBATCH; ++j) {
                const int iy = oy0 - a.pad + i0 + j;
                // zero padding as a multiplication by a 0/1 factor on a clamped (always valid, finite) address: a select
                // on the wave-uniform row test is compiled into a branch around every load, which serialises them
                const float ym = (iy >= 0 && iy < a.H) ? 1.f : 0.f;
                // buffer loads: the row offset is wave-uniform (SGPR), the lane part three fixed registers -- with plain
                // pointers the compiler hoists 126 per-lane addresses out of the channel loop and spills them
                const unsigned so = (unsigned)((c * a.H + min(max(iy, 0), a.H - 1)) * a.W) * 4u;
                const float t = ldb(rx, xo, so);
                // (t * xm) * ym, not t * (xm * ym): the latter is invariant in c and would be hoisted into 42 registers
                rows[j] = PADDED ? (t * xm) * ym : t;
            }
#pragma unroll
            for (int j = 0; j < RBATCH; ++j) {
                const float x0 = rows[j], x1 = lane_next(x0), x2 = lane_next(x1);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int r = i0 + j - ky;
                    if (r >= 0 && r < WRB) {
                        acc[ky * 3] = fmaf(gr[r], x0, acc[ky * 3]);
                        acc[ky * 3 + 1] = fmaf(gr[r], x1, acc[ky * 3 + 1]);
                        acc[ky * 3 + 2] = fmaf(gr[r], x2, acc[ky * 3 + 2]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the next batch's loads behind this batch's arithmetic: hoisting
                                                    // all 126 loads of a channel costs 512 registers (one wave per SIMD)
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v = wave_sum_dpp(acc[t]);
            if (lane == 0) pp[c * 9 + t] = v;
        }
    }
}

// out[j] = sum over strips of part[strip][j]: one workgroup per j, fixed order
__global__ __launch_bounds__(NT) void head_wrw_reduce_kernel(const float* __restrict__ part, int nstrips, int stride,
                                                             float* __restrict__ gw, float* __restrict__ gb) {
    __shared__ float red[NT / WAVE];
    const int j = blockIdx.x;
    float acc = 0.f;
    for (int s = threadIdx.x; s < nstrips; s += NT) acc += part[(size_t)s * stride + j];
    const float t = block_sum<NT>(acc, red);
    if (threadIdx.x == 0) {
        if (j < stride - 1)
            gw[j] = t;
        else if (gb)
            gb[0] = t;
    }
}

// DMH_HEAD_TILE=1 keeps the LDS-tile forward kernel (timing comparisons)
inline bool force_tile_kernel() {
    static const bool f = [] { const char* e = getenv("DMH_HEAD_TILE"); return e && e[0] == '1'; }();
    return f;
}

struct WrwGeo { int sx, sy, cg; long long strips; };
inline WrwGeo wrw_geo(int B, int C, int Ho, int Wo) {
    WrwGeo g;
    g.sx = (Wo + FCOLS - 1) / FCOLS;
    g.sy = (Ho + WRB - 1) / WRB;
    g.strips = (long long)B * g.sx * g.sy;
    g.cg = 1;
    while (g.strips * g.cg < 4096 && g.cg * 2 <= C && C % (g.cg * 2) == 0) g.cg *= 2;   // small maps: split the channels
    return g;
}

}  // namespace

extern "C" {

int dmh_conv3x3_head_bwd_data(const float* g, const float* w, int B, int C, int H, int W, float* g_x, void* stream) {
    DMH_REQUIRE(g && w && g_x, "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && H >= 3 && W >= 3, "C must be a multiple of 4, the input at least 3 x 3");
    DMH_REQUIRE((int64_t)B * C * H * W < ((int64_t)1 << 40), "tensor too large");
    HBArgs a;
    a.g = g;
    a.w = w;
    a.gx = g_x;
    a.B = B;
    a.C = C;
    a.H = H;
    a.W = W;
    a.Ho = H - 2;
    a.Wo = W - 2;
    a.sx = (W + FCOLS - 1) / FCOLS;
    a.sy = (H + BR - 1) / BR;
    const long long strips = (long long)B * a.sx * a.sy;
    a.nsplit = strips >= 2048 ? 1 : 4;          // few strips: split the channels over four waves to fill the chip
    DMH_REQUIRE(strips * a.nsplit < (1ll << 31), "grid too large");
    a.nwaves = (int)(strips * a.nsplit);
    hipLaunchKernelGGL(head_bwd_strip_kernel, dim3((unsigned)((a.nwaves + NT / 64 - 1) / (NT / 64))), dim3(NT), 0,
                       (hipStream_t)stream, a);
    return check_launch("dmh_conv3x3_head_bwd_data");
}

int64_t dmh_conv3x3_head_wrw_partials_size(int B, int C, int H, int W, int pad) {
    if (B <= 0 || C <= 0 || pad < 0 || pad > 2 || H + 2 * pad - 2 < 1 || W + 2 * pad - 2 < 1) return 0;
    return wrw_geo(B, C, H + 2 * pad - 2, W + 2 * pad - 2).strips * ((int64_t)C * 9 + 1);
}

int dmh_conv3x3_head_wrw(const float* x, const float* g, int B, int C, int H, int W, int pad, float* partials, float* g_w,
                         float* g_b, void* stream) {
    DMH_REQUIRE(x && g && partials && g_w, "null pointer");
    DMH_REQUIRE(B > 0 && C > 0 && pad >= 0 && pad <= 2, "bad sizes");
    DMH_REQUIRE((int64_t)C * H * W < ((int64_t)1 << 29), "image too large (32-bit byte offsets)");
    HWArgs a;
    a.x = x;
    a.g = g;
    a.part = partials;
    a.B = B;
    a.C = C;
    a.H = H;
    a.W = W;
    a.Ho = H + 2 * pad - 2;
    a.Wo = W + 2 * pad - 2;
    DMH_REQUIRE(a.Ho >= 1 && a.Wo >= 1, "image smaller than the filter");
    a.pad = pad;
    const WrwGeo geo = wrw_geo(B, C, a.Ho, a.Wo);
    a.sx = geo.sx;
    a.sy = geo.sy;
    a.cg = geo.cg;
    a.cpg = C / geo.cg;
    const long long waves = geo.strips * geo.cg;
    DMH_REQUIRE(waves < (1ll << 31) && geo.strips < (1ll << 24), "grid too large");
    const dim3 grid((unsigned)((waves + NT / 64 - 1) / (NT / 64)));
    if (pad == 0)
        hipLaunchKernelGGL(head_wrw_kernel<false>, grid, dim3(NT), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(head_wrw_kernel<true>, grid, dim3(NT), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(head_wrw_reduce_kernel, dim3((unsigned)(C * 9 + 1)), dim3(NT), 0, (hipStream_t)stream, partials,
                       (int)geo.strips, C * 9 + 1, g_w, g_b);
    return check_launch("dmh_conv3x3_head_wrw");
}

int dmh_conv3x3_head(const float* x, const float* w, const float* bias, int B, int C, int H, int W, int pad, float* y,
                     void* stream) {
    DMH_REQUIRE(x && w && y, "null pointer");
    DMH_REQUIRE(pad >= 0 && pad <= 2, "pad must be 0, 1 or 2");
    DMH_REQUIRE(B > 0 && C > 0 && (C % KC == 0 || (pad == 0 && C % 4 == 0 && !force_tile_kernel())),
                "input channels must be a multiple of 16 (pad 0: of 4)");
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    DMH_REQUIRE(Ho >= 1 && Wo >= 1, "image smaller than the filter");
    DMH_REQUIRE((int64_t)C * H * W < ((int64_t)1 << 31), "image too large");
    if (pad == 0 && (int64_t)C * H * W < ((int64_t)1 << 29) && !force_tile_kernel()) {     // the strip kernel
        HFArgs a;
        a.x = x;
        a.w = w;
        a.bias = bias;
        a.y = y;
        a.B = B;
        a.C = C;
        a.H = H;
        a.W = W;
        a.Ho = Ho;
        a.Wo = Wo;
        a.sx = (Wo + FCOLS - 1) / FCOLS;
        a.sy = (Ho + FR - 1) / FR;
        a.sigmoid = 0;
        const long long strips = (long long)B * a.sx * a.sy;
        DMH_REQUIRE(strips < (1ll << 31), "grid too large");
        hipLaunchKernelGGL(head_fwd_strip_kernel, dim3((unsigned)strips), dim3(NT), 0, (hipStream_t)stream, a);
        return check_launch("dmh_conv3x3_head");
    }
    const int gx = (Wo + TW - 1) / TW, gy = (Ho + TH - 1) / TH;
    const long long blocks = (long long)B * gx * gy;
    DMH_REQUIRE(blocks < (1ll << 31), "grid too large");
    hipLaunchKernelGGL(head_conv_kernel, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, x, w, bias, C, H, W, Ho,
                       Wo, pad, gx, gy, y);
    return check_launch("dmh_conv3x3_head");
}

}  // extern "C"
