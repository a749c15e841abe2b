// K16 -- weight gradient of the 3x3 stride-1 convolutions with 16 output channels at full image resolution: the last
// decoder stage, MD2/networks/depth_decoder.py:38-41 upconv(0,1) 16->16 @320x1024 and upconv(0,0) 32->16 @160x512
// (MD2/layers.py:127-141 Conv3x3), train pass:
//     dW[k][c][ky][kx] = sum_{b,y,x} g[b][k][y][x] * zero_pad(x)[b][c][y+ky][x+kx],     db[k] = sum g[b][k][y][x]
// A reduction over 10 M pixels per (k, c, tap) that MIOpen's implicit-GEMM weight-gradient kernels run at 25-28 TFLOP/s
// (1.9 and 0.87 ms per train pass).  Here the pixel axis is the k dimension of v_mfma_f32_16x16x4_f32 (exact fp32):
//     D[k 16][c 16] += G[k 16][4 pixels] * X[4 pixels][c 16]                              per tap and 16-channel block,
// four consecutive pixels of a row per MFMA, the gradient operand shared by the nine taps.  A workgroup stages a
// TR x 64 pixel tile of g (16 planes) and the matching (TR+2) x 66 tile of x (C planes, zero padding by out-of-range
// buffer offsets) in LDS with plane pitches == 4 (mod 64) floats, so that the 16 channels x 4 pixels of an operand read
// hit 64 different banks; every operand is one ds_read_b32 at an immediate offset.  Workgroups are persistent: the
// 9 x C/16 accumulator blocks (36 registers per 16 input channels) stay in registers over all tiles of a workgroup, the
// next tile is fetched into registers while the current one is multiplied, and only at the end the four waves are
// added in LDS and one partial per workgroup goes to memory; small_wrw_reduce_kernel adds the partials in a fixed
// order (deterministic, no atomics).
#include "common.hpp"

using namespace dmh;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256, NWV = NT / 64;
constexpr int TW = 64;
constexpr unsigned OOB = 0x80000000u;       // byte offset beyond any tensor here: the buffer load returns 0

struct WArgs {
    const float *x, *g;
    float* part;                    // [workgroup][NCB * 9 * 4 * 64 + 64]
    int B, C, H, W, Ho, Wo, pad, tx, ty, ntiles;
};

template <int NCB, int TR>
struct Geo {
    static constexpr int C = 16 * NCB, RH = TR + 2, RW = TW + 2;
    static constexpr int XPLANE = ((RH * RW - 4 + 63) / 64) * 64 + 4;      // >= RH * RW and == 4 (mod 64)
    static constexpr int GPLANE = TR * TW + 4;                             // == 4 (mod 64)
    static constexpr int XROWS = C * RH, GROWS = 16 * TR;
    static constexpr int XPW = (XROWS + NWV - 1) / NWV, GPW = GROWS / NWV; // rows per wave
    static constexpr int XTAIL = (2 * XROWS + NT - 1) / NT;                // the two columns 64, 65 of every x row
    static constexpr int ACC = NCB * 9 * 4;                                // accumulator registers per lane
    static constexpr int TILE_FLOATS = C * XPLANE + 16 * GPLANE;
    static constexpr int RED_FLOATS = NWV * ACC * 64 + NWV * 64;           // the final cross-wave sum reuses the tile memory
    static constexpr int LDS_FLOATS = TILE_FLOATS > RED_FLOATS ? TILE_FLOATS : RED_FLOATS;
    static_assert(GROWS % NWV == 0, "g rows per wave");
};

template <int NCB, int TR>
__global__ __launch_bounds__(NT, 2) void small_wrw_kernel(const WArgs a) {
    using G = Geo<NCB, TR>;
    extern __shared__ float lds[];
    float* xt = lds;                        // [C][XPLANE]
    float* gt = lds + G::C * G::XPLANE;     // [16][GPLANE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned HW = (unsigned)(a.H * a.W), HWo = (unsigned)(a.Ho * a.Wo);

    f32x4 acc[NCB][9];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[cb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    constexpr int GQ_UNROLL = NCB == 1 ? TW / 4 : 4;    // two channel blocks: a full unroll preloads operands past 256 registers

    float px[G::XPW], pg[G::GPW], ptail[G::XTAIL];
    // fetch tile t into registers (zero outside the image / beyond the output)
    auto fetch = [&](const int t) __attribute__((always_inline)) {
        int q = t;
        const int txi = q % a.tx;
        q /= a.tx;
        const int tyi = q % a.ty, b = q / a.ty;
        const int oy0 = tyi * TR, ox0 = txi * TW;
        const rsrc_t rx = make_rsrc(a.x + (size_t)b * a.C * HW, (unsigned)a.C * HW * 4u);
        const rsrc_t rg = make_rsrc(a.g + (size_t)b * 16 * HWo, 16u * HWo * 4u);
        const int ix = ox0 - a.pad + lane;
        const unsigned xo = (ix >= 0 && ix < a.W) ? (unsigned)ix * 4u : OOB;
        const unsigned go = (ox0 + lane < a.Wo) ? (unsigned)(ox0 + lane) * 4u : OOB;
#pragma unroll
        for (int k = 0; k < G::XPW; ++k) {
            const int rr = wv + NWV * k, c = rr / G::RH, r = rr - c * G::RH, iy = oy0 - a.pad + r;
            const bool ok = rr < G::XROWS && iy >= 0 && iy < a.H;                    // wave-uniform
            const unsigned so = ok ? ((unsigned)c * HW + (unsigned)(iy * a.W)) * 4u : 0u;
            px[k] = ldb(rx, xo | (ok ? 0u : OOB), so);
        }
#pragma unroll
        for (int k = 0; k < G::GPW; ++k) {
            const int rr = wv + NWV * k, kk = rr / TR, r = rr - kk * TR, oy = oy0 + r;
            const bool ok = oy < a.Ho;
            const unsigned so = ok ? ((unsigned)kk * HWo + (unsigned)(oy * a.Wo)) * 4u : 0u;
            pg[k] = ldb(rg, go | (ok ? 0u : OOB), so);
        }
#pragma unroll
        for (int k = 0; k < G::XTAIL; ++k) {
            const int e = tid + NT * k, rr = e >> 1, c = rr / G::RH, r = rr - c * G::RH;
            const int iy = oy0 - a.pad + r, ix2 = ox0 - a.pad + TW + (e & 1);
            const bool ok = rr < G::XROWS && iy >= 0 && iy < a.H && ix2 >= 0 && ix2 < a.W;
            ptail[k] = ldb(rx, ok ? ((unsigned)c * HW + (unsigned)(iy * a.W + ix2)) * 4u : OOB, 0u);
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < G::XPW; ++k) {
            const int rr = wv + NWV * k, c = rr / G::RH, r = rr - c * G::RH;
            if (rr < G::XROWS) xt[c * G::XPLANE + r * G::RW + lane] = px[k];
        }
#pragma unroll
        for (int k = 0; k < G::GPW; ++k) {
            const int rr = wv + NWV * k, kk = rr / TR, r = rr - kk * TR;
            gt[kk * G::GPLANE + r * TW + lane] = pg[k];
        }
#pragma unroll
        for (int k = 0; k < G::XTAIL; ++k) {
            const int e = tid + NT * k, rr = e >> 1, c = rr / G::RH, r = rr - c * G::RH;
            if (rr < G::XROWS) xt[c * G::XPLANE + r * G::RW + TW + (e & 1)] = ptail[k];
        }
    };

    // operand bases: A = g[k = lane & 15][pixel lane >> 4], B = x[c = lane & 15][pixel lane >> 4]
    const float* ga = gt + (lane & 15) * G::GPLANE + (lane >> 4);
    const float* xb = xt + (lane & 15) * G::XPLANE + (lane >> 4);

    int t = blockIdx.x;
    if (t < a.ntiles) fetch(t);
    for (; t < a.ntiles; t += gridDim.x) {
        __syncthreads();                    // the previous tile has been consumed
        commit();
        __syncthreads();
        if (t + (int)gridDim.x < a.ntiles) fetch(t + gridDim.x);
#pragma unroll 1
        for (int r = wv; r < TR; r += NWV) {
            const float* gr = ga + r * TW;
            const float* xr = xb + r * G::RW;
#pragma unroll GQ_UNROLL
            for (int gq = 0; gq < TW / 4; ++gq) {
                const float av = gr[4 * gq];
                bsum += av;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) {
                        const float bv = xr[cb * 16 * G::XPLANE + (tp / 3) * G::RW + 4 * gq + (tp % 3)];
                        acc[cb][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[cb][tp], 0, 0, 0);
                    }
            }
        }
    }
    // ---- add the four waves (fixed order) and write this workgroup's partial
    __syncthreads();
    float* red = lds;                       // [wave][ACC][64] then [wave][64] for the bias sums
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) red[(wv * G::ACC + (cb * 9 + tp) * 4 + rr) * 64 + lane] = acc[cb][tp][rr];
    red[NWV * G::ACC * 64 + wv * 64 + lane] = bsum;
    __syncthreads();
    float* pp = a.part + (size_t)blockIdx.x * (G::ACC * 64 + 64);
    for (int e = tid; e < G::ACC * 64 + 64; e += NT) {
        float s = 0.f;
        if (e < G::ACC * 64) {
#pragma unroll
            for (int w = 0; w < NWV; ++w) s += red[w * G::ACC * 64 + e];
        } else {
#pragma unroll
            for (int w = 0; w < NWV; ++w) s += red[NWV * G::ACC * 64 + w * 64 + (e - G::ACC * 64)];
        }
        pp[e] = s;
    }
}

// g_w[k][c][tap] = sum over workgroups of part[wg][((cb * 9 + tap) * 4 + (k & 3)) * 64 + (c & 15) + 16 * (k >> 2)],
// cb = c / 16;  g_b[k] = sum over workgroups and the four pixel slots of part[wg][ACC * 64 + k + 16 * slot].
// One workgroup per 64 outputs: 64 lanes x 4 groups of workgroup partials, fixed order.
__global__ __launch_bounds__(NT) void small_wrw_reduce_kernel(const float* __restrict__ part, int nwg, int C,
                                                              float* __restrict__ g_w, float* __restrict__ g_b) {
    __shared__ float red[NT];
    const int P = (C / 16) * 9 * 4 * 64 + 64;
    const int nout = 16 * C * 9 + 16;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float s = 0.f;
    if (j < nout) {
        if (j < 16 * C * 9) {
            const int k = j / (C * 9), rem = j - k * (C * 9), c = rem / 9, tap = rem - c * 9;
            const int idx = (((c >> 4) * 9 + tap) * 4 + (k & 3)) * 64 + (c & 15) + 16 * (k >> 2);
            for (int w = grp; w < nwg; w += NT / 64) s += part[(size_t)w * P + idx];
        } else {
            const int k = j - 16 * C * 9;
            for (int w = grp; w < nwg; w += NT / 64) {
                const float* p = part + (size_t)w * P + (P - 64);
                s += (p[k] + p[k + 16]) + (p[k + 32] + p[k + 48]);
            }
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && j < nout) {
        const float v = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
        if (j < 16 * C * 9)
            g_w[j] = v;
        else if (g_b)
            g_b[j - 16 * C * 9] = v;
    }
}

constexpr int MAX_WG = 512;                 // two persistent workgroups per CU

template <int NCB, int TR>
int launch(WArgs& a, float* g_w, float* g_b, hipStream_t st) {
    using G = Geo<NCB, TR>;
    constexpr size_t smem = (size_t)G::LDS_FLOATS * sizeof(float);
    static std::atomic<uint64_t> configured{0};     // per device, see configure_dynamic_lds
    if (configure_dynamic_lds(small_wrw_kernel<NCB, TR>, smem, configured) != hipSuccess)
        return fail(DMH_ELAUNCH, "%s: cannot raise the dynamic LDS limit", "dmh_conv3x3_small_wrw");
    a.tx = (a.Wo + TW - 1) / TW;
    a.ty = (a.Ho + TR - 1) / TR;
    const long long tiles = (long long)a.B * a.tx * a.ty;
    if (tiles >= (1ll << 31)) return fail(DMH_EINVAL, "%s: grid too large", "dmh_conv3x3_small_wrw");
    a.ntiles = (int)tiles;
    const int nwg = (int)(tiles < MAX_WG ? tiles : MAX_WG);
    hipLaunchKernelGGL((small_wrw_kernel<NCB, TR>), dim3(nwg), dim3(NT), smem, st, a);
    hipLaunchKernelGGL(small_wrw_reduce_kernel, dim3((16 * a.C * 9 + 16 + 63) / 64), dim3(NT), 0, st, a.part, nwg, a.C, g_w,
                       g_b);
    return check_launch("dmh_conv3x3_small_wrw");
}

}  // namespace

extern "C" {

int64_t dmh_conv3x3_small_wrw_partials_size(int C) {
    if (C != 16 && C != 32) return 0;
    return (int64_t)MAX_WG * ((C / 16) * 9 * 4 * 64 + 64);
}

int dmh_conv3x3_small_wrw(const float* x, const float* g, int B, int C, int H, int W, int pad, float* partials, float* g_w,
                          float* g_b, void* stream) {
    DMH_REQUIRE(x && g && partials && g_w, "null pointer");
    DMH_REQUIRE(B > 0 && (C == 16 || C == 32) && pad >= 0 && pad <= 2, "16 or 32 input channels (16 output channels), pad 0..2");
    WArgs a;
    a.x = x;
    a.g = g;
    a.part = partials;
    a.B = B;
    a.C = C;
    a.H = H;
    a.W = W;
    a.pad = pad;
    a.Ho = H + 2 * pad - 2;
    a.Wo = W + 2 * pad - 2;
    DMH_REQUIRE(a.Ho >= 1 && a.Wo >= 1, "image smaller than the filter");
    DMH_REQUIRE((int64_t)C * H * W < ((int64_t)1 << 28) && (int64_t)16 * a.Ho * a.Wo < ((int64_t)1 << 28),
                "image too large (32-bit byte offsets)");
    if (C == 16) return launch<1, 8>(a, g_w, g_b, (hipStream_t)stream);
    return launch<2, 4>(a, g_w, g_b, (hipStream_t)stream);
}

}  // extern "C"
