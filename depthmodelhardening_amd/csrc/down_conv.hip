// K15 -- the two convolutions that open a down-sampling ResNet block, in ONE launch each way, on the fp32 matrix cores:
//     y3 = conv3x3(x, w3, stride 2, padding 1)        torchvision BasicBlock.conv1      (layer2.0 / layer3.0 / layer4.0)
//     yd = conv1x1(x, wd, stride 2)                   BasicBlock.downsample[0]
// under MD2/networks/resnet_encoder.py:85-98 (`self.encoder.layerN(...)`).  MIOpen serves these with its stride-2
// Winograd (f3x2) / dilated Winograd for the gradient and NHWC implicit GEMMs wrapped in layout transposes: 12 % of an
// attack step.  Both convolutions read the same input, and the 1x1 tap IS the centre tap of the 3x3 window
// (input pixel (2 oy, 2 ox)), so the 1x1 result costs one extra MFMA per nine on operands that are already in LDS.
//
// Forward, GEMM view: M = 64 output channels per workgroup, N = 4 output rows x 32 columns (one row per wave; rows of
// the flattened (image, row) index, so short maps fill their tiles),
// K = C_in x 9 walked in chunks of 8 channels:
//   * v_mfma_f32_32x32x2_f32 (exact fp32): the two k of an MFMA are the SAME tap of two adjacent input channels, so the
//     half-waves differ by one LDS plane and every B operand is one ds_read_b32 at an immediate offset from a per-lane
//     base (lane stride 2 floats, odd plane pitch: conflict-free);
//   * A operand: the filter chunk in LDS as [row][k parity][tap][4 channel pairs]: one ds_read_b128 feeds four MFMAs
//     (row pitch 36 floats: conflict-free for 16-lane b128 groups);
//   * the next chunk's input rows (8 channels x 4 x 3 rows x 65, zero padding by out-of-range buffer offsets) and filter slice are
//     fetched into registers with buffer loads (per-thread offsets formed once, chunk offset in an SGPR) while the
//     current chunk's 72 (+8) MFMAs run; no address arithmetic inside the loop.
// Backward-data: g_x = conv3x3_s2^T(g3) + conv1x1_s2^T(gd).  The four parity classes (iy & 1, ix & 1) of g_x are four
// small stride-1 correlations with 1, 2, 2 and 4 of the nine taps; a workgroup owns 64 input channels x an 8 x 64
// pixel block = the four classes of a 4 x 32 class tile, 8 accumulators, and the four neighbouring g3 values a lane
// needs are read once per channel pair and shared by the nine taps.  The 1x1 gradient lands on class (0, 0) only: one
// more MFMA.  The filter is taken pre-transposed ([C_in][C_out][3][3], once per attack under ops.frozen_weights()).
// MFMA-bound: 2 x 9.06 GFLOP per direction at each of the three encoder shapes of an attack step (12 x 64 x 80 x 256,
// 12 x 128 x 40 x 128, 12 x 256 x 20 x 64).
#include "common.hpp"

using namespace dmh;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;
constexpr int TR = 4, TC = 32;            // rows (one per wave) x columns of the N tile
constexpr int CK = 8, NPAIR = CK / 2;     // K chunk: 8 channels = 4 MFMA channel pairs
constexpr unsigned OOB = 0x80000000u;     // byte offset beyond any tensor here: the buffer load returns 0

constexpr int A_PITCH = 36;                       // floats per (row, k parity): 9 taps x 4 pairs
constexpr int A_ELEMS = 64 * 2 * A_PITCH;         // 4608 for a 64-row M tile
constexpr int A_NW = A_ELEMS / NT;                // 18 per thread
constexpr int D_ELEMS = 64 * 2 * NPAIR;           // 512: the 1x1 filter slice, [row][k parity][4 pairs]
constexpr int D_NW = D_ELEMS / NT;                // 2

// per-thread source offsets of the filter slice: LDS element d = tid + i * NT  <-  w[(r0 + row) * inner + kk][tap]
// (w is [outer][inner][9]; forward: outer = C_out, inner = C_in; backward: the transposed filter, outer = C_in)
template <int NW, int ND>
__device__ __forceinline__ void filter_offsets(int tid, int r0, int inner, unsigned (&wo)[NW], unsigned (&dofs)[ND]) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int d = tid + i * NT, rowh = d / A_PITCH, q = d - rowh * A_PITCH;
        const int row = rowh >> 1, hh = rowh & 1, t = q >> 2, p = q & 3;
        wo[i] = (unsigned)(((r0 + row) * inner + 2 * p + hh) * 9 + t) * 4u;
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int d = tid + i * NT, rowh = d >> 2, p = d & 3;
        dofs[i] = (unsigned)((r0 + (rowh >> 1)) * inner + 2 * p + (rowh & 1)) * 4u;
    }
}

// ---- round 5: the filters as an LDS IMAGE, the operands by 16-byte words -------------------------------------------------
// The loaders above cost a chunk 45 (forward) / 33 (backward) dword buffer loads and as many LDS writes per thread -- and a
// vector-memory instruction costs a wave ~64 cycles among the fp32 MFMAs whatever its width (profiles/README.md, K10): 2,900 of
// a chunk's ~9,100 cycles beside 5,120 of matrix pipe (tools/isa_census.py).  The WIDE kernels read
//   * the filter slice of a chunk as a CONTIGUOUS image of the LDS layout, [row group of 32][chunk][A: 32 x 2 x 36 | D: 32 x 2 x
//     4] floats, written once per weight tensor by down_weight_image_kernel: 640 16-byte words per (group, chunk), linear in
//     global memory and in LDS -> 6 buffer_load_dwordx4 + 6 ds_write_b128 per thread instead of 20 + 20;
//   * the input rows as the aligned 16-byte words that cover them (17 per 65-column row from column 64 bx - 4; backward: 9 per
//     33-column row): 7 (3 + 1) loads instead of 25 (9 + 4).  A word lies inside or outside the image as a whole when the row
//     length is a multiple of 4 -- the launcher takes the WIDE kernels for such shapes only.  Forward: the columns of a row are
//     stored DE-INTERLEAVED (even | odd, two 8-byte LDS writes per word), so that the stride-2 taps of consecutive lanes are
//     consecutive floats; plane pitches = 32 (mod 64) floats put the two half-waves on complementary banks.
// Same MFMAs in the same order: bit-identical to the dword kernels (tests/test_gpu_conv_anchor.py).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int IMG_A = 32 * 2 * A_PITCH;           // 2304 floats: 3x3 filter slice of 32 rows
constexpr int IMG_D = 32 * 2 * NPAIR;             // 256 floats: 1x1 filter slice of 32 rows
constexpr int IMG_G = IMG_A + IMG_D;              // 2560 floats per (row group, chunk)
constexpr int IMG_WORDS = IMG_G / 4;              // 640
constexpr int IMG_NW = (IMG_WORDS + NT - 1) / NT; // 3 words per thread and row group (the third for the first 128 threads)

// w3: [rows][inner][9], wd: [rows][inner] or null (zeros) -> image [rows / 32][inner / 8][IMG_G]
__global__ __launch_bounds__(NT) void down_weight_image_kernel(const float* __restrict__ w3, const float* __restrict__ wd, int rows,
                                                               int inner, float* __restrict__ img) {
    const size_t idx = (size_t)blockIdx.x * NT + threadIdx.x;
    const int nch = inner / CK;
    if (idx >= (size_t)(rows / 32) * nch * IMG_G) return;
    const int e = (int)(idx % IMG_G);
    const int gc = (int)(idx / IMG_G), rg = gc / nch, ch = gc - rg * nch;
    float v;
    if (e < IMG_A) {
        const int rowh = e / A_PITCH, q = e - rowh * A_PITCH, t = q >> 2, p = q & 3;
        v = w3[((size_t)(rg * 32 + (rowh >> 1)) * inner + ch * CK + 2 * p + (rowh & 1)) * 9 + t];
    } else {
        const int d = e - IMG_A, rowh = d >> 2, p = d & 3;
        v = wd ? wd[(size_t)(rg * 32 + (rowh >> 1)) * inner + ch * CK + 2 * p + (rowh & 1)] : 0.f;
    }
    img[idx] = v;
}

__device__ __forceinline__ f32x4 ldb4(rsrc_t rs, unsigned byte_off, unsigned s_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, s_off, 0));
}

// ---------------------------------------------------------------------------------------------- forward
// The N tile is 4 rows of the FLATTENED (image, output row) index, so a 10-row map (layer4 at 320 x 1024) fills its
// tiles and 12 images x 10 rows x 8 channel groups make 240 workgroups = one round of the 256 CUs.  Each output row
// therefore keeps its own three input rows in LDS (rows of one tile may belong to two images).
constexpr int F_PR = 3 * TR, F_PC = 2 * TC + 1, F_PEL = F_PR * F_PC, F_PLANE = F_PEL + 1;   // 12 x 65 = 780 (+1: odd pitch)
constexpr int F_NP = (CK * F_PEL + NT - 1) / NT;                                            // 25 per thread

struct FArgs {
    const float *x, *w3, *wd;
    const float *shift3, *shiftd;     // optional per-channel addends of y3 / yd (an eval-mode BatchNorm whose scale is in the filter)
    int relu3;                        // clamp y3 at zero after the shift
    float *y3, *yd;
    int B, Cin, Cout, H, W, Ho, Wo, gx, gy, gk;     // gy = row tiles over B * Ho
};

// D[i][n]: lane holds pixel n, channels i = 8 * (v / 4) + 4 * h + v % 4 (+ 32)
template <bool DOWN>
__device__ __forceinline__ void fwd_store(const FArgs& a, const f32x16& accA, const f32x16& accB, const f32x16& accDA,
                                          const f32x16& accDB, int R, int ox, int h, int k0, int NR) {
    const int b = R / a.Ho, oy = R - b * a.Ho;
    if (R < NR && ox < a.Wo) {
        const size_t HWo = (size_t)a.Ho * a.Wo;
        const size_t o = ((size_t)b * a.Cout + k0) * HWo + (size_t)oy * a.Wo + ox;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int i = 8 * (v >> 2) + 4 * h + (v & 3);
            float ya = accA[v], yb = accB[v];
            if (a.shift3) {
                ya += a.shift3[k0 + i];
                yb += a.shift3[k0 + i + 32];
            }
            if (a.relu3) {
                ya = fmaxf(ya, 0.f);
                yb = fmaxf(yb, 0.f);
            }
            a.y3[o + (size_t)i * HWo] = ya;
            a.y3[o + (size_t)(i + 32) * HWo] = yb;
            if constexpr (DOWN) {
                float da = accDA[v], db = accDB[v];
                if (a.shiftd) {
                    da += a.shiftd[k0 + i];
                    db += a.shiftd[k0 + i + 32];
                }
                a.yd[o + (size_t)i * HWo] = da;
                a.yd[o + (size_t)(i + 32) * HWo] = db;
            }
        }
    }
}

template <bool DOWN>
__global__ __launch_bounds__(NT, 2) void down_conv_fwd_kernel(const FArgs a) {
    __shared__ float patch[CK * F_PLANE];    // [channel][output row of the tile][3 input rows][65 columns]
    __shared__ __attribute__((aligned(16))) float wt[A_ELEMS];
    __shared__ __attribute__((aligned(16))) float wdl[D_ELEMS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, h = lane >> 5;
    int q = blockIdx.x;
    const int kt = q % a.gk;       // channel groups of one pixel tile are neighbours: the input rectangle stays in L2
    q /= a.gk;
    const int bxi = q % a.gx, rt = q / a.gx;
    const int R0 = rt * TR, ox0 = bxi * TC, k0 = kt * 64;
    const int ix0 = 2 * ox0 - 1, NR = a.B * a.Ho;
    const unsigned HW = (unsigned)(a.H * a.W);
    const rsrc_t rx = make_rsrc(a.x, (unsigned)a.B * (unsigned)a.Cin * HW * 4u);
    const rsrc_t rw = make_rsrc(a.w3, (unsigned)a.Cout * (unsigned)a.Cin * 36u);
    const rsrc_t rd = make_rsrc(DOWN ? a.wd : a.w3, (unsigned)a.Cout * (unsigned)a.Cin * 4u);

    unsigned po[F_NP], wo[A_NW], dofs[D_NW];
#pragma unroll
    for (int i = 0; i < F_NP; ++i) {
        const int e = tid + i * NT, c = e / F_PEL, rem = e - c * F_PEL, r = rem / F_PC, cc = rem - r * F_PC;
        const int R = R0 + r / 3, bb = R / a.Ho, oy = R - bb * a.Ho;
        const int iy = 2 * oy - 1 + r % 3, ix = ix0 + cc;
        const bool ok = e < CK * F_PEL && R < NR && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        po[i] = ok ? ((unsigned)(bb * a.Cin + c) * HW + (unsigned)(iy * a.W + ix)) * 4u : OOB;
    }
    filter_offsets(tid, k0, a.Cin, wo, dofs);

    float rp[F_NP], rwv[A_NW], rdv[D_NW];
    auto fetch = [&](const int c0) __attribute__((always_inline)) {
        const unsigned sp = (unsigned)c0 * HW * 4u, sw = (unsigned)c0 * 36u, sd = (unsigned)c0 * 4u;
#pragma unroll
        for (int i = 0; i < F_NP; ++i) rp[i] = ldb(rx, po[i], sp);
#pragma unroll
        for (int i = 0; i < A_NW; ++i) rwv[i] = ldb(rw, wo[i], sw);
        if constexpr (DOWN) {
#pragma unroll
            for (int i = 0; i < D_NW; ++i) rdv[i] = ldb(rd, dofs[i], sd);
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < F_NP; ++i) {
            const int e = tid + i * NT;
            if (i < F_NP - 1 || e < CK * F_PEL) patch[e + e / F_PEL] = rp[i];
        }
#pragma unroll
        for (int i = 0; i < A_NW; ++i) wt[tid + i * NT] = rwv[i];
        if constexpr (DOWN) {
#pragma unroll
            for (int i = 0; i < D_NW; ++i) wdl[tid + i * NT] = rdv[i];
        }
    };

    f32x16 accA, accB, accDA, accDB;
#pragma unroll
    for (int v = 0; v < 16; ++v) accA[v] = accB[v] = accDA[v] = accDB[v] = 0.f;
    const float* pb = patch + h * F_PLANE + (3 * wv) * F_PC + 2 * n;
    const float* aA = wt + (n * 2 + h) * A_PITCH;
    const float* aB = aA + 32 * 2 * A_PITCH;
    const float* dA = wdl + (n * 2 + h) * NPAIR;
    const float* dB = dA + 32 * 2 * NPAIR;

    const int chunks = a.Cin / CK;
    fetch(0);
    for (int ch = 0; ch < chunks; ++ch) {
        __syncthreads();                     // the previous chunk's operands have been read
        commit();
        __syncthreads();
        if (ch + 1 < chunks) fetch((ch + 1) * CK);
        float4 da = make_float4(0.f, 0.f, 0.f, 0.f), db = da;
        if constexpr (DOWN) {
            da = *reinterpret_cast<const float4*>(dA);
            db = *reinterpret_cast<const float4*>(dB);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 wa = *reinterpret_cast<const float4*>(aA + 4 * t);
            const float4 wb = *reinterpret_cast<const float4*>(aB + 4 * t);
            const float was[4] = {wa.x, wa.y, wa.z, wa.w}, wbs[4] = {wb.x, wb.y, wb.z, wb.w};
            const float das[4] = {da.x, da.y, da.z, da.w}, dbs[4] = {db.x, db.y, db.z, db.w};
#pragma unroll
            for (int p = 0; p < NPAIR; ++p) {
                const float bv = pb[2 * p * F_PLANE + (t / 3) * F_PC + (t % 3)];
                accA = __builtin_amdgcn_mfma_f32_32x32x2f32(was[p], bv, accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_32x32x2f32(wbs[p], bv, accB, 0, 0, 0);
                if (DOWN && t == 4) {
                    accDA = __builtin_amdgcn_mfma_f32_32x32x2f32(das[p], bv, accDA, 0, 0, 0);
                    accDB = __builtin_amdgcn_mfma_f32_32x32x2f32(dbs[p], bv, accDB, 0, 0, 0);
                }
            }
        }
    }
    fwd_store<DOWN>(a, accA, accB, accDA, accDB, R0 + wv, ox0 + n, h, k0, NR);
}

// ---- forward, WIDE loaders (see "round 5" above).  a.w3 is the filter IMAGE here (rows = C_out, inner = C_in); a.wd only says
//      whether the shortcut convolution is wanted.
constexpr int FW_WPR = 17, FW_SUB = 2 * FW_WPR, FW_ROW = 2 * FW_SUB;      // words per row; floats per even / odd half row; per row
constexpr int FW_PLANE = 864;                                             // 12 x 68 = 816 -> the next pitch = 32 (mod 64)
constexpr int FW_NWORD = CK * F_PR * FW_WPR;                              // 1632 words per chunk
constexpr int FW_NP = (FW_NWORD + NT - 1) / NT;                           // 7 per thread
static_assert(FW_PLANE >= F_PR * FW_ROW && FW_PLANE % 64 == 32, "plane pitch");

template <bool DOWN>
__global__ __launch_bounds__(NT, 2) void down_conv_fwd_wide_kernel(const FArgs a) {
    __shared__ __attribute__((aligned(16))) float patch[CK * FW_PLANE];   // [channel][tile row x 3 input rows][even | odd][34]
    __shared__ f32x4 wimg[2 * IMG_WORDS];                                  // two row groups of 32: [A | D] each
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, h = lane >> 5;
    int q = blockIdx.x;
    const int kt = q % a.gk;
    q /= a.gk;
    const int bxi = q % a.gx, rt = q / a.gx;
    const int R0 = rt * TR, ox0 = bxi * TC, k0 = kt * 64;
    const int NR = a.B * a.Ho, nch = a.Cin / CK;
    const unsigned HW = (unsigned)(a.H * a.W);
    const rsrc_t rx = make_rsrc(a.x, (unsigned)a.B * (unsigned)a.Cin * HW * 4u);
    const rsrc_t ri = make_rsrc(a.w3, (unsigned)(a.Cout / 32) * (unsigned)nch * (unsigned)(IMG_G * 4));

    unsigned po[FW_NP];
    int lo[FW_NP];
#pragma unroll
    for (int i = 0; i < FW_NP; ++i) {
        const int e = tid + i * NT, c = e / (F_PR * FW_WPR), rem = e - c * (F_PR * FW_WPR), r = rem / FW_WPR, w = rem - r * FW_WPR;
        const int R = R0 + r / 3, bb = R / a.Ho, oy = R - bb * a.Ho;
        const int iy = 2 * oy - 1 + r % 3, ix = 2 * ox0 - 4 + 4 * w;       // W is a multiple of 4: the word is inside or outside
        const bool ok = e < FW_NWORD && R < NR && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        po[i] = ok ? ((unsigned)(bb * a.Cin + c) * HW + (unsigned)(iy * a.W + ix)) * 4u : OOB;
        lo[i] = c * FW_PLANE + r * FW_ROW + 2 * w;
    }
    const unsigned wofs = (unsigned)tid * 16u;
    const unsigned gstep = (unsigned)nch * (unsigned)(IMG_G * 4);         // bytes from a row group's image to the next group's
    const unsigned g0 = (unsigned)(2 * kt) * gstep;

    f32x4 rp[FW_NP], rw[2 * IMG_NW];
    auto fetch = [&](const int ch) __attribute__((always_inline)) {
        const unsigned sp = (unsigned)(ch * CK) * HW * 4u;
#pragma unroll
        for (int i = 0; i < FW_NP; ++i) rp[i] = ldb4(rx, po[i], sp);
        const unsigned sw = g0 + (unsigned)ch * (unsigned)(IMG_G * 4);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < IMG_NW; ++j)
                rw[g * IMG_NW + j] = ldb4(ri, ((j + 1) * NT <= IMG_WORDS || j * NT + tid < IMG_WORDS) ? wofs + (unsigned)(j * NT * 16) : OOB,
                                          sw + (unsigned)g * gstep);
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FW_NP; ++i) {
            if (i < FW_NP - 1 || tid + i * NT < FW_NWORD) {
                *reinterpret_cast<float2*>(patch + lo[i]) = make_float2(rp[i][0], rp[i][2]);             // even columns
                *reinterpret_cast<float2*>(patch + lo[i] + FW_SUB) = make_float2(rp[i][1], rp[i][3]);    // odd columns
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < IMG_NW; ++j)
                if ((j + 1) * NT <= IMG_WORDS || j * NT + tid < IMG_WORDS) wimg[g * IMG_WORDS + j * NT + tid] = rw[g * IMG_NW + j];
    };

    f32x16 accA, accB, accDA, accDB;
#pragma unroll
    for (int v = 0; v < 16; ++v) accA[v] = accB[v] = accDA[v] = accDB[v] = 0.f;
    // lane n = output column ox0 + n reads input columns 2 n + 3 + kx of the staged row (it starts at column 2 ox0 - 4):
    // kx = 0 -> odd[n + 1], kx = 1 -> even[n + 2], kx = 2 -> odd[n + 2]
    const float* pb = patch + h * FW_PLANE + (3 * wv) * FW_ROW + n;
    const float* wf = reinterpret_cast<const float*>(wimg);
    const float* aA = wf + (n * 2 + h) * A_PITCH;
    const float* aB = aA + IMG_G;
    const float* dA = wf + IMG_A + (n * 2 + h) * NPAIR;
    const float* dB = dA + IMG_G;

    fetch(0);
    for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();                     // the previous chunk's operands have been read
        commit();
        __syncthreads();
        if (ch + 1 < nch) fetch(ch + 1);
        float4 da = make_float4(0.f, 0.f, 0.f, 0.f), db = da;
        if constexpr (DOWN) {
            da = *reinterpret_cast<const float4*>(dA);
            db = *reinterpret_cast<const float4*>(dB);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 wa = *reinterpret_cast<const float4*>(aA + 4 * t);
            const float4 wb = *reinterpret_cast<const float4*>(aB + 4 * t);
            const float was[4] = {wa.x, wa.y, wa.z, wa.w}, wbs[4] = {wb.x, wb.y, wb.z, wb.w};
            const float das[4] = {da.x, da.y, da.z, da.w}, dbs[4] = {db.x, db.y, db.z, db.w};
            const int kx = t % 3, off = (t / 3) * FW_ROW + (kx == 1 ? 2 : FW_SUB + (kx == 0 ? 1 : 2));
#pragma unroll
            for (int p = 0; p < NPAIR; ++p) {
                const float bv = pb[2 * p * FW_PLANE + off];
                accA = __builtin_amdgcn_mfma_f32_32x32x2f32(was[p], bv, accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_32x32x2f32(wbs[p], bv, accB, 0, 0, 0);
                if (DOWN && t == 4) {
                    accDA = __builtin_amdgcn_mfma_f32_32x32x2f32(das[p], bv, accDA, 0, 0, 0);
                    accDB = __builtin_amdgcn_mfma_f32_32x32x2f32(dbs[p], bv, accDB, 0, 0, 0);
                }
            }
        }
    }
    fwd_store<DOWN>(a, accA, accB, accDA, accDB, R0 + wv, ox0 + n, h, k0, NR);
}

// ---------------------------------------------------------------------------------------------- backward-data
constexpr int B_PR = 2 * TR, B_PC = TC + 1, B_PEL = B_PR * B_PC, B_PLANE = B_PEL + 1;   // 8 x 33 = 264 (+1: odd pitch)
constexpr int B_NP = (CK * B_PEL + NT - 1) / NT;                                        // 9 per thread
constexpr int G_PLANE = TR * TC + 1;                                          // gd tile plane: 129 (odd)
constexpr int G_NP = CK * TR * TC / NT;                                       // 4 per thread

struct BArgs {
    const float *g3, *gd, *w3t, *wdt;
    const float* gadd;           // optional: a gradient of the same tensor from another consumer, added in the epilogue
    float* gx;
    int B, Cin, Cout, H, W, Ho, Wo, tx, ty, gc;
};

// lane: class pixel (a, b0 + n) -> g_x rows 2a (+1), columns 2b, 2b + 1 as one 8-byte store per row
template <int MH>
__device__ __forceinline__ void bwd_store(const BArgs& a, const f32x16 (&acc)[4][MH], int R, int bc, int h, int c0m, int NR) {
    const int b = R / a.Ho, ar = R - b * a.Ho;
    if (R < NR && bc < a.Wo) {
        const size_t HW = (size_t)a.H * a.W;
        const size_t o = ((size_t)b * a.Cin + c0m) * HW + (size_t)(2 * ar) * a.W + 2 * bc;
#pragma unroll
        for (int half = 0; half < MH; ++half)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = 8 * (v >> 2) + 4 * h + (v & 3) + 32 * half;
                float* g = a.gx + o + (size_t)i * HW;
                float2 r0 = make_float2(acc[0][half][v], acc[1][half][v]), r1 = make_float2(acc[2][half][v], acc[3][half][v]);
                if (a.gadd) {
                    const float* q = a.gadd + o + (size_t)i * HW;
                    const float2 q0 = *reinterpret_cast<const float2*>(q), q1 = *reinterpret_cast<const float2*>(q + a.W);
                    r0.x += q0.x; r0.y += q0.y; r1.x += q1.x; r1.y += q1.y;
                }
                *reinterpret_cast<float2*>(g) = r0;
                *reinterpret_cast<float2*>(g + a.W) = r1;
            }
    }
}

// MH = 32-channel halves of the M tile (2: 64 input channels per workgroup; 1: 32, for maps too small to fill the chip
// with 64-channel tiles -- layer4 at 12 scenes has 30 row tiles x 4 channel groups).
template <bool DOWN, int MH>
__global__ __launch_bounds__(NT, 2) void down_conv_bwd_kernel(const BArgs a) {
    constexpr int W_NW = A_NW * MH / 2, W_ND = D_NW * MH / 2;
    __shared__ float patch[CK * B_PLANE];
    __shared__ float gdl[CK * G_PLANE];
    __shared__ __attribute__((aligned(16))) float wt[A_ELEMS * MH / 2];
    __shared__ __attribute__((aligned(16))) float wdl[D_ELEMS * MH / 2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, h = lane >> 5;
    int q = blockIdx.x;
    const int ct = q % a.gc;
    q /= a.gc;
    const int bxi = q % a.tx, rt = q / a.tx;
    const int R0 = rt * TR, b0 = bxi * TC, c0m = ct * 32 * MH, NR = a.B * a.Ho;      // rows of the flattened (image, class row)
    const unsigned HWo = (unsigned)(a.Ho * a.Wo);
    const rsrc_t rg = make_rsrc(a.g3, (unsigned)a.B * (unsigned)a.Cout * HWo * 4u);
    const rsrc_t rgd = make_rsrc(DOWN ? a.gd : a.g3, (unsigned)a.B * (unsigned)a.Cout * HWo * 4u);
    const rsrc_t rw = make_rsrc(a.w3t, (unsigned)a.Cout * (unsigned)a.Cin * 36u);
    const rsrc_t rd = make_rsrc(DOWN ? a.wdt : a.w3t, (unsigned)a.Cout * (unsigned)a.Cin * 4u);

    unsigned po[B_NP], go[G_NP], wo[W_NW], dofs[W_ND];
#pragma unroll
    for (int i = 0; i < B_NP; ++i) {
        const int e = tid + i * NT, c = e / B_PEL, rem = e - c * B_PEL, r = rem / B_PC, cc = rem - r * B_PC;
        const int R = R0 + (r >> 1), bb = R / a.Ho, oy = R - bb * a.Ho + (r & 1), ox = b0 + cc;
        const bool ok = e < CK * B_PEL && R < NR && oy < a.Ho && ox < a.Wo;
        po[i] = ok ? ((unsigned)(bb * a.Cout + c) * HWo + (unsigned)(oy * a.Wo + ox)) * 4u : OOB;
    }
#pragma unroll
    for (int i = 0; i < G_NP; ++i) {
        const int e = tid + i * NT, c = e / (TR * TC), rem = e - c * (TR * TC), r = rem / TC, cc = rem - r * TC;
        const int R = R0 + r, bb = R / a.Ho, oy = R - bb * a.Ho, ox = b0 + cc;
        go[i] = (R < NR && ox < a.Wo) ? ((unsigned)(bb * a.Cout + c) * HWo + (unsigned)(oy * a.Wo + ox)) * 4u : OOB;
    }
    filter_offsets(tid, c0m, a.Cout, wo, dofs);

    float rp[B_NP], rg4[G_NP], rwv[W_NW], rdv[W_ND];
    auto fetch = [&](const int k0) __attribute__((always_inline)) {
        const unsigned sp = (unsigned)k0 * HWo * 4u, sw = (unsigned)k0 * 36u, sd = (unsigned)k0 * 4u;
#pragma unroll
        for (int i = 0; i < B_NP; ++i) rp[i] = ldb(rg, po[i], sp);
#pragma unroll
        for (int i = 0; i < W_NW; ++i) rwv[i] = ldb(rw, wo[i], sw);
        if constexpr (DOWN) {
#pragma unroll
            for (int i = 0; i < G_NP; ++i) rg4[i] = ldb(rgd, go[i], sp);
#pragma unroll
            for (int i = 0; i < W_ND; ++i) rdv[i] = ldb(rd, dofs[i], sd);
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < B_NP; ++i) {
            const int e = tid + i * NT;
            if (i < B_NP - 1 || e < CK * B_PEL) patch[e + e / B_PEL] = rp[i];
        }
#pragma unroll
        for (int i = 0; i < W_NW; ++i) wt[tid + i * NT] = rwv[i];
        if constexpr (DOWN) {
#pragma unroll
            for (int i = 0; i < G_NP; ++i) {
                const int e = tid + i * NT, c = e / (TR * TC);
                gdl[e + c] = rg4[i];                         // plane pitch TR * TC + 1
            }
#pragma unroll
            for (int i = 0; i < W_ND; ++i) wdl[tid + i * NT] = rdv[i];
        }
    };

    f32x16 acc[4][MH];     // [parity class 2 * py + px][channel half]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[c][m][v] = 0.f;
    const float* pg = patch + h * B_PLANE + (2 * wv) * B_PC + n;
    const float* pd = gdl + h * G_PLANE + wv * TC + n;
    const float* aA = wt + (n * 2 + h) * A_PITCH;
    const float* aB = aA + 32 * 2 * A_PITCH;
    const float* dA = wdl + (n * 2 + h) * NPAIR;
    const float* dB = dA + 32 * 2 * NPAIR;

    const int chunks = a.Cout / CK;
    fetch(0);
    for (int ch = 0; ch < chunks; ++ch) {
        __syncthreads();
        commit();
        __syncthreads();
        if (ch + 1 < chunks) fetch((ch + 1) * CK);
        float bv[NPAIR][2][2];            // g3 at (a + dy, b + dx) of the lane's class pixel, per channel pair
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {
            bv[p][0][0] = pg[2 * p * B_PLANE];
            bv[p][0][1] = pg[2 * p * B_PLANE + 1];
            bv[p][1][0] = pg[2 * p * B_PLANE + B_PC];
            bv[p][1][1] = pg[2 * p * B_PLANE + B_PC + 1];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // tap (ky, kx) feeds the class with py = (ky != 1), px = (kx != 1) from g3[a + (ky == 0)][b + (kx == 0)]
            const int ky = t / 3, kx = t % 3;
            const int cls = 2 * (ky != 1 ? 1 : 0) + (kx != 1 ? 1 : 0), dy = ky == 0 ? 1 : 0, dx = kx == 0 ? 1 : 0;
            const float4 wa = *reinterpret_cast<const float4*>(aA + 4 * t);
            const float4 wb = *reinterpret_cast<const float4*>((MH == 2 ? aB : aA) + 4 * t);
            const float was[4] = {wa.x, wa.y, wa.z, wa.w}, wbs[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
            for (int p = 0; p < NPAIR; ++p) {
                acc[cls][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(was[p], bv[p][dy][dx], acc[cls][0], 0, 0, 0);
                if constexpr (MH == 2)
                    acc[cls][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wbs[p], bv[p][dy][dx], acc[cls][1], 0, 0, 0);
            }
        }
        if constexpr (DOWN) {
            const float4 da = *reinterpret_cast<const float4*>(dA), db = *reinterpret_cast<const float4*>(MH == 2 ? dB : dA);
            const float das[4] = {da.x, da.y, da.z, da.w}, dbs[4] = {db.x, db.y, db.z, db.w};
#pragma unroll
            for (int p = 0; p < NPAIR; ++p) {
                const float gv = pd[2 * p * G_PLANE];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(das[p], gv, acc[0][0], 0, 0, 0);
                if constexpr (MH == 2) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dbs[p], gv, acc[0][1], 0, 0, 0);
            }
        }
    }
    bwd_store<MH>(a, acc, R0 + wv, b0 + n, h, c0m, NR);
}

// ---- backward-data, WIDE loaders.  a.w3t is the IMAGE of the transposed filters (rows = C_in, inner = C_out); a.wdt only says
//      whether the shortcut's gradient is there.  g3 rows of 33 columns from column 32 bx (aligned): 9 words, LDS row pitch 36,
//      plane 8 x 36 = 288 = 32 (mod 64); gd rows of 32 columns: 8 words, plane 128 -> 160.
constexpr int BW_WPR = 9, BW_ROW = 4 * BW_WPR, BW_PLANE = B_PR * BW_ROW;      // 9, 36, 288
constexpr int BW_NWORD = CK * B_PR * BW_WPR;                                  // 576 words per chunk
constexpr int BW_NP = (BW_NWORD + NT - 1) / NT;                               // 3 per thread
constexpr int GW_PLANE = 160;                                                 // 4 x 32 = 128 -> 160 = 32 (mod 64)
static_assert(BW_PLANE % 64 == 32 && GW_PLANE % 64 == 32 && CK * TR * (TC / 4) == NT, "wide backward layout");

template <bool DOWN, int MH>
__global__ __launch_bounds__(NT, 2) void down_conv_bwd_wide_kernel(const BArgs a) {
    __shared__ f32x4 patch4[CK * BW_PLANE / 4];
    __shared__ f32x4 gdl4[CK * GW_PLANE / 4];
    __shared__ f32x4 wimg[MH * IMG_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, h = lane >> 5;
    int q = blockIdx.x;
    const int ct = q % a.gc;
    q /= a.gc;
    const int bxi = q % a.tx, rt = q / a.tx;
    const int R0 = rt * TR, b0 = bxi * TC, c0m = ct * 32 * MH, NR = a.B * a.Ho, nch = a.Cout / CK;
    const unsigned HWo = (unsigned)(a.Ho * a.Wo);
    const rsrc_t rg = make_rsrc(a.g3, (unsigned)a.B * (unsigned)a.Cout * HWo * 4u);
    const rsrc_t rgd = make_rsrc(DOWN ? a.gd : a.g3, (unsigned)a.B * (unsigned)a.Cout * HWo * 4u);
    const rsrc_t ri = make_rsrc(a.w3t, (unsigned)(a.Cin / 32) * (unsigned)nch * (unsigned)(IMG_G * 4));

    unsigned po[BW_NP], go;
#pragma unroll
    for (int i = 0; i < BW_NP; ++i) {
        const int e = tid + i * NT, c = e / (B_PR * BW_WPR), rem = e - c * (B_PR * BW_WPR), r = rem / BW_WPR, w = rem - r * BW_WPR;
        const int R = R0 + (r >> 1), bb = R / a.Ho, oy = R - bb * a.Ho + (r & 1), ox = b0 + 4 * w;    // Wo is a multiple of 4
        const bool ok = e < BW_NWORD && R < NR && oy < a.Ho && ox < a.Wo;
        po[i] = ok ? ((unsigned)(bb * a.Cout + c) * HWo + (unsigned)(oy * a.Wo + ox)) * 4u : OOB;
    }
    {
        const int c = tid / (TR * (TC / 4)), rem = tid - c * (TR * (TC / 4)), r = rem / (TC / 4), w = rem - r * (TC / 4);
        const int R = R0 + r, bb = R / a.Ho, oy = R - bb * a.Ho, ox = b0 + 4 * w;
        go = (R < NR && ox < a.Wo) ? ((unsigned)(bb * a.Cout + c) * HWo + (unsigned)(oy * a.Wo + ox)) * 4u : OOB;
    }
    const int gslot = (tid / (TR * (TC / 4))) * (GW_PLANE / 4) + (tid % (TR * (TC / 4)));      // word index in gdl4
    const unsigned wofs = (unsigned)tid * 16u;
    const unsigned gstep = (unsigned)nch * (unsigned)(IMG_G * 4);
    const unsigned g0 = (unsigned)(ct * MH) * gstep;

    f32x4 rp[BW_NP], rg4, rw[MH * IMG_NW];
    auto fetch = [&](const int ch) __attribute__((always_inline)) {
        const unsigned sp = (unsigned)(ch * CK) * HWo * 4u;
#pragma unroll
        for (int i = 0; i < BW_NP; ++i) rp[i] = ldb4(rg, po[i], sp);
        if constexpr (DOWN) rg4 = ldb4(rgd, go, sp);
        const unsigned sw = g0 + (unsigned)ch * (unsigned)(IMG_G * 4);
#pragma unroll
        for (int g = 0; g < MH; ++g)
#pragma unroll
            for (int j = 0; j < IMG_NW; ++j)
                rw[g * IMG_NW + j] = ldb4(ri, ((j + 1) * NT <= IMG_WORDS || j * NT + tid < IMG_WORDS) ? wofs + (unsigned)(j * NT * 16) : OOB,
                                          sw + (unsigned)g * gstep);
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < BW_NP; ++i)
            if ((i + 1) * NT <= BW_NWORD || tid + i * NT < BW_NWORD) patch4[tid + i * NT] = rp[i];
        if constexpr (DOWN) gdl4[gslot] = rg4;
#pragma unroll
        for (int g = 0; g < MH; ++g)
#pragma unroll
            for (int j = 0; j < IMG_NW; ++j)
                if ((j + 1) * NT <= IMG_WORDS || j * NT + tid < IMG_WORDS) wimg[g * IMG_WORDS + j * NT + tid] = rw[g * IMG_NW + j];
    };

    f32x16 acc[4][MH];     // [parity class 2 * py + px][channel half]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[c][m][v] = 0.f;
    const float* pg = reinterpret_cast<const float*>(patch4) + h * BW_PLANE + (2 * wv) * BW_ROW + n;
    const float* pd = reinterpret_cast<const float*>(gdl4) + h * GW_PLANE + wv * TC + n;
    const float* wf = reinterpret_cast<const float*>(wimg);
    const float* aA = wf + (n * 2 + h) * A_PITCH;
    const float* aB = aA + (MH == 2 ? IMG_G : 0);
    const float* dA = wf + IMG_A + (n * 2 + h) * NPAIR;
    const float* dB = dA + (MH == 2 ? IMG_G : 0);

    fetch(0);
    for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();
        commit();
        __syncthreads();
        if (ch + 1 < nch) fetch(ch + 1);
        float bv[NPAIR][2][2];            // g3 at (a + dy, b + dx) of the lane's class pixel, per channel pair
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {
            bv[p][0][0] = pg[2 * p * BW_PLANE];
            bv[p][0][1] = pg[2 * p * BW_PLANE + 1];
            bv[p][1][0] = pg[2 * p * BW_PLANE + BW_ROW];
            bv[p][1][1] = pg[2 * p * BW_PLANE + BW_ROW + 1];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t % 3;
            const int cls = 2 * (ky != 1 ? 1 : 0) + (kx != 1 ? 1 : 0), dy = ky == 0 ? 1 : 0, dx = kx == 0 ? 1 : 0;
            const float4 wa = *reinterpret_cast<const float4*>(aA + 4 * t);
            const float4 wb = *reinterpret_cast<const float4*>(aB + 4 * t);
            const float was[4] = {wa.x, wa.y, wa.z, wa.w}, wbs[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
            for (int p = 0; p < NPAIR; ++p) {
                acc[cls][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(was[p], bv[p][dy][dx], acc[cls][0], 0, 0, 0);
                if constexpr (MH == 2)
                    acc[cls][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wbs[p], bv[p][dy][dx], acc[cls][1], 0, 0, 0);
            }
        }
        if constexpr (DOWN) {
            const float4 da = *reinterpret_cast<const float4*>(dA), db = *reinterpret_cast<const float4*>(dB);
            const float das[4] = {da.x, da.y, da.z, da.w}, dbs[4] = {db.x, db.y, db.z, db.w};
#pragma unroll
            for (int p = 0; p < NPAIR; ++p) {
                const float gv = pd[2 * p * GW_PLANE];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(das[p], gv, acc[0][0], 0, 0, 0);
                if constexpr (MH == 2) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dbs[p], gv, acc[0][1], 0, 0, 0);
            }
        }
    }
    bwd_store<MH>(a, acc, R0 + wv, b0 + n, h, c0m, NR);
}

int check_sizes(int B, int Cin, int Cout, int H, int W) {
    DMH_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H >= 2 && W >= 2 && (H & 1) == 0 && (W & 1) == 0,
                "input height and width must be even");
    const int64_t cmax = Cin > Cout ? Cin : Cout;
    DMH_REQUIRE((int64_t)B * cmax * H * W < ((int64_t)1 << 29) && (int64_t)Cin * Cout < ((int64_t)1 << 24),
                "tensor too large (32-bit byte offsets)");
    return DMH_OK;
}

}  // namespace

extern "C" {

int dmh_down_conv_fwd_act(const float* x, const float* w3, const float* wd, const float* shift3, const float* shiftd,
                          int relu3, int B, int Cin, int Cout, int H, int W, float* y3, float* yd, void* stream);

int dmh_down_conv_fwd(const float* x, const float* w3, const float* wd, int B, int Cin, int Cout, int H, int W,
                      float* y3, float* yd, void* stream) {
    return dmh_down_conv_fwd_act(x, w3, wd, nullptr, nullptr, 0, B, Cin, Cout, H, W, y3, yd, stream);
}

int dmh_down_conv_fwd_act(const float* x, const float* w3, const float* wd, const float* shift3, const float* shiftd,
                          int relu3, int B, int Cin, int Cout, int H, int W, float* y3, float* yd, void* stream) {
    DMH_REQUIRE(x && w3 && y3 && ((wd == nullptr) == (yd == nullptr)), "null pointer");
    DMH_REQUIRE(shiftd == nullptr || wd != nullptr, "shiftd without the shortcut convolution");
    if (int rc = check_sizes(B, Cin, Cout, H, W)) return rc;
    DMH_REQUIRE(Cin % CK == 0 && Cout % 64 == 0, "C_in must be a multiple of 8 and C_out of 64");
    FArgs a;
    a.x = x;
    a.w3 = w3;
    a.wd = wd;
    a.shift3 = shift3;
    a.shiftd = shiftd;
    a.relu3 = relu3 ? 1 : 0;
    a.y3 = y3;
    a.yd = yd;
    a.B = B;
    a.Cin = Cin;
    a.Cout = Cout;
    a.H = H;
    a.W = W;
    a.Ho = H / 2;
    a.Wo = W / 2;
    a.gx = (a.Wo + TC - 1) / TC;
    a.gy = (B * a.Ho + TR - 1) / TR;
    a.gk = Cout / 64;
    const long long blocks = (long long)a.gx * a.gy * a.gk;
    DMH_REQUIRE(blocks < (1ll << 31), "grid too large");
    if (wd)
        hipLaunchKernelGGL(down_conv_fwd_kernel<true>, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(down_conv_fwd_kernel<false>, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, a);
    return check_launch("dmh_down_conv_fwd_act");
}

int64_t dmh_down_conv_image_size(int rows, int inner) {
    if (rows <= 0 || inner <= 0 || rows % 32 || inner % CK) return -1;
    return (int64_t)(rows / 32) * (inner / CK) * IMG_G;
}

int dmh_down_conv_weight_image(const float* w3, const float* wd, int rows, int inner, float* image, void* stream) {
    DMH_REQUIRE(w3 && image, "null pointer");
    const int64_t n = dmh_down_conv_image_size(rows, inner);
    DMH_REQUIRE(n > 0 && n < ((int64_t)1 << 29), "rows must be a multiple of 32 and inner of 8");
    hipLaunchKernelGGL(down_weight_image_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, w3, wd, rows,
                       inner, image);
    return check_launch("dmh_down_conv_weight_image");
}

int dmh_down_conv_fwd_img(const float* x, const float* image, int has_down, const float* shift3, const float* shiftd, int relu3,
                          int B, int Cin, int Cout, int H, int W, float* y3, float* yd, void* stream) {
    DMH_REQUIRE(x && image && y3 && ((has_down != 0) == (yd != nullptr)), "null pointer");
    DMH_REQUIRE(shiftd == nullptr || has_down, "shiftd without the shortcut convolution");
    if (int rc = check_sizes(B, Cin, Cout, H, W)) return rc;
    DMH_REQUIRE(Cin % CK == 0 && Cout % 64 == 0, "C_in must be a multiple of 8 and C_out of 64");
    DMH_REQUIRE(W % 4 == 0, "the image form reads 16-byte words: W must be a multiple of 4 (use dmh_down_conv_fwd_act otherwise)");
    FArgs a;
    a.x = x;
    a.w3 = image;
    a.wd = has_down ? image : nullptr;
    a.shift3 = shift3;
    a.shiftd = shiftd;
    a.relu3 = relu3 ? 1 : 0;
    a.y3 = y3;
    a.yd = yd;
    a.B = B;
    a.Cin = Cin;
    a.Cout = Cout;
    a.H = H;
    a.W = W;
    a.Ho = H / 2;
    a.Wo = W / 2;
    a.gx = (a.Wo + TC - 1) / TC;
    a.gy = (B * a.Ho + TR - 1) / TR;
    a.gk = Cout / 64;
    const long long blocks = (long long)a.gx * a.gy * a.gk;
    DMH_REQUIRE(blocks < (1ll << 31), "grid too large");
    if (has_down)
        hipLaunchKernelGGL(down_conv_fwd_wide_kernel<true>, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(down_conv_fwd_wide_kernel<false>, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, a);
    return check_launch("dmh_down_conv_fwd_img");
}

int dmh_down_conv_bwd_data(const float* g3, const float* gd, const float* w3t, const float* wdt, int B, int Cin, int Cout,
                           int H, int W, float* g_x, void* stream) {
    return dmh_down_conv_bwd_data_acc(g3, gd, w3t, wdt, nullptr, B, Cin, Cout, H, W, g_x, stream);
}

int dmh_down_conv_bwd_data_acc(const float* g3, const float* gd, const float* w3t, const float* wdt, const float* g_add, int B,
                               int Cin, int Cout, int H, int W, float* g_x, void* stream) {
    DMH_REQUIRE(g3 && w3t && g_x && ((gd == nullptr) == (wdt == nullptr)), "null pointer");
    if (int rc = check_sizes(B, Cin, Cout, H, W)) return rc;
    DMH_REQUIRE(Cout % CK == 0 && Cin % 64 == 0, "C_out must be a multiple of 8 and C_in of 64");
    const long long tiles = (long long)((W / 2 + TC - 1) / TC) * ((B * (H / 2) + TR - 1) / TR);
    const int mh = tiles * (Cin / 64) < 200 ? 1 : 2;       // too few 64-channel tiles for 256 CUs: 32-channel tiles
    BArgs a;
    a.g3 = g3;
    a.gd = gd;
    a.w3t = w3t;
    a.wdt = wdt;
    a.gadd = g_add;
    a.gx = g_x;
    a.B = B;
    a.Cin = Cin;
    a.Cout = Cout;
    a.H = H;
    a.W = W;
    a.Ho = H / 2;
    a.Wo = W / 2;
    a.tx = (a.Wo + TC - 1) / TC;
    a.ty = (B * a.Ho + TR - 1) / TR;
    a.gc = Cin / (32 * mh);
    const long long blocks = (long long)a.tx * a.ty * a.gc;
    DMH_REQUIRE(blocks < (1ll << 31), "grid too large");
    if (gd && mh == 2)
        hipLaunchKernelGGL((down_conv_bwd_kernel<true, 2>), dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, a);
    else if (gd)
        hipLaunchKernelGGL((down_conv_bwd_kernel<true, 1>), dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, a);
    else if (mh == 2)
        hipLaunchKernelGGL((down_conv_bwd_kernel<false, 2>), dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((down_conv_bwd_kernel<false, 1>), dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, a);
    return check_launch("dmh_down_conv_bwd_data");
}

int dmh_down_conv_bwd_data_img(const float* g3, const float* gd, const float* image, const float* g_add, int B, int Cin, int Cout,
                               int H, int W, float* g_x, void* stream) {
    DMH_REQUIRE(g3 && image && g_x, "null pointer");
    if (int rc = check_sizes(B, Cin, Cout, H, W)) return rc;
    DMH_REQUIRE(Cout % CK == 0 && Cin % 64 == 0, "C_out must be a multiple of 8 and C_in of 64");
    DMH_REQUIRE(W % 8 == 0, "the image form reads 16-byte words: W must be a multiple of 8 (use dmh_down_conv_bwd_data_acc otherwise)");
    const long long tiles = (long long)((W / 2 + TC - 1) / TC) * ((B * (H / 2) + TR - 1) / TR);
    const int mh = tiles * (Cin / 64) < 200 ? 1 : 2;       // as dmh_down_conv_bwd_data_acc
    BArgs a;
    a.g3 = g3;
    a.gd = gd;
    a.w3t = image;
    a.wdt = gd ? image : nullptr;
    a.gadd = g_add;
    a.gx = g_x;
    a.B = B;
    a.Cin = Cin;
    a.Cout = Cout;
    a.H = H;
    a.W = W;
    a.Ho = H / 2;
    a.Wo = W / 2;
    a.tx = (a.Wo + TC - 1) / TC;
    a.ty = (B * a.Ho + TR - 1) / TR;
    a.gc = Cin / (32 * mh);
    const long long blocks = (long long)a.tx * a.ty * a.gc;
    DMH_REQUIRE(blocks < (1ll << 31), "grid too large");
    if (gd && mh == 2)
        hipLaunchKernelGGL((down_conv_bwd_wide_kernel<true, 2>), dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, a);
    else if (gd)
        hipLaunchKernelGGL((down_conv_bwd_wide_kernel<true, 1>), dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, a);
    else if (mh == 2)
        hipLaunchKernelGGL((down_conv_bwd_wide_kernel<false, 2>), dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((down_conv_bwd_wide_kernel<false, 1>), dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, a);
    return check_launch("dmh_down_conv_bwd_data_img");
}

}  // extern "C"
