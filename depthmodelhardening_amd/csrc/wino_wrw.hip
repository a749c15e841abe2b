// K18 -- weight gradient of the 3x3 stride-1 convolutions with >= 64 channels on both sides, in the Winograd F(2x2, 3x3)
// domain on v_mfma_f32_32x32x2_f32 (the train pass of MD2/networks/resnet_encoder.py:85-98 BasicBlocks and of the decoder's
// Conv3x3, MD2/layers.py:127-141; autograd's aten.convolution_backward(..., [False, True, False])).
//
// Why: through round 2 these weight gradients were MIOpen's NHWC implicit GEMMs (12.2 ms per step at 75-114 TFLOP/s) wrapped
// in layout transposes (2.9 ms), and their atomics made the training step irreproducible in the last bits.  With
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        the gradient is       dg = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G,
// i.e. per transform position p one GEMM  dU_p[k][c] = sum_t dM_p[k][t] * V_p[c][t]  whose reduction axis is the TILE index:
// 16/36 of the multiplications of the direct form, on the same fp32 MFMA and with the same LDS images as K10.
//
//   * workgroup   : one (64 output channels) x (64 input channels) block of dU, all 16 positions = 65,536 accumulators = 256
//                   registers per lane (wave w: channel sub-block (w & 1, w >> 1)); it walks a contiguous slice of the
//                   tile chunks and never leaves its accumulators: no per-item epilogue at all.
//   * chunk       : 8 consecutive tiles of one tile row (2 x 16 output-gradient pixels, 4 x 18 input pixels) of the 64 + 64
//                   channels.  Raw rows through registers into LDS (buffer loads: uniform row offsets in SGPRs, an
//                   out-of-range offset = zero padding = masked ragged edge), both transforms LDS -> LDS into the images
//                   dM[p][h][k][4], V[p][h][c][4] (h + 2s = tile: a 16-byte operand read feeds 4 k-steps, as in K10),
//                   64 MFMAs per wave.  Images single-buffered, raw rows double-buffered: MFMA phase / transform phase with
//                   two barriers per chunk (K17's scheme; the fp32 MFMA shares the vector pipe, so nothing is lost by
//                   taking the transforms out from between the MFMAs).
//   * split       : the 256 workgroups are dealt over the (k-block, c-block) pairs, each pair's chunks split into equal
//                   contiguous slices; every workgroup stores its partial dU, and wino_wrw_reduce_kernel adds the slices in
//                   a fixed order and applies G^T . G  -> dw[K][C][3][3].  No atomics: bitwise reproducible.
#include <stdlib.h>

#include "common.hpp"

using namespace dmh;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef DMH_WRW_ABLATE      // timing experiments (tools/wrw_ablate.sh): 1 no global loads, 2 no transforms, 4 no raw LDS stores,
#define DMH_WRW_ABLATE 0    // 8 no MFMAs, 16 no barriers in the loop (results are garbage)
#endif
constexpr int NT = 256;
constexpr int TPC = 8;                       // tiles per chunk
constexpr int XC = 24;                       // input columns staged per row: the 18 of the chunk inside six aligned 16-byte words
constexpr int DC = 2 * TPC;                  // 16 gradient columns, 2 rows
constexpr int XCS = 4 * XC + 4;              // raw channel strides in floats (multiples of 4: 16-byte LDS writes)
constexpr int DCS = 2 * DC + 4;

// Block shapes: KS x CS sub-blocks of 32 x 32 (output x input channels) per workgroup.  <2,2> = 64 x 64 (the >= 64-channel
// layers), <1,3> = 32 x 96 (upconv(1,1)), <1,2> = 32 x 64 (upconv(1,0)): the 32-output-channel decoder layers.
template <int KS, int CS>
struct Cfg {
    static constexpr int KCH = 32 * KS, CCH = 32 * CS;
    static constexpr int XRAW = CCH * XCS, DRAW = KCH * DCS;
    static constexpr int RAWBUF = XRAW + DRAW;                      // floats per raw buffer
    static constexpr int IMGM = 16 * 2 * KCH, IMGV = 16 * 2 * CCH;  // f32x4 words of the two images
    // 16-byte raw loads per thread and chunk (a vector-memory instruction costs the wave ~64 cycles among the MFMAs whatever
    // its width: few, wide loads)
    static constexpr int NXL = CCH * 24 / NT, NDL = KCH * 8 / NT;
    static constexpr int NACC = 4 * KS * CS;                        // accumulator tiles per wave: 4 positions x the sub-blocks
    static constexpr size_t SMEM = (size_t)(IMGM + IMGV) * 16 + (size_t)2 * RAWBUF * 4;
    static_assert(CCH * 24 % NT == 0 && KCH * 8 % NT == 0, "whole load items per thread");
    static_assert(NACC * 16 <= 256 && SMEM <= 160 * 1024, "accumulators / LDS");
};

struct RArgs {
    const float* x;
    const float* dy;
    float* ws;                  // [pairs][S][16][KCH][CCH]
    int B, C, K, H, W, Ho, Wo, pad;
    int Ht, cpr;                // rows of tiles per image, chunks per tile row
    int nchunks, nk, nc, S, cps;   // chunks in total; channel blocks; slices per pair; chunks per slice
};

// IL (round 6): the interleaved form of the chunk loop -- see "interleaved form" below.  Same sums in the same order: the two
// forms give bit-identical results (tests/test_gpu_conv_anchor.py).
template <int KS, int CS, bool IL>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_wrw_kernel(RArgs a) {
    using G = Cfg<KS, CS>;
    constexpr int KCH = G::KCH, CCH = G::CCH, NXL = G::NXL, NDL = G::NDL, XRAW = G::XRAW, RAWBUF = G::RAWBUF;
    extern __shared__ f32x4 smem[];
    f32x4* M_lds = smem;                                              // dM image [16][2][KCH]
    f32x4* V_lds = smem + G::IMGM;                                    // V image  [16][2][CCH]
    float* raw = reinterpret_cast<float*>(smem + G::IMGM + G::IMGV);  // [2][ x: CCH x XCS | dy: KCH x DCS ]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wv_s = __builtin_amdgcn_readfirstlane(wv);
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;
    const int pair = (int)blockIdx.x / a.S, slice = (int)blockIdx.x - pair * a.S;
    const int kb = pair / a.nc, cb = pair - kb * a.nc;
    const int c_first = slice * a.cps, c_end = min(c_first + a.cps, a.nchunks);
    const int n = c_end - c_first;                               // may be <= 0: the block then only stores zeros

    const rsrc_t xrs = make_rsrc(a.x, (unsigned)((size_t)a.B * a.C * HW * 4));
    const rsrc_t drs = make_rsrc(a.dy, (unsigned)((size_t)a.B * a.K * HoWo * 4));
    // raw loads, 16 bytes each, lanes running ALONG the rows (six adjacent lanes read the 96 bytes of one x row, four the 64
    // bytes of one dy row: every cache line is touched by one instruction; with one row per lane the lines were re-fetched
    // for each word).  x: item = tid + 256 k -> (channel item / 24, row (item % 24) / 6, word item % 6), the words being the
    // aligned 16-byte words of the row from the chunk's first column rounded down to a multiple of 4 (coff = columns in
    // front of the chunk: 0 / 3 for pad 0 / 1; an image row holds whole words on either side of its zero padding: W is a
    // multiple of 4 where pad = 1 matters, the encoder maps; the pre-padded decoder inputs, pad 0, have no padding).
    // dy: item -> (channel item / 8, row (item % 8) / 4, word item % 4).  Everything about an item is a per-thread constant.
    const int coff = (4 - (a.pad & 3)) & 3;
    unsigned xg[NXL], xl[NXL], xm[NXL], dg[NDL], dl[NDL];
#pragma unroll
    for (int k = 0; k < NXL; ++k) {
        const int item = tid + NT * k, ch = item / 24, rem = item - ch * 24, r = rem / 6, f = rem - r * 6;
        xg[k] = (unsigned)(ch * HW + r * a.W + 4 * f);
        xl[k] = (unsigned)(ch * XCS + r * XC + 4 * f);
        xm[k] = (r == 0 ? 1u : 0u) | (r == 3 ? 2u : 0u) | (f == 0 ? 4u : 0u) | (f == 5 ? 8u : 0u);
    }
#pragma unroll
    for (int k = 0; k < NDL; ++k) {
        const int item = tid + NT * k, ch = item >> 3, r = (item >> 2) & 1, f = item & 3;
        dg[k] = (unsigned)(ch * HoWo + r * a.Wo + 4 * f);
        dl[k] = (unsigned)(XRAW + ch * DCS + r * DC + 4 * f);
    }

    f32x4 rx[NXL], rd[NDL];
    // raw loads of chunk `ch` (clamped to the slice: past its end the last chunk is re-read and never used): chunk_at() forms the
    // wave-uniform part, load_x(k) / load_d(k) issue one 16-byte load each, store_x(k) / store_d(k) put a register into the raw
    // buffer -- one at a time, so that the steady-state loop can place each of them in the shadow of a group of MFMAs (a
    // vector-memory instruction issued in a block of its own costs the wave ~64 cycles of nothing else)
    struct Chunk { unsigned xbase, dbase, bad; int wrem; };
    auto chunk_at = [&](int ch) __attribute__((always_inline)) {
        ch = min(max(ch, c_first), max(c_end - 1, c_first));
        ch = min(ch, a.nchunks - 1);
        // (chunks walk along the rows of tiles; walking an image in column strips, so that consecutive chunks share two of their
        // four input rows, measured 4 % SLOWER: the row pieces of consecutive chunks are then no longer neighbours in memory)
        const int b = ch / (a.Ht * a.cpr), rem = ch - b * (a.Ht * a.cpr), ty = rem / a.cpr, tx0 = (rem - ty * a.cpr) * TPC;
        const int iy0 = 2 * ty - a.pad, ixa = 2 * tx0 - a.pad - coff;          // ixa: multiple of 4 (tx0 is a multiple of 8)
        Chunk c;
        // which border rows / words of this chunk lie in the zero padding (uniform); an item in the padding reads at an
        // out-of-range offset, i.e. 0
        c.bad = (iy0 < 0 ? 1u : 0u) | (iy0 + 3 >= a.H ? 2u : 0u) | (ixa < 0 ? 4u : 0u) | (ixa + 20 >= a.W ? 8u : 0u);
        c.xbase = (unsigned)(((b * a.C + cb * CCH) * a.H + iy0) * a.W + ixa);     // may wrap below 0: sums are mod 2^32
        c.dbase = (unsigned)(((b * a.K + kb * KCH) * a.Ho + 2 * ty) * a.Wo + 2 * tx0);
        c.wrem = a.Wo - 2 * tx0;                                // gradient columns of the chunk inside the row
        return c;
    };
    auto load_x = [&](const Chunk& c, const int k) __attribute__((always_inline)) {
        const unsigned vo = (xm[k] & c.bad) ? 0xFFFFFF00u : (c.xbase + xg[k]) * 4u;      // beyond num_records (<= 0xFFFFFF00, checked on the host): reads 0
        rx[k] = (DMH_WRW_ABLATE & 1) ? f32x4{1.f, 1.f, 1.f, 1.f}
                                     : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, 0, 0));
    };
    // (the ragged right edge of dy -- columns beyond the row are other rows' -- is masked when the register is stored)
    auto load_d = [&](const Chunk& c, const int k) __attribute__((always_inline)) {
        rd[k] = (DMH_WRW_ABLATE & 1) ? f32x4{1.f, 1.f, 1.f, 1.f}
                                     : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, (c.dbase + dg[k]) * 4u, 0, 0));
    };
    auto store_x = [&](const int buf, const int k) __attribute__((always_inline)) {
        *reinterpret_cast<f32x4*>(raw + buf * RAWBUF + xl[k]) = rx[k];
    };
    auto store_d = [&](const int buf, const int k, const int wrem) __attribute__((always_inline)) {
        f32x4 v = rd[k];
        if (wrem < DC) {                                        // uniform
            const int c0 = 4 * ((tid + NT * k) & 3);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (c0 + j < wrem) ? v[j] : 0.f;
        }
        *reinterpret_cast<f32x4*>(raw + buf * RAWBUF + dl[k]) = v;
    };
    int wrem_held = DC;          // of the chunk whose dy words sit in rd[]
    auto load_chunk = [&](int ch) __attribute__((always_inline)) {
        const Chunk c = chunk_at(ch);
#pragma unroll
        for (int k = 0; k < NXL; ++k) load_x(c, k);
#pragma unroll
        for (int k = 0; k < NDL; ++k) load_d(c, k);
        wrem_held = c.wrem;
    };
    auto store_raw = [&](const int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NXL; ++k) store_x(buf, k);
#pragma unroll
        for (int k = 0; k < NDL; ++k) store_d(buf, k, wrem_held);
    };
    // transforms of one chunk: work item = (channel, tile pair tp): tiles h + 4j, h + 4j + 2 with h = tp & 1, j = tp >> 1; the
    // two tiles are components 2j, 2j + 1 of the channel's 16-byte image word -> one 8-byte write per position.  Items are
    // dealt channel-fastest over the threads (64 channels: lane = channel, wave = tile pair).
    auto xform_d = [&](const int buf, const int ch, const int tp) __attribute__((always_inline)) {
        // dM = A dY A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]
        const int th = tp & 1, tj = tp >> 1;
        const float* ds = raw + buf * RAWBUF + XRAW + ch * DCS + 2 * (th + 4 * tj);
        float* const md = reinterpret_cast<float*>(M_lds + th * KCH + ch) + 2 * tj;
        float m[2][4][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float2 r0 = *reinterpret_cast<const float2*>(ds + 4 * t), r1 = *reinterpret_cast<const float2*>(ds + DC + 4 * t);
            const float u[4][2] = {{r0.x, r0.y}, {r0.x + r1.x, r0.y + r1.y}, {r0.x - r1.x, r0.y - r1.y}, {-r1.x, -r1.y}};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                m[t][i][0] = u[i][0];
                m[t][i][1] = u[i][0] + u[i][1];
                m[t][i][2] = u[i][0] - u[i][1];
                m[t][i][3] = -u[i][1];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float2*>(md + (i * 4 + j) * (8 * KCH)) = make_float2(m[0][i][j], m[1][i][j]);
    };
    auto xform_x = [&](const int buf, const int ch, const int tp) __attribute__((always_inline)) {
        // V = B^T d B
        const int th = tp & 1, tj = tp >> 1;
        const float* xs = raw + buf * RAWBUF + ch * XCS + coff + 2 * (th + 4 * tj);  // tile A; tile B: + 4 columns
        float* const vd = reinterpret_cast<float*>(V_lds + th * CCH + ch) + 2 * tj;
        float v[2][4][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float d[4][4], q[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* row = xs + i * XC + 4 * t;      // 4-byte aligned only (coff is odd for pad 1): four dword reads,
                d[i][0] = row[0]; d[i][1] = row[1]; d[i][2] = row[2]; d[i][3] = row[3];   // paired by the compiler (ds_read2_b32)
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                q[0][j] = d[0][j] - d[2][j];
                q[1][j] = d[1][j] + d[2][j];
                q[2][j] = d[2][j] - d[1][j];
                q[3][j] = d[1][j] - d[3][j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[t][i][0] = q[i][0] - q[i][2];
                v[t][i][1] = q[i][1] + q[i][2];
                v[t][i][2] = q[i][2] - q[i][1];
                v[t][i][3] = q[i][1] - q[i][3];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float2*>(vd + (i * 4 + j) * (8 * CCH)) = make_float2(v[0][i][j], v[1][i][j]);
    };
    auto transform = [&](const int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < (KCH * 4 + NT - 1) / NT; ++r) {
            const int it = tid + NT * r;
            if ((KCH * 4) % NT == 0 || it < KCH * 4) xform_d(buf, it % KCH, it / KCH);
        }
#pragma unroll
        for (int r = 0; r < (CCH * 4 + NT - 1) / NT; ++r) {
            const int it = tid + NT * r;
            if ((CCH * 4) % NT == 0 || it < CCH * 4) xform_x(buf, it % CCH, it / CCH);
        }
    };

    // MFMA phase: wave w owns positions 4w .. 4w+3 of every sub-block: per position KS + CS operand words (16 bytes = the 4
    // k-steps of a chunk) feed 4 KS CS MFMAs, and consecutive MFMAs run on different accumulators.
    f32x16 acc[G::NACC];
#pragma unroll
    for (int p = 0; p < G::NACC; ++p)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[p][v] = 0.f;
    const f32x4* const Mw = M_lds + (4 * wv_s) * (2 * KCH) + (lane >> 5) * KCH + (lane & 31);
    const f32x4* const Vw = V_lds + (4 * wv_s) * (2 * CCH) + (lane >> 5) * CCH + (lane & 31);

    if (IL && n > 0) {
        // ---- interleaved form.  The two-phase loop below it leaves the matrix pipe idle while the transforms run (LDS read ->
        // arithmetic -> LDS write, a chain of latencies that one wave per SIMD cannot cover: 107 of 388 us at layer1's shape,
        // tools/wrw_ablate.py) because the images are single-buffered and there is no LDS for a second pair (2 x 64 KB).  But an
        // image word holds FOUR tiles (its components = the 4 k-steps of a chunk), and a k-step only reads its own component:
        // the chunk is cut in two halves by k-step, J = 0: components .xy (tiles h, h + 2), J = 1: .zw (tiles h + 4, h + 6):
        //     phase A(g):  MFMAs of chunk g, k-steps 0-1 (read .xy)   beside   transform of chunk g,     tiles of J = 1 (write .zw)
        //     phase B(g):  MFMAs of chunk g, k-steps 2-3 (read .zw)   beside   transform of chunk g + 1, tiles of J = 0 (write .xy)
        //                  + raw registers (chunk g + 2) -> raw[g & 1], refill with chunk g + 3
        // one barrier after each phase (as many as before), every accumulator still receives k-steps 0, 1, 2, 3 of every chunk
        // in that order.  A transform work item is ONE tile of one channel (4-byte image writes instead of 8-byte ones), the
        // d-tiles first, then the x-tiles, dealt over the threads; a phase is 8 fenced slots of KS CS MFMAs, the transform's
        // reads in slot 0, its arithmetic in slots 2-3, its writes in slots 4-7.
        constexpr int NDT = KCH * 4, NXT = CCH * 4;                 // d- and x-tiles of one half J
        static_assert(NDT <= NT && NDT % 64 == 0 && NXT % 64 == 0, "tile items are dealt by whole waves");
        constexpr int NXI = (NXT + NT - 1) / NT;                    // x-tiles per thread and phase (1 or 2)
        constexpr int XTAIL = NXT % NT;                             // the last round's items go to the UPPER threads (the lower
                                                                    // ones hold the d-tiles when those do not fill the workgroup)
        float* const Mf = reinterpret_cast<float*>(M_lds);
        float* const Vf = reinterpret_cast<float*>(V_lds);
        const bool has_d = NDT == NT || wv_s * 64 < NDT;            // wave-uniform
        const int d_ch = tid % KCH, d_q = (tid / KCH) & 3;
        const int d_off_r = XRAW + d_ch * DCS + 2 * ((d_q & 1) + 2 * (d_q >> 1));           // + 8 J (tile = th + 4 J + 2 s)
        // the images of this form are [position][h][J][channel] 8-BYTE words (the two tiles of half J a k-step pair reads):
        // an operand read is one lane-linear ds_read_b64, and the 4-byte writes of 32 consecutive channels fall on 32 different
        // banks two by two (in the 16-byte words of the other form they met four by four)
        const int d_off_w = (d_q & 1) * (4 * KCH) + 2 * d_ch + (d_q >> 1);                  // + 2 KCH J
        int x_off_r[NXI], x_off_w[NXI];
        bool has_x[NXI];
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            const bool tail = XTAIL != 0 && k == NXI - 1;
            const int e = tail ? NT * k + tid - (NT - XTAIL) : NT * k + tid;
            has_x[k] = !tail || wv_s * 64 >= NT - XTAIL;
            const int ee = has_x[k] ? e : 0, ch = ee % CCH, q = ee / CCH;
            x_off_r[k] = ch * XCS + coff + 2 * ((q & 1) + 2 * (q >> 1));
            x_off_w[k] = (q & 1) * (4 * CCH) + 2 * ch + (q >> 1);
        }
        // Round 6: both transforms on packed adds (common.hpp: the fp32 MFMA shadows no vector instruction, and this loop carried
        // 88 scalar additions per chunk beside its 64 MFMAs).  A patch row arrives as column pairs -- ds_read2_b32 puts two
        // adjacent columns into a register pair --, the row pass works on such pairs, the column pass and the 2 x 2 butterflies of
        // the d-tile use the operand selects of v_pk_add_f32.  The same additions on the same operands: bit-identical sums.
        f32x2 dr0, dr1, xp[NXI][4][2];                              // the tiles' patches: d rows; x rows as (columns 0-1, columns 2-3)
        float dt[16], xt[NXI][16];                                  // the 16 transformed values of a tile
        auto t_read = [&](const int buf, const int J) __attribute__((always_inline)) {
            if (has_d) {
                const float* ds = raw + buf * RAWBUF + d_off_r + 8 * J;
                dr0 = *reinterpret_cast<const f32x2*>(ds);
                dr1 = *reinterpret_cast<const f32x2*>(ds + DC);
            }
#pragma unroll
            for (int k = 0; k < NXI; ++k)
                if (has_x[k]) {
                    const float* xs = raw + buf * RAWBUF + x_off_r[k] + 8 * J;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float* row = xs + i * XC;             // 4-byte aligned only (coff is odd for pad 1)
                        xp[k][i][0] = f32x2{row[0], row[1]};
                        xp[k][i][1] = f32x2{row[2], row[3]};
                    }
                }
        };
        auto t_math_d = [&]() __attribute__((always_inline)) {          // dM = A dY A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]
            if (has_d) {
                // rows of A dY: u(i) = (u0[i], u1[i]) = dr0, dr0 + dr1, dr0 - dr1, -dr1; then (u0, u0 + u1, u0 - u1, -u1) of each
                const f32x2 u1 = pk_add(dr0, dr1), u2 = pk_sub(dr0, dr1);
                const f32x2 m0 = pk_bfly(dr0, dr0), m1 = pk_bfly(u1, u1), m2 = pk_bfly(u2, u2), m3 = pk_bfly_neg(dr1, dr1);
                dt[0] = dr0.x;  dt[1] = m0.x;  dt[2] = m0.y;  dt[3] = -dr0.y;
                dt[4] = u1.x;   dt[5] = m1.x;  dt[6] = m1.y;  dt[7] = -u1.y;
                dt[8] = u2.x;   dt[9] = m2.x;  dt[10] = m2.y; dt[11] = -u2.y;
                dt[12] = -dr1.x; dt[13] = m3.x; dt[14] = m3.y; dt[15] = dr1.y;
            }
        };
        auto t_math_x = [&](const int k) __attribute__((always_inline)) {     // V = B^T d B
            if (has_x[k]) {
                f32x2 q[4][2];
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    q[0][jp] = pk_sub(xp[k][0][jp], xp[k][2][jp]);
                    q[1][jp] = pk_add(xp[k][1][jp], xp[k][2][jp]);
                    q[2][jp] = pk_sub(xp[k][2][jp], xp[k][1][jp]);
                    q[3][jp] = pk_sub(xp[k][1][jp], xp[k][3][jp]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2 o01 = pk_col01(q[i][0], q[i][1]), o23 = pk_col23(q[i][0], q[i][1]);
                    xt[k][4 * i + 0] = o01.x;       // q0 - q2
                    xt[k][4 * i + 1] = o01.y;       // q1 + q2
                    xt[k][4 * i + 2] = o23.x;       // q2 - q1
                    xt[k][4 * i + 3] = o23.y;       // q1 - q3
                }
            }
        };
        // image writes, 4 bytes each: position pp of the tile -> component s of word [pp][h][J][channel]
        auto t_write_d = [&](const int J, const int part) __attribute__((always_inline)) {      // positions 8 part .. 8 part + 7
            if (has_d) {
                float* md = Mf + d_off_w + 2 * KCH * J;
#pragma unroll
                for (int pp = 8 * part; pp < 8 * part + 8; ++pp) md[pp * (8 * KCH)] = dt[pp];
            }
        };
        auto t_write_x = [&](const int J, const int k, const int part) __attribute__((always_inline)) {
            if (has_x[k]) {
                float* vd = Vf + x_off_w[k] + 2 * CCH * J;
#pragma unroll
                for (int pp = 8 * part; pp < 8 * part + 8; ++pp) vd[pp * (8 * CCH)] = xt[k][pp];
            }
        };
        const float2* const Mw2 = reinterpret_cast<const float2*>(M_lds) + (4 * wv_s) * (4 * KCH) + (lane >> 5) * (2 * KCH) + (lane & 31);
        const float2* const Vw2 = reinterpret_cast<const float2*>(V_lds) + (4 * wv_s) * (4 * CCH) + (lane >> 5) * (2 * CCH) + (lane & 31);
        constexpr int NST = NXL + NDL;
        // one phase: the 8 (position, k-step) MFMA groups of half J on the images, with the transform of (tbuf, tJ) and --
        // STAGE -- the raw stores / refill loads dealt over its slots
        auto phase = [&](const int J, const int tbuf, const int tJ, const bool stage, const int sbuf, const int g)
            __attribute__((always_inline)) {
            float2 ua[2][KS], vb[2][CS];
#pragma unroll
            for (int i = 0; i < KS; ++i) ua[0][i] = Mw2[J * KCH + i * 32];
#pragma unroll
            for (int i = 0; i < CS; ++i) vb[0][i] = Vw2[J * CCH + i * 32];
            const int wrem_st = wrem_held;
            Chunk cn;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int sl = q * 2 + ks;
#pragma unroll
                    for (int ci = 0; ci < CS; ++ci)
#pragma unroll
                        for (int ki = 0; ki < KS; ++ki) {
                            f32x16& c = acc[(q * CS + ci) * KS + ki];
                            const float av = ks ? ua[q & 1][ki].y : ua[q & 1][ki].x, bv = ks ? vb[q & 1][ci].y : vb[q & 1][ci].x;
                            if (!(DMH_WRW_ABLATE & 8)) c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
                            else c[ks] += av * bv;
                        }
                    if (q + 1 < 4) {                          // operands of the next position, spread over the two k-steps
#pragma unroll
                        for (int r = ks; r < KS + CS; r += 2) {
                            if (r < KS) ua[(q + 1) & 1][r] = Mw2[(q + 1) * (4 * KCH) + J * KCH + r * 32];
                            else vb[(q + 1) & 1][r - KS] = Vw2[(q + 1) * (4 * CCH) + J * CCH + (r - KS) * 32];
                        }
                    }
                    if (!(DMH_WRW_ABLATE & 2)) {            // the transform of (tbuf, tJ): reads, arithmetic, writes
                        if (sl == 0) t_read(tbuf, tJ);
                        if (sl == 2) { t_math_d(); t_math_x(0); }
                        if (sl == 3 && NXI > 1) t_math_x(1);
                        if (sl == 3) t_write_d(tJ, 0);
                        if (sl == 4) t_write_d(tJ, 1);
                        if (sl == 4) t_write_x(tJ, 0, 0);
                        if (sl == 5) t_write_x(tJ, 0, 1);
                        if (sl == 6 && NXI > 1) t_write_x(tJ, 1, 0);
                        if (sl == 7 && NXI > 1) t_write_x(tJ, 1, 1);
                    }
                    if (stage) {
                        if (sl < 4) {                           // raw registers (chunk g + 2) -> raw[sbuf]
                            if (!(DMH_WRW_ABLATE & 4)) {
#pragma unroll
                                for (int k = sl * NST / 4; k < (sl + 1) * NST / 4; ++k) {
                                    if (k < NXL) store_x(sbuf, k);
                                    else store_d(sbuf, k - NXL, wrem_st);
                                }
                            }
                        } else {                                // refill with chunk g + 3
                            if (sl == 4) cn = chunk_at(c_first + g + 3);
#pragma unroll
                            for (int k = (sl - 4) * NST / 4; k < (sl - 3) * NST / 4; ++k) {
                                if (k < NXL) load_x(cn, k);
                                else load_d(cn, k - NXL);
                            }
                            if (sl == 7) wrem_held = cn.wrem;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // raw s_barrier + lgkmcnt only: __syncthreads() would also wait for the loads just issued (vmcnt(0))
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(DMH_WRW_ABLATE & 16)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        // prologue: chunk 0 -> raw[0]; chunk 1 -> raw[1]; chunk 2 in registers; the J = 0 half of chunk 0's images
        load_chunk(c_first);
        store_raw(0);
        load_chunk(c_first + 1);
        __syncthreads();
        t_read(0, 0);
        t_math_d();
        t_write_d(0, 0);
        t_write_d(0, 1);
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            t_math_x(k);
            t_write_x(0, k, 0);
            t_write_x(0, k, 1);
        }
        store_raw(1);
        load_chunk(c_first + 2);
        __syncthreads();
        // two chunks per trip: the raw buffer of every phase is then a constant, and its LDS addresses are immediates instead of
        // 16 vector additions per chunk
        int g = 0;
        for (; g + 1 < n; g += 2) {
            phase(0, 0, 1, false, 0, g);            // A: k-steps 0-1 of chunk g      | tiles J = 1 of chunk g     (raw[cur])
            phase(1, 1, 0, true, 0, g);             // B: k-steps 2-3 of chunk g      | tiles J = 0 of chunk g + 1 (raw[nxt]); staging
            phase(0, 1, 1, false, 0, g + 1);
            phase(1, 0, 0, true, 1, g + 1);
        }
        if (g < n) {
            phase(0, 0, 1, false, 0, g);
            phase(1, 1, 0, true, 0, g);
        }
    }
    if (!IL && n > 0) {
        // prologue: chunk 0 -> raw[0] -> images; chunk 1 -> raw[1]; chunk 2 in registers
        load_chunk(c_first);
        store_raw(0);
        load_chunk(c_first + 1);
        __syncthreads();
        transform(0);
        store_raw(1);
        load_chunk(c_first + 2);
        __syncthreads();
        for (int g = 0; g < n; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            f32x4 ua[2][KS], vb[2][CS];
#pragma unroll
            for (int i = 0; i < KS; ++i) ua[0][i] = Mw[i * 32];
#pragma unroll
            for (int i = 0; i < CS; ++i) vb[0][i] = Vw[i * 32];
            // staging work of this iteration, dealt over the 16 MFMA groups below: groups 0-7 the raw registers (chunk g+2)
            // -> raw[cur] (read by transform(g) one iteration ago), groups 8-15 the refill of those registers with chunk g+3
            constexpr int NST = NXL + NDL;
            const int wrem_st = wrem_held;
            Chunk cn;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                    for (int ci = 0; ci < CS; ++ci)
#pragma unroll
                        for (int ki = 0; ki < KS; ++ki) {
                            f32x16& c = acc[(q * CS + ci) * KS + ki];
                            if (!(DMH_WRW_ABLATE & 8)) c = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[q & 1][ki][ks], vb[q & 1][ci][ks], c, 0, 0, 0);
                            else c[ks] += ua[q & 1][ki][ks] * vb[q & 1][ci][ks];
                        }
                    if (q + 1 < 4) {                          // operands of the next position, spread over the four k-steps
#pragma unroll
                        for (int r = ks; r < KS + CS; r += 4) {
                            if (r < KS) ua[(q + 1) & 1][r] = Mw[(q + 1) * (2 * KCH) + r * 32];
                            else vb[(q + 1) & 1][r - KS] = Vw[(q + 1) * (2 * CCH) + (r - KS) * 32];
                        }
                    }
                    const int sl = q * 4 + ks;
                    if (sl < 8) {
                        if (!(DMH_WRW_ABLATE & 4)) {
#pragma unroll
                            for (int k = sl * NST / 8; k < (sl + 1) * NST / 8; ++k) {
                                if (k < NXL) store_x(cur, k);
                                else store_d(cur, k - NXL, wrem_st);
                            }
                        }
                    } else {
                        if (sl == 8) cn = chunk_at(c_first + g + 3);
#pragma unroll
                        for (int k = (sl - 8) * NST / 8; k < (sl - 7) * NST / 8; ++k) {
                            if (k < NXL) load_x(cn, k);
                            else load_d(cn, k - NXL);
                        }
                        if (sl == 15) wrem_held = cn.wrem;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // raw s_barrier + lgkmcnt only: __syncthreads() would also wait for the loads just issued (vmcnt(0))
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(DMH_WRW_ABLATE & 16)) __builtin_amdgcn_s_barrier();   // every wave has read the images of chunk g
            asm volatile("" ::: "memory");
            if (!(DMH_WRW_ABLATE & 2)) transform(nxt);      // chunk g+1 (written to raw[nxt] one iteration ago)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(DMH_WRW_ABLATE & 16)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    // ---- partial dU of this slice: ws[pair][slice][p][k KCH][c CCH]; D row (k) = 8 (v >> 2) + 4 (lane >> 5) + (v & 3), col (c) = lane & 31
    float* wsb = a.ws + (((size_t)pair * a.S + slice) * 16 + 4 * wv) * (KCH * CCH) + (size_t)(4 * (lane >> 5)) * CCH + (lane & 31);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ci = 0; ci < CS; ++ci)
#pragma unroll
            for (int ki = 0; ki < KS; ++ki) {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    wsb[(size_t)q * (KCH * CCH) + (ki * 32 + (v & 3) + 8 * (v >> 2)) * CCH + ci * 32] = acc[(q * CS + ci) * KS + ki][v];
                __builtin_amdgcn_sched_barrier(0);
            }
}

// dw[k][c][3][3] = G^T (sum over the slices of dU[.][k][c]) G,  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; slices in order.
// A block = 16 consecutive input channels of one output channel x the 16 positions: every thread adds the S slices of its
// (c, p) (64-byte segments per slice), the 4 x 4 sums meet in LDS and 144 threads form one filter tap each.
__global__ __launch_bounds__(NT) void wino_wrw_reduce_kernel(const float* __restrict__ ws, int K, int C, int kch, int cch, int nc,
                                                             int S, float* __restrict__ dw) {
    __shared__ float su[16][17];                        // [position][channel], padded
    const int cl = threadIdx.x & 15, p = threadIdx.x >> 4;
    const int cblocks = C >> 4;
    const int k = (int)blockIdx.x / cblocks, c = ((int)blockIdx.x - k * cblocks) * 16 + cl;
    const int pair = (k / kch) * nc + c / cch;
    const size_t blk = (size_t)kch * cch;
    const float* src = ws + (size_t)pair * S * (16 * blk) + (size_t)p * blk + (size_t)(k % kch) * cch + (c % cch);
    float s = 0.f;
    for (int q = 0; q < S; ++q) s += src[(size_t)q * (16 * blk)];
    su[p][cl] = s;
    __syncthreads();
    if (threadIdx.x < 144) {
        const int cc = threadIdx.x / 9, tap = threadIdx.x - cc * 9, r = tap / 3, col = tap - r * 3;
        // G^T row r and G column `col`: coefficients over the 4 transform rows / columns
        const float gr[3][4] = {{1.f, 0.5f, 0.5f, 0.f}, {0.f, 0.5f, -0.5f, 0.f}, {0.f, 0.5f, 0.5f, 1.f}};
        float o = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float t = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) t += su[a * 4 + b][cc] * gr[col][b];
            o += gr[r][a] * t;
        }
        dw[((size_t)k * C + ((int)blockIdx.x - k * cblocks) * 16 + cc) * 9 + tap] = o;
    }
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
    }
    return n;
}

// block shape for (K, C): 0 = none
int pick_shape(int K, int C, int& kch, int& cch) {
    if (K % 64 == 0 && C % 64 == 0) { kch = 64; cch = 64; return 1; }
    if (K % 32 == 0 && C % 96 == 0) { kch = 32; cch = 96; return 2; }
    if (K % 32 == 0 && C % 64 == 0) { kch = 32; cch = 64; return 3; }
    return 0;
}

void plan(RArgs& a, int kch, int cch) {
    a.Ht = a.Ho / 2;
    a.cpr = (a.Wo / 2 + TPC - 1) / TPC;
    a.nchunks = a.B * a.Ht * a.cpr;
    a.nk = a.K / kch;
    a.nc = a.C / cch;
    const int pairs = a.nk * a.nc;
    int S = num_cus() / pairs;
    if (S < 1) S = 1;
    if (S > a.nchunks) S = a.nchunks;
    a.S = S;
    a.cps = (a.nchunks + S - 1) / S;
}

template <int KS, int CS, bool IL>
int launch_form(const RArgs& a, hipStream_t st) {
    constexpr size_t smem = Cfg<KS, CS>::SMEM;
    static std::atomic<uint64_t> configured{0};     // per device, see configure_dynamic_lds
    if (configure_dynamic_lds(wino_wrw_kernel<KS, CS, IL>, smem, configured) != hipSuccess)
        return fail(DMH_ELAUNCH, "%s: cannot raise the dynamic LDS limit", "dmh_wino_wrw");
    hipLaunchKernelGGL((wino_wrw_kernel<KS, CS, IL>), dim3((unsigned)(a.nk * a.nc * a.S)), dim3(NT), smem, st, a);
    return check_launch("dmh_wino_wrw");
}

template <int KS, int CS>
int launch(const RArgs& a, hipStream_t st) {
    // DMH_WRW_FORM=0: the two-phase loop of rounds 3-5 (A/B switch; bit-identical results)
    static const bool il = !(getenv("DMH_WRW_FORM") && atoi(getenv("DMH_WRW_FORM")) == 0);
    return il ? launch_form<KS, CS, true>(a, st) : launch_form<KS, CS, false>(a, st);
}

}  // namespace

extern "C" {

int64_t dmh_wino_wrw_workspace_size(int B, int C, int K, int H, int W, int pad) {
    int kch, cch;
    if (B <= 0 || C <= 0 || K <= 0 || !pick_shape(K, C, kch, cch) || pad < 0 || pad > 1 || (pad == 1 && W % 16)) return -1;
    RArgs a;
    a.B = B; a.C = C; a.K = K; a.H = H; a.W = W; a.pad = pad; a.Ho = H + 2 * pad - 2; a.Wo = W + 2 * pad - 2;
    if (a.Ho < 2 || a.Wo < 2 || (a.Ho & 1) || (a.Wo & 1)) return -1;
    plan(a, kch, cch);
    return (int64_t)a.nk * a.nc * a.S * 16 * kch * cch;
}

int dmh_wino_wrw(const float* x, const float* dy, int B, int C, int K, int H, int W, int pad, float* workspace, float* dw,
                 void* stream) {
    DMH_REQUIRE(x && dy && workspace && dw, "null pointer");
    int kch = 0, cch = 0;
    const int shape = (B > 0 && C > 0 && K > 0) ? pick_shape(K, C, kch, cch) : 0;
    DMH_REQUIRE(shape != 0, "channel counts: K and C multiples of 64, or K a multiple of 32 with C a multiple of 96 or 64");
    DMH_REQUIRE(pad == 0 || pad == 1, "pad must be 0 or 1");
    DMH_REQUIRE(pad == 0 || W % 16 == 0, "with pad 1 the width must be a multiple of 16 (whole 8-tile chunks: the zero\n"
                "padding column then lies in the last 16-byte word of a chunk row)");
    RArgs a;
    a.x = x; a.dy = dy; a.ws = workspace;
    a.B = B; a.C = C; a.K = K; a.H = H; a.W = W; a.pad = pad; a.Ho = H + 2 * pad - 2; a.Wo = W + 2 * pad - 2;
    DMH_REQUIRE(a.Ho >= 2 && a.Wo >= 2 && (a.Ho & 1) == 0 && (a.Wo & 1) == 0, "output height and width must be even");
    DMH_REQUIRE((int64_t)B * C * H * W * 4 <= (int64_t)0xFFFFFF00 && (int64_t)B * K * a.Ho * a.Wo < ((int64_t)1 << 30),
                "tensor larger than 4 GB (32-bit byte offsets of the buffer loads)");
    plan(a, kch, cch);
    int rc;
    if (shape == 1) rc = launch<2, 2>(a, (hipStream_t)stream);
    else if (shape == 2) rc = launch<1, 3>(a, (hipStream_t)stream);
    else rc = launch<1, 2>(a, (hipStream_t)stream);
    if (rc) return rc;
    hipLaunchKernelGGL(wino_wrw_reduce_kernel, dim3((unsigned)(K * (C / 16))), dim3(NT), 0, (hipStream_t)stream,
                       workspace, K, C, kch, cch, a.nc, a.S, dw);
    return check_launch("dmh_wino_wrw (reduce)");
}

}  // extern "C"
