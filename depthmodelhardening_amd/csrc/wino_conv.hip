// K10 -- 3x3 stride-1 convolution, Winograd F(2x2, 3x3) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Why: after K1-K9 the adversarial-training step was ~75 % MIOpen `miopenSp3AsmConv_v30_3_1_gfx9_fp32_f2x3`
// (564 launches, 133 ms of a 248 ms step): a gfx9-generic Winograd on the VECTOR ALU at 80-105 TFLOP/s
// direct-equivalent.  gfx950 has an exact-fp32 MFMA; the 16 transform-domain GEMMs
//     M_p[k][tile] = sum_c U_p[k][c] * V_p[c][tile]
// map onto it directly.  This is the forward AND the backward-data pass (the same kernel on the flipped/transposed
// filter with pad' = 2 - pad) of the 3x3/1 convolutions of the ResNet encoder and the depth decoder
// (MD2/networks/resnet_encoder.py:85-98 via torchvision BasicBlock, MD2/layers.py:127-141 Conv3x3).  The weight
// gradient stays on MIOpen (train pass only).
//
// Structure (measured background in profiles/README.md):
//   * work item   : 64 output channels x 64 Winograd tiles (2 x 32 or 4 x 16 tiles) x all input channels.  4 waves,
//                   one per SIMD, each 32 channels x 32 tiles x ALL 16 transform positions = 16 accumulators of 32x32
//                   (256 registers per lane), so the output transform A^T M A is per-lane register arithmetic.
//   * persistent  : 256 workgroups (154 KB LDS + 512 registers per lane fill a CU) walk contiguous item ranges; the
//                   staging pipeline runs across item boundaries, only the output transform + store is serial.
//   * per chunk of 8 input channels: filter chunk (pre-transformed U = G w G^T, stored as the LDS image) by LDS-DMA;
//                   raw input region through registers (coalesced rows, zero padding = clamped address + mask) -> LDS;
//                   input transform B^T d B LDS -> LDS; 64 MFMAs per wave with one ds_read_b128 per operand and
//                   position (the channel order inside a chunk is permuted so that a 16-byte read feeds 4 k-steps).
//                   U, V and raw are double-buffered, ONE barrier per chunk, software pipeline 3 chunks deep.
//   * LDS images  : U[p][h][kout][4], V[p][h][tile][4] (h = lane >> 5 half of the MFMA k pair): consecutive lanes read
//                   consecutive 16-byte words = conflict-free ds_read_b128 (MI355X_MICROARCH.md, LDS).
//   * the fp32 MFMA shadows no other instruction (tools/micro/mfma_shadow.hip): the main loop is written in issue
//                   order as 32 fenced slots of 2 MFMAs + a few staging instructions, to keep the non-MFMA
//                   instruction count low and evenly spread.
//   * variants    : FLAT (4 x 16 regions whose tile rows are taken from the whole batch, for 10x32-sized images), EPI
//                   (conv + eval BatchNorm + identity + ReLU in one launch), 2-way channel split for small launches.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

using namespace dmh;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));    // native vector: arrays of it stay in registers

#ifndef DMH_WINO_ABLATE          // tools/wino_ablate.sh: 1 no global loads, 2 no input transform, 4 no LDS staging writes,
#define DMH_WINO_ABLATE 0        // 16 no MFMAs in the steady-state loop, 32 no stores in the epilogue, 64 the epilogue's transform + stores twice,
                                 // 128 no barrier at the end of a chunk, 256 no counted wait before it (timing only)
#endif
constexpr int CK = 8;            // input channels per chunk
constexpr int NT = 256;

struct WArgs {
    const float* x;
    const f32x4* U;              // [C/8][16][2][Kp][4 floats], Kp = K rounded up to 64
    const float* bias;           // [K] or null
    const float* res;            // EPI kernels: tensor added to the output before the activation (or, flag 2, its mask), or null
    int relu;                    // EPI kernels: bit 0 clamp the output at zero; bit 1 `res` is a ReLU mask, not an addend
    float* y;
    int B, C, K, Kp, H, W, Ho, Wo, pad;
    int gx, gy, kg;              // tile-region groups along x / y, output-channel groups
    int csplit;                  // 1, or 2: the input channels of a region are split over two items that add
                                 // their halves into a zeroed y (fills the chip when there are few regions)
    int bitems;                  // B (2x32 regions: per image) or 1 (4x16 regions: tile rows flattened over the batch)
    int nitems;                  // bitems * gy * gx * kg * csplit work items
    float* part;                 // SK: workspace of the partial items (2 * sk_grid slots of SK_SLOT floats)
    // stream-K decomposition (SK instantiations): the (item, channel chunk) space of sk_units = regions * C/8 units is dealt
    // to the sk_grid workgroups in contiguous, equal ranges; a range that starts or ends inside an item leaves a PARTIAL item,
    // whose sums go to slot 2 * wg (the workgroup's first piece) / 2 * wg + 1 (its last) of `part` (SK_SLOT floats each) and are
    // added in chunk order by wino_sk_fixup_kernel
    int sk_units, sk_grid, sk_per, sk_rem;   // sk_per = sk_units / sk_grid, sk_rem = sk_units % sk_grid
};
constexpr int SK_SLOT = 16 * NT * 4;      // floats of one partial item: [output channel v 16][thread 256] float4 (y00, y01, y10, y11)

// pre-transform the filter: U = G g G^T, scattered into the chunked layout the kernel streams.
// mode 0: forward   u[k][c] from w[k][c][ky][kx]
// mode 1: backward  u[c][k] from w[k][c][2-ky][2-kx]   (output channels of the pass = C of the filter)
// scale (optional): per-channel factor of the convolution OUTPUT of the forward pass (an eval-mode BatchNorm folded
// into the filter): multiplies filter row ko in mode 0 and filter column ci (the same channel) in mode 1.
// One thread = one output channel x FOUR consecutive input channels of the pass (the four components s of an image word): the
// 16 transformed values of each go out as 16-byte stores that consecutive threads (ko fastest) write back to back.  (Through
// round 5 a thread owned one (ko, ci) and wrote 16 single floats 16 bytes apart: a quarter of every store's lanes' bytes.)
__device__ __forceinline__ void wino_weight_element(const float* __restrict__ w, int Kw, int Cw, int mode,
                                                    const float* __restrict__ scale, float* __restrict__ U, int Kp, int i) {
    // "out" / "in" are the channel roles of the pass this transform is for
    const int n_out = mode ? Cw : Kw, n_in = mode ? Kw : Cw;
    if (i >= Kp * (n_in >> 2)) return;
    const int ko = i % Kp, cq = i / Kp;          // cq: quad of input channels 4 cq .. 4 cq + 3
    float u[4][16];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int ci = 4 * cq + s;
        float g[3][3];
        if (ko < n_out) {
            const float* src = mode ? w + ((size_t)ci * Cw + ko) * 9 : w + ((size_t)ko * Cw + ci) * 9;
            const float sc = scale ? scale[mode ? ci : ko] : 1.f;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) g[a][b] = sc * (mode ? src[(2 - a) * 3 + (2 - b)] : src[a * 3 + b]);
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) g[a][b] = 0.f;
        }
        float t[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t[0][b] = g[0][b];
            t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            t[3][b] = g[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            u[s][a * 4 + 0] = t[a][0];
            u[s][a * 4 + 1] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
            u[s][a * 4 + 2] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
            u[s][a * 4 + 3] = t[a][2];
        }
    }
    const int cc = cq >> 1, h = cq & 1;          // chunk of 8 channels, half of the MFMA k pair
    f32x4* Uw = reinterpret_cast<f32x4*>(U);
#pragma unroll
    for (int p = 0; p < 16; ++p) Uw[(((size_t)cc * 16 + p) * 2 + h) * Kp + ko] = f32x4{u[0][p], u[1][p], u[2][p], u[3][p]};
}

__global__ __launch_bounds__(NT) void wino_weight_kernel(const float* __restrict__ w, int Kw, int Cw, int mode,
                                                         const float* __restrict__ scale, float* __restrict__ U, int Kp) {
    wino_weight_element(w, Kw, Cw, mode, scale, U, Kp, (int)blockIdx.x * NT + (int)threadIdx.x);    // over Kp * n_in / 4
}

// Many filters in ONE launch (dmh_wino_weight_transform_batch): the jobs travel by value in the kernel arguments, a block finds
// its job by a scan of the jobs' first blocks (uniform: scalar code).  A step transforms ~76 filters of 37 K ... 2.4 M elements
// each -- as 76 launches of 5-20 us they cost 0.67 ms of an 85 ms step, most of it launch ramp.
constexpr int WT_MAX_JOBS = 32;
struct WtBatch {
    const float* w[WT_MAX_JOBS];
    const float* scale[WT_MAX_JOBS];
    float* U[WT_MAX_JOBS];
    int K[WT_MAX_JOBS], C[WT_MAX_JOBS], mode[WT_MAX_JOBS], Kp[WT_MAX_JOBS];
    int first[WT_MAX_JOBS + 1];     // first block of job j; first[n] = grid size
    int n;
};
__global__ __launch_bounds__(NT) void wino_weight_batch_kernel(const WtBatch b) {
    int j = 0;
    while (j + 1 < b.n && (int)blockIdx.x >= b.first[j + 1]) ++j;
    wino_weight_element(b.w[j], b.K[j], b.C[j], b.mode[j], b.scale[j], b.U[j], b.Kp[j],
                        ((int)blockIdx.x - b.first[j]) * NT + (int)threadIdx.x);
}

// DMH_WINO_PK (round 6): the input transform B^T d B on PACKED fp32 adds.  A thread transforms the patches of TWO channels of
// one tile; the raw region keeps the rows of such a channel pair INTERLEAVED ([pair][row][channel of the pair][word]), so that the
// same patch element of both channels is one ds_read2_b32 (offsets j, j + row pitch) into a register pair, both passes of the
// transform are v_pk_add_f32 on such pairs (32 instructions instead of 64 -- the fp32 MFMA shadows no vector instruction, so
// they come straight off the chunk time), and the results are the (channel, channel + 1) pairs the 8-byte image writes want.
// Same additions in the same order: bit-identical results.
#ifndef DMH_WINO_PK
#define DMH_WINO_PK 1
#endif
// 16-byte words of one channel's raw region (rows x words per row), and the channel pitch in words
template <int TRW, bool FLAT>
struct RawGeo {
    static constexpr int TRH = 64 / TRW;
    static constexpr int NWR = (2 * TRW + 2 + 6) / 4;
    static constexpr int RH = FLAT ? 4 * TRH : 2 * TRH + 2;
    static constexpr int CHW = RH * NWR;
    static constexpr int CHPW = CHW;
    static constexpr int RAWN = CK * CHPW;
    static constexpr size_t SMEM = (size_t)4 * 16 * 2 * 64 * 16 + (size_t)(FLAT ? 1 : 2) * RAWN * 16;
    static_assert(SMEM <= 160 * 1024, "LDS");
};

// Decoded work item: 64 output channels x one TRH x TRW tile region of one image.  Items are numbered with the
// output-channel group fastest and every workgroup takes a CONTIGUOUS range, so the channel groups of one region run
// back to back on the same CU and re-read its input from L1/L2.
struct Item {
    int b, ty0, tx0, k0, c0;     // c0: first channel chunk of the item (channel split)
    int si;                      // which part of the channel split
};
template <int TRW>
__device__ __forceinline__ Item decode_item(const WArgs& a, int item) {
    constexpr int TRH = 64 / TRW;
    Item it;
    it.si = item % a.csplit;
    it.c0 = it.si * (a.C / CK / a.csplit);  item /= a.csplit;
    it.k0 = (item % a.kg) * 64;  item /= a.kg;
    it.tx0 = (item % a.gx) * TRW;  item /= a.gx;
    it.ty0 = (item % a.gy) * TRH;
    it.b = item / a.gy;
    return it;
}
// the item after `it` (items are numbered with the channel split fastest, then the channel group, the region column, its row,
// the image): a workgroup walks a contiguous range, so only its first item is decoded by division (round 6: two decodes per
// item were ~300 scalar instructions of an epilogue that has no MFMA to hide them behind)
template <int TRW>
__device__ __forceinline__ Item next_item(const WArgs& a, Item it) {
    constexpr int TRH = 64 / TRW;
    if (++it.si < a.csplit) {
        it.c0 += a.C / CK / a.csplit;
        return it;
    }
    it.si = 0; it.c0 = 0;
    if ((it.k0 += 64) < a.kg * 64) return it;
    it.k0 = 0;
    if ((it.tx0 += TRW) < a.gx * TRW) return it;
    it.tx0 = 0;
    if ((it.ty0 += TRH) < a.gy * TRH) return it;
    it.ty0 = 0;
    ++it.b;
    return it;
}

template <int TRW, bool FLAT, bool EPI, bool SK = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_conv_kernel(WArgs a) {
    constexpr int TRH = 64 / TRW;
    // Tile regions normally lie inside one image and their tile rows share input rows.  FLAT (4 x 16 regions on images
    // with few rows of tiles): the region's 4 tile rows come from the rows of tiles of the WHOLE batch (row R = b * Ht +
    // ty), so a 5-row image wastes nothing; consecutive tile rows may then belong to different images, so each stages
    // its own 4 input rows, and to stay inside 160 KB of LDS the raw region is single-buffered with a second barrier
    // per chunk.  Measured: 10x32 images 178 -> 127 us, but 10-30 % slower where the plain regions waste < 15 %.
    // The raw region is staged in aligned 16-BYTE WORDS (round 3): a vector-memory instruction costs the wave ~64 cycles
    // among the MFMAs whatever its width (tools/wrw_ablate.py), so a row of the region -- 2 TRW + 2 columns from x0 = 2 tx0 -
    // pad -- is fetched as the NWR aligned words that cover it (x0 rounded down to a multiple of 4: `coff` = 0 / 3 / 2
    // columns in front for pad 0 / 1 / 2), lanes running along the rows: 4-5 loads per thread and chunk instead of 11-13.
    // A word is inside or outside the image as a whole when W is a multiple of 4 (every layer of the network); otherwise
    // the one word that straddles the right edge is masked when it is written to LDS (`partial`).
    constexpr int RW = 2 * TRW + 2;
    constexpr int NWR = (RW + 6) / 4;                       // 16-byte words per staged row (18 / 10)
    constexpr int RWA = 4 * NWR;                            // LDS row pitch in floats
    constexpr int RH = FLAT ? 4 * TRH : 2 * TRH + 2;        // raw input rows of a work item (per channel)
    constexpr int TRS = (FLAT ? 4 : 2) * RWA;               // raw floats from one tile row to the next
    constexpr int CHPW = RawGeo<TRW, FLAT>::CHPW;           // words of one channel's raw region
    constexpr int CHP = 4 * CHPW;                           // ... floats
    constexpr int PKI = DMH_WINO_PK ? 2 : 1;                // DMH_WINO_PK: the rows of a channel pair are interleaved
    constexpr int RAW_N = CK * CHPW;                        // words of one raw buffer
    static_assert(RH * NWR == RawGeo<TRW, FLAT>::CHW, "raw geometry");
    constexpr int RAW_PER_T = (RAW_N + NT - 1) / NT;
    constexpr int RAW_BUF = FLAT ? 0 : RAW_N;               // words from raw buffer 0 to buffer 1 (FLAT: one buffer)
    constexpr int BUF = 16 * 2 * 64;                        // f32x4 words of one U or V image (32 KB)
    extern __shared__ f32x4 smem[];
    f32x4* U_lds = smem;                                    // [2 buffers][16][2][64]
    f32x4* V_lds = smem + 2 * BUF;                          // [2 buffers][16][2][64]
    f32x4* raw4 = smem + 4 * BUF;                           // [1 or 2 buffers][CK][RH][NWR] words
    float* raw = reinterpret_cast<float*>(raw4);
    const int Ht = a.Ho >> 1, NR = a.B * Ht;                // FLAT: rows of tiles per image / in the batch

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wv_s = __builtin_amdgcn_readfirstlane(wv);    // wave index as a scalar
    const size_t HW = (size_t)a.H * a.W;
    const int nch = a.C / CK / a.csplit;                    // channel chunks per item

    // ---- this workgroup's contiguous item range, flattened with the channel chunks into one iteration space
    int item0, nmine, cb0 = 0, ce_last = nch;   // SK: the first piece starts at chunk cb0 of item0, the last ends before ce_last
    if (SK) {       // equal ranges of (item, chunk) units: a range may begin and end inside an item (see WArgs)
        const int u0 = sk_boundary(a.sk_per, a.sk_rem, nch, (int)blockIdx.x);
        const int u1 = sk_boundary(a.sk_per, a.sk_rem, nch, (int)blockIdx.x + 1);
        if (u0 >= u1) return;
        item0 = u0 / nch;
        const int il = (u1 - 1) / nch;
        nmine = il - item0 + 1;
        cb0 = u0 - item0 * nch;
        ce_last = u1 - il * nch;
    } else {
        const int q = a.nitems / (int)gridDim.x, r = a.nitems % (int)gridDim.x;
        item0 = (int)blockIdx.x * q + min((int)blockIdx.x, r);
        nmine = q + ((int)blockIdx.x < r ? 1 : 0);
    }
    const int item_last = item0 + nmine - 1;
    int pn = nch;           // channel chunks of the current piece (SK: a partial item has fewer)

    // transform role: wave wv owns chunk channels {2wv, 2wv+1} = (h = wv>>1, s = 2(wv&1) + {0,1}); lane = tile
    const int tly = lane / TRW, tlx = lane - tly * TRW;
    const int coff = (4 - (a.pad & 3)) & 3;                 // columns of the first word in front of the region
    const bool partial = a.pad > 0 && (a.W & 3) != 0;       // a word can straddle the right image edge
    const float* rsrc = raw + (2 * wv) * CHP + tly * (PKI * TRS) + 2 * tlx + coff;
    float* vdst = reinterpret_cast<float*>(V_lds + (wv >> 1) * 64 + lane) + 2 * (wv & 1);
    const int kb = wv & 1, tb = wv >> 1;
    const int aidx = (lane >> 5) * 64 + kb * 32 + (lane & 31);
    const int bidx = (lane >> 5) * 64 + tb * 32 + (lane & 31);

    // ---- raw-load stage constants of an item: offsets inside one channel chunk + validity bits (zero padding =
    //      clamped address + masked value).  The load stages run ahead of the MFMAs across the item boundary, so the
    //      constants of the NEXT item are kept beside the current ones and selected per iteration (no branch).
    // Input and filter are read through buffer resources: a 32-bit per-thread byte offset + a wave-uniform SGPR offset
    // (chunk / filter row) per load, no 64-bit address arithmetic in the loop; an offset beyond the resource reads 0, which
    // IS the zero padding -- padded elements get the offset 0xFFFFFFFF and need no mask (round 2 clamped the address and
    // masked the value: ~55 vector instructions per chunk that the fp32 MFMA, sharing the vector pipe, could not shadow).
    const rsrc_t xrs = make_rsrc(a.x, (unsigned)((size_t)a.B * a.C * HW * 4));
    const rsrc_t urs = make_rsrc(a.U, (unsigned)((size_t)(a.C / CK) * 32 * a.Kp * 16));
    const rsrc_t prs = make_rsrc(SK ? (const void*)a.part : (const void*)a.U, SK ? (unsigned)(2 * a.sk_grid) * (unsigned)(SK_SLOT * 4) : 16u);
    unsigned roff[RAW_PER_T], roff_n[RAW_PER_T];
    unsigned uoff = 0, uoff_n = 0;           // byte offset of the item's first filter chunk (uniform)
    int ixa = 0, ixa_n = 0;                  // first staged column of the item (uniform; `partial` only)
#define DMH_WINO_ITEM_CONSTS(ITEM, IT, ROFF, UOFF, IXA)                                           \
    {                                                                                             \
        Item it = (IT);                                                                           \
        if (SK) it.c0 = ((ITEM) == item0) ? cb0 : 0;      /* a workgroup's first piece may start inside its item */ \
        const int ix0 = 2 * it.tx0 - a.pad - coff;        /* multiple of 4: tx0 is a multiple of 16 */ \
        IXA = ix0;                                                                                \
        /* thread index rebuilt from v_mbcnt + the scalar wave index: a copy of `tid` kept from kernel entry is  */ \
        /* spilled, and its reload's vmcnt(0) waits for the previous item's stores (also keeps the slot          */ \
        /* decomposition below from being hoisted and held in registers)                                         */ \
        int tid_o;      /* volatile asm: the builtin form is loop-invariant, gets hoisted and spilled all the same   */ \
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tid_o));           \
        tid_o += wv_s * 64;                                                                       \
        UOFF = (unsigned)(((size_t)it.k0 + (size_t)it.c0 * 32 * a.Kp) * 16);                      \
        const int cbase = (it.b * a.C + it.c0 * CK) * (int)HW;    /* first element of the item's first chunk */ \
        /* FLAT: image and tile row of the region's 4 rows of tiles (uniform: scalar divisions) */ \
        int fb[4], fy[4];                                                                         \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                           \
            const int Rg = min(it.ty0 + t, NR - 1);                                               \
            fb[t] = FLAT ? Rg / Ht : 0;                                                           \
            fy[t] = FLAT ? Rg - fb[t] * Ht : 0;                                                   \
        }                                                                                         \
        _Pragma("unroll") for (int k = 0; k < RAW_PER_T; ++k) {                                   \
            const int e = tid_o + NT * k;                                                         \
            /* LDS order: [channel][row][word]; DMH_WINO_PK: [channel pair][row][channel of the pair][word] */ \
            const int cq = e / (PKI * CHPW), r1 = e - cq * (PKI * CHPW), rr = r1 / (PKI * NWR), r2 = r1 - rr * (PKI * NWR); \
            const int c = PKI * cq + r2 / NWR, xx = 4 * (r2 % NWR);                               \
            int iy, bofs = cbase;                                                                 \
            bool okr = e < RAW_N;                                                                 \
            if (FLAT) {                                                                           \
                const int t = rr >> 2;                                                            \
                const int bb = t == 0 ? fb[0] : t == 1 ? fb[1] : t == 2 ? fb[2] : fb[3];          \
                const int ty = t == 0 ? fy[0] : t == 1 ? fy[1] : t == 2 ? fy[2] : fy[3];          \
                iy = 2 * ty - a.pad + (rr & 3);                                                   \
                bofs = (bb * a.C + it.c0 * CK) * (int)HW;                                         \
                okr = okr && it.ty0 + t < NR;                                                     \
            } else {                                                                              \
                iy = 2 * it.ty0 - a.pad + rr;                                                     \
            }                                                                                     \
            const int ix = ix0 + xx;                       /* first column of the word */         \
            const bool ok = okr && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;                    \
            ROFF[k] = ok ? (unsigned)(bofs + c * (int)HW + iy * a.W + ix) * 4u : 0xFFFFFFFFu;     \
        }                                                                                         \
    }

    f32x4 rreg[RAW_PER_T];
    // CHB: wave-uniform byte offset of the chunk inside the item (chunk index * 8 channels)
#define DMH_WINO_LOAD1(K, CHB, OFF) rreg[K] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (OFF), (CHB), 0));
#define DMH_WINO_LOAD_RAW(CHB, ROFF)                                                              \
    _Pragma("unroll") for (int k = 0; k < RAW_PER_T; ++k) DMH_WINO_LOAD1(k, CHB, ROFF[k])
    // registers -> LDS (word e of the buffer: the region is enumerated in LDS order); IXW: first staged column of the item
    // the data belongs to -- with `partial` the columns at and beyond W are zeroed (they hold the next row's first pixels)
#define DMH_WINO_WRITE1(K, BUFI, IXW)                                                             \
    {                                                                                             \
        const int e_ = tid + NT * (K);                                                            \
        f32x4 v_ = rreg[K];                                                                       \
        if (partial) {                                                                            \
            const int n_ = a.W - ((IXW) + 4 * (e_ % NWR));                                        \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) v_[j] = (j < n_) ? v_[j] : 0.f;         \
        }                                                                                         \
        if (RAW_PER_T * NT == RAW_N || e_ < RAW_N) raw4[(BUFI) * RAW_BUF + e_] = v_;              \
    }
    // filter chunk: 32 rows (p, h) of 64 x 16 B, lane-linear both in global memory and in the LDS image -> LDS-DMA
    // (buffer_load_dwordx4 ... lds: no registers, no ds_write; the row's byte offset is an SGPR, the per-lane part
    // lane * 16 a loop-invariant register); wave wv moves rows wv, wv+4, ...; K = row slot 0..7
    // Written as asm: through the builtin hipcc treats every later LDS read as a possible alias of the DMA and drains
    // vmcnt(0) -- the register-staged raw loads included -- a few slots later.  The asm DMA is invisible to hipcc's
    // s_waitcnt bookkeeping, so its completion is counted by hand (the vmcnt before each barrier below); hipcc's own
    // counted waits for the raw loads only become stricter (the DMAs are younger than the loads they wait for).
    const unsigned lane16 = (unsigned)lane * 16u;
#define DMH_WINO_GLDS_U_ROW(UCB, BUFI, K)                                                         \
    {                                                                                             \
        const unsigned srow = __builtin_amdgcn_readfirstlane((UCB) + (unsigned)((wv_s + 4 * (K)) * a.Kp) * 16u);   \
        const unsigned ldst = __builtin_amdgcn_readfirstlane(                                     \
            (unsigned)(uintptr_t)(U_lds + (BUFI) * BUF) + (unsigned)((wv_s + 4 * (K)) * 64 * 16)); \
        unsigned keep;                                                                            \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep) : "v"(lane16), "s"(urs), "s"(ldst), "s"(srow) : "memory");      \
    }
#define DMH_WINO_GLDS_U(UCB, BUFI)                                                                \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) DMH_WINO_GLDS_U_ROW(UCB, BUFI, k)
#define DMH_WINO_WRITE_RAW(BUFI, IXW)                                                             \
    _Pragma("unroll") for (int k = 0; k < RAW_PER_T; ++k) DMH_WINO_WRITE1(k, BUFI, IXW)

    f32x16 acc[16];         // never cleared: the first chunk of an item multiplies onto a zero C operand (an inline constant)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // Software pipeline over the flattened (item, chunk) sequence, ONE barrier and ONE fenced block per chunk.
    // Iteration g = item * nch + ch runs
    //   M(g)    64 MFMAs on U[g&1], V[g&1]
    //   T(g+1)  input transform raw[(g+1)&1] -> V[(g+1)&1]
    //   D       LDS-DMA (global_load_lds_dwordx4) of filter chunk g+1 -> U[(g+1)&1], issued first
    //   W       prefetch registers (raw chunk g+2) -> raw[g&1]
    //   L       global loads of raw chunk g+3 into the registers W just freed: in flight for a whole iteration
    // The stages run across item boundaries (nch >= 3), so the next item's pipeline fill overlaps this item's last
    // chunks; only the output transform + store of an item is serial.  Past the last item the stages re-stage its
    // first chunks into buffers nobody reads any more.
    const unsigned chunk_bytes = (unsigned)(CK * HW * 4);          // one chunk of 8 input channels
    const unsigned uchunk_bytes = (unsigned)(32 * a.Kp * 16);      // one filter chunk
    Item it_cur = decode_item<TRW>(a, item0);
    DMH_WINO_ITEM_CONSTS(item0, it_cur, roff, uoff, ixa)
    DMH_WINO_LOAD_RAW(0u, roff)
    DMH_WINO_GLDS_U(uoff, 0)
    DMH_WINO_WRITE_RAW(0, ixa)
    DMH_WINO_LOAD_RAW(chunk_bytes, roff)
    __syncthreads();
    {   // T(0)
        float t[2][4][4];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            float d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {       // 4-byte aligned only (coff is odd for pad 1): dword reads, paired by the compiler
                const float* row = DMH_WINO_PK ? rsrc + (2 * i + ch) * RWA : rsrc + ch * CHP + i * RWA;
                d[i][0] = row[0]; d[i][1] = row[1]; d[i][2] = row[2]; d[i][3] = row[3];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[ch][0][j] = d[0][j] - d[2][j];
                t[ch][1][j] = d[1][j] + d[2][j];
                t[ch][2][j] = d[2][j] - d[1][j];
                t[ch][3][j] = d[1][j] - d[3][j];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float2*>(vdst + (i * 4 + 0) * 512) = make_float2(t[0][i][0] - t[0][i][2], t[1][i][0] - t[1][i][2]);
            *reinterpret_cast<float2*>(vdst + (i * 4 + 1) * 512) = make_float2(t[0][i][1] + t[0][i][2], t[1][i][1] + t[1][i][2]);
            *reinterpret_cast<float2*>(vdst + (i * 4 + 2) * 512) = make_float2(t[0][i][2] - t[0][i][1], t[1][i][2] - t[1][i][1]);
            *reinterpret_cast<float2*>(vdst + (i * 4 + 3) * 512) = make_float2(t[0][i][1] - t[0][i][3], t[1][i][1] - t[1][i][3]);
        }
    }
    if (FLAT) __syncthreads();          // single raw buffer: every wave has transformed chunk 0 out of it
    DMH_WINO_WRITE_RAW(1, ixa)
    DMH_WINO_LOAD_RAW(2u * chunk_bytes, roff)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RAW_PER_T) : "memory");   // the LDS-DMA of U[0] has landed
    __syncthreads();

    int g = 0;
    for (int mi = 0; mi < nmine; ++mi) {
        const int item = item0 + mi;
        const Item it_nxt = item < item_last ? next_item<TRW>(a, it_cur) : it_cur;
        if (item < item_last) {
            DMH_WINO_ITEM_CONSTS(item + 1, it_nxt, roff_n, uoff_n, ixa_n)
        } else {    // the last item (the only one of most attack launches): the stages past its end re-stage its own first chunks
#pragma unroll
            for (int k = 0; k < RAW_PER_T; ++k) roff_n[k] = roff[k];
            uoff_n = uoff;
            ixa_n = ixa;
        }
        // one chunk; FIRST: the item's first chunk, whose first MFMA per position starts the accumulation from zero
        auto chunk = [&](const int ch, auto first_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const int cur = g & 1, nxt = cur ^ 1;
            const f32x4* Uc = U_lds + cur * BUF + aidx;
            const f32x4* Vc = V_lds + cur * BUF + bidx;
            const float* rs = rsrc + nxt * RAW_BUF * 4;
            const int ixw = (ch + 2 >= pn) ? ixa_n : ixa;   // the registers written to LDS below hold chunk ch+2
            float* vd = vdst + nxt * BUF * 4;
            // load-stage operands of this iteration: raw chunk ch+3 and filter chunk ch+1, possibly of the next item;
            // the raw registers written to LDS in this iteration hold chunk ch+2
            const bool r_next = ch + 3 >= pn, u_next = ch + 1 >= pn;
            unsigned xcb = (unsigned)__builtin_amdgcn_readfirstlane(   // raw chunk ch+3: byte offset in its item
                (int)((unsigned)(r_next ? ch + 3 - pn : ch + 3) * chunk_bytes));
            // pinned to an SGPR: a scalar offset the compiler parks in a VGPR turns every load that uses it into a waterfall
            // loop (readfirstlane + compare + branch) -- the FLAT instantiations did, 17 loops per chunk
            asm volatile("" : "+s"(xcb));
            const unsigned ucb = u_next ? uoff_n : uoff + (unsigned)(ch + 1) * uchunk_bytes;
            // The block is written in issue order and fenced (sched_barrier) per SLOT = 2 MFMAs on two alternating
            // accumulators (128 cycles of matrix pipe) + one operand read for the next position pair + a few
            // instructions of staging work, so that no gap between MFMAs carries more than the pipe can shadow.
            f32x4 ua[16], vb[16];
            ua[0] = Uc[0]; vb[0] = Vc[0];
            ua[1] = Uc[128]; vb[1] = Vc[128];
#if DMH_WINO_PK
            f32x2 D[4][4], T[4][4];         // (channel, channel + 1) pairs: patch elements, then B^T d
#else
            float d[4][4], t[2][4][4];
#endif
            __builtin_amdgcn_sched_barrier(0);
#if DMH_WINO_ABLATE & 16
#define DMH_MFMA(A, B, C) (C)
#else
#define DMH_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, C, 0, 0, 0)
#endif
#pragma unroll
            for (int sl = 0; sl < 32; ++sl) {
                const int p0 = 2 * (sl >> 2), p1 = p0 + 1, ks = sl & 3;
                if (FIRST && ks == 0) {
                    acc[p0] = DMH_MFMA(ua[p0][ks], vb[p0][ks], zero16);
                    acc[p1] = DMH_MFMA(ua[p1][ks], vb[p1][ks], zero16);
                } else {
                    acc[p0] = DMH_MFMA(ua[p0][ks], vb[p0][ks], acc[p0]);
                    acc[p1] = DMH_MFMA(ua[p1][ks], vb[p1][ks], acc[p1]);
                }
                if (p0 + 2 < 16) {                          // operands of the next position pair, one read per slot
                    if (ks == 0) ua[p0 + 2] = Uc[(p0 + 2) * 128];
                    if (ks == 1) vb[p0 + 2] = Vc[(p0 + 2) * 128];
                    if (ks == 2) ua[p1 + 2] = Uc[(p1 + 2) * 128];
                    if (ks == 3) vb[p1 + 2] = Vc[(p1 + 2) * 128];
                }
                if (sl < 8) {                               // filter chunk g+1 -> U[nxt] by LDS-DMA, one row per slot
                    if (!(DMH_WINO_ABLATE & 1)) DMH_WINO_GLDS_U_ROW(ucb, nxt, sl)
                } else if (DMH_WINO_ABLATE & 2) {
#if DMH_WINO_PK
                } else if (sl < 22) {
                    if (sl < 12) {              // column sl - 8 of BOTH channels' patches: one read per row = (channel, channel + 1)
                        const int j = sl - 8;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float* e = rs + 2 * i * RWA + j;
                            D[i][j] = f32x2{e[0], e[RWA]};
                        }
                    }
                    if (sl >= 10 && sl < 14) {  // B^T d of the column read two slots earlier: four packed adds
                        const int j = sl - 10;
                        T[0][j] = pk_sub(D[0][j], D[2][j]);
                        T[1][j] = pk_add(D[1][j], D[2][j]);
                        T[2][j] = pk_sub(D[2][j], D[1][j]);
                        T[3][j] = pk_sub(D[1][j], D[3][j]);
                    }
                    if (sl >= 14) {             // (B^T d) B: half an output row (2 positions) per slot -> V
                        const int rr = (sl - 14) >> 1;
                        if ((sl & 1) == 0) {
                            *reinterpret_cast<f32x2*>(vd + (rr * 4 + 0) * 512) = pk_sub(T[rr][0], T[rr][2]);
                            *reinterpret_cast<f32x2*>(vd + (rr * 4 + 1) * 512) = pk_add(T[rr][1], T[rr][2]);
                        } else {
                            *reinterpret_cast<f32x2*>(vd + (rr * 4 + 2) * 512) = pk_sub(T[rr][2], T[rr][1]);
                            *reinterpret_cast<f32x2*>(vd + (rr * 4 + 3) * 512) = pk_sub(T[rr][1], T[rr][3]);
                        }
                    }
#else
                } else if ((sl >= 8 && sl < 12) || (sl >= 14 && sl < 18)) {   // raw patch row of channel 0 / 1
                    const int c2 = sl >= 14, i = c2 ? sl - 14 : sl - 8;
                    const float* row = rs + c2 * CHP + i * RWA;
                    d[i][0] = row[0]; d[i][1] = row[1]; d[i][2] = row[2]; d[i][3] = row[3];
                } else if (sl == 12 || sl == 13 || sl == 18 || sl == 19) {    // rows of B^T d, two columns per slot
                    const int c2 = sl >= 18, j0 = 2 * (sl & 1);
#pragma unroll
                    for (int j = j0; j < j0 + 2; ++j) {
                        t[c2][0][j] = d[0][j] - d[2][j];
                        t[c2][1][j] = d[1][j] + d[2][j];
                        t[c2][2][j] = d[2][j] - d[1][j];
                        t[c2][3][j] = d[1][j] - d[3][j];
                    }
                } else if (sl >= 20 && sl < 28) {           // columns: half an output row (2 positions) per slot -> V
                    const int rr = (sl - 20) >> 1;
                    if ((sl & 1) == 0) {
                        *reinterpret_cast<float2*>(vd + (rr * 4 + 0) * 512) = make_float2(t[0][rr][0] - t[0][rr][2], t[1][rr][0] - t[1][rr][2]);
                        *reinterpret_cast<float2*>(vd + (rr * 4 + 1) * 512) = make_float2(t[0][rr][1] + t[0][rr][2], t[1][rr][1] + t[1][rr][2]);
                    } else {
                        *reinterpret_cast<float2*>(vd + (rr * 4 + 2) * 512) = make_float2(t[0][rr][2] - t[0][rr][1], t[1][rr][2] - t[1][rr][1]);
                        *reinterpret_cast<float2*>(vd + (rr * 4 + 3) * 512) = make_float2(t[0][rr][1] - t[0][rr][3], t[1][rr][1] - t[1][rr][3]);
                    }
#endif
                }
                if (FLAT && sl == 28) {     // single raw buffer: every wave has read this chunk's patches (slots 8-17)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                if (sl >= 28) {                             // raw registers -> LDS, then refill them (4 slots)
#pragma unroll
                    for (int k = (sl - 28) * ((RAW_PER_T + 3) / 4); k < min((sl - 27) * ((RAW_PER_T + 3) / 4), RAW_PER_T); ++k) {
                        if (!(DMH_WINO_ABLATE & 4)) DMH_WINO_WRITE1(k, cur, ixw)
                        if (!(DMH_WINO_ABLATE & 1)) DMH_WINO_LOAD1(k, xcb, r_next ? roff_n[k] : roff[k])
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // all older vector-memory operations (the LDS-DMA of U[nxt] and the loads consumed above) have landed when
            // only this iteration's RAW_PER_T raw loads are still outstanding; a raw s_barrier keeps those in flight
            // (__syncthreads() would drain them: its fence waits vmcnt(0) once an LDS-DMA has been issued)
            if (!(DMH_WINO_ABLATE & 256)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(RAW_PER_T) : "memory");
            if (!(DMH_WINO_ABLATE & 128)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ++g;
        };
        if (SK) pn = (item == item_last ? ce_last : nch) - (item == item0 ? cb0 : 0);
        chunk(0, std::true_type());
        for (int ch = 1; ch < pn; ++ch) chunk(ch, std::false_type());
        // ---- item done: output transform  Y = A^T M A, store; lane -> tile,
        //      register -> output channel
        {
            const Item it = it_cur;
            // lane index from v_mbcnt (not a register kept since kernel entry: see DMH_WINO_ITEM_CONSTS); opaque, so the
            // per-lane store addresses are not hoisted out of the item loop
            int lane_o;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_o));
            const int tl = tb * 32 + (lane_o & 31);
            const int Rt = it.ty0 + tl / TRW;                       // row of tiles: in the image, or (FLAT) in the batch
            const int ob = FLAT ? min(Rt, NR - 1) / Ht : it.b;
            const int oy = 2 * (FLAT ? Rt - ob * Ht : Rt), ox = 2 * (it.tx0 + tl % TRW);
            const bool inside = (FLAT ? Rt < NR : oy < a.Ho) && ox < a.Wo;
            float* yb = a.y + (size_t)ob * a.K * a.Ho * a.Wo + (size_t)oy * a.Wo + ox;
            const int kbase = it.k0 + kb * 32 + 4 * (lane_o >> 5);
            // EPI: the residual / mask values of channel v + RPF are requested while channel v is transformed (a load
            // issued and consumed inside one fenced iteration costs its full latency 16 times per item)
            constexpr int RPF = 4;
            float2 rq0[RPF], rq1[RPF];
            const bool use_res = EPI && a.res != nullptr;
            // The accumulators are read below by v_accvgpr_read_b32 written as asm, which the compiler's hazard recognizer does not
            // see into: a 16-pass MFMA's result may be read 19 wait states after its issue at the earliest.  The item's last MFMA
            // lies a barrier and the item decode behind us; these 20 wait states make that independent of how the code is laid out.
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
            auto res_fetch = [&](const int v, float2& q0, float2& q1) __attribute__((always_inline)) {
                const int ko = kbase + (v & 3) + 8 * (v >> 2);
                q0 = q1 = make_float2(0.f, 0.f);
                if (use_res && inside && ko < a.K) {
                    const float* rp = a.res + ((yb - a.y) + (size_t)ko * a.Ho * a.Wo);
                    q0 = *reinterpret_cast<const float2*>(rp);
                    q1 = *reinterpret_cast<const float2*>(rp + a.Wo);
                }
            };
            if (EPI) {
#pragma unroll
                for (int v = 0; v < RPF; ++v) res_fetch(v, rq0[v], rq1[v]);
            }
#pragma unroll 1
            for (int rep = 0; rep < ((DMH_WINO_ABLATE & 64) ? 2 : 1); ++rep)      // timing experiment: the transform + stores twice
#pragma unroll
            for (int vp = 0; vp < 8; ++vp) {
                // Round 6: the output transform on packed adds over the channel PAIR (v, v + 1) = two adjacent registers of every
                // accumulator (channels ko, ko + 1); one v_pk_mov_b32 per store regroups (y00 of both channels, y01 of both) into a
                // channel's pixel pair.  The same additions in the same order as the scalar form: bit-identical.
                const int v0 = 2 * vp, ko0 = kbase + (v0 & 3) + 8 * (v0 >> 2);
                f32x2 S0[4], S1[4];
                // every accumulator value is read ONCE, by a volatile v_accvgpr_read_b32 (left to itself hipcc re-reads the
                // accumulator file for every use -- 610 reads per item for 256 values -- and shuffles register pairs with v_mov_b64)
                f32x2 A2[16];
#pragma unroll
                for (int pq = 0; pq < 16; ++pq) {
                    float lo_, hi_;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(lo_) : "a"(acc[pq][v0]));
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(hi_) : "a"(acc[pq][v0 + 1]));
                    A2[pq] = f32x2{lo_, hi_};
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    S0[j] = pk_add(pk_add(A2[j], A2[4 + j]), A2[8 + j]);
                    S1[j] = pk_sub(pk_sub(A2[4 + j], A2[8 + j]), A2[12 + j]);
                }
                f32x2 Y00 = pk_add(pk_add(S0[0], S0[1]), S0[2]), Y01 = pk_sub(pk_sub(S0[1], S0[2]), S0[3]);
                f32x2 Y10 = pk_add(pk_add(S1[0], S1[1]), S1[2]), Y11 = pk_sub(pk_sub(S1[1], S1[2]), S1[3]);
                const bool whole = !(SK && pn < nch);
                if (whole) {
                    f32x2 BS = {0.f, 0.f};
                    if (a.bias && it.c0 == 0) {
                        BS.x = ko0 < a.K ? a.bias[ko0] : 0.f;
                        BS.y = ko0 + 1 < a.K ? a.bias[ko0 + 1] : 0.f;
                    }
                    Y00 = pk_add(Y00, BS); Y01 = pk_add(Y01, BS); Y10 = pk_add(Y10, BS); Y11 = pk_add(Y11, BS);
                }
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const int v = v0 + c2, ko = ko0 + c2;
                    float2 r0 = make_float2(0.f, 0.f), r1 = r0;
                    if (EPI) {
                        r0 = rq0[v % RPF];
                        r1 = rq1[v % RPF];
                        if (v + RPF < 16) res_fetch(v + RPF, rq0[v % RPF], rq1[v % RPF]);
                    }
                    // this channel's two pixel pairs (y00, y01), (y10, y11)
                    f32x2 P0 = c2 ? pk_hi_hi(Y00, Y01) : pk_lo_lo(Y00, Y01), P1 = c2 ? pk_hi_hi(Y10, Y11) : pk_lo_lo(Y10, Y11);
                    if (!whole) {       // a partial item: its raw sums to the workgroup's slot, finished by wino_sk_fixup_kernel
                        // buffer stores: the slot's base is a descriptor in SGPRs, the (slot, channel) offset an SGPR, the thread's
                        // 16 bytes (y00, y01 | y10, y11) one VGPR offset -- no 64-bit per-lane address beside the 256 live accumulators
                        const int slot = 2 * (int)blockIdx.x + ((item != item0 && item == item_last) ? 1 : 0);
                        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((slot * 16 + v) * (NT * 16));
                        const f32x4 pv = {P0.x, P0.y, P1.x, P1.y};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, pv),
                                                               prs, (unsigned)(wv_s * 64 + lane_o) * 16u, soff, 0);
                        // the next instruction that writes these four registers is a v_pk_mov_b32 written as asm: the compiler's
                        // hazard recognizer leaves the 16-byte store's data alone for its own vector instructions but does not see
                        // into the asm -- without the two wait states the store sent the NEXT channel's values now and then
                        asm volatile("s_nop 1" ::: "memory");
                    } else if (inside && ko < a.K) {
                        float* yp = yb + (size_t)ko * a.Ho * a.Wo;
                        if (EPI) {      // fused eval-mode BatchNorm (scale in the filter, shift = bias) + identity + ReLU
                            if (a.res) {
                                if (a.relu & 2) {   // the tensor is a saved ReLU output: pass the gradient where it was positive
                                    P0.x = r0.x > 0.f ? P0.x : 0.f; P0.y = r0.y > 0.f ? P0.y : 0.f;
                                    P1.x = r1.x > 0.f ? P1.x : 0.f; P1.y = r1.y > 0.f ? P1.y : 0.f;
                                } else {
                                    P0 = pk_add(P0, f32x2{r0.x, r0.y});
                                    P1 = pk_add(P1, f32x2{r1.x, r1.y});
                                }
                            }
                            if (a.relu & 1) {
                                P0.x = fmaxf(P0.x, 0.f); P0.y = fmaxf(P0.y, 0.f); P1.x = fmaxf(P1.x, 0.f); P1.y = fmaxf(P1.y, 0.f);
                            }
                        }
                        if ((DMH_WINO_ABLATE & 32) && a.K > 0) {     // timing experiment: everything but the stores
                            asm volatile("" ::"v"(P0), "v"(P1));
                        } else if (EPI || a.csplit == 1) {
                            *reinterpret_cast<f32x2*>(yp) = P0;
                            *reinterpret_cast<f32x2*>(yp + a.Wo) = P1;
                        } else {    // two partial sums into zeros: 0 + a + b is the same in either order (deterministic)
                            unsafeAtomicAdd(yp, P0.x);
                            unsafeAtomicAdd(yp + 1, P0.y);
                            unsafeAtomicAdd(yp + a.Wo, P1.x);
                            unsafeAtomicAdd(yp + a.Wo + 1, P1.y);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // one channel pair at a time: hoisted accumulator reads spill
            }
        }
#pragma unroll
        for (int k = 0; k < RAW_PER_T; ++k) roff[k] = roff_n[k];
        uoff = uoff_n;
        ixa = ixa_n;
        it_cur = it_nxt;
    }
}

__global__ __launch_bounds__(NT) void zero_fill_kernel(f32x4* __restrict__ p, size_t n4) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i < n4) p[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Stream-K second stage: four workgroups (blockIdx.y: four of the sixteen output channels of a lane each) per range boundary
// that cuts an item.  The workgroups of the FIRST cut of an item add the item's partial pieces in chunk order (fixed:
// deterministic, no atomics), apply the bias and write the item's outputs with the main kernel's own thread -> (tile, channel)
// map.  The next piece is requested before the current one is added (the pieces of an item cut five times would otherwise cost
// five dependent round trips to memory).
template <int TRW, bool FLAT>
__global__ __launch_bounds__(NT) void wino_sk_fixup_kernel(WArgs a) {
    constexpr int TRH = 64 / TRW;
    const int nch = a.C / CK;
    const int w = (int)blockIdx.x + 1, v0 = 4 * (int)blockIdx.y;
    const int b = sk_boundary(a.sk_per, a.sk_rem, nch, w);
    const int item = b / nch;
    if (b == item * nch) return;                                 // the boundary falls between two items
    const int bp = sk_boundary(a.sk_per, a.sk_rem, nch, w - 1);
    if (bp > item * nch) return;                                 // an earlier boundary cuts this item: its workgroups sum
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, kb = wv & 1, tb = wv >> 1;
    const f32x4* part = reinterpret_cast<const f32x4*>(a.part) + (size_t)v0 * NT + tid;
    // the pieces in chunk order: the last piece of workgroup w-1 (its only one if it starts exactly at the item), then the first
    // piece of every workgroup whose range starts inside the item
    f32x4 acc[4], nx[4];
    {
        const size_t s0 = (size_t)(2 * (w - 1) + (bp == item * nch ? 0 : 1)) * 16 * NT, s1 = (size_t)(2 * w) * 16 * NT;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            acc[v] = part[s0 + (size_t)v * NT];
            nx[v] = part[s1 + (size_t)v * NT];
        }
    }
    const int end = (item + 1) * nch;
    for (int ww = w;; ++ww) {
        const bool last = sk_boundary(a.sk_per, a.sk_rem, nch, ww + 1) >= end;
        f32x4 n2[4];
        if (!last) {
            const size_t s2 = (size_t)(2 * (ww + 1)) * 16 * NT;
#pragma unroll
            for (int v = 0; v < 4; ++v) n2[v] = part[s2 + (size_t)v * NT];
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] += nx[v];
        if (last) break;
#pragma unroll
        for (int v = 0; v < 4; ++v) nx[v] = n2[v];
    }
    const Item it = decode_item<TRW>(a, item);
    const int Ht = a.Ho >> 1, NR = a.B * Ht;
    const int tl = tb * 32 + (lane & 31);
    const int Rt = it.ty0 + tl / TRW;
    const int ob = FLAT ? min(Rt, NR - 1) / Ht : it.b;
    const int oy = 2 * (FLAT ? Rt - ob * Ht : Rt), ox = 2 * (it.tx0 + tl % TRW);
    const bool inside = (FLAT ? Rt < NR : oy < a.Ho) && ox < a.Wo;
    float* yb = a.y + (size_t)ob * a.K * a.Ho * a.Wo + (size_t)oy * a.Wo + ox;
    const int kbase = it.k0 + kb * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int ko = kbase + v + 8 * (int)blockIdx.y;          // channel v0 + v of the lane: (vv & 3) + 8 (vv >> 2), vv = v0 + v
        if (inside && ko < a.K) {
            const float bs = a.bias ? a.bias[ko] : 0.f;
            float* yp = yb + (size_t)ko * a.Ho * a.Wo;
            float y00 = acc[v][0] + bs, y01 = acc[v][1] + bs, y10 = acc[v][2] + bs, y11 = acc[v][3] + bs;
            if (a.res) {        // the fused epilogue of the main kernel (eval-mode BatchNorm shift = bias, identity / ReLU mask, ReLU)
                const float* rp = a.res + (yp - a.y);
                const float2 r0 = *reinterpret_cast<const float2*>(rp), r1 = *reinterpret_cast<const float2*>(rp + a.Wo);
                if (a.relu & 2) {
                    y00 = r0.x > 0.f ? y00 : 0.f; y01 = r0.y > 0.f ? y01 : 0.f;
                    y10 = r1.x > 0.f ? y10 : 0.f; y11 = r1.y > 0.f ? y11 : 0.f;
                } else {
                    y00 += r0.x; y01 += r0.y; y10 += r1.x; y11 += r1.y;
                }
            }
            if (a.relu & 1) {
                y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f);
            }
            *reinterpret_cast<float2*>(yp) = make_float2(y00, y01);
            *reinterpret_cast<float2*>(yp + a.Wo) = make_float2(y10, y11);
        }
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

// CUs a K10 launch may occupy.  DMH_K10_RESERVE_CUS=n leaves n CUs without a K10 workgroup: a contingency for multi-GPU runs,
// where RCCL's kernels cannot become resident on a CU this kernel occupies; default 0.  Applies to the
// whole-item grid and to the stream-K grid alike.
int usable_cus() {
    static const int reserve = getenv("DMH_K10_RESERVE_CUS") ? atoi(getenv("DMH_K10_RESERVE_CUS")) : 0;
    const int n = num_cus();
    return n - (reserve > 0 && reserve < n ? reserve : 0);
}

// The stream-K decision of launch_split(), also what dmh_wino_conv3x3_plan() reports: taken when a workspace of 2 G slots is
// there and the cost model (3.05 us per chunk, ~4 us per item epilogue, ~6 us for the second launch: profiles/README.md) says
// it is >= 8 % faster than whole items.  With the fused epilogue only launches of FEW regions take it (< 200).  The count is the
// LAUNCH's own: at the attack batch of the benchmark (12 scenes) the windowed encoder launches have >= 240 regions and keep whole
// items like their whole-frame twins (bit-identical, tests/test_roi.py); at a small attack batch (e.g. 2 scenes per rank) a
// windowed launch can fall below 200 and be decomposed while its whole-frame twin is not -- the partial sums are then added in
// another order and the two agree to fp32 rounding, not bit for bit.
bool sk_decide(long long regions, int nch, bool epi, bool have_ws, int64_t ws_floats, int& G, long long& units) {
    if (!have_ws || (epi && regions >= 200)) return false;
    const int cus = usable_cus();
    units = regions * nch;
    G = (int)(units / 8 < cus ? units / 8 : cus);
    if (!(G >= 2 && units < ((long long)1 << 30) && (long long)2 * G * SK_SLOT <= ws_floats)) return false;
    const int cs = (!epi && regions < (3 * cus) / 4 && nch % 2 == 0 && nch >= 6) ? 2 : 1;   // what the legacy path would do
    const long long items = regions * cs;
    // microseconds per chunk, per item epilogue, for the second launch; the margin (DMH_SK_MODEL="chunk,epilogue,launch,margin": A/B)
    static const struct Model { double c = 3.05, e = 4.0, l = 6.0, m = 0.92; Model() {
        if (const char* v = getenv("DMH_SK_MODEL")) sscanf(v, "%lf,%lf,%lf,%lf", &c, &e, &l, &m); } } M;
    const double t_cur = (double)((items + cus - 1) / cus) * ((nch / cs) * M.c + M.e) + (cs > 1 ? M.l : 0.0);
    const double per = (double)units / G;
    const double t_sk = per * M.c + M.e * (per / nch + 1.5) + M.l;
    return t_sk < M.m * t_cur;
}

template <int TRW, bool FLAT, bool EPI>
int launch(WArgs& a, hipStream_t st) {
    // U and V images (double-buffered, 128 KB) + the raw input region(s): see FLAT in the kernel
    constexpr size_t smem = RawGeo<TRW, FLAT>::SMEM;
    static std::atomic<uint64_t> configured{0};     // per device, see configure_dynamic_lds
    if (configure_dynamic_lds(wino_conv_kernel<TRW, FLAT, EPI>, smem, configured) != hipSuccess)
        return fail(DMH_ELAUNCH, "%s: cannot raise the dynamic LDS limit", "dmh_wino_conv3x3");
    // persistent: one workgroup per CU (its 154 KB of LDS and 512 registers per lane fill the CU), each walking a
    // contiguous range of work items
    const int cus = usable_cus();
    const int grid = a.nitems < cus ? a.nitems : cus;
    hipLaunchKernelGGL((wino_conv_kernel<TRW, FLAT, EPI>), dim3((unsigned)grid), dim3(NT), smem, st, a);
    return check_launch("dmh_wino_conv3x3");
}

// few regions (small images): split the channels of every region over two items so that the launch covers the chip
template <int TRW, bool FLAT>
int launch_split(WArgs& a, hipStream_t st, bool epi, float* ws, int64_t ws_floats) {
    const int64_t regions = (int64_t)a.bitems * a.gx * a.gy * a.kg;
    if (regions >= ((int64_t)1 << 30)) return fail(DMH_EINVAL, "%s: too many work items", "dmh_wino_conv3x3");
    const int nch = a.C / CK;
    a.part = nullptr; a.sk_units = 0; a.sk_grid = 0;
    // Stream-K with a caller-provided workspace (sk_decide()).  A launch is as long as its slowest workgroup:
    // with whole items that is ceil(items / CUs) x (one item's channel loop); here every workgroup gets the same number of
    // (item, chunk) units and at most two partial items, whose sums meet in wino_sk_fixup_kernel.  Taken when the model below
    // (3.05 us per chunk, ~4 us per item epilogue, ~6 us for the second launch: profiles/README.md) says it is >= 8 % faster.
    // With the fused epilogue only launches of FEW regions take it (< 200: the ones the no-split rule of ops.py turned away,
    // layer4 at the attack batch): the fix-up kernel then applies shift / identity / ReLU.  At the benchmark's attack batch the
    // windowed encoder launches have more regions and keep whole items (see sk_decide() for smaller batches).
    {
        int G = 0;
        long long units = 0;
        if (sk_decide(regions, nch, epi, ws != nullptr, ws_floats, G, units)) {
            a.csplit = 1; a.nitems = (int)regions; a.part = ws; a.sk_units = (int)units; a.sk_grid = G; a.sk_per = (int)(units / G); a.sk_rem = (int)(units % G);
            static std::atomic<uint64_t> configured{0};
            constexpr size_t smem = RawGeo<TRW, FLAT>::SMEM;
            static std::atomic<uint64_t> configured_epi{0};
            if ((epi ? configure_dynamic_lds(wino_conv_kernel<TRW, FLAT, true, true>, smem, configured_epi)
                     : configure_dynamic_lds(wino_conv_kernel<TRW, FLAT, false, true>, smem, configured)) != hipSuccess)
                return fail(DMH_ELAUNCH, "%s: cannot raise the dynamic LDS limit", "dmh_wino_conv3x3");
            if (epi) hipLaunchKernelGGL((wino_conv_kernel<TRW, FLAT, true, true>), dim3((unsigned)G), dim3(NT), smem, st, a);
            else hipLaunchKernelGGL((wino_conv_kernel<TRW, FLAT, false, true>), dim3((unsigned)G), dim3(NT), smem, st, a);
            if (int rc = check_launch("dmh_wino_conv3x3 (stream-K)")) return rc;
            hipLaunchKernelGGL((wino_sk_fixup_kernel<TRW, FLAT>), dim3((unsigned)(G - 1), 4), dim3(NT), 0, st, a);
            return check_launch("dmh_wino_conv3x3 (stream-K fix-up)");
        }
    }
    // (a fused activation needs the complete sum in one item: no split)
    a.csplit = (!epi && regions < (3 * usable_cus()) / 4 && nch % 2 == 0 && nch >= 6) ? 2 : 1;
    a.nitems = (int)regions * a.csplit;
    if (a.csplit > 1) {
        // a fill KERNEL, not hipMemsetAsync: inside a stream capture (torch.cuda.graph) the runtime's memset was not replayed
        // with the graph on this stack (ROCm 7.2) -- the two halves then accumulated onto the previous replay's output
        const size_t n4 = ((size_t)a.B * a.K * a.Ho * a.Wo) / 4;        // Ho, Wo even: a multiple of 4 floats
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n4 + NT - 1) / NT)), dim3(NT), 0, st,
                           reinterpret_cast<f32x4*>(a.y), n4);
        if (int rc = check_launch("dmh_wino_conv3x3 (zero fill)")) return rc;
    }
    return epi ? launch<TRW, FLAT, true>(a, st) : launch<TRW, FLAT, false>(a, st);
}

}  // namespace

extern "C" {

int64_t dmh_wino_weight_size(int n_out, int n_in) {
    if (n_out <= 0 || n_in <= 0 || n_in % CK) return -1;
    const int64_t Kp = ((int64_t)n_out + 63) / 64 * 64;
    return (int64_t)(n_in / CK) * 16 * 2 * Kp * 4;
}

int dmh_wino_weight_transform_scaled(const float* w, int K, int C, int backward, const float* scale, float* U,
                                     void* stream) {
    DMH_REQUIRE(w && U, "null pointer");
    const int n_out = backward ? C : K, n_in = backward ? K : C;
    DMH_REQUIRE(K > 0 && C > 0 && n_in % CK == 0, "the pass's input channel count must be a multiple of 8");
    DMH_REQUIRE(((uintptr_t)U & 15) == 0, "U must be 16-byte aligned");
    const int Kp = (n_out + 63) / 64 * 64;
    const long long n = (long long)Kp * (n_in / 4);
    hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, w, K, C,
                       backward ? 1 : 0, scale, U, Kp);
    return check_launch("dmh_wino_weight_transform");
}

int dmh_wino_weight_transform(const float* w, int K, int C, int backward, float* U, void* stream) {
    return dmh_wino_weight_transform_scaled(w, K, C, backward, nullptr, U, stream);
}

int dmh_wino_weight_transform_batch(const dmh_wino_wt_job* jobs, int n, void* stream) {
    DMH_REQUIRE(n >= 0 && (jobs || n == 0), "null pointer");
    for (int lo = 0; lo < n; lo += WT_MAX_JOBS) {
        WtBatch b;
        b.n = n - lo < WT_MAX_JOBS ? n - lo : WT_MAX_JOBS;
        long long blocks = 0;
        for (int j = 0; j < b.n; ++j) {
            const dmh_wino_wt_job& q = jobs[lo + j];
            DMH_REQUIRE(q.w && q.U && ((uintptr_t)q.U & 15) == 0, "null pointer / U not 16-byte aligned");
            const int n_out = q.backward ? q.C : q.K, n_in = q.backward ? q.K : q.C;
            DMH_REQUIRE(q.K > 0 && q.C > 0 && n_in % CK == 0, "the pass's input channel count must be a multiple of 8");
            b.w[j] = q.w; b.scale[j] = q.scale; b.U[j] = q.U; b.K[j] = q.K; b.C[j] = q.C; b.mode[j] = q.backward ? 1 : 0;
            b.Kp[j] = (n_out + 63) / 64 * 64;
            b.first[j] = (int)blocks;
            blocks += ((long long)b.Kp[j] * (n_in / 4) + NT - 1) / NT;
            DMH_REQUIRE(blocks < ((long long)1 << 30), "too many elements in one batch");
        }
        b.first[b.n] = (int)blocks;
        hipLaunchKernelGGL(wino_weight_batch_kernel, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, b);
        if (int rc = check_launch("dmh_wino_weight_transform_batch")) return rc;
    }
    return DMH_OK;
}

// tile-region geometry of a launch (a.B, a.Ho, a.Wo, ... set): fills gx, gy, bitems; returns 0 = 2 x 32 regions per image,
// 1 = 4 x 16 regions per image, 2 = 4 x 16 regions with the rows of tiles flattened over the batch (FLAT)
static int tile_geometry(WArgs& a) {
    const int Ht = a.Ho / 2, Wt = a.Wo / 2;
    // narrow images: 4 x 16 tile regions waste fewer lanes than 2 x 32 ones
    const bool narrow = (Wt % 32) != 0 && (Wt <= 16 || ((Wt + 15) / 16 * 16 - Wt) < ((Wt + 31) / 32 * 32 - Wt));
    if (narrow) {
        a.gx = (Wt + 15) / 16;
        // rows of tiles flattened over the batch when per-image regions would waste >= 1/5 of their tile rows
        if (5 * Ht <= 4 * ((Ht + 3) / 4 * 4) && (int64_t)a.B * a.C * a.H * a.W < ((int64_t)1 << 31)) {
            a.gy = (a.B * Ht + 3) / 4; a.bitems = 1;
            return 2;
        }
        a.gy = (Ht + 3) / 4; a.bitems = a.B;
        return 1;
    }
    a.gx = (Wt + 31) / 32; a.gy = (Ht + 1) / 2; a.bitems = a.B;
    return 0;
}

static int wino_conv_common(const float* x, const float* U, const float* bias, const float* residual, int relu, bool epi,
                            int B, int C, int K, int H, int W, int pad, float* y, void* stream, float* ws = nullptr,
                            int64_t ws_floats = 0) {
    DMH_REQUIRE(x && U && y, "null pointer");
    DMH_REQUIRE(B > 0 && C >= 3 * CK && K > 0 && C % CK == 0, "input channels must be a multiple of 8, at least 24");
    DMH_REQUIRE(pad >= 0 && pad <= 2, "pad must be 0, 1 or 2");
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    DMH_REQUIRE(Ho >= 2 && Wo >= 2 && (Ho & 1) == 0 && (Wo & 1) == 0, "output height and width must be even");
    DMH_REQUIRE((int64_t)C * H * W < ((int64_t)1 << 31) && (int64_t)K * Ho * Wo < ((int64_t)1 << 31), "image too large");
    DMH_REQUIRE((int64_t)B * C * H * W < ((int64_t)1 << 30), "input larger than 4 GB (32-bit byte offsets of the buffer loads)");
    WArgs a;
    a.x = x; a.U = reinterpret_cast<const f32x4*>(U); a.bias = bias; a.res = residual; a.relu = relu; a.y = y;
    a.B = B; a.C = C; a.K = K; a.Kp = (K + 63) / 64 * 64; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.pad = pad;
    a.kg = a.Kp / 64;
    switch (tile_geometry(a)) {
        case 2: return launch_split<16, true>(a, (hipStream_t)stream, epi, ws, ws_floats);
        case 1: return launch_split<16, false>(a, (hipStream_t)stream, epi, ws, ws_floats);
        default: return launch_split<32, false>(a, (hipStream_t)stream, epi, ws, ws_floats);
    }
}

int dmh_wino_conv3x3_plan(int B, int C, int K, int H, int W, int pad, int epilogue, int64_t workspace_floats) {
    if (!(B > 0 && C >= 3 * CK && K > 0 && C % CK == 0 && pad >= 0 && pad <= 2)) return -1;
    WArgs a;
    a.B = B; a.C = C; a.K = K; a.Kp = (K + 63) / 64 * 64; a.H = H; a.W = W; a.Ho = H + 2 * pad - 2; a.Wo = W + 2 * pad - 2; a.pad = pad;
    if (a.Ho < 2 || a.Wo < 2 || (a.Ho & 1) || (a.Wo & 1) || (int64_t)B * C * H * W >= ((int64_t)1 << 30)) return -1;
    a.kg = a.Kp / 64;
    const int form = tile_geometry(a);
    const long long regions = (long long)a.bitems * a.gx * a.gy * a.kg;
    if (regions >= ((long long)1 << 22)) return -1;
    const int nch = C / CK;
    int G = 0;
    long long units = 0;
    const bool sk = sk_decide(regions, nch, epilogue != 0, workspace_floats > 0, workspace_floats, G, units);
    const int csplit = (!sk && !epilogue && regions < (3 * usable_cus()) / 4 && nch % 2 == 0 && nch >= 6) ? 2 : 1;
    return (sk ? 1 : 0) | (csplit == 2 ? 2 : 0) | (form << 2) | ((int)(regions * csplit) << 8);
}

int dmh_wino_conv3x3(const float* x, const float* U, const float* bias, int B, int C, int K, int H, int W, int pad,
                     float* y, void* stream) {
    return wino_conv_common(x, U, bias, nullptr, 0, false, B, C, K, H, W, pad, y, stream);
}

int dmh_wino_conv3x3_ws(const float* x, const float* U, const float* bias, int B, int C, int K, int H, int W, int pad,
                        float* y, float* workspace, int64_t workspace_floats, void* stream) {
    DMH_REQUIRE(workspace == nullptr || workspace_floats > 0, "a workspace needs its size");
    DMH_REQUIRE(workspace == nullptr || ((uintptr_t)workspace & 15) == 0, "the workspace must be 16-byte aligned");
    return wino_conv_common(x, U, bias, nullptr, 0, false, B, C, K, H, W, pad, y, stream, workspace, workspace_floats);
}

int dmh_wino_conv3x3_act(const float* x, const float* U, const float* bias, const float* residual, int relu, int B, int C,
                         int K, int H, int W, int pad, float* y, void* stream) {
    return wino_conv_common(x, U, bias, residual, relu, true, B, C, K, H, W, pad, y, stream);
}

int dmh_wino_conv3x3_act_ws(const float* x, const float* U, const float* bias, const float* residual, int relu, int B, int C,
                            int K, int H, int W, int pad, float* y, float* workspace, int64_t workspace_floats, void* stream) {
    DMH_REQUIRE(workspace == nullptr || workspace_floats > 0, "a workspace needs its size");
    DMH_REQUIRE(workspace == nullptr || ((uintptr_t)workspace & 15) == 0, "the workspace must be 16-byte aligned");
    return wino_conv_common(x, U, bias, residual, relu, true, B, C, K, H, W, pad, y, stream, workspace, workspace_floats);
}

}  // extern "C"
