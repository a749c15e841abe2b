// K17 -- the 32-output-channel form of K10: 3x3 stride-1 convolution, Winograd F(2x2, 3x3) on v_mfma_f32_32x32x2_f32, for
// layers whose output channel count is a multiple of 32 but not of 64 (MD2/networks/depth_decoder.py:51-63: upconv(1,0)
// 64 -> 32 and upconv(1,1) 96 -> 32 forward, and the 32 -> 96 backward-data pass of the latter).  K10's work item is 64
// output channels x 64 tiles; with 32 channels half of its MFMA rows multiply zeros, which is why these layers stayed on
// MIOpen's vector-ALU Winograd through round 2 (16.5 ms of a 158 ms step).
//
//   * work item   : 32 output channels x 128 Winograd tiles (4 rows x 32 columns of tiles = 8 x 64 output pixels) x all
//                   input channels.  4 waves, one per SIMD; wave w owns tile row w: 32 channels x 32 tiles x all 16 transform
//                   positions = 16 accumulators of 32x32 (256 registers per lane), so the output transform A^T M A is
//                   per-lane register arithmetic exactly as in K10.
//   * LDS         : the transformed-input image V of a chunk is 64 KB for 128 tiles, so it is SINGLE-buffered: a chunk is
//                   an MFMA phase (64 MFMAs per wave on U[cur], V) and a transform phase (raw[next] -> V) separated by two
//                   barriers.  The fp32 MFMA shares the vector pipe (tools/micro/mfma_shadow.hip), so de-interleaving the
//                   transform from the MFMAs costs barrier skew only.  U (16 KB per chunk: 32 channels) and the raw input
//                   region (8 channels x 10 x 66) stay double-buffered: 32 + 64 + 43 KB = 139 KB.
//   * per chunk   : filter chunk by LDS-DMA (global_load_lds_dwordx4), raw region through registers one chunk ahead of the
//                   LDS write and two ahead of its transform; loads of chunk g+3 in flight during the MFMAs of chunk g; the
//                   staging runs across item boundaries (the next item's offsets are kept beside the current ones).
//   * traffic     : per 8 input channels a CU fetches 16 KB of filter + 21 KB of input for 4096 MFMA cycles = 9 B/cycle
//                   (K10: 32 + 13 KB = 11 B/cycle).
// Filter layout: U[C/8][16 positions][2 halves][Kp][4 floats], Kp = K rounded up to 32 (dmh_wino32_weight_transform).
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

using namespace dmh;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// DMH_W32_PK (round 6, as DMH_WINO_PK in wino_conv.hip): the input transform on packed fp32 adds -- the raw rows of a thread's
// channel pair are interleaved in LDS ([pair][row][channel of the pair][word]), one ds_read2_b32 fetches a patch element of both
// channels into a register pair, both stages are v_pk_add_f32 on such pairs (56 instead of 112 per chunk), and the results are
// the (channel, channel + 1) pairs the 8-byte image writes want.  Same additions in the same order: bit-identical results.
#ifndef DMH_W32_PK
#define DMH_W32_PK 1
#endif
constexpr int CK = 8;                       // input channels per chunk
constexpr int NT = 256;
constexpr int TRW = 32, TRH = 4;            // tile region of an item
constexpr int RW = 2 * TRW + 2, RH = 2 * TRH + 2;
constexpr int NWR = (RW + 6) / 4;           // the raw rows are staged as aligned 16-byte words (see K10): 18 per row
constexpr int RWA = 4 * NWR;                // LDS row pitch in floats (72)
constexpr int RAW_N = CK * RH * NWR;        // 1440 words
constexpr int RAW_PER_T = (RAW_N + NT - 1) / NT;   // 6 loads per thread and chunk (dword loads: 21)
constexpr int RAW_BUF = RAW_N * 4;          // floats per raw buffer (23 KB)
constexpr int PKI = DMH_W32_PK ? 2 : 1;     // DMH_W32_PK: the rows of a channel pair are interleaved
constexpr int UBUF = 16 * 2 * 32;           // f32x4 words of one U chunk image (16 KB)
constexpr int VBUF = 16 * 2 * 128;          // f32x4 words of the V image (64 KB)

struct W32Args {
    const float* x;
    const f32x4* U;
    const float* bias;
    float* y;
    int B, C, K, Kp, H, W, Ho, Wo, pad;
    int gx, gy, kg;              // tile-region groups along x / y, output-channel groups of 32
    int nitems;                  // B * gy * gx * kg
    float* part;                 // SK: workspace of the partial items (2 * sk_grid slots of SK_SLOT floats), see K10
    int sk_units, sk_grid;       // SK: nitems * C/8 (item, chunk) units dealt to sk_grid workgroups in equal contiguous ranges
    int sk_per, sk_rem;          // sk_units / sk_grid, sk_units % sk_grid
};
constexpr int SK_SLOT = 16 * NT * 4;      // floats of one partial item: [output channel v 16][thread 256] float4 (y00, y01, y10, y11)

struct Item { int b, ty0, tx0, k0; };

__device__ __forceinline__ Item decode_item(const W32Args& a, int item) {
    Item it;
    it.k0 = (item % a.kg) * 32;  item /= a.kg;          // channel groups fastest: the groups of a region run back to back
    it.tx0 = (item % a.gx) * TRW;  item /= a.gx;
    it.ty0 = (item % a.gy) * TRH;
    it.b = item / a.gy;
    return it;
}
// the item after `it`: a workgroup walks a contiguous range, so only its first item is decoded by division (see K10)
__device__ __forceinline__ Item next_item(const W32Args& a, Item it) {
    if ((it.k0 += 32) < a.kg * 32) return it;
    it.k0 = 0;
    if ((it.tx0 += TRW) < a.gx * TRW) return it;
    it.tx0 = 0;
    if ((it.ty0 += TRH) < a.gy * TRH) return it;
    it.ty0 = 0;
    ++it.b;
    return it;
}

template <bool SK>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino32_conv_kernel(W32Args a) {
    extern __shared__ f32x4 smem[];
    f32x4* U_lds = smem;                                        // [2][16][2][32]
    f32x4* V_lds = smem + 2 * UBUF;                             // [16][2][128]
    float* raw = reinterpret_cast<float*>(smem + 2 * UBUF + VBUF);   // [2][CK][RH][RW]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wv_s = __builtin_amdgcn_readfirstlane(wv);
    const size_t HW = (size_t)a.H * a.W;
    const int nch = a.C / CK;

    int item0, nmine, cb0 = 0, ce_last = nch;   // SK: the first piece starts at chunk cb0 of item0, the last ends before ce_last
    if (SK) {       // equal ranges of (item, chunk) units: a range may begin and end inside an item (K10, wino_conv.hip)
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

    // transform role: wave wv owns chunk channels {2wv, 2wv+1} = (h = wv>>1, s = 2(wv&1) + {0,1}); a lane transforms the two
    // vertically adjacent tiles (rows 2 (lane >> 5), + 1; column lane & 31) of the region (tile t = row t/32, column t%32)
    const int tl0y = lane >> 5, tlx = lane & 31;
    const int coff = (4 - (a.pad & 3)) & 3;                 // columns of the first word in front of the region
    const bool partial = a.pad > 0 && (a.W & 3) != 0;       // a word can straddle the right image edge
    f32x4* const raw4 = reinterpret_cast<f32x4*>(raw);
    const float* rsrc0 = raw + (2 * wv) * (RH * RWA) + (4 * tl0y) * (PKI * RWA) + 2 * tlx + coff;  // upper tile; the lower one: + 2 rows
    float* vdst0 = reinterpret_cast<float*>(V_lds + (wv >> 1) * 128 + 64 * tl0y + tlx) + 2 * (wv & 1);   // +32 words: the tile below
    // MFMA role: wave wv multiplies the 32 channels by tiles [32 wv, 32 wv + 32)
    const int aidx = (lane >> 5) * 32 + (lane & 31);
    const int bidx = (lane >> 5) * 128 + wv * 32 + (lane & 31);

    // input and filter through buffer resources (see K10): 32-bit per-thread byte offsets + a wave-uniform SGPR offset per
    // load; padded elements carry the offset 0xFFFFFFFF and read 0 (no clamp, no mask)
    const rsrc_t xrs = make_rsrc(a.x, (unsigned)((size_t)a.B * a.C * HW * 4));
    const rsrc_t urs = make_rsrc(a.U, (unsigned)((size_t)nch * 32 * a.Kp * 16));
    const rsrc_t prs = make_rsrc(SK ? (const void*)a.part : (const void*)a.U, SK ? (unsigned)(2 * a.sk_grid) * (unsigned)(SK_SLOT * 4) : 16u);
    unsigned roff[RAW_PER_T], roff_n[RAW_PER_T];
    unsigned uoff = 0, uoff_n = 0;
    int ixa = 0, ixa_n = 0;
    // raw-load constants of an item.  The thread index is rebuilt from v_mbcnt so that nothing of this is hoisted and spilled.
#define DMH_W32_ITEM_CONSTS(ITEM, IT, ROFF, UOFF, IXA)                                             \
    {                                                                                             \
        const Item it = (IT);                                                                     \
        const int ix0 = 2 * it.tx0 - a.pad - coff, iy0 = 2 * it.ty0 - a.pad;                      \
        IXA = ix0;                                                                                \
        int tid_o;                                                                                \
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tid_o)); \
        tid_o += wv_s * 64;                                                                       \
        const int cb_ = (SK && (ITEM) == item0) ? cb0 : 0;    /* a workgroup's first piece may start inside its item */ \
        UOFF = (unsigned)it.k0 * 16u + (unsigned)cb_ * (unsigned)(32 * a.Kp * 16);                \
        const int cbase = (it.b * a.C + cb_ * CK) * (int)HW;                                      \
        _Pragma("unroll") for (int k = 0; k < RAW_PER_T; ++k) {                                   \
            const int e = tid_o + NT * k;                                                         \
            /* LDS order: [channel][row][word]; DMH_W32_PK: [channel pair][row][channel of the pair][word] */ \
            const int cq = e / (PKI * RH * NWR), r1 = e - cq * (PKI * RH * NWR), rr = r1 / (PKI * NWR), r2 = r1 - rr * (PKI * NWR); \
            const int c = PKI * cq + r2 / NWR, xx = 4 * (r2 % NWR);                               \
            const int iy = iy0 + rr, ix = ix0 + xx;                                               \
            const bool ok = e < RAW_N && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;              \
            ROFF[k] = ok ? (unsigned)(cbase + c * (int)HW + iy * a.W + ix) * 4u : 0xFFFFFFFFu;    \
        }                                                                                         \
    }
    f32x4 rreg[RAW_PER_T];
#define DMH_W32_LOAD1(K, CHB, OFF) rreg[K] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (OFF), (CHB), 0));
#define DMH_W32_LOAD_RAW(CHB, ROFF)                                                               \
    _Pragma("unroll") for (int k = 0; k < RAW_PER_T; ++k) DMH_W32_LOAD1(k, CHB, ROFF[k])
#define DMH_W32_WRITE1(K, BUFI, IXW)                                                              \
    {                                                                                             \
        const int e_ = tid + NT * (K);                                                            \
        f32x4 v_ = rreg[K];                                                                       \
        if (partial) {        /* columns at and beyond W hold the next row's first pixels */      \
            const int n_ = a.W - ((IXW) + 4 * (e_ % NWR));                                        \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) v_[j] = (j < n_) ? v_[j] : 0.f;         \
        }                                                                                         \
        if (RAW_PER_T * NT == RAW_N || e_ < RAW_N) raw4[(BUFI) * RAW_N + e_] = v_;                \
    }
#define DMH_W32_WRITE_RAW(BUFI, IXW)                                                              \
    _Pragma("unroll") for (int k = 0; k < RAW_PER_T; ++k) DMH_W32_WRITE1(k, BUFI, IXW)
    // filter chunk: 16 positions x (2 halves x 32 channels x 16 B = 1 KB, contiguous in LDS): one LDS-DMA instruction per
    // position, four per wave; lanes 0-31 read half 0, lanes 32-63 half 1 of the position's rows (Kp * 16 B apart).
    // asm: see K10 (hipcc would drain vmcnt(0) at every later LDS read); completion is counted by hand before the barriers.
    const unsigned lane_u = (unsigned)((lane >> 5) * a.Kp + (lane & 31)) * 16u;
#define DMH_W32_GLDS_U_ROW(UCB, BUFI, Q)                                                          \
    {                                                                                             \
        const int p_ = wv_s + 4 * (Q);                                                            \
        const unsigned srow = __builtin_amdgcn_readfirstlane((UCB) + (unsigned)(p_ * 2 * a.Kp) * 16u);   \
        const unsigned ldst = __builtin_amdgcn_readfirstlane(                                     \
            (unsigned)(uintptr_t)(U_lds + (BUFI) * UBUF) + (unsigned)(p_ * 64 * 16));              \
        unsigned keep;                                                                            \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep) : "v"(lane_u), "s"(urs), "s"(ldst), "s"(srow) : "memory");      \
    }
    // input transform B^T d B of this thread's two channels for TWO vertically adjacent tiles (rows 2q..2q+3 and 2q+2..2q+5 of
    // the raw patch at RS): the horizontal stage d B is taken once per raw row -- six rows instead of two times four (24
    // instead of 32 LDS words read, 56 instead of 64 additions per channel) --, the vertical stage B^T (.) per tile.
    // V words at VD (upper tile) and VD + 32 words (the tile below it).
#if DMH_W32_PK
#define DMH_W32_TRANSFORM_PAIR(RS, VD)                                                            \
    {                                                                                             \
        f32x2 h_[6][4];         /* (channel, channel + 1) pairs */                                \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                           \
            const float* row_ = (RS) + 2 * i * RWA;                       /* 4-byte aligned only */ \
            const f32x2 d0_ = {row_[0], row_[RWA]}, d1_ = {row_[1], row_[RWA + 1]};               \
            const f32x2 d2_ = {row_[2], row_[RWA + 2]}, d3_ = {row_[3], row_[RWA + 3]};           \
            h_[i][0] = pk_sub(d0_, d2_);                                                          \
            h_[i][1] = pk_add(d1_, d2_);                                                          \
            h_[i][2] = pk_sub(d2_, d1_);                                                          \
            h_[i][3] = pk_sub(d1_, d3_);                                                          \
        }                                                                                         \
        _Pragma("unroll") for (int tv_ = 0; tv_ < 2; ++tv_) {                                     \
            float* const vd_ = (VD) + tv_ * (32 * 4);                                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                       \
                *reinterpret_cast<f32x2*>(vd_ + (0 * 4 + j) * 1024) = pk_sub(h_[2 * tv_][j], h_[2 * tv_ + 2][j]); \
                *reinterpret_cast<f32x2*>(vd_ + (1 * 4 + j) * 1024) = pk_add(h_[2 * tv_ + 1][j], h_[2 * tv_ + 2][j]); \
                *reinterpret_cast<f32x2*>(vd_ + (2 * 4 + j) * 1024) = pk_sub(h_[2 * tv_ + 2][j], h_[2 * tv_ + 1][j]); \
                *reinterpret_cast<f32x2*>(vd_ + (3 * 4 + j) * 1024) = pk_sub(h_[2 * tv_ + 1][j], h_[2 * tv_ + 3][j]); \
            }                                                                                     \
        }                                                                                         \
    }
#else
#define DMH_W32_TRANSFORM_PAIR(RS, VD)                                                            \
    {                                                                                             \
        float h_[2][6][4];                                                                        \
        _Pragma("unroll") for (int ch_ = 0; ch_ < 2; ++ch_) {                                     \
            _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                       \
                const float* row_ = (RS) + ch_ * (RH * RWA) + i * RWA;    /* 4-byte aligned only */ \
                const float d0_ = row_[0], d1_ = row_[1], d2_ = row_[2], d3_ = row_[3];           \
                h_[ch_][i][0] = d0_ - d2_;                                                        \
                h_[ch_][i][1] = d1_ + d2_;                                                        \
                h_[ch_][i][2] = d2_ - d1_;                                                        \
                h_[ch_][i][3] = d1_ - d3_;                                                        \
            }                                                                                     \
        }                                                                                         \
        _Pragma("unroll") for (int tv_ = 0; tv_ < 2; ++tv_) {                                     \
            float* const vd_ = (VD) + tv_ * (32 * 4);                                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                       \
                const float a0_ = h_[0][2 * tv_][j], a1_ = h_[0][2 * tv_ + 1][j], a2_ = h_[0][2 * tv_ + 2][j], a3_ = h_[0][2 * tv_ + 3][j]; \
                const float b0_ = h_[1][2 * tv_][j], b1_ = h_[1][2 * tv_ + 1][j], b2_ = h_[1][2 * tv_ + 2][j], b3_ = h_[1][2 * tv_ + 3][j]; \
                *reinterpret_cast<float2*>(vd_ + (0 * 4 + j) * 1024) = make_float2(a0_ - a2_, b0_ - b2_); \
                *reinterpret_cast<float2*>(vd_ + (1 * 4 + j) * 1024) = make_float2(a1_ + a2_, b1_ + b2_); \
                *reinterpret_cast<float2*>(vd_ + (2 * 4 + j) * 1024) = make_float2(a2_ - a1_, b2_ - b1_); \
                *reinterpret_cast<float2*>(vd_ + (3 * 4 + j) * 1024) = make_float2(a1_ - a3_, b1_ - b3_); \
            }                                                                                     \
        }                                                                                         \
    }
#endif
#define DMH_W32_TRANSFORM(BUFI) DMH_W32_TRANSFORM_PAIR(rsrc0 + (BUFI) * RAW_BUF, vdst0)

    f32x16 acc[16];         // never cleared: the first chunk of a piece multiplies onto a zero C operand (an inline constant), as in K10
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // Pipeline over the flattened (item, chunk) sequence.  Iteration g runs
    //   M(g)    64 MFMAs on U[g&1] and V; interleaved: LDS-DMA of filter chunk g+1 -> U[(g+1)&1]; registers (raw chunk
    //           g+2) -> raw[g&1]; global loads of raw chunk g+3 into the registers just freed
    //   barrier, T(g+1): raw[(g+1)&1] -> V, barrier
    const unsigned chunk_bytes = (unsigned)(CK * HW * 4);
    const unsigned uchunk_bytes = (unsigned)(32 * a.Kp * 16);
    Item it_cur = decode_item(a, item0);
    DMH_W32_ITEM_CONSTS(item0, it_cur, roff, uoff, ixa)
    DMH_W32_LOAD_RAW(0u, roff)
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) DMH_W32_GLDS_U_ROW(uoff, 0, k4)
    DMH_W32_WRITE_RAW(0, ixa)
    DMH_W32_LOAD_RAW(chunk_bytes, roff)
    __syncthreads();
    DMH_W32_TRANSFORM(0)
    DMH_W32_WRITE_RAW(1, ixa)
    DMH_W32_LOAD_RAW(2u * chunk_bytes, roff)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RAW_PER_T) : "memory");   // the LDS-DMA of U[0] has landed
    __syncthreads();

    int g = 0;
    for (int mi = 0; mi < nmine; ++mi) {
        const int item = item0 + mi;
        const Item it_nxt = item < item_last ? next_item(a, it_cur) : it_cur;
        if (item < item_last) {
            DMH_W32_ITEM_CONSTS(item + 1, it_nxt, roff_n, uoff_n, ixa_n)
        } else {    // the last item: the stages past its end re-stage its own first chunks
#pragma unroll
            for (int k = 0; k < RAW_PER_T; ++k) roff_n[k] = roff[k];
            uoff_n = uoff;
            ixa_n = ixa;
        }
        if (SK) pn = (item == item_last ? ce_last : nch) - (item == item0 ? cb0 : 0);
        // one chunk; FIRST: the piece's first chunk, whose first MFMA per position starts the accumulation from zero (round 6: the
        // 256 accumulator writes per item that cleared them are gone)
        auto chunk = [&](const int ch, auto first_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const int cur = g & 1, nxt = cur ^ 1;
            const f32x4* Uc = U_lds + cur * UBUF + aidx;
            const f32x4* Vc = V_lds + bidx;
            const bool r_next = ch + 3 >= pn, u_next = ch + 1 >= pn;
            unsigned xcb = (unsigned)(r_next ? ch + 3 - pn : ch + 3) * chunk_bytes;
            asm volatile("" : "+s"(xcb));          // a scalar offset parked in a VGPR would make every load a waterfall loop
            const unsigned ucb = u_next ? uoff_n : uoff + (unsigned)(ch + 1) * uchunk_bytes;
            f32x4 ua[16], vb[16];
            ua[0] = Uc[0]; vb[0] = Vc[0];
            ua[1] = Uc[64]; vb[1] = Vc[256];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sl = 0; sl < 32; ++sl) {
                const int p0 = 2 * (sl >> 2), p1 = p0 + 1, ks = sl & 3;
                if (FIRST && ks == 0) {
                    acc[p0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[p0][ks], vb[p0][ks], zero16, 0, 0, 0);
                    acc[p1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[p1][ks], vb[p1][ks], zero16, 0, 0, 0);
                } else {
                    acc[p0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[p0][ks], vb[p0][ks], acc[p0], 0, 0, 0);
                    acc[p1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[p1][ks], vb[p1][ks], acc[p1], 0, 0, 0);
                }
                if (p0 + 2 < 16) {                          // operands of the next position pair, one read per slot
                    if (ks == 0) ua[p0 + 2] = Uc[(p0 + 2) * 64];
                    if (ks == 1) vb[p0 + 2] = Vc[(p0 + 2) * 256];
                    if (ks == 2) ua[p1 + 2] = Uc[(p1 + 2) * 64];
                    if (ks == 3) vb[p1 + 2] = Vc[(p1 + 2) * 256];
                }
                if (sl < 4) {                               // filter chunk g+1 -> U[nxt] by LDS-DMA, one position per slot
                    DMH_W32_GLDS_U_ROW(ucb, nxt, sl)
                } else if (sl >= 8 && sl < 8 + RAW_PER_T) {  // raw registers (chunk g+2) -> raw[cur], then refill (chunk g+3)
                    const int k = sl - 8;
                    DMH_W32_WRITE1(k, cur, (ch + 2 >= pn) ? ixa_n : ixa)
                    DMH_W32_LOAD1(k, xcb, r_next ? roff_n[k] : roff[k])
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // every wave has read V; U[nxt] has landed when only this iteration's raw loads are outstanding
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(RAW_PER_T) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            DMH_W32_TRANSFORM(nxt)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ++g;
        };
        chunk(0, std::true_type());
        for (int ch = 1; ch < pn; ++ch) chunk(ch, std::false_type());
        // ---- item done: output transform Y = A^T M A, store; lane -> tile, register -> channel
        {
            const Item it = it_cur;
            int lane_o;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_o));
            const int oy = 2 * (it.ty0 + wv_s), ox = 2 * (it.tx0 + (lane_o & 31));
            const bool inside = oy < a.Ho && ox < a.Wo;
            float* yb = a.y + (size_t)it.b * a.K * a.Ho * a.Wo + (size_t)oy * a.Wo + ox;
            const int kbase = it.k0 + 4 * (lane_o >> 5);
            // The output transform as in K10 (round 6): every accumulator value read ONCE by an asm v_accvgpr_read_b32 (hipcc re-reads
            // the accumulator file for every use and spills around it), packed additions over the channel pair (v0, v0 + 1) = two
            // adjacent registers of every accumulator, one v_pk_mov_b32 per store regroups the pairs into a channel's pixel pairs.
            // The same additions in the same order: bit-identical.  The asm reads are invisible to the compiler's hazard recognizer:
            // 20 wait states put the item's last 16-pass MFMA behind us whatever the code layout (a barrier and the item decode lie
            // between anyway).
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
            for (int vp = 0; vp < 8; ++vp) {
                const int v0 = 2 * vp, ko0 = kbase + (v0 & 3) + 8 * (v0 >> 2);
                f32x2 A2[16], S0[4], S1[4];
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
                    const f32x2 BS = {(a.bias && ko0 < a.K) ? a.bias[ko0] : 0.f, (a.bias && ko0 + 1 < a.K) ? a.bias[ko0 + 1] : 0.f};
                    Y00 = pk_add(Y00, BS); Y01 = pk_add(Y01, BS); Y10 = pk_add(Y10, BS); Y11 = pk_add(Y11, BS);
                }
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const int v = v0 + c2, ko = ko0 + c2;
                    const f32x2 P0 = c2 ? pk_hi_hi(Y00, Y01) : pk_lo_lo(Y00, Y01), P1 = c2 ? pk_hi_hi(Y10, Y11) : pk_lo_lo(Y10, Y11);
                    if (!whole) {       // a partial item: its raw sums to the workgroup's slot (wino32_sk_fixup_kernel adds them)
                        const int slot = 2 * (int)blockIdx.x + ((item != item0 && item == item_last) ? 1 : 0);
                        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((slot * 16 + v) * (NT * 16));
                        const f32x4 pv = {P0.x, P0.y, P1.x, P1.y};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, pv),
                                                               prs, (unsigned)(wv_s * 64 + lane_o) * 16u, soff, 0);
                        // the next writer of these four registers is an asm v_pk_mov_b32 the hazard recognizer does not see: two wait
                        // states keep it off the 16-byte store's data (see K10)
                        asm volatile("s_nop 1" ::: "memory");
                    } else if (inside && ko < a.K) {
                        float* yp = yb + (size_t)ko * a.Ho * a.Wo;
                        *reinterpret_cast<f32x2*>(yp) = P0;
                        *reinterpret_cast<f32x2*>(yp + a.Wo) = P1;
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

// Stream-K second stage (see wino_sk_fixup_kernel in wino_conv.hip): the workgroups (four per cut: four output channels of a
// lane each) of an item's FIRST cut add the item's partial pieces in chunk order, add the bias and write the outputs with the
// main kernel's thread -> (tile, channel) map.
__global__ __launch_bounds__(NT) void wino32_sk_fixup_kernel(W32Args a) {
    const int nch = a.C / CK;
    const int w = (int)blockIdx.x + 1, v0 = 4 * (int)blockIdx.y;
    const int b = sk_boundary(a.sk_per, a.sk_rem, nch, w);
    const int item = b / nch;
    if (b == item * nch) return;
    const int bp = sk_boundary(a.sk_per, a.sk_rem, nch, w - 1);
    if (bp > item * nch) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const f32x4* part = reinterpret_cast<const f32x4*>(a.part) + (size_t)v0 * NT + tid;
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
    const Item it = decode_item(a, item);
    const int oy = 2 * (it.ty0 + wv), ox = 2 * (it.tx0 + (lane & 31));
    const bool inside = oy < a.Ho && ox < a.Wo;
    float* yb = a.y + (size_t)it.b * a.K * a.Ho * a.Wo + (size_t)oy * a.Wo + ox;
    const int kbase = it.k0 + 4 * (lane >> 5);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int ko = kbase + v + 8 * (int)blockIdx.y;
        if (inside && ko < a.K) {
            const float bs = a.bias ? a.bias[ko] : 0.f;
            float* yp = yb + (size_t)ko * a.Ho * a.Wo;
            *reinterpret_cast<float2*>(yp) = make_float2(acc[v][0] + bs, acc[v][1] + bs);
            *reinterpret_cast<float2*>(yp + a.Wo) = make_float2(acc[v][2] + bs, acc[v][3] + bs);
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

// same transform as K10's (U = G g G^T, chunked layout), with the channel rows padded to a multiple of 32
__global__ __launch_bounds__(NT) void wino32_weight_kernel(const float* __restrict__ w, int Kw, int Cw, int mode,
                                                           float* __restrict__ U, int Kp) {
    const int n_out = mode ? Cw : Kw, n_in = mode ? Kw : Cw;
    const int i = blockIdx.x * NT + threadIdx.x;       // over Kp * n_in
    if (i >= Kp * n_in) return;
    const int ko = i % Kp, ci = i / Kp;
    float g[3][3];
#pragma unroll
    for (int aa = 0; aa < 3; ++aa)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[aa][b] = 0.f;
    if (ko < n_out) {
        // mode 0: u[k][c] from w[k][c][ky][kx];  mode 1 (backward-data): u[c][k] from w[k][c][2-ky][2-kx]
        const float* src = mode ? w + ((size_t)ci * Cw + ko) * 9 : w + ((size_t)ko * Cw + ci) * 9;
#pragma unroll
        for (int aa = 0; aa < 3; ++aa)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[aa][b] = mode ? src[(2 - aa) * 3 + (2 - b)] : src[aa * 3 + b];
    }
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const int cc = ci / CK, cl = ci % CK, h = cl >> 2, s = cl & 3;
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) {
        const float u[4] = {t[aa][0], 0.5f * (t[aa][0] + t[aa][1] + t[aa][2]), 0.5f * (t[aa][0] - t[aa][1] + t[aa][2]), t[aa][2]};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int p = aa * 4 + b;
            U[((((size_t)cc * 16 + p) * 2 + h) * Kp + ko) * 4 + s] = u[b];
        }
    }
}

}  // namespace

extern "C" {

int64_t dmh_wino32_weight_size(int n_out, int n_in) {
    if (n_out <= 0 || n_in <= 0 || n_in % CK) return -1;
    const int64_t Kp = ((int64_t)n_out + 31) / 32 * 32;
    return (int64_t)(n_in / CK) * 16 * 2 * Kp * 4;
}

int dmh_wino32_weight_transform(const float* w, int K, int C, int backward, float* U, void* stream) {
    DMH_REQUIRE(w && U, "null pointer");
    const int n_out = backward ? C : K, n_in = backward ? K : C;
    DMH_REQUIRE(K > 0 && C > 0 && n_in % CK == 0, "the pass's input channel count must be a multiple of 8");
    const int Kp = (n_out + 31) / 32 * 32;
    const long long n = (long long)Kp * n_in;
    hipLaunchKernelGGL(wino32_weight_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, w, K, C,
                       backward ? 1 : 0, U, Kp);
    return check_launch("dmh_wino32_weight_transform");
}

static int wino32_common(const float* x, const float* U, const float* bias, int B, int C, int K, int H, int W, int pad,
                         float* y, float* ws, int64_t ws_floats, void* stream) {
    DMH_REQUIRE(x && U && y, "null pointer");
    DMH_REQUIRE(B > 0 && C >= 3 * CK && K > 0 && C % CK == 0, "input channels must be a multiple of 8, at least 24");
    DMH_REQUIRE(pad >= 0 && pad <= 2, "pad must be 0, 1 or 2");
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    DMH_REQUIRE(Ho >= 2 && Wo >= 2 && (Ho & 1) == 0 && (Wo & 1) == 0, "output height and width must be even");
    DMH_REQUIRE((int64_t)C * H * W < ((int64_t)1 << 31) && (int64_t)K * Ho * Wo < ((int64_t)1 << 31), "image too large");
    DMH_REQUIRE((int64_t)B * C * H * W < ((int64_t)1 << 30), "input larger than 4 GB (32-bit byte offsets of the buffer loads)");
    W32Args a;
    a.x = x; a.U = reinterpret_cast<const f32x4*>(U); a.bias = bias; a.y = y;
    a.B = B; a.C = C; a.K = K; a.Kp = (K + 31) / 32 * 32; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.pad = pad;
    a.kg = a.Kp / 32;
    const int Ht = Ho / 2, Wt = Wo / 2;
    a.gx = (Wt + TRW - 1) / TRW;
    a.gy = (Ht + TRH - 1) / TRH;
    const int64_t items = (int64_t)B * a.gx * a.gy * a.kg;
    DMH_REQUIRE(items < ((int64_t)1 << 30), "too many work items");
    a.nitems = (int)items;
    constexpr size_t smem = (size_t)(2 * UBUF + VBUF) * 16 + (size_t)2 * RAW_BUF * 4;
    a.part = nullptr; a.sk_units = 0; a.sk_grid = 0;
    const int cus = num_cus();
    // stream-K with a caller-provided workspace, where the cost model (3.4 us per chunk of this kernel, 4 us per item epilogue,
    // 6 us for the second launch) predicts >= 8 % over whole items: see launch_split() in wino_conv.hip
    if (ws) {
        const int nch = C / CK;
        const long long units = items * nch;
        const int G = (int)(units / 8 < cus ? units / 8 : cus);
        if (G >= 2 && units < ((long long)1 << 30) && (long long)2 * G * SK_SLOT <= ws_floats) {
            const double t_cur = (double)((items + cus - 1) / cus) * (nch * 3.4 + 4.0);
            const double per = (double)units / G;
            const double t_sk = per * 3.4 + 4.0 * (per / nch + 1.5) + 6.0;
            if (t_sk < 0.92 * t_cur) {
                a.part = ws; a.sk_units = (int)units; a.sk_grid = G; a.sk_per = (int)(units / G); a.sk_rem = (int)(units % G);
                static std::atomic<uint64_t> configured_sk{0};
                if (configure_dynamic_lds(wino32_conv_kernel<true>, smem, configured_sk) != hipSuccess)
                    return fail(DMH_ELAUNCH, "%s: cannot raise the dynamic LDS limit", "dmh_wino32_conv3x3");
                hipLaunchKernelGGL(wino32_conv_kernel<true>, dim3((unsigned)G), dim3(NT), smem, (hipStream_t)stream, a);
                if (int rc = check_launch("dmh_wino32_conv3x3 (stream-K)")) return rc;
                hipLaunchKernelGGL(wino32_sk_fixup_kernel, dim3((unsigned)(G - 1), 4), dim3(NT), 0, (hipStream_t)stream, a);
                return check_launch("dmh_wino32_conv3x3 (stream-K fix-up)");
            }
        }
    }
    static std::atomic<uint64_t> configured{0};     // per device, see configure_dynamic_lds
    if (configure_dynamic_lds(wino32_conv_kernel<false>, smem, configured) != hipSuccess)
        return fail(DMH_ELAUNCH, "%s: cannot raise the dynamic LDS limit", "dmh_wino32_conv3x3");
    const int grid = a.nitems < cus ? a.nitems : cus;
    hipLaunchKernelGGL(wino32_conv_kernel<false>, dim3((unsigned)grid), dim3(NT), smem, (hipStream_t)stream, a);
    return check_launch("dmh_wino32_conv3x3");
}

int dmh_wino32_conv3x3(const float* x, const float* U, const float* bias, int B, int C, int K, int H, int W, int pad,
                       float* y, void* stream) {
    return wino32_common(x, U, bias, B, C, K, H, W, pad, y, nullptr, 0, stream);
}

int dmh_wino32_conv3x3_ws(const float* x, const float* U, const float* bias, int B, int C, int K, int H, int W, int pad,
                          float* y, float* workspace, int64_t workspace_floats, void* stream) {
    DMH_REQUIRE(workspace == nullptr || workspace_floats > 0, "a workspace needs its size");
    DMH_REQUIRE(workspace == nullptr || ((uintptr_t)workspace & 15) == 0, "the workspace must be 16-byte aligned");
    return wino32_common(x, U, bias, B, C, K, H, W, pad, y, workspace, workspace_floats, stream);
}

}  // extern "C"
