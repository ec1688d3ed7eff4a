// bf16x3 implicit-GEMM convolution, 256 x 256 tile, FOUR waves of 128 x 128: one wave per SIMD, hand-pipelined K loop.
//
// Same arithmetic, operand formats, LDS image (XOR swizzle on the DMA source address) and epilogue as the X3 body of
// conv_igemm_glds.hip (f32 activations split into bf16 (hi, lo) in registers, pre-split pair-format weights, three
// v_mfma_f32_32x32x16_bf16 per product: a_lo*b_hi, a_hi*b_lo, a_hi*b_hi, f32 accumulate) -- results are bit-identical
// to that kernel's.  What differs is who overlaps what.  The 8-wave tile keeps two waves per SIMD that run in lock step:
// both read / split / issue DMA together, then both queue on the matrix pipe (its loop takes read phase + MFMA phase,
// DESIGN 3 "What bounds the main loop").  Here ONE wave owns a SIMD and a 128 x 128 accumulator block (256 AGPRs), and
// the instruction stream itself is the pipeline: every MFMA is followed by a fixed, small group of the other work
// (<= ~6 single-issue instructions per 32-cycle MFMA: MI355X_MICROARCH "one wave per SIMD" row), so the matrix pipe
// never waits for a phase to end:
//   * a K step (16 f32 of K) is TN = 4 groups (one per 32-wide output column block) of 12 MFMAs (3 terms x 4 row blocks);
//   * behind the MFMAs of group g run: the two ds_read_b128 of the NEXT K step's activation fragment of row block g and its
//     split into (hi, lo) (24 VALU, six per MFMA gap), the refill of the weight fragment registers the group has just
//     finished with (lo half after the 8th MFMA, hi half after the 12th), and two 1 KiB LDS-DMA pieces of a tile two
//     K tiles ahead;
//   * one s_barrier per K tile (32 f32 of K), placed after the first MFMAs of the tile's SECOND K step: it publishes
//     the next tile half a tile before its first fragment read, so the barrier wait is covered by queued MFMAs;
//   * LDS ring: three activation stages + two weight stages (3 x 32 + 2 x 32 KiB = 160 KiB), counted `s_waitcnt vmcnt(8)`.
// MFMAs, fragment reads and waits are inline asm (program order = issue order; hipcc does not model them, so the
// hazards it would pad are spaced by construction or by explicit s_nop); the split and the address arithmetic are
// plain C++ pinned into their gap by empty asm statements on their inputs / outputs.
// Contract (dispatcher in conv_igemm_glds.hip): dense conv, f32 storage, Cin % 32 == 0, Cout % 256 == 0, KH*KW <= 31.
#include <stdlib.h>

#include "conv_common.h"

namespace tt {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef TT_PIPE_DEBUG
#define TT_PIPE_DEBUG 0
#endif

// APAIR: pre-split (pair-format) activations, tt_conv_desc.in_pair -- the fragment reads fetch the bf16 hi and lo halves directly
// (chunks 4 kc + h and that ^ 2, like the weights) and the eight split stages of every row-block window are simply not there.
template <int WAVES_M, int WAVES_N, int BN = 256, bool APAIR = false>
__global__ __launch_bounds__(256, 1) void conv_x3_pipe_kernel(const ConvArgs p, const void* zero_page, int tiles_m,
                                                             int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 256, BKB = 128, BK = 32;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(NW == 4 && (TM * TN == 16 || TM * TN == 8) && TN % TM == 0,
                  "four waves of 128 x 128 (2 x 2) or 64 x 256 (4 x 1) on a 256 x 256 tile; of 64 x 128 (4 x 1) on a 256 x 128 tile");
    constexpr int A_BYTES = BM * BKB, B_BYTES = BN * BKB;
    constexpr int NIA = BM * 8 / 64 / NW, NIB = BN * 8 / 64 / NW;      // 1 KiB DMA pieces per wave per tile (8 + 8)
    constexpr int MG = 3 * TM;            // MFMAs per group (one 32-wide column block x TM row blocks x 3 terms)
    constexpr int NGR = TN / TM;          // groups per row-block window (the window in which one row block's next fragment is made)
    constexpr int WIN = NGR * MG;         // gaps per window
    constexpr int DPA = NIA / TN, DPB = NIB / TN;     // DMA pieces per group: activations (first K step of a tile) / weights
    static_assert(NIA % TN == 0 && NIB % TN == 0 && DPA >= 1 && DPA <= 2 && DPB >= 1 && DPB <= 2, "one or two DMA pieces per group");
    static_assert(WIN >= 12, "a row-block window needs 12 gaps for the barrier, the fragment reads and the split");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int Mlim = p.M;

    // XCD-aware tile order (bijective), as in conv_igemm_glds.hip
    const int nblk = tiles_m * tiles_n;
    if ((int)blockIdx.x >= nblk) return;
    int L;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nblk >> 3, r = nblk & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tile_n = L % tiles_n, tile_m = L / tiles_n;
    const int m0 = p.m_begin + tile_m * BM, n0 = tile_n * BN;
    if (m0 >= Mlim) return;

    const float* __restrict__ in = reinterpret_cast<const float*>(p.in);
    const float* __restrict__ wgt = reinterpret_cast<const float*>(p.weight);
    const float* zp = reinterpret_cast<const float*>(zero_page);
    auto swz = [](int row) { return (row >> 1) & 7; };

    // ---- DMA slots.  Activation slot j of this wave = 1 KiB piece (wave + 4 j) of the tile: 8 rows x 8 chunks.
    const float* a_ptr[NIA];            // the chunk's address for tap (0, 0), channel 0 (possibly outside the image)
    unsigned a_mask[NIA];               // bit t: tap t of this row lies inside the image (bit 31 never set)
#pragma unroll
    for (int j = 0; j < NIA; ++j) {
        const int g = (wave + NW * j) * 64 + lane;
        const int row = g >> 3, pos = g & 7;
        const int c = (pos ^ swz(row)) * 4;
        const int m = m0 + row;
        const bool ok = m < Mlim;
        const int mm = ok ? m : 0;
        const int n = mm / (p.OH * p.OW);
        const int r = mm - n * (p.OH * p.OW);
        const int oh = r / p.OW, ow = r - oh * p.OW;
        const int h0 = oh * p.stride - p.pad, w0 = ow * p.stride - p.pad;
        a_ptr[j] = in + (long long)n * p.in_nstride + p.in_coff + ((long long)h0 * p.W + w0) * p.in_cstride + c;
        unsigned mk = 0;
        if (ok) {
            int tbit = 0;
            for (int kh = 0; kh < p.KH; ++kh) {
                const int ih = h0 + kh * p.dil;
                for (int kw = 0; kw < p.KW; ++kw, ++tbit) {
                    const int iw = w0 + kw * p.dil;
                    if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) mk |= 1u << tbit;
                }
            }
        }
        a_mask[j] = mk;
    }
    // Weight slot j = piece (wave + 4 j) of the 256-row weight tile: rows 32 j apart, one pointer + a uniform stride.
    const float* b_ptr0;
    {
        const int g = wave * 64 + lane;
        const int row = g >> 3, pos = g & 7;
        b_ptr0 = wgt + (long long)(n0 + row) * p.K + (pos ^ swz(row)) * 4;
    }
    const long long b_jstride = (long long)(NW * 8) * p.K;        // elements between consecutive slots of a wave

    const int nk = p.K / BK;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // ring: activation stage s at s * 32 KiB (s = tile % 3), weight stage s at 96 KiB + s * 32 KiB (s = tile % 2)
    const unsigned ldsA = lds_base, ldsB = lds_base + 3u * A_BYTES;

    // ---- the two DMA walkers (wave-uniform, SALU).  K order: channel chunk outer, filter tap inner.  Everything a tile
    // needs is kept incrementally -- tap index, element offset from the slot pointer, ring stage -- and advanced by
    // compare + select (no branches: the loop body stays one basic block, its instruction order is the schedule).
    const int ntaps = p.KH * p.KW;
    const long long a_d1 = (long long)p.dil * p.in_cstride;                                         // kw + 1
    const long long a_d2 = ((long long)p.dil * p.W - (long long)(p.KW - 1) * p.dil) * p.in_cstride;  // kw -> 0, kh + 1
    const long long a_d3 = BK - ((long long)(p.KH - 1) * p.dil * p.W + (long long)(p.KW - 1) * p.dil) * p.in_cstride;
    const long long a_e2 = a_d2 - a_d1, a_e3 = a_d3 - a_d2;       // (selects below pick between VALUES: a `c ? x : y` of two
    const long long b_d1 = p.Cin;                                  // captured variables selects their addresses and pins them
    const long long b_e2 = (BK - (long long)(ntaps - 1) * p.Cin) - b_d1;   // to memory)
    int a_kw = 0, a_tap = 0, a_rem = nk;          // a_rem: tiles not yet issued, this one included
    long long a_off = 0;
    unsigned a_st = ldsA;
    int b_tap = 0, b_rem = nk, b_w = 0;
    long long b_off = 0;
    unsigned b_st = ldsB;
    long long a_d = 0;
    auto a_walk1 = [&]() {        // kw + 1 (wrap: kh + 1)
        const int kw1 = a_kw + 1;
        const bool w1 = kw1 == p.KW;
        a_kw = w1 ? 0 : kw1;
        a_d = a_d1 + (w1 ? a_e2 : 0ll);
    };
    auto a_walk2 = [&]() {        // tap + 1 (wrap: next channel chunk; a tap wrap is a kw wrap too)
        const int tp1 = a_tap + 1;
        const bool w2 = tp1 == ntaps;
        a_tap = w2 ? 0 : tp1;
        a_off += a_d + (w2 ? a_e3 : 0ll);
    };
    auto a_walk3 = [&]() {
        a_rem -= 1;
        a_st = a_st == ldsA + 2u * A_BYTES ? ldsA : a_st + A_BYTES;
    };
    auto b_walk1 = [&]() {
        // beyond the last tile the walker stands still: the last tile again (valid memory) into a free stage
        const int tp1 = b_tap + 1;
        const bool adv = b_rem > 1;
        b_w = tp1 == ntaps;
        b_off += (adv ? b_d1 : 0ll) + ((adv && b_w) ? b_e2 : 0ll);
        b_tap = adv ? (b_w ? 0 : tp1) : b_tap;
    };
    auto b_walk2 = [&]() {
        b_rem -= 1;
        b_st = b_st == ldsB ? ldsB + B_BYTES : ldsB;
    };
    // beyond the last tile every activation row reads the zero page (bit 31 of a_mask is never set)
    auto a_emit = [&](int j) {
        const int tapbit = a_rem > 0 ? a_tap : 31;
        const float* src = ((a_mask[j] >> tapbit) & 1u) ? a_ptr[j] + a_off : zp;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(uintptr_t)(a_st + (unsigned)(wave_s + NW * j) * 1024u), 16, 0, 0);
    };
    auto b_emit = [&](int j) {
        const float* src = b_ptr0 + b_off + (long long)j * b_jstride;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(uintptr_t)(b_st + (unsigned)(wave_s + NW * j) * 1024u), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- fragment offsets inside a stage.  A: row = lane & 31 of the row block, lane half h owns floats 8h .. 8h+7 of the
    // 16-wide K step = 16 B chunks 4 kc + 2 h and that + 1 (address ^ 16); B: hi chunk 4 kc + h, lo chunk that + 2 (^ 32).
    const unsigned hi = lane >> 5;
    unsigned fa_pre[2][TM], fb_pre[2][TN];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * WTM + i * 32 + (lane & 31);
            fa_pre[kc][i] = row * BKB + (((4u * kc + (APAIR ? hi : 2u * hi)) ^ swz(row)) << 4);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = wn * WTN + j * 32 + (lane & 31);
            fb_pre[kc][j] = row * BKB + (((4u * kc + hi) ^ swz(row)) << 4);
        }
    }
    auto lds_read = [](unsigned addr) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
        return v;
    };
// Timing ablations (tools/build_pipe_debug.sh; results are wrong by design): TT_PIPE_DEBUG bit 0 no DMA in the loop, bit 1 no
// operand split, bit 2 no MFMA, bit 3 no fragment reads in the loop, bit 4 no tile barrier
#if TT_PIPE_DEBUG & 4
#define TT_MFMA(c, a, b) asm volatile("" : "+a"(c) : "v"(a), "v"(b))
#else
#define TT_MFMA(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#endif
    // the split of one element pair in two 3-VALU halves (cvt_pk, shift, and | sub, sub, cvt_pk): bf16 hi = rne(x),
    // lo = rne(x - hi) with the subtraction exact in f32 -- the arithmetic of conv_igemm_glds.hip's split_frag
    auto split_a = [](float x0, float x1, uint32_t& h, float& t0, float& t1) {
        h = pack_bf16x2(x0, x1);
        t0 = __uint_as_float(h << 16);
        t1 = __uint_as_float(h & 0xffff0000u);
    };
    auto split_b = [](float x0, float x1, float t0, float t1) { return pack_bf16x2(x0 - t0, x1 - t1); };

    u32x4 ah[2][TM], al[2][TM];       // [K step parity][row block]: split activation fragments
    u32x4 bh[TN], bl[TN];             // weight fragments of the current K step (refilled behind their last use)
    u32x4 ra0, ra1;                   // raw f32 fragment in flight (one row block)

    // ---- prologue: tiles 0 and 1 go out whole, tile 0's fragments are read and split before the loop
    {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int j = 0; j < NIA; ++j) a_emit(j);
            a_walk1(); a_walk2(); a_walk3();
#pragma unroll
            for (int j = 0; j < NIB; ++j) b_emit(j);
            b_walk1(); b_walk2();
        }
        // tile 0 has landed for this wave once only tile 1's pieces (NIA + NIB) are outstanding
        if constexpr (NIA + NIB == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if constexpr (NIA + NIB == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (APAIR) {
                ah[0][i] = lds_read(ldsA + fa_pre[0][i]);
                al[0][i] = lds_read(ldsA + (fa_pre[0][i] ^ 32u));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(ah[0][i]), "+v"(al[0][i]));
                continue;
            }
            ra0 = lds_read(ldsA + fa_pre[0][i]);
            ra1 = lds_read(ldsA + (fa_pre[0][i] ^ 16u));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(ra0), "+v"(ra1));
            const float x[8] = {__uint_as_float(ra0.x), __uint_as_float(ra0.y), __uint_as_float(ra0.z), __uint_as_float(ra0.w),
                                __uint_as_float(ra1.x), __uint_as_float(ra1.y), __uint_as_float(ra1.z), __uint_as_float(ra1.w)};
            uint32_t h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t0, t1;
                split_a(x[2 * e], x[2 * e + 1], h[e], t0, t1);
                l[e] = split_b(x[2 * e], x[2 * e + 1], t0, t1);
            }
            ah[0][i] = u32x4{h[0], h[1], h[2], h[3]};
            al[0][i] = u32x4{l[0], l[1], l[2], l[3]};
        }
#pragma unroll
        for (int j = 0; j < TN - 1; ++j) {        // column TN-1's hi half is read in the first gap of the K step itself
            bh[j] = lds_read(ldsB + fb_pre[0][j]);
            bl[j] = lds_read(ldsB + (fb_pre[0][j] ^ 32u));
        }
        bl[TN - 1] = lds_read(ldsB + (fb_pre[0][TN - 1] ^ 32u));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bh[j]), "+v"(bl[j]));
    }

    // stage addresses of tile kt (cur) and kt + 1 (nxt), advanced once per tile in a late gap of the tile's last group
    unsigned sA_cur = ldsA, sA_nxt = ldsA + A_BYTES, sB_cur = ldsB, sB_nxt = ldsB + B_BYTES;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            // K step (kt, kc): operands ah/al[kc], bh/bl.  It prepares K step (kt, 1) (kc = 0) or (kt + 1, 0) (kc = 1) and
            // issues the DMA of the activations (kc = 0) / the weights (kc = 1) of tile kt + 2.
            const int kn = kc ^ 1;                                  // kc index of the next K step inside its tile
            const unsigned srcA = kc == 0 ? sA_cur : sA_nxt;       // stage the next K step's operands lie in
            const unsigned srcB = kc == 0 ? sB_cur : sB_nxt;
            auto dma = [&](int q) {
                if (TT_PIPE_DEBUG & 1) return;
                asm volatile("" ::: "memory");
                if (kc == 0) a_emit(q);
                else b_emit(q);
                asm volatile("" ::: "memory");
            };
            uint32_t sh[4], sl[4];
            float st0 = 0.f, st1 = 0.f;
            // split stage s = 2 * pair + half of the fragment in ra0 / ra1
            auto split_stage = [&](int s) {
                // pinned on both sides: the stage's 3 VALU cannot start before this point (input pin) nor end after it
                const int e = s >> 1;
                asm volatile("" : "+v"(ra0), "+v"(ra1));
                const float x0 = __uint_as_float(e == 0 ? ra0.x : e == 1 ? ra0.z : e == 2 ? ra1.x : ra1.z);
                const float x1 = __uint_as_float(e == 0 ? ra0.y : e == 1 ? ra0.w : e == 2 ? ra1.y : ra1.w);
                if (TT_PIPE_DEBUG & 2) {
                    if ((s & 1) == 0) sh[e] = __float_as_uint(x0);
                    else sl[e] = __float_as_uint(x1);
                    return;
                }
                if ((s & 1) == 0) {
                    split_a(x0, x1, sh[e], st0, st1);
                    asm volatile("" : "+v"(sh[e]), "+v"(st0), "+v"(st1));
                } else {
                    asm volatile("" : "+v"(st0), "+v"(st1));
                    sl[e] = split_b(x0, x1, st0, st1);
                    asm volatile("" : "+v"(sl[e]));
                }
            };
#pragma unroll
            for (int g = 0; g < TN; ++g) {
                const int rb = g / NGR;                             // row block whose next fragment this group works on
                const bool barw = (kc == 1 && rb == 0);             // window that carries the tile barrier
                const bool bar = barw && (g % NGR) == 0;            // ... and the group inside it
                const bool last = g == TN - 1;
                // window slots: reads @1, wait @4, eight split stages from @4 on; barrier window: barrier @3, reads @4, wait @7
                constexpr int R0 = 1, W0 = 4, RB = 4, WB = 7;
#pragma unroll
                for (int m = 0; m < MG; ++m) {
                    const int t = m / TM, i = m % TM;
                    if (t == 0) TT_MFMA(acc[i][g], al[kc][i], bh[g]);
                    else if (t == 1) TT_MFMA(acc[i][g], ah[kc][i], bl[g]);
                    else TT_MFMA(acc[i][g], ah[kc][i], bh[g]);
                    // ---- the gap behind MFMA m
                    const int ws = (g % NGR) * MG + m;              // slot inside the row-block window
                    if (m == 0 && !(TT_PIPE_DEBUG & 8)) {
                        // hi half of the column block the previous group finished with.  g = 0: column TN-1 of THIS K step
                        // (its registers were busy until the previous K step's last MFMA); else column g-1 of the next one
                        if (g == 0) bh[TN - 1] = lds_read(sB_cur + fb_pre[kc][TN - 1]);
                        else bh[g - 1] = lds_read(srcB + fb_pre[kn][g - 1]);
                    }
                    if (bar && m == 3) {
                        // tile kt + 1 has landed for this wave (the NIA youngest loads = activations of tile kt + 2 stay in
                        // flight); every fragment read of tile kt is complete: publish tile kt + 1, free tile kt's stages
                        if (TT_PIPE_DEBUG & 16) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    }
                    if (ws == (barw ? RB : R0) && !(TT_PIPE_DEBUG & 8)) {
                        if constexpr (APAIR) {      // (the registers' last readers, the previous K step's MFMAs, were issued long ago)
                            ah[kn][rb] = lds_read(srcA + fa_pre[kn][rb]);
                            al[kn][rb] = lds_read(srcA + (fa_pre[kn][rb] ^ 32u));
                        } else {
                            ra0 = lds_read(srcA + fa_pre[kn][rb]);
                            ra1 = lds_read(srcA + (fa_pre[kn][rb] ^ 16u));
                        }
                    }
                    // DMA: behind the fragment reads; in the barrier group after the barrier (the weight stage it frees)
                    {
                        constexpr int DPG = 2;                      // slots per group (a weight step may use one)
                        const int npieces = kc == 0 ? DPA : DPB;
                        const int s0 = bar ? (MG >= 8 ? 5 : 4) : 2; // barrier group: after the barrier at gap 3
                        if (m == s0) dma(npieces * g);
                        if (m == s0 + 1 && npieces == DPG) dma(npieces * g + 1);
                    }
                    if (ws == (barw ? WB : W0)) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if constexpr (APAIR) asm volatile("" : "+v"(ah[kn][rb]), "+v"(al[kn][rb]));
                        else asm volatile("" : "+v"(ra0), "+v"(ra1));
                        if (g == 0) asm volatile("" : "+v"(bh[TN - 1]));     // the gap-0 read of this K step has landed too
                    }
                    if (m == 2 * TM - 1 && !(TT_PIPE_DEBUG & 8)) bl[g] = lds_read(srcB + (fb_pre[kn][g] ^ 32u));   // lo half: free now
                    if constexpr (!APAIR) {
                        // the eight 3-VALU split stages, spread over what is left of the window
                        const int first = barw ? WB : W0;
                        const int avail = WIN - first;
                        int lastslot;
                        if (avail >= 16) {                           // every other gap
                            if (ws >= first && ((ws - first) & 1) == 0 && (ws - first) / 2 < 8) split_stage((ws - first) / 2);
                            lastslot = first + 14;
                        } else if (avail >= 8) {                     // every gap
                            if (ws >= first && ws - first < 8) split_stage(ws - first);
                            lastslot = first + 7;
                        } else {                                     // two stages per gap
                            if (ws >= first && ws - first < 4) {
                                split_stage(2 * (ws - first));
                                split_stage(2 * (ws - first) + 1);
                            }
                            lastslot = first + 3;
                        }
                        if (ws == lastslot) {
                            ah[kn][rb] = u32x4{sh[0], sh[1], sh[2], sh[3]};
                            al[kn][rb] = u32x4{sl[0], sl[1], sl[2], sl[3]};
                            asm volatile("" : "+v"(ah[kn][rb]), "+v"(al[kn][rb]));
                        }
                    }
                    // the walker of the DMA stream this K step has just finished issuing, and the tile's stage rotation
                    if (last && kc == 0) {
                        if (m == MG - 3) {
                            asm volatile("" : "+s"(a_kw));
                            a_walk1();
                            asm volatile("" : "+s"(a_kw), "+s"(a_d));
                        }
                        if (m == MG - 2) {
                            asm volatile("" : "+s"(a_tap), "+s"(a_d), "+s"(a_off));
                            a_walk2();
                            asm volatile("" : "+s"(a_tap), "+s"(a_off));
                        }
                        if (m == MG - 1) {
                            asm volatile("" : "+s"(a_rem), "+s"(a_st));
                            a_walk3();
                            asm volatile("" : "+s"(a_rem), "+s"(a_st));
                        }
                    }
                    if (last && kc == 1) {
                        if (m == MG - 3) {
                            asm volatile("" : "+s"(b_tap), "+s"(b_off));
                            b_walk1();
                            asm volatile("" : "+s"(b_tap), "+s"(b_off));
                        }
                        if (m == MG - 2) {
                            asm volatile("" : "+s"(b_rem), "+s"(b_st));
                            b_walk2();
                            asm volatile("" : "+s"(b_rem), "+s"(b_st));
                        }
                        if (m == MG - 1) {
                            asm volatile("" : "+s"(sA_cur), "+s"(sA_nxt), "+s"(sB_cur), "+s"(sB_nxt));
                            sA_cur = sA_nxt;
                            sA_nxt = sA_nxt == ldsA + 2u * A_BYTES ? ldsA : sA_nxt + A_BYTES;
                            const unsigned tb = sB_cur;
                            sB_cur = sB_nxt;
                            sB_nxt = tb;
                            asm volatile("" : "+s"(sA_cur), "+s"(sA_nxt), "+s"(sB_cur), "+s"(sB_nxt));
                        }
                    }
                }
            }
        }
    }
    // MFMA results -> any other reader: the hazard hipcc would pad for a builtin (8-pass XDL: 12+ states)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // zero-page DMA of the two tiles beyond the end
    __syncthreads();
    // one 32 x 128 row block at a time through the shared epilogue: statically indexed accumulator blocks only (a rolled
    // row-block loop anywhere in it would pin all 256 accumulator registers to scratch for the whole kernel)
    conv_epilogue<float, 1, TN, 32, WTN>(p, *reinterpret_cast<f32x16(*)[1][TN]>(&acc[0]), smem, wave, lane, wm * TM + 0, wn, m0, n0, Mlim);
    conv_epilogue<float, 1, TN, 32, WTN>(p, *reinterpret_cast<f32x16(*)[1][TN]>(&acc[1]), smem, wave, lane, wm * TM + 1, wn, m0, n0, Mlim);
    if constexpr (TM == 4) {
        conv_epilogue<float, 1, TN, 32, WTN>(p, *reinterpret_cast<f32x16(*)[1][TN]>(&acc[2]), smem, wave, lane, wm * TM + 2, wn, m0, n0, Mlim);
        conv_epilogue<float, 1, TN, 32, WTN>(p, *reinterpret_cast<f32x16(*)[1][TN]>(&acc[3]), smem, wave, lane, wm * TM + 3, wn, m0, n0, Mlim);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// RUN3: the same pipeline for 3 x 3 (KW = 3) stride-1 / dilation-1 "same" convolutions over a dense batch, with the
// activation stream staged as pixel RUNS.  In channel-last memory the three kw taps of one filter row read the same
// pixels shifted by one: for the 256 consecutive output pixels of a tile, tap (kh, kw) of output row r is the pixel with
// linear index  m0 + r + (kh - 1) W + (kw - 1)  (global over the batch: H W pixels per image, OH = H, OW = W).  So ONE staged
// run of 258 pixels x 32 channels per (channel chunk, kh) serves three K tiles -- activation DMA 8 -> 3 pieces per wave and
// tile, L2 -> LDS activation bytes / 3 -- and a fragment of tile kw reads LDS row r + kw.  What the per-tap tiles got from
// the zero page at DMA time (padding) is decided at READ time instead: a lane whose (output row, tap) lies outside the
// image reads a 256 B zero row in LDS (ds_read takes per-lane addresses).  Rows of the run outside the tensor read the
// zero page (first / last image).  The K loop is unrolled over the three tiles of a run, so the tile's kw, its vmcnt and
// its share of the next run's DMA (5 + 4 + 0 pieces) are compile-time.  LDS: 2 run buffers x 36 KiB (288 rows: 36 pieces)
// + 2 weight stages + the zero row.  Results are bit-identical to the per-tap kernels (same K order, same operands).
template <int BN, bool APAIR = false>
__global__ __launch_bounds__(256, 1) void conv_x3_run3_kernel(const ConvArgs p, const void* zero_page, int tiles_m,
                                                             int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 256, BKB = 128, BK = 32, NW = 4;
    constexpr int WTM = 64, WTN = BN, TM = 2, TN = BN / 32;
    constexpr int RUN_ROWS = 288, RUN_BYTES = RUN_ROWS * BKB, B_BYTES = BN * BKB;
    constexpr int NIR = RUN_ROWS * 8 / 64 / NW;        // 9 DMA pieces per wave per run
    constexpr int NIB = BN * 8 / 64 / NW;              // weight pieces per wave per tile (8 / 4)
    constexpr int MG = 3 * TM, NGR = TN / TM, WIN = NGR * MG, DPB = NIB / TN;
    static_assert(NIR == 9 && (TN == 8 || TN == 4) && WIN >= 12 && DPB == 1, "256 x 256 or 256 x 128 tile, four 64-row waves");
    constexpr unsigned Z_OFF = 2u * RUN_BYTES + 2u * B_BYTES;          // the zero row

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave;
    const int Mlim = p.M;
    const int nblk = tiles_m * tiles_n;
    if ((int)blockIdx.x >= nblk) return;
    int L;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nblk >> 3, r = nblk & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tile_n = L % tiles_n, tile_m = L / tiles_n;
    const int m0 = p.m_begin + tile_m * BM, n0 = tile_n * BN;
    if (m0 >= Mlim) return;

    const float* __restrict__ in = reinterpret_cast<const float*>(p.in);
    const float* __restrict__ wgt = reinterpret_cast<const float*>(p.weight);
    const float* zp = reinterpret_cast<const float*>(zero_page);
    auto swz = [](int row) { return (row >> 1) & 7; };
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned ldsB = lds_base + 2u * RUN_BYTES;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);

    // ---- run DMA slots: piece (wave + 4 j), j = 0..8, = run rows 8 (wave + 4 j) .. + 7; run row q holds the pixel with global
    // linear index m0 - W - 1 + kh W + q.  Slot pointer = that pixel for kh = 0, channel 0 (+ the lane's swizzled 16 B chunk);
    // a_mask bit kh: the pixel lies inside the tensor and the row inside the 258 used rows.
    const long long npix = (long long)p.N * p.H * p.W;
    const float* a_ptr[NIR];
    unsigned a_mask[NIR];
#pragma unroll
    for (int j = 0; j < NIR; ++j) {
        const int g = (wave + NW * j) * 64 + lane;
        const int q = g >> 3, pos = g & 7;
        const long long G0 = (long long)m0 - p.W - 1 + q;
        a_ptr[j] = in + p.in_coff + G0 * p.in_cstride + (pos ^ swz(q)) * 4;
        unsigned mk = 0;
        for (int kh = 0; kh < p.KH; ++kh) {
            const long long G = G0 + (long long)kh * p.W;
            if (q < BM + 2 && G >= 0 && G < npix) mk |= 1u << kh;
        }
        a_mask[j] = mk;
    }
    const float* b_ptr0;
    {
        const int g = wave * 64 + lane;
        const int row = g >> 3, pos = g & 7;
        b_ptr0 = wgt + (long long)(n0 + row) * p.K + (pos ^ swz(row)) * 4;
    }
    const long long b_jstride = (long long)(NW * 8) * p.K;

    // ---- per fragment row: which of the KH x 3 taps lie inside the image (padding is applied when the fragment is read)
    unsigned f_mask[TM];
    unsigned fa_run[3][2][TM], fb_pre[2][TN];
    const unsigned hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 32 + (lane & 31);
        const int m = m0 + row;
        unsigned mk = 0;
        if (m < Mlim) {
            const int n = m / (p.OH * p.OW);
            const int r = m - n * (p.OH * p.OW);
            const int oh = r / p.OW, ow = r - oh * p.OW;
            int tbit = 0;
            for (int kh = 0; kh < p.KH; ++kh)
                for (int kw = 0; kw < 3; ++kw, ++tbit) {
                    const int ih = oh + kh - 1, iw = ow + kw - 1;
                    if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) mk |= 1u << tbit;
                }
        }
        f_mask[i] = mk;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const int q = row + kw;
                fa_run[kw][kc][i] = q * BKB + (((4u * kc + (APAIR ? hi : 2u * hi)) ^ swz(q)) << 4);
            }
    }
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = j * 32 + (lane & 31);
            fb_pre[kc][j] = row * BKB + (((4u * kc + hi) ^ swz(row)) << 4);
        }

    const int nk = p.K / BK, nruns = nk / 3;
    const int ntaps = p.KH * 3;
    // run walker: kh inner, channel chunk outer.  off = element offset added to the slot pointers
    const long long r_d1 = (long long)p.W * p.in_cstride, r_e2 = BK - (long long)p.KH * p.W * p.in_cstride;
    int r_kh = 0, r_rem = nruns;
    long long r_off = 0;
    unsigned r_st = lds_base;                        // buffer of the run being issued
    auto run_walk = [&]() {
        const int k1 = r_kh + 1;
        const bool w = k1 == p.KH;
        r_kh = w ? 0 : k1;
        r_off += r_d1 + (w ? r_e2 : 0ll);
        r_rem -= 1;
        r_st = r_st == lds_base ? lds_base + RUN_BYTES : lds_base;
    };
    auto run_emit = [&](int j) {
        const int bit = r_rem > 0 ? r_kh : 31;
        const float* src = ((a_mask[j] >> bit) & 1u) ? a_ptr[j] + r_off : zp;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(uintptr_t)(r_st + (unsigned)(wave_s + NW * j) * 1024u), 16, 0, 0);
    };
    // weight walker: per tile, tap inner
    const long long b_d1 = p.Cin, b_e2 = (BK - (long long)(ntaps - 1) * p.Cin) - b_d1;
    int b_tap = 0, b_rem = nk;
    long long b_off = 0;
    unsigned b_st = ldsB;
    auto b_walk1 = [&]() {
        const int tp1 = b_tap + 1;
        const bool adv = b_rem > 1;
        const bool w = tp1 == ntaps;
        b_off += (adv ? b_d1 : 0ll) + ((adv && w) ? b_e2 : 0ll);
        b_tap = adv ? (w ? 0 : tp1) : b_tap;
    };
    auto b_walk2 = [&]() {
        b_rem -= 1;
        b_st = b_st == ldsB ? ldsB + B_BYTES : ldsB;
    };
    auto b_emit = [&](int j) {
        const float* src = b_ptr0 + b_off + (long long)j * b_jstride;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(uintptr_t)(b_st + (unsigned)(wave_s + NW * j) * 1024u), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto lds_read = [](unsigned addr) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
        return v;
    };
    auto split_a = [](float x0, float x1, uint32_t& h, float& t0, float& t1) {
        h = pack_bf16x2(x0, x1);
        t0 = __uint_as_float(h << 16);
        t1 = __uint_as_float(h & 0xffff0000u);
    };
    auto split_b = [](float x0, float x1, float t0, float t1) { return pack_bf16x2(x0 - t0, x1 - t1); };

    u32x4 ah[2][TM], al[2][TM], bh[TN], bl[TN], ra0, ra1;
    const unsigned zaddr = lds_base + Z_OFF;

    // ---- prologue: zero row, run 0, weight tiles 0 and 1; tile 0's fragments
    if (tid < 16) *reinterpret_cast<float4*>(smem + Z_OFF + tid * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    {
#pragma unroll
        for (int j = 0; j < NIR; ++j) run_emit(j);
        run_walk();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int j = 0; j < NIB; ++j) b_emit(j);
            b_walk1(); b_walk2();
        }
        if constexpr (NIB == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // only weight tile 1 outstanding
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned a0 = ((f_mask[i] >> 0) & 1u) ? lds_base + fa_run[0][0][i] : zaddr;
            if constexpr (APAIR) {
                ah[0][i] = lds_read(a0);
                al[0][i] = lds_read(a0 ^ 32u);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(ah[0][i]), "+v"(al[0][i]));
                continue;
            }
            ra0 = lds_read(a0);
            ra1 = lds_read(a0 ^ 16u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(ra0), "+v"(ra1));
            const float x[8] = {__uint_as_float(ra0.x), __uint_as_float(ra0.y), __uint_as_float(ra0.z), __uint_as_float(ra0.w),
                                __uint_as_float(ra1.x), __uint_as_float(ra1.y), __uint_as_float(ra1.z), __uint_as_float(ra1.w)};
            uint32_t h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t0, t1;
                split_a(x[2 * e], x[2 * e + 1], h[e], t0, t1);
                l[e] = split_b(x[2 * e], x[2 * e + 1], t0, t1);
            }
            ah[0][i] = u32x4{h[0], h[1], h[2], h[3]};
            al[0][i] = u32x4{l[0], l[1], l[2], l[3]};
        }
#pragma unroll
        for (int j = 0; j < TN - 1; ++j) {
            bh[j] = lds_read(ldsB + fb_pre[0][j]);
            bl[j] = lds_read(ldsB + (fb_pre[0][j] ^ 32u));
        }
        bl[TN - 1] = lds_read(ldsB + (fb_pre[0][TN - 1] ^ 32u));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bh[j]), "+v"(bl[j]));
    }

    unsigned sR_cur = lds_base, sR_nxt = lds_base + RUN_BYTES;        // run buffers of the current / next run
    unsigned sB_cur = ldsB, sB_nxt = ldsB + B_BYTES;
    int tap0 = 0;                                                      // tap index of the current run's kw = 0 tile
    for (int rr = 0; rr < nruns; ++rr) {
        const int tap0_nxt = (tap0 + 3 == ntaps) ? 0 : tap0 + 3;
#pragma unroll
        for (int t = 0; t < 3; ++t) {                                  // tile kw = t of the run
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const int kn = kc ^ 1;
                // the K step being prepared: (this tile, kc = 1), or the next tile's kc = 0 -- kw + 1 of this run or kw = 0 of the next
                const int nkw = kc == 0 ? t : (t == 2 ? 0 : t + 1);
                const unsigned srcA = (kc == 1 && t == 2) ? sR_nxt : sR_cur;
                const int ntap = kc == 0 ? tap0 + t : (t == 2 ? tap0_nxt : tap0 + t + 1);
                const unsigned srcB = kc == 0 ? sB_cur : sB_nxt;
                // DMA of this K step: kc = 0 -> the next run's pieces (5 with the first tile, 4 with the second); kc = 1 -> the
                // weights of tile + 2
                const int a_first = t == 0 ? 0 : 5, a_count = t == 0 ? 5 : (t == 1 ? 4 : 0);
                auto dma = [&](int q) {
                    if (TT_PIPE_DEBUG & 1) return;
                    asm volatile("" ::: "memory");
                    if (kc == 0) run_emit(a_first + q);
                    else b_emit(q);
                    asm volatile("" ::: "memory");
                };
                uint32_t sh[4], sl[4];
                float st0 = 0.f, st1 = 0.f;
                auto split_stage = [&](int s) {
                    const int e = s >> 1;
                    asm volatile("" : "+v"(ra0), "+v"(ra1));
                    const float x0 = __uint_as_float(e == 0 ? ra0.x : e == 1 ? ra0.z : e == 2 ? ra1.x : ra1.z);
                    const float x1 = __uint_as_float(e == 0 ? ra0.y : e == 1 ? ra0.w : e == 2 ? ra1.y : ra1.w);
                    if ((s & 1) == 0) {
                        split_a(x0, x1, sh[e], st0, st1);
                        asm volatile("" : "+v"(sh[e]), "+v"(st0), "+v"(st1));
                    } else {
                        asm volatile("" : "+v"(st0), "+v"(st1));
                        sl[e] = split_b(x0, x1, st0, st1);
                        asm volatile("" : "+v"(sl[e]));
                    }
                };
#pragma unroll
                for (int g = 0; g < TN; ++g) {
                    const int rb = g / NGR;
                    const bool barw = (kc == 1 && rb == 0);
                    const bool bar = barw && (g % NGR) == 0;
                    const bool last = g == TN - 1;
                    constexpr int R0 = 1, W0 = 4, RB = 4, WB = 7;
#pragma unroll
                    for (int m = 0; m < MG; ++m) {
                        const int tt = m / TM, i = m % TM;
                        if (tt == 0) TT_MFMA(acc[i][g], al[kc][i], bh[g]);
                        else if (tt == 1) TT_MFMA(acc[i][g], ah[kc][i], bl[g]);
                        else TT_MFMA(acc[i][g], ah[kc][i], bh[g]);
                        const int ws = (g % NGR) * MG + m;
                        if (m == 0) {
                            if (g == 0) bh[TN - 1] = lds_read(sB_cur + fb_pre[kc][TN - 1]);
                            else bh[g - 1] = lds_read(srcB + fb_pre[kn][g - 1]);
                        }
                        if (bar && m == 3) {
                            // the next tile's operands have landed for this wave: its weights, and -- third tile of a run --
                            // the next run (issued with the first two tiles).  Younger loads = this tile's share of the run
                            if (t == 0) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                            else if (t == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                        }
                        if (ws == (barw ? RB : R0)) {
                            // padding: a lane whose (row, tap) is outside the image reads the zero row
                            const unsigned a0 = ((f_mask[rb] >> ntap) & 1u) ? srcA + fa_run[nkw][kn][rb] : zaddr;
                            if constexpr (APAIR) {
                                ah[kn][rb] = lds_read(a0);
                                al[kn][rb] = lds_read(a0 ^ 32u);
                            } else {
                                ra0 = lds_read(a0);
                                ra1 = lds_read(a0 ^ 16u);
                            }
                        }
                        {
                            // DMA slots: gaps 2 and 3 of a group (4 and 5 behind the barrier).  Weights: one piece per group;
                            // run pieces: spread from group 0 on, two per group only where the count needs it (TN = 4)
                            const int s0 = bar ? (MG >= 8 ? 5 : 4) : 2;
                            if (kc == 1) {
                                if (m == s0) dma(g);
                            } else if (a_count > 0) {
                                const int per = (a_count + TN - 1) / TN;                    // 1 (TN = 8) or 2 (TN = 4, five pieces)
                                const int q0 = g * per;
                                if (m == s0 && q0 < a_count) dma(q0);
                                if (per == 2 && m == s0 + 1 && q0 + 1 < a_count) dma(q0 + 1);
                            }
                        }
                        if (ws == (barw ? WB : W0)) {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            if constexpr (APAIR) asm volatile("" : "+v"(ah[kn][rb]), "+v"(al[kn][rb]));
                            else asm volatile("" : "+v"(ra0), "+v"(ra1));
                            if (g == 0) asm volatile("" : "+v"(bh[TN - 1]));
                        }
                        if (m == 2 * TM - 1) bl[g] = lds_read(srcB + (fb_pre[kn][g] ^ 32u));
                        if constexpr (!APAIR) {
                            const int first = barw ? WB : W0;
                            const int avail = WIN - first;
                            int lastslot;
                            if (avail >= 16) {
                                if (ws >= first && ((ws - first) & 1) == 0 && (ws - first) / 2 < 8) split_stage((ws - first) / 2);
                                lastslot = first + 14;
                            } else if (avail >= 8) {
                                if (ws >= first && ws - first < 8) split_stage(ws - first);
                                lastslot = first + 7;
                            } else {
                                if (ws >= first && ws - first < 4) {
                                    split_stage(2 * (ws - first));
                                    split_stage(2 * (ws - first) + 1);
                                }
                                lastslot = first + 3;
                            }
                            if (ws == lastslot) {
                                ah[kn][rb] = u32x4{sh[0], sh[1], sh[2], sh[3]};
                                al[kn][rb] = u32x4{sl[0], sl[1], sl[2], sl[3]};
                                asm volatile("" : "+v"(ah[kn][rb]), "+v"(al[kn][rb]));
                            }
                        }
                        // walkers: the run walker after the run's last piece went out (second tile), the weight walker every tile
                        if (last && kc == 0 && t == 1 && m == MG - 2) run_walk();     // (a dozen SALU once per run: not pinned)
                        if (last && kc == 1) {
                            if (m == MG - 3) {
                                asm volatile("" : "+s"(b_tap), "+s"(b_off));
                                b_walk1();
                                asm volatile("" : "+s"(b_tap), "+s"(b_off));
                            }
                            if (m == MG - 2) {
                                asm volatile("" : "+s"(b_rem), "+s"(b_st));
                                b_walk2();
                                asm volatile("" : "+s"(b_rem), "+s"(b_st));
                            }
                            if (m == MG - 1) {
                                asm volatile("" : "+s"(sB_cur), "+s"(sB_nxt));
                                const unsigned tb = sB_cur;
                                sB_cur = sB_nxt;
                                sB_nxt = tb;
                                asm volatile("" : "+s"(sB_cur), "+s"(sB_nxt));
                            }
                        }
                    }
                }
            }
        }
        const unsigned tr = sR_cur;
        sR_cur = sR_nxt;
        sR_nxt = tr;
        tap0 = tap0_nxt;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    conv_epilogue<float, 1, TN, 32, WTN>(p, *reinterpret_cast<f32x16(*)[1][TN]>(&acc[0]), smem, wave, lane, wm * TM + 0, 0, m0, n0, Mlim);
    conv_epilogue<float, 1, TN, 32, WTN>(p, *reinterpret_cast<f32x16(*)[1][TN]>(&acc[1]), smem, wave, lane, wm * TM + 1, 0, m0, n0, Mlim);
#endif
}

static const void* pipe_zero_page() {
    static void* z = nullptr;
    if (!z) {
        if (hipMalloc(&z, 256) != hipSuccess) return nullptr;
        (void)hipMemset(z, 0, 256);
    }
    return z;
}

// Returns 1 if the launch was taken.  `m_tiles_limit` > 0: only that many row tiles from a.m_begin (tail split).
int try_launch_conv_x3_pipe(ConvArgs& a, hipStream_t st, int m_tiles_limit, int bn) {
    const bool apair = (a.flags & 32) != 0;          // pre-split activations (tt_conv_desc.in_pair)
    if (a.gather || a.m_dev || (bn != 256 && bn != 128) || a.Cout % bn != 0 || a.Cin % 32 != 0 || a.KH * a.KW > 31 || a.K < 64) return 0;
    const void* zp = pipe_zero_page();
    if (!zp) return 0;
    int tiles_m = div_up(a.M - a.m_begin, 256);
    if (m_tiles_limit > 0 && m_tiles_limit < tiles_m) tiles_m = m_tiles_limit;
    const int tiles_n = a.Cout / bn;
    // 3 x 3 stride-1 "same" convolutions over a dense batch: the run-staged form (one staged pixel run per filter row serves
    // its three taps).  TT_X3_RUN3=0 (test hook: tests/test_conv.py compares the two forms bit for bit): the per-tap form everywhere
    static const bool run3 = [] { const char* e = getenv("TT_X3_RUN3"); return e ? atoi(e) != 0 : true; }();
    if (run3 && a.KW == 3 && a.KH <= 5 && a.stride == 1 && a.dil == 1 && a.pad == 1 && a.OH == a.H && a.OW == a.W &&
        (a.N == 1 || a.in_nstride == (long long)a.H * a.W * a.in_cstride)) {
        const size_t smem_r = (size_t)2 * 288 * 128 + (size_t)2 * bn * 128 + 256;
        auto kr = bn == 128 ? (apair ? conv_x3_run3_kernel<128, true> : conv_x3_run3_kernel<128>)
                            : (apair ? conv_x3_run3_kernel<256, true> : conv_x3_run3_kernel<256>);
        static bool attr_r = false;
        if (!attr_r) {
            const int s128 = 2 * 288 * 128 + 2 * 128 * 128 + 256, s256 = 2 * 288 * 128 + 2 * 256 * 128 + 256;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_x3_run3_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, s128);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_x3_run3_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, s256);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_x3_run3_kernel<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, s128);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_x3_run3_kernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, s256);
            attr_r = true;
        }
        a.tiles_n = tiles_n;
        a.splits = 1;
        a.ws = nullptr;
        if (a.m_begin == 0)
            snprintf(g_conv_kernel, sizeof(g_conv_kernel), "conv_x3_run3_kernel<%d>%s%s", bn, apair ? " pre-split A" : "",
                     m_tiles_limit > 0 ? " + tail" : "");
        hipLaunchKernelGGL(kr, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), smem_r, st, a, zp, tiles_m, tiles_n);
        return 1;
    }
    // activation ring 3 x 32 KiB + weight ring 2 x (bn x 128 B); the epilogue stages 4 x 32 x (WTN + 4) floats
    size_t smem = (size_t)(3 * 256 + 2 * bn) * 128;
    const size_t epi = (size_t)4 * 32 * ((bn == 128 ? 128 : 256) + 4) * 4;
    if (smem < epi) smem = epi;
    // wave grid 4 x 1: 64 x 256 (64 x 128 on the 128-wide tile) per wave -- every activation fragment is split by ONE wave (the
    // 2 x 2 grid of 128 x 128 waves measured slower, profiles/r04_pipe_ab_grids.txt)
    auto kern = bn == 128 ? (apair ? conv_x3_pipe_kernel<4, 1, 128, true> : conv_x3_pipe_kernel<4, 1, 128>)
                          : (apair ? conv_x3_pipe_kernel<4, 1, 256, true> : conv_x3_pipe_kernel<4, 1>);
    static bool attr_set = false;
    if (!attr_set) {
        const int full = (3 * 256 + 2 * 256) * 128;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_x3_pipe_kernel<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, full);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_x3_pipe_kernel<4, 1, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, full);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_x3_pipe_kernel<4, 1, 256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, full);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_x3_pipe_kernel<4, 1, 128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, full);
        attr_set = true;
    }
    a.tiles_n = tiles_n;
    a.splits = 1;
    a.ws = nullptr;
    if (a.m_begin == 0)
        snprintf(g_conv_kernel, sizeof(g_conv_kernel), "conv_x3_pipe_kernel<%s>%s%s", bn == 128 ? "4, 1, 128" : "4, 1", apair ? " pre-split A" : "",
                 m_tiles_limit > 0 ? " + tail" : "");
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), smem, st, a, zp, tiles_m, tiles_n);
    return 1;
}

}  // namespace tt
