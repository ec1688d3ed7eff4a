// "h2" implicit-GEMM convolution: IEEE-half activations AS STORED x an f16 (hi, lo) weight pair, two MFMAs per product.
//
// The mixed precision / storage mode of the camera trunk (the PAFPN -- the one stage whose half storage the
// storage-level emulation clears, profiles/r06_precision_mix_storage.txt) keeps its activations in IEEE half in HBM.  A stored half is exact as an MFMA operand, so the only rounding left in a product is
// the weight's: it is carried as hi = f16(w), lo = f16(w - hi) (22 mantissa bits) and each product is
//     v_mfma_f32_32x32x16_f16(a, w_lo) + v_mfma_f32_32x32x16_f16(a, w_hi)          (f32 accumulation)
// -- two thirds of the bf16x3 form's matrix work, no operand split on the VALU, and half the activation / output bytes
// of f32 storage for the layers that sit at the HBM roof.
//
// Data movement is the LDS-DMA pipeline of conv_igemm_glds.hip (global_load_lds_dwordx4, counted vmcnt, raw s_barrier,
// XOR swizzle on the DMA source address, zero page for padding, XCD-aware tile order) with operand rows of different width:
//   * K tile = 64 halves.  Activation rows are 128 B (whole cache lines per DMA lane group), 256 rows per tile = 32 KiB;
//     weight rows are 256 B (per 16 K elements 64 B = [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15],
//     thinktwice_amd/weights.py::split_pairs_h2), BN rows per tile.
//   * swizzle: activation chunk c of row r sits at slot c ^ ((r >> 1) & 7); weight chunk c (16 per row) at c ^ (r & 15): the
//     16 lanes ds_read_b128 serves together touch 16 distinct 16 B slots of the 256 B bank row either way.
//   * ring: SA activation stages + SB weight stages (3 + 2 for the 128-wide tile = 160 KiB: the weights are L2-resident for
//     every workgroup of a launch, the activations are the stream that misses, so they get the second tile of lookahead);
//     a layer with fewer K tiles than stages only allocates what it uses (K = 64: 64 KiB, two workgroups per CU).
//   * K order: channel chunk outer, filter tap inner (the taps of a chunk re-read the same pixels: L2 hits).
// Epilogue: the shared conv_epilogue (folded BN, residual in half, ReLU, half or f32 output, optional second f32 copy).
// Contract (tt_conv2d_fwd dispatches here whenever tt_conv_desc.weight_h2 is set; anything else is refused there): dense
// convolution, Cin % 64 == 0, KH*KW <= 31.
#include <stdlib.h>

#include "conv_common.h"

namespace tt {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int BN, int WAVES_M, int WAVES_N, int SA, int SB>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, (WAVES_M * WAVES_N == 8) ? 2 : 1)
void conv_h2_kernel(const ConvArgs p, const void* zero_page, int tiles_m, int tiles_n, int sa_used, int sb_used) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 256, BK = 64;
    constexpr int A_ROWB = 128, B_ROWB = 256;
    constexpr int A_BYTES = BM * A_ROWB, B_BYTES = BN * B_ROWB;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int NIA = A_BYTES / 1024 / NW, NIB = B_BYTES / 1024 / NW;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NKC = 4;                                  // 16-wide K steps per tile
    static_assert(NIA >= 1 && NIB >= 1 && A_BYTES % (1024 * NW) == 0 && B_BYTES % (1024 * NW) == 0, "tile too small for the wave count");
    static_assert((SA == 2 && SB == 2) || (SA == 3 && SB == 2) || (SA == 3 && SB == 3), "ring shapes with a counted wait below");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int Mlim = p.M;

    // XCD-aware tile order (bijective for any grid size): hardware places block b on XCD b % 8
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

    const uint16_t* __restrict__ in = reinterpret_cast<const uint16_t*>(p.in);
    const uint16_t* __restrict__ wgt = reinterpret_cast<const uint16_t*>(p.weight);
    const uint16_t* zp = reinterpret_cast<const uint16_t*>(zero_page);

    // ---- DMA slots.  Activation slot j = 1 KiB piece (wave + NW j): 8 rows x 8 chunks; ONE pointer (tap (0,0), channel 0,
    // possibly outside the image) and ONE tap-validity mask per slot
    const uint16_t* a_ptr[NIA];
    unsigned a_mask[NIA];
#pragma unroll
    for (int j = 0; j < NIA; ++j) {
        const int g = (wave + NW * j) * 64 + lane;
        const int row = g >> 3, pos = g & 7;
        const int c = (pos ^ ((row >> 1) & 7)) * 8;
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
    // Weight slot j = piece (wave + NW j): 4 rows x 16 chunks of the [BN][256 B] tile; rows of 2 K halves in memory
    const uint16_t* b_ptr[NIB];
    bool b_ok[NIB];
#pragma unroll
    for (int j = 0; j < NIB; ++j) {
        const int g = (wave + NW * j) * 64 + lane;
        const int row = g >> 4, pos = g & 15;
        b_ok[j] = (n0 + row) < p.Cout;
        b_ptr[j] = wgt + (long long)(b_ok[j] ? n0 + row : 0) * (2ll * p.K) + (pos ^ (row & 15)) * 8;
    }

    const int nk = p.K / BK;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned ldsA = lds_base, ldsB = lds_base + (unsigned)sa_used * A_BYTES;
    (void)sb_used;

    struct KWalk {
        int kh, kw, ci;
    };
    KWalk wa{0, 0, 0}, wb{0, 0, 0};
    auto advance = [&](KWalk& w) {
        if (++w.kw == p.KW) {
            w.kw = 0;
            if (++w.kh == p.KH) {
                w.kh = 0;
                w.ci += BK;
            }
        }
    };
    struct DmaCtx {
        unsigned st;
        int tap;
        long long off;
        bool on;
    };
    auto a_begin = [&](int kt) {
        DmaCtx c{0u, 0, 0, kt < nk};
        if (!c.on) return c;
        c.st = ldsA + (unsigned)(kt % SA) * A_BYTES;
        c.tap = wa.kh * p.KW + wa.kw;
        c.off = ((long long)(wa.kh * p.dil) * p.W + wa.kw * p.dil) * p.in_cstride + wa.ci;
        advance(wa);
        return c;
    };
    auto a_emit = [&](const DmaCtx& c, int j) {
        const uint16_t* src = ((a_mask[j] >> c.tap) & 1u) ? a_ptr[j] + c.off : zp;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(uintptr_t)(c.st + (unsigned)(wave_s + NW * j) * 1024u), 16, 0, 0);
    };
    auto b_begin = [&](int kt) {
        DmaCtx c{0u, 0, 0, kt < nk};
        if (!c.on) return c;
        c.st = ldsB + (unsigned)(kt % SB) * B_BYTES;
        c.off = 2ll * ((long long)(wb.kh * p.KW + wb.kw) * p.Cin + wb.ci);
        advance(wb);
        return c;
    };
    auto b_emit = [&](const DmaCtx& c, int j) {
        const uint16_t* src = b_ok[j] ? b_ptr[j] + c.off : zp;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(uintptr_t)(c.st + (unsigned)(wave_s + NW * j) * 1024u), 16, 0, 0);
    };
    auto issue_a = [&](int kt) {
        const DmaCtx c = a_begin(kt);
        if (!c.on) return;
#pragma unroll
        for (int j = 0; j < NIA; ++j) a_emit(c, j);
    };
    auto issue_b = [&](int kt) {
        const DmaCtx c = b_begin(kt);
        if (!c.on) return;
#pragma unroll
        for (int j = 0; j < NIB; ++j) b_emit(c, j);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: issue order = the order of the loop below (weights of a tile before the activations issued with it)
    //   (2,2): A0 B0            (3,2): A0 B0 A1            (3,3): A0 B0 A1 B1
    issue_a(0);
    issue_b(0);
    if (SA == 3) issue_a(1);
    if (SB == 3) issue_b(1);

    // ---- fragment offsets inside a stage (ds_read_b128 per lane: row = lane & 31 of the block, K half = lane >> 5)
    const unsigned hi = lane >> 5;
    unsigned fa_pre[NKC][TM], fb_pre[NKC][TN];
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * WTM + i * 32 + (lane & 31);
            fa_pre[kc][i] = row * A_ROWB + (((2u * kc + hi) ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = wn * WTN + j * 32 + (lane & 31);
            fb_pre[kc][j] = row * B_ROWB + (((4u * kc + hi) ^ (row & 15)) << 4);     // hi half; the lo half is this ^ 32
        }
    }
    auto lds_read = [](unsigned addr) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
        return v;
    };

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed for THIS wave once only the loads issued after its last piece are outstanding
        if (SA == 3 && SB == 2 && kt + 1 < nk) {                       // in flight behind B(kt): A(kt+1)
            if constexpr (NIA == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (NIA == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (SA == 3 && SB == 3 && kt + 1 < nk) {                // behind A(kt): B(kt+1) A(kt+1)
            if constexpr (NIA + NIB == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (NIA + NIB == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr (NIA + NIB == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (NIA + NIB == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");       // publishes tile kt; everyone is done reading tile kt - 1

        const unsigned sa = ldsA + (unsigned)(kt % SA) * A_BYTES, sb = ldsB + (unsigned)(kt % SB) * B_BYTES;
        u32x4 fa[2][TM], fbh[2][TN], fbl[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = lds_read(sa + fa_pre[0][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            fbh[0][j] = lds_read(sb + fb_pre[0][j]);
            fbl[0][j] = lds_read(sb + (fb_pre[0][j] ^ 32u));
        }
        // the DMA this iteration owes the ring, one piece group behind each K step's MFMAs (all eight waves issuing a whole
        // tile at once queues the CU's one texture path: conv_igemm_glds.hip "spread")
        const DmaCtx cb = b_begin(kt + SB - 1);
        const DmaCtx ca = a_begin(kt + SA - 1);
        constexpr int PER = (NIA + NIB + NKC - 1) / NKC;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const int cur = kc & 1, nxt = cur ^ 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[cur][i]));
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fbh[cur][j]), "+v"(fbl[cur][j]));
#pragma unroll
            for (int q = kc * PER; q < (kc + 1) * PER && q < NIA + NIB; ++q) {
                if (q < NIB) { if (cb.on) b_emit(cb, q); }
                else if (ca.on) a_emit(ca, q - NIB);
            }
            if (kc + 1 < NKC) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[nxt][i] = lds_read(sa + fa_pre[kc + 1][i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    fbh[nxt][j] = lds_read(sb + fb_pre[kc + 1][j]);
                    fbl[nxt][j] = lds_read(sb + (fb_pre[kc + 1][j] ^ 32u));
                }
            }
            // term-major: consecutive MFMAs write different accumulators; the small term first
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    Mfma<f16_t>::run(__builtin_bit_cast(uint4, fa[cur][i]), __builtin_bit_cast(uint4, fbl[cur][j]), acc[i][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    Mfma<f16_t>::run(__builtin_bit_cast(uint4, fa[cur][i]), __builtin_bit_cast(uint4, fbh[cur][j]), acc[i][j]);
        }
    }
    __syncthreads();   // all waves done with the last stage before the epilogue reuses LDS
    conv_epilogue<f16_t, TM, TN, WTM, WTN, true>(p, acc, smem, wave, lane, wm, wn, m0, n0, Mlim);     // (with out2 / res1_f32)
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// The hand-pipelined form for the long-K layers (the PAFPN's 3 x 3 convolutions): 256 x 128 tile, FOUR waves of 64 x 128, one
// wave per SIMD, the instruction stream is the pipeline -- the schedule of conv_x3_pipe.hip with nothing to split.  Same
// operands, LDS image, K order and per-accumulator product order (lo term, hi term, K steps in sequence) as conv_h2_kernel:
// bit-identical results.  The compiler-scheduled eight-wave kernel keeps the matrix pipe 0.50 busy on these layers
// (profiles/r06_conv_sq_counters.txt: its two waves per SIMD read, issue DMA and then queue on the pipe in lock step).
//   * a K tile (64 halves) is four K steps of four groups (one per 32-wide column block) of four MFMAs (2 terms x 2 row blocks);
//   * behind the MFMAs of group g run, one per gap: the ds_read_b128 of the weight hi fragment the previous group has just
//     finished with, of this group's lo fragment (both for the NEXT K step), of the next K step's activation fragment of row
//     block g (g < 2), and one or two 1 KiB LDS-DMA pieces: the activations of tile kt + 2 behind K steps 0 and 1, its weights
//     behind K step 3; the walkers' scalar arithmetic sits in K step 2;
//   * fragment reads are waited for with COUNTED lgkmcnt at the head of each group (LDS returns in order; the counts are the
//     reads issued after the youngest one the group needs: 4, 6, 7, 7);
//   * one s_barrier per K tile, behind the second MFMA of K step 3: every read of tile kt has been issued by then (the last one,
//     the hi fragment of column block 3, one gap earlier), so `vmcnt(8) lgkmcnt(0)` + barrier publishes tile kt + 1 and frees
//     tile kt's stages half a K step before the first read of tile kt + 1;
//   * LDS ring: three activation stages + two weight stages of 32 KiB (160 KiB).
// Contract (dispatcher below): dense conv, Cin % 64 == 0, Cout % 128 == 0, KH*KW <= 31.
#ifndef TT_H2_DEBUG
#define TT_H2_DEBUG 0      // timing ablations (results wrong by design): bit 0 no DMA in the loop, bit 1 no fragment reads, bit 2 no barrier
#endif
__global__ __launch_bounds__(256, 1) void conv_h2_pipe_kernel(const ConvArgs p, const void* zero_page, int tiles_m, int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 256, BN = 128, BK = 64, NW = 4;
    constexpr int A_ROWB = 128, B_ROWB = 256;
    constexpr int A_BYTES = BM * A_ROWB, B_BYTES = BN * B_ROWB;
    constexpr int NIA = 8, NIB = 8;                        // 1 KiB DMA pieces per wave per tile
    constexpr int TM = 2, TN = 4, WTM = 64, WTN = 128, NKC = 4;

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

    const uint16_t* __restrict__ in = reinterpret_cast<const uint16_t*>(p.in);
    const uint16_t* __restrict__ wgt = reinterpret_cast<const uint16_t*>(p.weight);
    const uint16_t* zp = reinterpret_cast<const uint16_t*>(zero_page);

    // ---- DMA slots (conv_h2_kernel's): activation slot j = piece (wave + 4 j): 8 rows x 8 chunks; weight slot j = piece
    // (wave + 4 j): 4 rows x 16 chunks, rows 16 j apart: one pointer + a uniform stride
    const uint16_t* a_ptr[NIA];
    unsigned a_mask[NIA];
#pragma unroll
    for (int j = 0; j < NIA; ++j) {
        const int g = (wave + NW * j) * 64 + lane;
        const int row = g >> 3, pos = g & 7;
        const int c = (pos ^ ((row >> 1) & 7)) * 8;
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
    const uint16_t* b_ptr0;
    {
        const int g = wave * 64 + lane;
        const int row = g >> 4, pos = g & 15;
        b_ptr0 = wgt + (long long)(n0 + row) * (2ll * p.K) + (pos ^ (row & 15)) * 8;
    }
    const long long b_jstride = (long long)(NW * 4) * (2ll * p.K);      // halves between consecutive slots of a wave

    const int nk = p.K / BK;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned ldsA = lds_base, ldsB = lds_base + 3u * A_BYTES;

    // ---- the two DMA walkers (wave-uniform, branch-free: conv_x3_pipe.hip).  K order: channel chunk outer, filter tap inner
    const int ntaps = p.KH * p.KW;
    const long long a_d1 = (long long)p.dil * p.in_cstride;
    const long long a_d2 = ((long long)p.dil * p.W - (long long)(p.KW - 1) * p.dil) * p.in_cstride;
    const long long a_d3 = BK - ((long long)(p.KH - 1) * p.dil * p.W + (long long)(p.KW - 1) * p.dil) * p.in_cstride;
    const long long a_e2 = a_d2 - a_d1, a_e3 = a_d3 - a_d2;
    const long long b_d1 = 2ll * p.Cin;
    const long long b_e2 = 2ll * (BK - (long long)(ntaps - 1) * p.Cin) - b_d1;
    int a_kw = 0, a_tap = 0, a_rem = nk;
    long long a_off = 0;
    unsigned a_st = ldsA;
    int b_tap = 0, b_rem = nk, b_w = 0;
    long long b_off = 0;
    unsigned b_st = ldsB;
    long long a_d = 0;
    auto a_walk1 = [&]() {
        const int kw1 = a_kw + 1;
        const bool w1 = kw1 == p.KW;
        a_kw = w1 ? 0 : kw1;
        a_d = a_d1 + (w1 ? a_e2 : 0ll);
    };
    auto a_walk2 = [&]() {
        const int tp1 = a_tap + 1;
        const bool w2 = tp1 == ntaps;
        a_tap = w2 ? 0 : tp1;
        a_off += a_d + (w2 ? a_e3 : 0ll);
    };
    auto a_walk3 = [&]() {
        a_rem -= 1;
        a_st = a_st == ldsA + 2u * A_BYTES ? ldsA : a_st + A_BYTES;
    };
    auto b_walk1 = [&]() {     // beyond the last tile the walker stands still: the last tile again (valid memory) into a free stage
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
    auto a_emit = [&](int j) {      // beyond the last tile every activation row reads the zero page (bit 31 is never set)
        const int tapbit = a_rem > 0 ? a_tap : 31;
        const uint16_t* src = ((a_mask[j] >> tapbit) & 1u) ? a_ptr[j] + a_off : zp;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(uintptr_t)(a_st + (unsigned)(wave_s + NW * j) * 1024u), 16, 0, 0);
    };
    auto b_emit = [&](int j) {
        const uint16_t* src = b_ptr0 + b_off + (long long)j * b_jstride;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(uintptr_t)(b_st + (unsigned)(wave_s + NW * j) * 1024u), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned hi = lane >> 5;
    unsigned fa_pre[NKC][TM], fb_pre[NKC][TN];
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * WTM + i * 32 + (lane & 31);
            fa_pre[kc][i] = row * A_ROWB + (((2u * kc + hi) ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = j * 32 + (lane & 31);
            fb_pre[kc][j] = row * B_ROWB + (((4u * kc + hi) ^ (row & 15)) << 4);     // hi half; the lo half is this ^ 32
        }
    }
    auto lds_read = [](unsigned addr) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
        return v;
    };
#define TT_H2_MFMA(c, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))

    u32x4 fa[2][TM];                  // [K step parity][row block]
    u32x4 bh[TN], bl[TN];             // weight fragments of the current K step (refilled behind their last use)

    // ---- prologue: tiles 0 and 1 go out whole; tile 0's first fragments are read before the loop
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
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // tile 0 has landed for this wave
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = lds_read(ldsA + fa_pre[0][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (j < TN - 1) bh[j] = lds_read(ldsB + fb_pre[0][j]);       // column 3's hi half: first gap of the K step itself
            bl[j] = lds_read(ldsB + (fb_pre[0][j] ^ 32u));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[0][i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bh[j]), "+v"(bl[j]));
    }

#define LOOP_READ(addr) ((TT_H2_DEBUG & 2) ? u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u} : lds_read(addr))
    unsigned sA_cur = ldsA, sA_nxt = ldsA + A_BYTES, sB_cur = ldsB, sB_nxt = ldsB + B_BYTES;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int ks = 0; ks < NKC; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1, kn = (ks + 1) & 3;
            const unsigned srcA = ks < 3 ? sA_cur : sA_nxt;        // stage the next K step's operands lie in
            const unsigned srcB = ks < 3 ? sB_cur : sB_nxt;
            auto dma_a = [&](int j) {
                if (TT_H2_DEBUG & 1) return;
                asm volatile("" ::: "memory");
                a_emit(j);
                asm volatile("" ::: "memory");
            };
            auto dma_b = [&](int j) {
                if (TT_H2_DEBUG & 1) return;
                asm volatile("" ::: "memory");
                b_emit(j);
                asm volatile("" ::: "memory");
            };
#pragma unroll
            for (int g = 0; g < TN; ++g) {
                // the group's operands have landed: reads issued after the youngest of them (see the header)
                if (g == 0) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                else if (g == 1) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                if (g == 0) asm volatile("" : "+v"(fa[cur][0]), "+v"(fa[cur][1]));
                asm volatile("" : "+v"(bl[g]), "+v"(bh[g]));
                // ---- MFMA 0 + gap 0: hi half of the column block the previous group finished with (g = 0: column 3 of THIS K step)
                TT_H2_MFMA(acc[0][g], fa[cur][0], bl[g]);
                if (g == 0) bh[TN - 1] = LOOP_READ(sB_cur + fb_pre[ks][TN - 1]);
                else bh[g - 1] = LOOP_READ(srcB + fb_pre[kn][g - 1]);
                if (ks == 3 && g == 1) dma_b(2);
                if (ks == 3 && g == 2) dma_b(5);
                // ---- MFMA 1 + gap 1: the tile barrier (K step 3, group 0), then this group's lo half of the next K step
                TT_H2_MFMA(acc[1][g], fa[cur][1], bl[g]);
                if (ks == 3 && g == 0) { if (TT_H2_DEBUG & 4) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
                bl[g] = LOOP_READ(srcB + (fb_pre[kn][g] ^ 32u));
                // ---- MFMA 2 + gap 2: the next K step's activation fragment of row block g; a DMA piece
                TT_H2_MFMA(acc[0][g], fa[cur][0], bh[g]);
                if (g < TM) fa[nxt][g] = LOOP_READ(srcA + fa_pre[kn][g]);
                if (ks < 2) dma_a(4 * ks + g);
                if (ks == 3 && g == 0) dma_b(0);
                if (ks == 3 && g == 1) dma_b(3);
                if (ks == 3 && g == 2) dma_b(6);
                if (ks == 3 && g == 3) {
                    asm volatile("" : "+s"(b_tap), "+s"(b_off));
                    b_walk1();
                    asm volatile("" : "+s"(b_tap), "+s"(b_off));
                }
                // ---- MFMA 3 + gap 3
                TT_H2_MFMA(acc[1][g], fa[cur][1], bh[g]);
                if (ks == 3 && g == 0) dma_b(1);
                if (ks == 3 && g == 1) dma_b(4);
                if (ks == 3 && g == 2) dma_b(7);
                if (ks == 2 && g == 0) {
                    asm volatile("" : "+s"(a_kw));
                    a_walk1();
                    asm volatile("" : "+s"(a_kw), "+s"(a_d));
                }
                if (ks == 2 && g == 1) {
                    asm volatile("" : "+s"(a_tap), "+s"(a_d), "+s"(a_off));
                    a_walk2();
                    asm volatile("" : "+s"(a_tap), "+s"(a_off));
                }
                if (ks == 2 && g == 2) {
                    asm volatile("" : "+s"(a_rem), "+s"(a_st));
                    a_walk3();
                    asm volatile("" : "+s"(a_rem), "+s"(a_st));
                }
                if (ks == 3 && g == 3) {
                    asm volatile("" : "+s"(b_rem), "+s"(b_st), "+s"(sA_cur), "+s"(sA_nxt), "+s"(sB_cur), "+s"(sB_nxt));
                    b_walk2();
                    sA_cur = sA_nxt;
                    sA_nxt = sA_nxt == ldsA + 2u * A_BYTES ? ldsA : sA_nxt + A_BYTES;
                    const unsigned tb = sB_cur;
                    sB_cur = sB_nxt;
                    sB_nxt = tb;
                    asm volatile("" : "+s"(b_rem), "+s"(b_st), "+s"(sA_cur), "+s"(sA_nxt), "+s"(sB_cur), "+s"(sB_nxt));
                }
            }
        }
    }
#undef LOOP_READ
#undef TT_H2_MFMA
    // MFMA results -> any other reader: the hazard hipcc would pad for a builtin (8-pass XDL)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the DMA and fragment reads of the two tiles beyond the end
    __syncthreads();
    conv_epilogue<f16_t, 1, TN, 32, WTN, true>(p, *reinterpret_cast<f32x16(*)[1][TN]>(&acc[0]), smem, wave, lane, wm * TM + 0, 0, m0, n0, Mlim);
    conv_epilogue<f16_t, 1, TN, 32, WTN, true>(p, *reinterpret_cast<f32x16(*)[1][TN]>(&acc[1]), smem, wave, lane, wm * TM + 1, 0, m0, n0, Mlim);
#endif
}

static const void* h2_zero_page() {
    static void* z = nullptr;
    if (!z) {
        if (hipMalloc(&z, 256) != hipSuccess) return nullptr;
        (void)hipMemset(z, 0, 256);
    }
    return z;
}

template <int BN, int WAVES_M, int WAVES_N, int SA, int SB>
static int launch_h2(ConvArgs& a, hipStream_t st) {
    constexpr int BM = 256, A_BYTES = BM * 128, B_BYTES = BN * 256;
    constexpr int NW = WAVES_M * WAVES_N, WTN = BN / WAVES_N;
    const void* zp = h2_zero_page();
    if (!zp) return 0;
    const int tiles_m = div_up(a.M - a.m_begin, BM), tiles_n = div_up(a.Cout, BN);
    const int nk = a.K / 64;
    const int sa_used = nk < SA ? nk : SA, sb_used = nk < SB ? nk : SB;
    size_t smem = (size_t)sa_used * A_BYTES + (size_t)sb_used * B_BYTES;
    const size_t epi = (size_t)NW * 32 * (WTN + 4) * 4;
    if (smem < epi) smem = epi;
    auto kern = conv_h2_kernel<BN, WAVES_M, WAVES_N, SA, SB>;
    static bool attr_set = false;
    if (!attr_set) {
        size_t full = (size_t)SA * A_BYTES + (size_t)SB * B_BYTES;
        if (full < epi) full = epi;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)full);
        attr_set = true;
    }
    a.tiles_n = tiles_n;
    a.splits = 1;
    a.ws = nullptr;
    snprintf(g_conv_kernel, sizeof(g_conv_kernel), "conv_h2_kernel<%d, %d, %d, %d, %d>", BN, WAVES_M, WAVES_N, SA, SB);
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n)), dim3(NW * 64), smem, st, a, zp, tiles_m, tiles_n, sa_used, sb_used);
    return 1;
}

static int launch_h2_pipe(ConvArgs& a, hipStream_t st) {
    const void* zp = h2_zero_page();
    if (!zp) return 0;
    const int tiles_m = div_up(a.M - a.m_begin, 256), tiles_n = a.Cout / 128;
    const size_t smem = (size_t)5 * 32 * 1024;                       // 3 + 2 stages (the epilogue's 4 x 32 x 132 floats fit inside)
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    a.tiles_n = tiles_n;
    a.splits = 1;
    a.ws = nullptr;
    snprintf(g_conv_kernel, sizeof(g_conv_kernel), "conv_h2_pipe_kernel");
    hipLaunchKernelGGL(conv_h2_pipe_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), smem, st, a, zp, tiles_m, tiles_n);
    return 1;
}

int try_launch_conv_h2(ConvArgs& a, hipStream_t st) {
    if (a.gather || a.m_dev || a.ws || a.pixel_shuffle2 || a.Cin % 64 != 0 || a.KH * a.KW > 31 || a.K < 64) return 0;
    a.m_begin = 0;
    // long K, 128-wide column tiles: the hand-pipelined one-wave-per-SIMD kernel.  TT_H2_PIPE=0 (test hook: tests/test_conv.py
    // compares the two kernels bit for bit): the compiler-scheduled kernel everywhere
    static const bool pipe = [] { const char* e = getenv("TT_H2_PIPE"); return e ? atoi(e) != 0 : true; }();
    if (pipe && a.Cout % 128 == 0 && a.K >= 1152) return launch_h2_pipe(a, st);
    // 128-wide: 8 waves of 64 x 64 on a 3 + 2 ring (160 KiB); measured against four waves of 128 x 64 (+15 %) and a 2 + 2 ring (-0.5 %:
    // kept out, one variant less): profiles/r06_h2_microbench.txt.  64-wide: 8 waves of 32 x 64, 3 + 3 ring (four waves: +20 %)
    if (a.Cout > 64) return launch_h2<128, 4, 2, 3, 2>(a, st);
    return launch_h2<64, 8, 1, 3, 3>(a, st);
}

}  // namespace tt
