// Implicit-GEMM convolution, LDS-DMA pipelined kernel (gfx950 `global_load_lds_dwordx4`).
//
// Same math and epilogue as conv_igemm.hip; different data movement.  Carries every large dense layer of the
// camera trunk / SECOND / decoder value projections, the 7x7/2 stem in row-run form, and (GATHER) the sparse 3D
// convolutions of the LiDAR encoder:
//   * block tile 256 x {256, 128, 64, 32}, 4 or 8 waves with 64x64 or 128x64 register tiles (template parameters;
//     the dispatcher at the bottom of this file states which shape gets which and why).
//   * global -> LDS by DMA (no staging VGPRs).  STAGES = 3: tile k+2 is issued right after the barrier that
//     publishes tile k, with a COUNTED `s_waitcnt vmcnt(N)` (never 0 in the main loop) and a raw `s_barrier`, so
//     two K tiles of loads overlap the MFMA phase (64 B rows).  STAGES = 2: one tile in flight, 128 B rows =
//     whole cache lines per DMA lane group (the faster choice wherever 2 x tile fits the LDS budget).
//   * K order: channel chunk outer, filter tap inner (dense); natural [tap][channel] order, 1-4 taps per 128 B
//     row, rulebook entries fetched one tile ahead (gather).
//   * per DMA slot ONE precomputed pointer (tap (0,0)) and ONE tap-validity bitmask; per K tile a slot costs a
//     64-bit add of a wave-uniform offset, a bit test and a select.
//   * LDS rows are unpadded (the DMA writes wave-uniform base + lane*16); bank conflicts are removed
//     by an XOR swizzle applied to the SOURCE address: 16 B chunk c of row r is stored at chunk slot
//     c ^ f(r), f(r) = (r>>1)&7 for 128 B rows, (r>>2)&3 for 64 B rows -- every 16-lane
//     group of ds_read_b128 then touches 16 distinct slots of the 256 B bank row.
//   * rows that are padding / beyond M / beyond K read from a 16 B zero page instead of branching.
//   * XCD-aware tile order: workgroups that share an A row-block are consecutive on ONE XCD (its L2).
// Requires: Cin % BK == 0 (dense: one tap per K tile) or power-of-two Cin >= 16 (gather); KH*KW <= 32; no split-K.
#include <stdlib.h>

#include "conv_common.h"

namespace tt {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Timing experiments only (tools/conv_microbench.py, TT_MB_ACT=97/98 on the 2-stage variants): skip the weight /
// activation DMA after the first K tile to see how much of the loop time is the L2->LDS stream.  0 in the product.
#ifndef TT_GLDS_DEBUG
#define TT_GLDS_DEBUG 0
#endif

// X3 (T = float, 128 B rows): "bf16x3" arithmetic on f32 storage.  Activations stay f32 in HBM / LDS and are split
// into a bf16 (hi, lo) pair per element when a wave loads its A fragment; the weights arrive pre-split (per 16
// K elements 64 B = [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15], thinktwice_amd/weights.py::split_pairs_x3, same
// footprint as f32).  Each product is three v_mfma_f32_32x32x16_bf16: a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with f32
// accumulation -- 16 mantissa bits per operand (relative error ~1e-5 per dot product instead of bf16's 4e-3) at
// 16/3 of the exact-f32 MFMA rate.  This is the precision mode whose outputs meet the 1e-3 tolerance.
// APAIR (X3 only): the ACTIVATIONS arrive pre-split too -- the tensor holds, per 16 channels, 64 B = [hi c0-7 | hi c8-15 | lo c0-7 |
// lo c8-15] in bf16 (the weights' pair format along the channel axis, same footprint as f32), written by the producer's epilogue
// (tt_conv_desc.out_pair) or by an elementwise producer (tt_bilinear_up2_pair).  The hi / lo values are the ones split_frag computes,
// so the sums are bit-identical; what disappears is the split itself: 6 VALU per element pair per USE (a 3 x 3 layer splits every
// input element 9 x per column tile) against once per element at the producer.  For tensors whose ONLY readers are bf16x3 convolutions.
template <typename T, int BN, int WAVES_M, int WAVES_N, int BKB, int STAGES = 3, bool GATHER = false, bool X3 = false, bool APAIR = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64,
                             ((256 / WAVES_M / 32) * (BN / WAVES_N / 32) >= 16)  ? 1      // 128x128 per wave: 512 regs
                             : ((256 / WAVES_M / 32) * (BN / WAVES_N / 32) >= 8) ? 2      // 128x64 per wave: 256 regs
                             : (WAVES_M * WAVES_N == 16 || BKB == 64)            ? 4
                                                                                 : 2)
void conv_igemm_glds_kernel(const ConvArgs p, const void* zero_page,
                                                              int tiles_m, int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)   // amdgcn builtins / inline asm: keep the x86 host pass away from the body
    constexpr int BM = 256;
    constexpr int VEC = Elem<T>::kVec;
    constexpr int BK = BKB / (int)sizeof(T);          // BKB = K bytes per row per tile (64 or 128)
    constexpr int CPR = BKB / 16;                     // 16 B chunks per row
    // STAGES = 23: asymmetric ring -- THREE activation stages, TWO weight stages (x3 256x256 tile: 3 x 32 + 2 x 32 KiB =
    // the whole 160 KiB).  The weight tiles are L2-resident for every workgroup of the launch; the activation tiles are the
    // ones that miss (one tap in nine), so they get the second tile of lookahead.
    constexpr bool ASYM = STAGES == 23;
    constexpr int SA = ASYM ? 3 : STAGES, SB = ASYM ? 2 : STAGES;
    static_assert(STAGES == 2 || STAGES == 3 || ASYM, "2 or 3 LDS stages, or 23 = 3 activation + 2 weight stages");
    static_assert(!GATHER || STAGES == 2, "gather mode is written for the 2-stage pipeline");
    constexpr int A_BYTES = BM * BKB, B_BYTES = BN * BKB;
    constexpr int STAGE_BYTES = (BM + BN) * BKB;
    // byte offset of tile kt's activation / weight stage from the start of the dynamic LDS
    auto off_a = [](int kt) { return ASYM ? (kt % 3) * A_BYTES : (kt % SA) * STAGE_BYTES; };
    auto off_b = [](int kt) { return ASYM ? 3 * A_BYTES + (kt % 2) * B_BYTES : (kt % SB) * STAGE_BYTES + A_BYTES; };
    constexpr int NA_INSTR = BM * CPR / 64;           // 1 KiB wave-instructions in the A tile
    constexpr int NB_INSTR = BN * CPR / 64;
    constexpr int NW = WAVES_M * WAVES_N;             // waves per workgroup (8 or 16)
    constexpr int NIA = NA_INSTR / NW;                // A wave-instructions per wave per tile
    constexpr int NIB = (NB_INSTR + NW - 1) / NW;     // B  " (when NB_INSTR < NW some waves re-load a chunk
                                                      //       another wave also loads: same bytes, harmless,
                                                      //       and every wave keeps the same vmcnt arithmetic)
    constexpr int LPT = NIA + NIB;                    // DMA loads per thread per tile
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(!X3 || (sizeof(T) == 4 && BKB == 128), "x3: f32 storage, 128 B rows");
    static_assert(!APAIR || (X3 && !GATHER), "pre-split activations: dense bf16x3 only");
    static_assert(NW == 4 || NW == 8 || NW == 16, "4, 8 or 16 waves");
    static_assert(NA_INSTR % NW == 0 && NIA >= 1 && NIB >= 1, "tile too small for the wave count");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    int Mlim = p.M;
    if (GATHER && p.m_dev) {           // sparse conv: live output rows are only known on the device
        const int md = *p.m_dev;
        Mlim = md < Mlim ? md : Mlim;
    }
    // XCD-aware remap (bijective for any grid size): hardware places block b on XCD b % 8.  Sparse launches cover the
    // ALLOCATED rows: remap over the live row tiles only, or whole XCDs end up holding nothing but dead tiles
    const int live_m = (GATHER && p.m_dev) ? (Mlim - p.m_begin + BM - 1) / BM : tiles_m;
    const int nblk = (live_m < tiles_m ? (live_m > 0 ? live_m : 0) : tiles_m) * tiles_n;
    if ((int)blockIdx.x >= nblk) return;               // block-uniform, before any barrier
    int L;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nblk >> 3, r = nblk & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tile_n = L % tiles_n, tile_m = L / tiles_n;
    const int m0 = p.m_begin + tile_m * BM, n0 = tile_n * BN;
    if (m0 >= Mlim) return;            // block-uniform, before any barrier
    long long* const tr = (p.trace && tid == 0 && blockIdx.y == 0) ? p.trace + (long long)blockIdx.x * 4 : nullptr;
    if (tr) tr[0] = (long long)wall_clock64();

    const T* __restrict__ in = reinterpret_cast<const T*>(p.in);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(p.weight);
    const T* zp = reinterpret_cast<const T*>(zero_page);

    auto swz = [](int row) { return (BKB == 64) ? ((row >> 2) & 3) : ((row >> 1) & 7); };

    // ---- per-thread DMA slots: A slot j covers LDS chunk g = (wave + 8*j)*64 + lane of the A tile
    int a_h0[NIA], a_w0[NIA], a_c[NIA];
    long long a_base[NIA];
    bool a_ok[NIA];
#pragma unroll
    for (int j = 0; j < NIA; ++j) {
        const int g = (wave + NW * j) * 64 + lane;
        const int row = g / CPR, pos = g % CPR;
        a_c[j] = (pos ^ swz(row)) * VEC;               // element offset of the global chunk inside the K tile
        const int m = m0 + row;
        a_ok[j] = m < Mlim;
        const int mm = a_ok[j] ? m : 0;
        const int n = mm / (p.OH * p.OW);
        const int r = mm - n * (p.OH * p.OW);
        const int oh = r / p.OW, ow = r - oh * p.OW;
        a_h0[j] = oh * p.stride - p.pad;
        a_w0[j] = ow * p.stride - p.pad;
        // GATHER: rulebook row; with a tile plan the tile's slot `mm` stands for output row row_perm[mm]
        a_base[j] = GATHER ? (long long)((p.row_perm && a_ok[j]) ? p.row_perm[mm] : mm) * p.KW
                           : (long long)n * p.in_nstride + p.in_coff;
    }
    // Dense path: everything about a slot that does not change over the K loop is folded into ONE pointer (the
    // chunk's address for tap (0,0), possibly outside the image) and ONE bitmask (bit t = tap t of this row is
    // inside the image; KH*KW <= 32, dispatcher).  Per K tile a slot then costs a 64-bit add of a wave-uniform
    // tap offset, a bit test and a select -- the per-tile im2col arithmetic was ~2/3 of the kernel's VALU issue
    // (SQ_INSTS_VALU 5.6 per MFMA, profiles/r01_conv_sq_counters.txt).
    const T* a_ptr[NIA];
    unsigned a_mask[NIA];
    if (!GATHER) {
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            a_ptr[j] = in + a_base[j] + ((long long)a_h0[j] * p.W + a_w0[j]) * p.in_cstride + a_c[j];
            unsigned mk = 0;
            if (a_ok[j]) {
                int tbit = 0;
                for (int kh = 0; kh < p.KH; ++kh) {
                    const int ih = a_h0[j] + kh * p.dil;
                    for (int kw = 0; kw < p.KW; ++kw, ++tbit) {
                        const int iw = a_w0[j] + kw * p.dil;
                        if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) mk |= 1u << tbit;
                    }
                }
            }
            a_mask[j] = mk;
        }
    }
    int b_c[NIB];
    long long b_base[NIB];
    bool b_ok[NIB];
    const T* b_ptr[NIB];
#pragma unroll
    for (int j = 0; j < NIB; ++j) {
        const int g = ((wave + NW * j) % NB_INSTR) * 64 + lane;
        const int row = g / CPR, pos = g % CPR;
        b_c[j] = (pos ^ swz(row)) * VEC;
        b_ok[j] = (n0 + row) < p.Cout;
        b_base[j] = (long long)(n0 + row) * p.K;
        b_ptr[j] = wgt + b_base[j] + b_c[j];
    }

    int nk = (p.K + BK - 1) / BK;

    // K order: channel chunk OUTER, filter tap INNER (Cin % BK == 0: one tap per K tile).  The KH*KW taps of one
    // BK-channel chunk re-read the same (tile + halo) pixels back to back, so the re-reads hit the XCD's 4 MiB L2
    // (64 resident tiles x ~700 px x 128 B lines = ~2 MiB with the XCD-contiguous tile order).  With taps outer the
    // reuse distance was the whole channel extent (~8 MiB per XCD for Cin = 256) and 8 of 9 reads fell through to
    // the Infinity Cache.  Summation order differs from (tap, channel) only in f32 rounding.
    int it_kh = 0, it_kw = 0, it_ci = 0;               // running (kh, kw, ci) of the next tile to issue
    // GATHER (sparse conv) walks K in natural [tap][channel] order; a K tile covers BK/Cin taps when Cin < BK
    // (each 16 B chunk of a row then comes from its own tap's input row) or a BK-channel slice of one tap.
    // Cin is a power of two there (dispatcher): tap / channel by shift and mask, not integer division.
    int g_cur[NIA];                                    // rulebook entries of the next tile to issue
    int g_t = 0;                                       // K tile of the next rulebook fetch
    const int cin_shift = 31 - __clz(p.Cin), cin_mask = p.Cin - 1;
    // Tile plan (tt_sp_tile_plan): the rows of this tile were sorted by tap mask, so the tile only walks the UNION of
    // their taps.  K tile t of the compact walk covers tap(s) "the (t*TPT + s)-th set bit of the union" (TPT taps per
    // 128 B row when Cin < BK) or one BK-channel slice of one tap (Cin >= BK).  Both walkers below (rulebook fetch, one
    // tile ahead; DMA issue) pop the same bit sequence; everything is wave-uniform scalar work.
    constexpr int TPTMAX = 4;
    const bool planned = GATHER && p.row_mask != nullptr;
    unsigned umask = 0;
    if (planned) {
        unsigned* uw = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) *uw = 0u;
        __syncthreads();
        if (tid < BM && m0 + tid < Mlim) atomicOr(uw, p.row_mask[m0 + tid]);
        __syncthreads();
        umask = *uw;
        __syncthreads();                               // every wave has read it before the first DMA lands there
        umask = __builtin_amdgcn_readfirstlane(umask);
        const int nt = __popc(umask);
        nk = p.Cin >= BK ? nt * (p.Cin / BK) : (nt * p.Cin + BK - 1) / BK;
    }
    struct TapWalk {
        unsigned rem;
        int sub, cur;
    };
    TapWalk wf{umask, 0, -1}, wi{umask, 0, -1};
    auto next_taps = [&](TapWalk& w, int (&tp)[TPTMAX]) {
        if (p.Cin >= BK) {
            if (w.sub == 0) {
                w.cur = w.rem ? __builtin_ctz(w.rem) : -1;
                w.rem &= w.rem - 1;
            }
            tp[0] = w.cur;
            if (++w.sub == p.Cin / BK) w.sub = 0;
        } else {
            const int tpt = BK >> cin_shift;           // taps per K tile (<= TPTMAX, dispatcher)
#pragma unroll
            for (int i = 0; i < TPTMAX; ++i) {
                tp[i] = -1;
                if (i < tpt && w.rem) {
                    tp[i] = __builtin_ctz(w.rem);
                    w.rem &= w.rem - 1;
                }
            }
        }
    };
    auto pick = [&](const int (&tp)[TPTMAX], int slot) {    // tap of 16 B chunk `slot` (= element offset >> cin_shift)
        int t = tp[0];
#pragma unroll
        for (int i = 1; i < TPTMAX; ++i) t = (slot == i) ? tp[i] : t;
        return t;
    };
    auto fetch_rulebook = [&]() {
        if (planned) {
            int tp[TPTMAX];
            next_taps(wf, tp);
#pragma unroll
            for (int j = 0; j < NIA; ++j) {
                const int t = p.Cin >= BK ? tp[0] : pick(tp, a_c[j] >> cin_shift);
                g_cur[j] = (a_ok[j] && t >= 0) ? p.gather[a_base[j] + t] : -1;
            }
            ++g_t;
            return;
        }
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            const int k = g_t * BK + a_c[j];
            g_cur[j] = (a_ok[j] && k < p.K) ? p.gather[a_base[j] + (k >> cin_shift)] : -1;
        }
        ++g_t;
    };
    // The activation and the weight stream walk the same (channel chunk, tap) sequence; with the asymmetric ring the
    // activation walker runs one tile ahead of the weight walker, so each keeps its own position.
    // The LDS destination of a DMA instruction is wave-uniform (it goes to M0): keep it in scalar registers instead of
    // deriving it from threadIdx through a VALU add + v_readfirstlane per load.
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_dma_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    struct KWalk {
        int kh, kw, ci;
    };
    KWalk wa{0, 0, 0}, wb{0, 0, 0};
    if (!GATHER && p.splits > 1) {
        // split-K (few rows, long K: gridDim.y K ranges): this workgroup owns K tiles [kt0, kt1) of the (channel chunk outer, tap
        // inner) walk -- both walkers start there, the ring and the loop below count from 0; the epilogue stores the raw partial
        // sums into slice blockIdx.y of the workspace (conv_epilogue's p.ws branch), splitk_finalize_kernel adds the slices
        const int per = (nk + p.splits - 1) / p.splits;
        const int kt0 = (int)blockIdx.y * per;
        int kt1 = kt0 + per;
        kt1 = kt1 < nk ? kt1 : nk;
        const int taps = p.KH * p.KW;
        const int chunk = kt0 / taps, tap = kt0 - chunk * taps;
        wa.ci = wb.ci = chunk * BK;
        wa.kh = wb.kh = tap / p.KW;
        wa.kw = wb.kw = tap - (tap / p.KW) * p.KW;
        nk = kt1 > kt0 ? kt1 - kt0 : 0;
    }
    auto advance = [&](KWalk& w) {
        if (++w.kw == p.KW) {
            w.kw = 0;
            if (++w.kh == p.KH) {
                w.kh = 0;
                w.ci += BK;
            }
        }
    };
    auto issue_a = [&](int kt) {
        const unsigned st = lds_dma_base + (unsigned)off_a(kt);
        const int kh = wa.kh, kw = wa.kw, ci = wa.ci;
        const int k0 = kt * BK;                        // GATHER: position of this tile in the [KH][KW][Cin] weight row
        // wave-uniform (SALU): tap index and the element offset of tap (kh, kw), channel ci from tap (0, 0), channel 0
        const int tap = kh * p.KW + kw;
        const long long tap_off = ((long long)(kh * p.dil) * p.W + kw * p.dil) * p.in_cstride + ci;
        advance(wa);
        const bool dbg_skip_a = TT_GLDS_DEBUG && p.act == 98 && kt > 0;
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            if (dbg_skip_a) break;
            const T* src;
            if (GATHER) {
                // sparse conv: the activation row of (output row, tap) comes from the rulebook entry that was
                // fetched one iteration ago (g_cur), so the DMA does not sit behind a dependent index load
                const int gi = g_cur[j];
                const int kl = k0 + a_c[j];
                src = gi >= 0 ? in + (long long)gi * p.in_cstride + p.in_coff + (kl & cin_mask) : zp;
            } else {
                src = ((a_mask[j] >> tap) & 1u) ? a_ptr[j] + tap_off : zp;
            }
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(uintptr_t)(st + (unsigned)(wave_s + NW * j) * 1024u), 16, 0, 0);
        }
    };
    auto issue_b = [&](int kt) {
        const unsigned st = lds_dma_base + (unsigned)off_b(kt);
        const int k0 = GATHER ? kt * BK                // position of this tile in the [KH][KW][Cin] weight row
                              : (wb.kh * p.KW + wb.kw) * p.Cin + wb.ci;
        advance(wb);
        const bool dbg_skip_b = TT_GLDS_DEBUG && p.act == 97 && kt > 0;
        int tpi[TPTMAX];
        if (planned) next_taps(wi, tpi);
#pragma unroll
        for (int j = 0; j < NIB; ++j) {
            if (dbg_skip_b) break;
            // dense: K % BK == 0 (Cin % BK == 0); gather: the last K tile may be ragged (27 taps of 16/32 channels)
            bool ok = b_ok[j] && (!GATHER || (k0 + b_c[j] < p.K));
            const T* src = ok ? b_ptr[j] + k0 : zp;
            if (planned) {      // compact walk: weight row offset = tap * Cin + channel
                const int t = p.Cin >= BK ? tpi[0] : pick(tpi, b_c[j] >> cin_shift);
                ok = b_ok[j] && t >= 0;
                src = ok ? b_ptr[j] - b_c[j] + (long long)t * p.Cin + ((k0 + b_c[j]) & cin_mask) : zp;
            }
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(uintptr_t)(st + (unsigned)((wave_s + NW * j) % NB_INSTR) * 1024u),
                                             16, 0, 0);
        }
    };
    auto issue_tile = [&](int kt) {
        issue_a(kt);
        issue_b(kt);
    };
    // what goes out right after the barrier of tile kt
    auto issue_ahead = [&](int kt) {
        if (ASYM) {
            if (kt + 1 < nk) issue_b(kt + 1);          // weights first: the counted wait at the next tile's top lets the
            if (kt + 2 < nk) issue_a(kt + 2);          // younger activation loads stay in flight
        } else if (kt + STAGES - 1 < nk) {
            issue_tile(kt + STAGES - 1);
        }
    };

    // The same DMA, one instruction at a time (dense bf16x3 body): when all eight waves issue their 8 pieces of the next tile
    // together right after the barrier, the CU's one texture path (64 B/clk: 16 clk per 1 KiB piece) queues 64 pieces and every
    // wave sits ~1000 cycles in instruction issue before its first MFMA.  Spread over the sub-steps of the tile -- one piece
    // behind each group of MFMAs -- the queue never fills.  Order of issue is unchanged (weights of tile kt+1, then
    // activations of kt+2), so the counted waits at the top of the next tile still hold.
    struct DmaCtx {
        unsigned st;
        int tap, k0;
        long long tap_off;
        bool on;
    };
    auto a_begin = [&](int kt, bool on) {
        DmaCtx c{0u, 0, 0, 0, on};
        if (!on) return c;
        c.st = lds_dma_base + (unsigned)off_a(kt);
        c.tap = wa.kh * p.KW + wa.kw;
        c.tap_off = ((long long)(wa.kh * p.dil) * p.W + wa.kw * p.dil) * p.in_cstride + wa.ci;
        advance(wa);
        return c;
    };
    auto a_emit = [&](const DmaCtx& c, int j) {
        const T* src = ((a_mask[j] >> c.tap) & 1u) ? a_ptr[j] + c.tap_off : zp;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(uintptr_t)(c.st + (unsigned)(wave_s + NW * j) * 1024u), 16, 0, 0);
    };
    auto b_begin = [&](int kt, bool on) {
        DmaCtx c{0u, 0, 0, 0, on};
        if (!on) return c;
        c.st = lds_dma_base + (unsigned)off_b(kt);
        c.k0 = (wb.kh * p.KW + wb.kw) * p.Cin + wb.ci;
        advance(wb);
        return c;
    };
    auto b_emit = [&](const DmaCtx& c, int j) {
        const T* src = b_ok[j] ? b_ptr[j] + c.k0 : zp;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(uintptr_t)(c.st + (unsigned)((wave_s + NW * j) % NB_INSTR) * 1024u),
                                         16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (GATHER && nk > 0) fetch_rulebook();
    if (nk > 0) issue_tile(0);
    if (GATHER && nk > 1) fetch_rulebook();            // for tile 1; lands while tile 0 streams in
    if (STAGES == 3 && nk > 1) issue_tile(1);
    if (ASYM && nk > 1) issue_a(1);

    // fragment addressing: row = tile row + (lane&31); 16 B chunk c16 = 2*kc + (lane>>5), swizzled.
    // The fragment reads are INLINE ASM: hipcc treats every ds_read of this array as aliasing the
    // in-flight LDS-DMA and would insert `s_waitcnt vmcnt(0)` in front of it (draining the two tiles
    // of prefetch every iteration); asm reads are invisible to that pass, ordering is by the counted
    // vmcnt + barrier above (MI355X_MICROARCH.md "Two waves per SIMD" item 7).
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned fa_off[TM], fb_off[TN], fa_s[TM], fb_s[TN];
    const unsigned hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 32 + (lane & 31);
        fa_off[i] = row * BKB;
        fa_s[i] = swz(row);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * WTN + j * 32 + (lane & 31);
        fb_off[j] = BM * BKB + row * BKB;
        fb_s[j] = swz(row);
    }
    // swizzled fragment offsets inside a stage for every k-step: registers instead of 3 VALU per read per step
    // X3: a k-step is 16 f32 = 64 B.  A: lane half h owns floats 8h..8h+7 = chunks 4kc+2h and 4kc+2h+1 (the second is
    // the first with address bit 4 flipped, the swizzle being an XOR); B: hi chunk 4kc+h, lo chunk 4kc+2+h (bit 5).
    constexpr int NKC_ = X3 ? BKB / 64 : BKB / 32;
    // spread DMA issue (below): 256-wide tiles only -- a 64 x 128 wave tile has 6 MFMAs (192 cycles) per sub-step to put one
    // 16-cycle DMA piece behind; the narrow tiles' 3-MFMA sub-steps do not cover the texture path's time for 8 waves' pieces
    // (measured +3 % on them)
    const bool spread = BN == 256;
    unsigned fa_pre[NKC_][TM], fb_pre[NKC_][TN];
#pragma unroll
    for (int kc = 0; kc < NKC_; ++kc) {
        const unsigned ca = X3 ? (APAIR ? 4u * kc + hi : 4u * kc + 2u * hi) : 2u * kc + hi;    // APAIR: hi chunk; lo chunk = ^ 32
        const unsigned cb = X3 ? 4u * kc + hi : 2u * kc + hi;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa_pre[kc][i] = fa_off[i] + ((ca ^ fa_s[i]) << 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb_pre[kc][j] = fb_off[j] + ((cb ^ fb_s[j]) << 4);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto lds_read = [](unsigned addr) {
        u32x4 v;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
#else
        v = u32x4{addr, 0, 0, 0};   // host pass only parses this kernel
#endif
        return v;
    };

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed for THIS wave once at most one younger tile (LPT loads) is outstanding
#if defined(__HIP_DEVICE_COMPILE__)
        if (ASYM && kt + 1 < nk) {                     // outstanding in issue order: A(kt) B(kt) A(kt+1)
            if constexpr (NIA == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (NIA == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if constexpr (NIA == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (STAGES == 3 && kt + 1 < nk) {
            if constexpr (LPT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (LPT == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if constexpr (LPT == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (LPT == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if constexpr (LPT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (LPT == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");   // publishes tile kt; everyone is done reading tile kt-1
#endif
        if (tr && kt == 0) tr[1] = (long long)wall_clock64();

        // fragment offsets carry the activation / weight split of a symmetric stage (weights at + A_BYTES): the two bases
        // below make the same offsets address the asymmetric ring
        const unsigned sbase_a = lds_base + (unsigned)off_a(kt);
        const unsigned sbase_b = lds_base + (unsigned)(off_b(kt) - A_BYTES);
        const unsigned sbase = sbase_a;
        if constexpr (X3) {
            // bf16x3 body.  A k-step is 16 f32 of K: 2 reads per A fragment (raw f32), the split (6 VALU per element
            // pair, once per k-step, up front), then per output column block j: 2 reads (weights hi, lo) and 3 MFMAs per row
            // block.  The walk over (k-step, j) is software-pipelined inside the tile: the reads of sub-step s+1 are
            // issued before the MFMAs of sub-step s, into the registers sub-step s-1 has released (one B fragment pair
            // per buffer keeps the 64 x 128 wave tile inside 256 registers).
            constexpr int NS = NKC_ * TN;
            u32x4 ra0[TM], ra1[TM], bh[2], bl[2];
            uint4 ah[2][TM], al[2][TM];
            auto split_frag = [&](int i, uint4& hi_out, uint4& lo_out) {
                asm volatile("" : "+v"(ra0[i]));
                asm volatile("" : "+v"(ra1[i]));
                if (APAIR || (TT_GLDS_DEBUG && p.act == 96)) {   // pre-split activations: ra0 = hi, ra1 = lo (debug 96: raw bits)
                    hi_out = __builtin_bit_cast(uint4, ra0[i]);
                    lo_out = __builtin_bit_cast(uint4, ra1[i]);
                    return;
                }
                const float x[8] = {__uint_as_float(ra0[i].x), __uint_as_float(ra0[i].y), __uint_as_float(ra0[i].z),
                                    __uint_as_float(ra0[i].w), __uint_as_float(ra1[i].x), __uint_as_float(ra1[i].y),
                                    __uint_as_float(ra1[i].z), __uint_as_float(ra1[i].w)};
                uint32_t h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = pack_bf16x2(x[2 * e], x[2 * e + 1]);                       // round to nearest even
                    const float r0 = x[2 * e] - __uint_as_float(h[e] << 16);           // exact in f32
                    const float r1 = x[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u);
                    l[e] = pack_bf16x2(r0, r1);
                }
                hi_out = make_uint4(h[0], h[1], h[2], h[3]);
                lo_out = make_uint4(l[0], l[1], l[2], l[3]);
            };
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ra0[i] = lds_read(sbase + fa_pre[0][i]);
                ra1[i] = lds_read(sbase + (fa_pre[0][i] ^ (APAIR ? 32u : 16u)));
            }
            bh[0] = lds_read(sbase_b + fb_pre[0][0]);
            bl[0] = lds_read(sbase_b + (fb_pre[0][0] ^ 32u));
            constexpr bool SPREAD = !GATHER;
            constexpr int PER = (NIA + NIB + NS - 1) / NS;         // DMA pieces per sub-step
            DmaCtx ca{0u, 0, 0, 0, false}, cb{0u, 0, 0, 0, false};
            if constexpr (SPREAD) {
                if (spread) {
                    // first pieces issued: weights (ASYM: of tile kt+1; else of kt+STAGES-1, after its activations)
                    if (ASYM) {
                        cb = b_begin(kt + 1, kt + 1 < nk);
                        ca = a_begin(kt + 2, kt + 2 < nk);
                    } else {
                        ca = a_begin(kt + STAGES - 1, kt + STAGES - 1 < nk);
                        cb = b_begin(kt + STAGES - 1, kt + STAGES - 1 < nk);
                    }
                } else {
                    issue_ahead(kt);
                }
            } else {
                issue_ahead(kt);
                if (kt + 2 < nk) fetch_rulebook();
            }
#pragma unroll
            for (int ss = 0; ss < NS; ++ss) {
                const int kc = ss / TN, j = ss % TN, buf = ss & 1, ab = kc & 1;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(bh[buf]));
                asm volatile("" : "+v"(bl[buf]));
                if constexpr (SPREAD) {
                    if (spread) {
                        // this sub-step's share of the next tile's DMA (no LDS read is in flight here: the compiler's
                        // lgkmcnt(0) in front of a global_load_lds costs nothing)
#pragma unroll
                        for (int q = ss * PER; q < (ss + 1) * PER && q < NIA + NIB; ++q) {
                            if (ASYM) {
                                if (q < NIB) { if (cb.on) b_emit(cb, q); }
                                else if (ca.on) a_emit(ca, q - NIB);
                            } else {
                                if (q < NIA) { if (ca.on) a_emit(ca, q); }
                                else if (cb.on) b_emit(cb, q - NIA);
                            }
                        }
                    }
                }
                if (j == 0) {    // the k-step's activation fragments: split up front
#pragma unroll
                    for (int i = 0; i < TM; ++i) split_frag(i, ah[ab][i], al[ab][i]);
                }
                // reads under this sub-step's MFMAs: the next sub-step's weight pair, and the next k-step's activations
                if (j == TN - 1 && kc + 1 < NKC_) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        ra0[i] = lds_read(sbase + fa_pre[kc + 1][i]);
                        ra1[i] = lds_read(sbase + (fa_pre[kc + 1][i] ^ (APAIR ? 32u : 16u)));
                    }
                }
                if (ss + 1 < NS) {
                    const int kc2 = (ss + 1) / TN, j2 = (ss + 1) % TN;
                    if (!(TT_GLDS_DEBUG && p.act == 95)) {   // debug 95: weight fragments read once per tile
                        bh[buf ^ 1] = lds_read(sbase_b + fb_pre[kc2][j2]);
                        bl[buf ^ 1] = lds_read(sbase_b + (fb_pre[kc2][j2] ^ 32u));
                    } else {
                        bh[buf ^ 1] = bh[buf];
                        bl[buf ^ 1] = bl[buf];
                    }
                }
                const uint4 bhv = __builtin_bit_cast(uint4, bh[buf]);
                const uint4 blv = __builtin_bit_cast(uint4, bl[buf]);
                // term-major order: consecutive MFMAs write different accumulators (a back-to-back pair on the same
                // accumulator waits for the first one's last pass); small terms first
                if (!(TT_GLDS_DEBUG && p.act == 94)) {       // debug 94: one MFMA per fragment pair instead of three
#pragma unroll
                    for (int i = 0; i < TM; ++i) Mfma<uint16_t>::run(al[ab][i], bhv, acc[i][j]);
#pragma unroll
                    for (int i = 0; i < TM; ++i) Mfma<uint16_t>::run(ah[ab][i], blv, acc[i][j]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) Mfma<uint16_t>::run(ah[ab][i], bhv, acc[i][j]);
            }
            continue;
        }
        constexpr int NKC = BKB / 32;
        u32x4 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = lds_read(sbase + fa_pre[0][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = lds_read(sbase_b + fb_pre[0][j]);
        // the next tile's DMA goes out AFTER the first fragment reads: its address arithmetic (~90 VALU/SALU) then
        // runs under the LDS latency instead of in front of it (every wave of the block is in this phase together,
        // so nothing else would cover that latency)
        issue_ahead(kt);
        if (GATHER && kt + 2 < nk) fetch_rulebook();   // for tile kt+2, consumed at the top of the next iteration
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const int cur = kc & 1, nxt = cur ^ 1;
            // wait for the fragments of step kc; tie the wait to the registers the MFMAs read
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // volatile asms stay in order: these empty ones come after the wait, and the MFMAs below
            // consume their outputs, so no MFMA can be scheduled above the wait
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[cur][i]));
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fb[cur][j]));
#endif
            if (kc + 1 < NKC) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[nxt][i] = lds_read(sbase + fa_pre[kc + 1][i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[nxt][j] = lds_read(sbase_b + fb_pre[kc + 1][j]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const uint4 av = __builtin_bit_cast(uint4, fa[cur][i]);
                    const uint4 bv = __builtin_bit_cast(uint4, fb[cur][j]);
                    Mfma<T>::run(av, bv, acc[i][j]);
                }
        }
    }
    // Measured and NOT kept (profiles/r01_conv_microbench_tiles.txt): deferring each tile's last k-step past the next
    // barrier (to cover the prologue bubble) changed nothing, and skipping either DMA stream after the first tile
    // (TT_GLDS_DEBUG) did not shorten the loop either: at ~1.0 PF the loop is neither L2->LDS- nor bubble-bound.
    __syncthreads();   // all waves done with the last stage before the epilogue reuses LDS
    if (tr) tr[2] = (long long)wall_clock64();
    conv_epilogue<T, TM, TN, WTM, WTN>(p, acc, smem, wave, lane, wm, wn, m0, n0, Mlim);
    if (p.trace && blockIdx.y == 0) {                                               // block-uniform: every wave takes the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                            // (trace only) the slowest wave's epilogue
        if (tr) tr[3] = (long long)wall_clock64();
    }
#endif
}

static const void* zero_page() {
    static void* z = nullptr;
    if (!z) {
        if (hipMalloc(&z, 256) != hipSuccess) return nullptr;
        (void)hipMemset(z, 0, 256);
    }
    return z;
}

template <typename T, int BN, int WAVES_M, int WAVES_N, int BKB, int STAGES = 3, bool GATHER = false, bool X3 = false, bool APAIR = false>
static int launch_glds(ConvArgs& a, hipStream_t st, int m_tiles_limit = 0, int splits = 1, int slices = 1) {
    constexpr int BM = 256;
    constexpr int WTN = BN / WAVES_N;
    const void* zp = zero_page();
    if (!zp) return 0;
    // rows [m_begin, M) by default; `m_tiles_limit` > 0 restricts the launch to that many row tiles from m_begin
    int tiles_m = div_up(a.M - a.m_begin, BM);
    if (m_tiles_limit > 0 && m_tiles_limit < tiles_m) tiles_m = m_tiles_limit;
    const int tiles_n = div_up(a.Cout, BN);
    size_t smem = STAGES == 23 ? (size_t)(3 * BM + 2 * BN) * BKB : (size_t)STAGES * (BM + BN) * BKB;
    const size_t epi = (size_t)(WAVES_M * WAVES_N) * 32 * (WTN + 4) * 4;
    if (smem < epi) smem = epi;
    auto kern = conv_igemm_glds_kernel<T, BN, WAVES_M, WAVES_N, BKB, STAGES, GATHER, X3, APAIR>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)smem);
        attr_set = true;
    }
    a.tiles_n = tiles_n;
    a.splits = splits;
    if (splits <= 1) a.ws = nullptr;       // split-K: a.ws / a.ws_slices are the caller's (ordered slices, `slices` non-empty ranges)
    if (a.m_begin == 0)      // (the tail launch of a split keeps the main launch's label)
        snprintf(g_conv_kernel, sizeof(g_conv_kernel), "conv_igemm_glds_kernel<%s, %d, %d, %d, %d, %d, %s, %s>%s%s",
                 sizeof(T) == 4 ? "float" : "16-bit", BN, WAVES_M, WAVES_N, BKB, STAGES, GATHER ? "true" : "false",
                 X3 ? "true" : "false", APAIR ? " pre-split A" : "", splits > 1 ? " split-K" : (m_tiles_limit > 0 ? " + tail" : ""));
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n), (unsigned)(splits > 1 ? slices : 1)),
                       dim3(WAVES_M * WAVES_N * 64), smem, st, a, zp, tiles_m, tiles_n);
    return 1;
}

// Tail split of a 256x256-tile launch.  With T tiles on 256 CUs (one workgroup per CU: 128 KiB of LDS) the launch
// takes ceil(T / 256) rounds; when the last round is less than a quarter full (the 512 -> 512 DepthNet layers: 784
// tiles = 3 rounds + 16 tiles, i.e. 23 % of the launch spent on 2 % of the work) the row tiles of that remainder are
// peeled off into a second launch of 256x64 tiles (4x as many, 1/4 the work each) that fills the chip.
// Returns the number of row tiles the MAIN launch should cover (0: no split).
static int tail_split_rows(const ConvArgs& a) {
    if (a.Cout % 256 != 0) return 0;
    const int tiles_m = div_up(a.M, 256), tiles_n = a.Cout / 256;
    const long long T = (long long)tiles_m * tiles_n;
    const int rounds = (int)((T + kNumCU - 1) / kNumCU);
    const int last = (int)(T - (long long)kNumCU * (rounds - 1));
    if (rounds < 2 || rounds > 8 || last > kNumCU / 4) return 0;
    const int peel = div_up(last, tiles_n);            // row tiles moved to the tail launch
    return peel < tiles_m ? tiles_m - peel : 0;
}

// Split-K form of the 64-wide bf16x3 tile: few rows, long K (batch-1 ticks: ResNet layer 4 at M = 3,136, K = 2048 / 4608 --
// 13 row tiles are a twentieth of the chip, so those layers ran the exact-f32 register-staged kernel with a K split, 2.0 ms per
// tick).  Column blocks of 64 give tiles_m x Cout / 64 workgroups; the K tiles are dealt over up to 16 ranges of >= 8 tiles until
// ~512 workgroups (two per CU) are in flight.  Returns the number of non-empty K ranges (= workspace slices), 0 = not this path.
static int x3_splitk_plan(const ConvArgs& a, int* splits_out) {
    if (a.gather || a.m_dev || a.M < 512 || a.M > 8192 || a.Cout < 64 || a.KH * a.KW > 32) return 0;
    if (a.Cin % 32 != 0 || a.K < 1024 || a.pixel_shuffle2) return 0;
    const int tiles = div_up(a.M, 256) * div_up(a.Cout, 64);
    const int nk = a.K / 32;
    if (tiles >= 256) return 0;
    int sp = 512 / tiles;
    if (sp > nk / 8) sp = nk / 8;
    if (sp > 16) sp = 16;
    if (sp < 2) return 0;
    const int per = div_up(nk, sp);
    if (splits_out) *splits_out = sp;
    return div_up(nk, per);
}

int conv_glds_x3_splitk_slices(const ConvArgs& a) { return x3_splitk_plan(a, nullptr); }

// a.weight = the pre-split weights, a.ws = the workspace of >= conv_glds_x3_splitk_slices(a) [M][Cout] f32 slices.  Launches the
// tiles only; the caller runs splitk_finalize_kernel (a.ws_slices is set to the slices written).
int launch_conv_glds_x3_splitk(ConvArgs& a, hipStream_t st) {
    int sp = 0;
    const int slices = x3_splitk_plan(a, &sp);
    if (slices < 2 || !a.ws || a.ws_slices < slices) return 0;
    a.ws_slices = slices;
    return launch_glds<float, 64, 8, 1, 128, 2, false, true>(a, st, 0, sp, slices);
}

// bf16x3 arithmetic on f32 storage (see the kernel's template comment).  `a.weight` must already point at the
// pre-split weights.  Returns 0 when the shape is outside the DMA kernel's contract (the caller then runs the exact
// f32 path on the plain weights).
int try_launch_conv_glds_x3(ConvArgs& a, hipStream_t st) {
    if (a.gather) {
        if (try_launch_sp_conv_runs(a, st)) return 1;      // 3x3x3 rulebooks with 32+ channels: run-staged kernel
        const bool cin_ok = a.Cin >= 16 && (a.Cin & (a.Cin - 1)) == 0;
        if (!cin_ok || a.M < 2048 || a.Cout < 16 || a.Cout > 128) return 0;
        if (a.Cout <= 32) return launch_glds<float, 32, 8, 1, 128, 2, true, true>(a, st);
        if (a.Cout <= 64) return launch_glds<float, 64, 8, 1, 128, 2, true, true>(a, st);
        return launch_glds<float, 128, 8, 1, 128, 2, true, true>(a, st);
    }
    if (a.m_dev || a.M < 2048 || a.KH * a.KW > 32) return 0;
    if (a.Cin % 32 != 0 || a.K < 64) return 0;       // 128 B rows = 32 f32 of one tap per K tile, >= 2 tiles
    // pre-split activations (tt_conv_desc.in_pair): 16-channel pair groups must line up with the 32-channel K tiles
    const bool apair = (a.flags & 32) != 0;
    if (apair && (a.in_coff % 16 != 0 || a.in_cstride % 16 != 0)) return 0;
    // few output channels over many rows (the segmentation head: 3 x 3, 64 -> 12 at 224 x 448 per image; the deformable conv's
    // offset head: 3 x 3, 512 -> 18; seg_res_to_image_feature's 64 -> 16): a 256 x 32 tile, two workgroups per CU.  Below 2^16
    // rows the exact-f32 register-staged kernel keeps them
    if (a.Cout < 64) {
        if (a.Cout > 32 || a.Cout < 8) return 0;
        if (apair) return launch_glds<float, 32, 8, 1, 128, 2, false, true, true>(a, st);
        if (a.M < (1 << 16)) return 0;
        return launch_glds<float, 32, 8, 1, 128, 2, false, true>(a, st);
    }
    // Tile width along N.  The widest wave tile the layer allows is the most efficient per tile (the operand split costs
    // 8/TN VALU per MFMA; measured ~1.0 / 0.85 / 0.63 relative MFMA rate for the 256 / 128 / 64 wide tiles), but a
    // launch with fewer workgroups than the chip holds (batch-1 ticks: 49 row tiles x 2 on 256 CUs) is bound by its
    // rounds, not by the per-tile rate: pick the width with the smallest  rounds x (BN / rate).
    const int tiles_m = div_up(a.M, 256);
    auto cost = [&](int bn, double rate) {      // the busiest CU runs ceil(tiles / 256) tiles at the tile's measured rate
        const long long tiles = (long long)tiles_m * div_up(a.Cout, bn);
        return (double)((tiles + kNumCU - 1) / kNumCU) * bn / rate;
    };
    const bool wide = a.Cout % 256 == 0 || a.Cout > 512;
    // 256-wide with a tail split: the main launch's full rounds + the peeled row tiles as 256 x 64 tiles
    auto cost256 = [&]() {
        const int main_rows = tail_split_rows(a);
        if (!main_rows) return cost(256, 1.0);
        const long long tn = a.Cout / 256;
        const long long main_tiles = (long long)main_rows * tn, tail_tiles = (long long)(tiles_m - main_rows) * tn * 4;
        return (double)((main_tiles + kNumCU - 1) / kNumCU) * 256 / 1.0 + (double)((tail_tiles + kNumCU - 1) / kNumCU) * 64 / 0.63;
    };
    const double c256 = wide ? cost256() : 1e30;
    // (long-K layers run the 128-wide tile on the hand-pipelined kernel: 404 vs 423 TF/s for the 256-wide one, profiles/r04_run3_ab.txt)
    const double c128 = a.Cout > 64 ? cost(128, (a.K >= 1152 && a.Cout % 128 == 0) ? 0.95 : 0.85) : 1e30;
    const double c64 = cost(64, 0.63);
    const int bn = (c256 <= c128 && c256 <= c64) ? 256 : (c128 <= c64 ? 128 : 64);
    // Long-K layers (K >= 1152: every 3 x 3 of the trunks) on the 256- and 128-wide tiles: four hand-pipelined waves, one per SIMD
    // (csrc/conv_x3_pipe.hip: MFMA pipe 77 % busy against 58 %, profiles/r04_conv_sq_counters_noepilogue.txt).  Short K keeps the
    // 8-wave tile: there the tile's prologue + epilogue dominate and eight waves issue the output stores faster than four
    // (K = 1024: 0.203 vs 0.216 ms, K = 256 N = 1280: 0.52 vs 0.77 ms; profiles/r04_pipe_ab_first.txt).
    // TT_X3_PIPE=0 (test hook: tests/test_conv.py compares the two families bit for bit): the compiler-scheduled tiles everywhere
    static const bool pipe = [] { const char* e = getenv("TT_X3_PIPE"); return e ? atoi(e) != 0 : true; }();
    const bool hand = pipe && a.K >= 1152;
    if (bn == 256) {                                                                                           // 8 x (64 x 128)
        const int main_rows = tail_split_rows(a);
        if (hand && try_launch_conv_x3_pipe(a, st, main_rows)) { /* taken */ }
        else if (apair) launch_glds<float, 256, 4, 2, 128, 23, false, true, true>(a, st, main_rows);
        else launch_glds<float, 256, 4, 2, 128, 23, false, true>(a, st, main_rows);      // 3 activation + 2 weight stages = 160 KiB
        if (main_rows) {
            ConvArgs t = a;
            t.m_begin = main_rows * 256;
            return apair ? launch_glds<float, 64, 8, 1, 128, 2, false, true, true>(t, st)
                         : launch_glds<float, 64, 8, 1, 128, 2, false, true>(t, st);
        }
        return 1;
    }
    // Narrow tiles: two LDS stages (a third costs the 64-wide tile its second workgroup per CU: N=64 K=576 2.13 -> 2.66 ms), eight
    // waves (four waves of 64 x 64 on the 64-wide tile: 3-15 % slower, profiles/r05_x3_64wide_waves_ab.txt)
    if (bn == 128) {                                                                                           // 8 x (32 x 128)
        if (hand && a.Cout % 128 == 0 && try_launch_conv_x3_pipe(a, st, 0, 128)) return 1;
        return apair ? launch_glds<float, 128, 8, 1, 128, 2, false, true, true>(a, st)
                     : launch_glds<float, 128, 8, 1, 128, 2, false, true>(a, st);
    }
    // (pre-split activations: four waves of 64 x 64 measured 1-5 % slower than eight of 32 x 64 here too, profiles/r06_pair_format.txt)
    return apair ? launch_glds<float, 64, 8, 1, 128, 2, false, true, true>(a, st)                              // 8 x (32 x 64)
                 : launch_glds<float, 64, 8, 1, 128, 2, false, true>(a, st);
}

// T16 = uint16_t (bf16) or f16_t (IEEE half): same tiles, same MFMA rate.
template <typename T16>
static int launch_glds16(ConvArgs& a, int dtype, hipStream_t st, int min_tiles);

int try_launch_conv_glds(ConvArgs& a, int dtype, hipStream_t st) {
    constexpr int min_tiles = 2;      // K = 64 1x1 layers: 0.43 -> 0.27 ms against the register-staged kernel
    if (a.gather) {
        // sparse 3D conv as a gathered GEMM (rulebook rows): whole 128 B+ activation rows per DMA lane group
        const bool cin_ok = a.Cin >= 16 && (a.Cin & (a.Cin - 1)) == 0;   // power of two: taps tile the 128 B rows
        if (dtype == TT_F32 || !cin_ok || a.M < 2048 || a.Cout < 16 || a.Cout > 128) return 0;
        return dtype == TT_F16 ? launch_glds16<f16_t>(a, dtype, st, min_tiles) : launch_glds16<uint16_t>(a, dtype, st, min_tiles);
    }
    if (a.m_dev || a.M < 2048 || a.Cout < 64 || a.KH * a.KW > 32) return 0;
    if (dtype == TT_F32) {
        if (a.Cin % 16 != 0 || div_up(a.K, 16) < min_tiles) return 0;
        if (a.Cout > 64) return launch_glds<float, 128, 4, 2, 64>(a, st);
        return launch_glds<float, 64, 8, 1, 64>(a, st);
    }
    if (a.Cin % 32 != 0 || div_up(a.K, 32) < min_tiles) return 0;
    return dtype == TT_F16 ? launch_glds16<f16_t>(a, dtype, st, min_tiles) : launch_glds16<uint16_t>(a, dtype, st, min_tiles);
}

template <typename T16>
static int launch_glds16(ConvArgs& a, int dtype, hipStream_t st, int min_tiles) {
    (void)dtype; (void)min_tiles;
    if (a.gather) {
        if (a.Cout <= 32) return launch_glds<T16, 32, 8, 1, 128, 2, true>(a, st);
        if (a.Cout <= 64) return launch_glds<T16, 64, 8, 1, 128, 2, true>(a, st);
        return launch_glds<T16, 128, 4, 2, 128, 2, true>(a, st);
    }
    // Tile selection (profiles/r01_conv_microbench_tiles.txt).  Three things set the rate of these kernels:
    //  * L2->LDS bytes per FLOP = workgroup tile: 256x128 -> 85 FLOP/B, 256x256 -> 128 FLOP/B;
    //  * whether a DMA lane group consumes WHOLE 128 B cache lines: with 64 B rows every activation line is
    //    fetched twice from L2 (the other half is needed one K tile later and the 32 KiB L1 cannot hold a tile);
    //    128 B rows in 2 stages beat 64 B rows in 3 stages by 10-17 % on every layer whose LDS budget allows it;
    //  * LDS fragment bytes per MFMA = per-wave register tile (64x64: 1 KiB, 128x64: 0.75 KiB) -- second order.
    // Auto: Cout % 256 == 0 -> 256x256 tile of eight 128x64 waves (128 B rows if Cin % 64 == 0); short K -> four
    // 128x64 waves on 256x128; Cout <= 64 -> 256x64 tile with 128 B rows; else eight 64x64 waves on 256x128.
    // Retired after measurement (same file): 16-wave 256x256, 8x1 wave grid, 128 B rows x 3 stages (1 workgroup/CU),
    // 128 B x 2 stages on the 256x128 tile.
    if (a.Cout > 64) {
        int v;
        {
            const long long tiles256 = (long long)div_up(a.M, 256) * (a.Cout / 256);
            if (a.Cout % 256 == 0 && tiles256 >= 200) v = (a.Cin % 64 == 0) ? 6 : 2;
            else if (a.K <= 512) v = 1;
            else v = 0;
        }
        if (v == 6 && a.Cout % 256 == 0 && a.Cin % 64 == 0) {
            if (const int main_rows = tail_split_rows(a)) {
                launch_glds<T16, 256, 2, 4, 128, 2>(a, st, main_rows);
                ConvArgs t = a;
                t.m_begin = main_rows * 256;
                return launch_glds<T16, 64, 8, 1, 128, 2>(t, st);
            }
            return launch_glds<T16, 256, 2, 4, 128, 2>(a, st);
        }
        if (v == 1) return launch_glds<T16, 128, 2, 2, 64>(a, st);                         // 4 waves x 128x64
        if (v == 2 && a.Cout % 256 == 0) return launch_glds<T16, 256, 2, 4, 64>(a, st);    // 8 waves x 128x64
        return launch_glds<T16, 128, 4, 2, 64>(a, st);                                     // 8 waves x 64x64
    }
    // Cout <= 64 (the 224x448 UNet / stem-level layers): 128 B rows in 2 stages (80 KiB, 2 workgroups / CU)
    // measured +17 % over 64 B rows x 3 stages (1.61 vs 1.89 ms on M=6.4M K=1152)
    if (a.Cin % 64 == 0) return launch_glds<T16, 64, 8, 1, 128, 2>(a, st);
    return launch_glds<T16, 64, 8, 1, 64>(a, st);
}

}  // namespace tt
