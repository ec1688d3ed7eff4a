// Sparse 3-D convolution (spconv SubMConv3d / SparseConv3d, 3x3x3) as a RUN-STAGED gathered GEMM, bf16x3 arithmetic.
//
// Replaces, for the 32/64/128-channel levels of SparseEncoder_fp32 (backbones/lidarnet.py:41-53, configs/thinktwice.py
// :167-176), the per-(row, tap) gather of conv_igemm_glds_kernel<GATHER>: that kernel DMA'd one input row per output
// row and tap -- 27 x C x 4 B per output row through L2, i.e. every input row ~19x (profiles/r02_forward_bf16x3_pmc.json:
// 49 GB read per step for 5 GB of input rows) -- and was bound by that stream (4.3 TB/s of gather at 97 TF/s).
//
// Here the rows of a level are in cell order (b, z, y, x), x fastest (csrc/lidar.hip), so for a tile of TM consecutive
// output rows the neighbours of one (dz, dy) pair -- the three dx taps -- lie in ONE short contiguous range of input
// rows.  Per (tile, dz, dy) group the kernel
//   1. DMAs that row range ONCE into LDS as plain coalesced 128 B lines (32-channel slices, `global_load_lds_dwordx4`,
//      double buffered) together with the group's three weight taps,
//   2. lets every MFMA lane fetch its A operand from the LDS row named by ITS OWN rulebook entry (ds_read_b128 takes
//      per-lane addresses, so the gather is free once the range is staged); absent neighbours read a zero row,
//   3. skips a tap whose 32 rows of a wave have no neighbour at all (wave-uniform test).
// L2 -> LDS traffic drops from 27 to ~9 input rows per output row.  Correctness does not depend on the row order: the
// range of a group is the min / max of its present rulebook entries, and a range longer than the LDS stage is walked
// in chunks (entries outside the current chunk read the zero row), so any rulebook gives the same sums -- only the
// speed relies on neighbours being close.
//
// Arithmetic: bf16x3 on f32 storage exactly as conv_igemm_glds_kernel<X3> (activations split into bf16 hi + lo in
// registers, pre-split pair-format weights, a_lo*b_hi + a_hi*b_lo + a_hi*b_hi into an f32 accumulator); epilogue =
// conv_epilogue (folded BN1d, residual, ReLU).  Summation order: (dz, dy) groups, chunks, 32-channel slices, dx taps.
#include <limits.h>
#include <stdlib.h>

#include "conv_common.h"

#ifndef TT_SP_DEBUG
#define TT_SP_DEBUG 0
#endif

namespace tt {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace {
constexpr int kG = 3;      // taps per group (the dx run)
constexpr int kNG = 9;     // groups (dz, dy)
constexpr int kRowB = 128; // staged bytes per row: a 32-channel f32 slice
}  // namespace

template <int NCB, int WR, int WC>
__global__ __launch_bounds__(WR * WC * 64, 1) void sp_conv_runs_kernel(const ConvArgs p, int tiles_m) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WR * WC;
    constexpr int NT = NW * 64;
    constexpr int TM = WR * 32;                 // output rows per tile (one 32-row block per wave row)
    constexpr int BN = WC * NCB * 32;           // output channels (all of them: one column tile)
    constexpr int WTN = NCB * 32;
    constexpr int S = TM + 32;                  // staged input rows per chunk
    constexpr int ABUF = (S + 8) * kRowB;       // + the zero row (row S), 1 KiB aligned
    constexpr int BBUF = kG * BN * kRowB;
    constexpr int NA_INSTR = S / 8;             // 1 KiB DMA wave-instructions per A chunk
    constexpr int NB_INSTR = kG * BN / 8;
    constexpr int NIA = (NA_INSTR + NW - 1) / NW, NIB = (NB_INSTR + NW - 1) / NW;
    static_assert(S % 8 == 0, "stage rows");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ENT_INSTR = (TM * kG * kNG * 4 + 1023) / 1024;       // 1 KiB DMA pieces of the tile's rulebook block
    int* s_ent = reinterpret_cast<int*>(smem + 2 * ABUF + 2 * BBUF);   // the tile's rulebook rows [TM][27]
    int* s_rng = s_ent + ENT_INSTR * 256;                              // [0..8] lo, [16..24] hi

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WC, wn = wave % WC;
    int Mlim = p.M;
    if (p.m_dev) {
        const int md = *p.m_dev;
        Mlim = md < Mlim ? md : Mlim;
    }
    // XCD-aware remap over the LIVE tiles only (the launch covers the allocation; hardware places block b on XCD b % 8):
    // consecutive tiles, which share their halo rows, land on one XCD's L2, and every XCD gets an equal share of the live
    // ones.  (Remapping over the allocated tile count left whole XCDs with dead tiles: level 2 of the bench cloud has
    // 6,585 live of 16,384 allocated tiles -- 3.2 of 8 XCDs were working.)
    const int live_tiles = (Mlim + TM - 1) / TM;
    if ((int)blockIdx.x >= live_tiles) return;
    int L;
    {
        const int b = blockIdx.x, xcd = b & 7, q = live_tiles >> 3, r = live_tiles & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    (void)tiles_m;
    const int m0 = L * TM;

    const float* __restrict__ in = reinterpret_cast<const float*>(p.in);
    const float* __restrict__ wgt = reinterpret_cast<const float*>(p.weight);
    const int KV = p.KW;                        // 27
    const int nsl = p.Cin >> 5;                 // 32-channel slices

    // ---- prologue: zero rows, the tile's rulebook block, the input-row range of every (dz, dy) group of this tile
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    {
        // the tile's rulebook block is contiguous (TM x 27 dwords): DMA it into LDS in 1 KiB pieces (the first version
        // copied it with a load -> wait -> ds_write loop, 14 serialised L2 round trips = ~14 us per tile).  Rows beyond
        // the live count hold garbage and are masked where they are read; the source is clamped to the allocation.
        const char* src = reinterpret_cast<const char*>(p.gather + (long long)m0 * KV);
        const long long avail = ((long long)p.M - m0) * KV * 4 - 16;            // last readable 16 B chunk of the allocation
        const unsigned se = lds_base + 2u * ABUF + 2u * BBUF;
#pragma unroll
        for (int j = 0; j < (ENT_INSTR + NW - 1) / NW; ++j) {
            const int i = wave_s + NW * j;
            if (i < ENT_INSTR) {
                long long off = (long long)i * 1024 + lane * 16;
                off = off < avail ? off : avail;
                __builtin_amdgcn_global_load_lds(src + off, (lds_ptr_t)(uintptr_t)(se + (unsigned)i * 1024u), 16, 0, 0);
            }
        }
    }
    if (tid < 16) *reinterpret_cast<uint4*>(smem + (tid >> 3) * ABUF + S * kRowB + (tid & 7) * 16) = uint4{0, 0, 0, 0};
    if (tid < 32) s_rng[tid] = (tid < 16) ? INT_MAX : -1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave * 64 < TM) {
        int lo[kNG], hi[kNG];
        const bool live_row = tid < TM && m0 + tid < Mlim;
        const int* e = s_ent + tid * KV;           // stride 27 dwords: conflict-free
#pragma unroll
        for (int g = 0; g < kNG; ++g) {
            lo[g] = INT_MAX;
            hi[g] = -1;
#pragma unroll
            for (int t = 0; t < kG; ++t) {
                const int v = live_row ? e[g * kG + t] : -1;
                if (v >= 0) {
                    lo[g] = v < lo[g] ? v : lo[g];
                    hi[g] = v + 1 > hi[g] ? v + 1 : hi[g];
                }
            }
        }
        // wave min / max of the 18 values: DPP inside 16-lane rows (xor 1, xor 2, half-row mirror, row mirror), then two
        // cross-row exchanges issued for all values at once
        auto dpp_step = [&](auto ctrl) {
            constexpr int C = decltype(ctrl)::value;
#pragma unroll
            for (int g = 0; g < kNG; ++g) {
                const int a = __builtin_amdgcn_update_dpp(0, lo[g], C, 0xF, 0xF, false);
                const int b = __builtin_amdgcn_update_dpp(0, hi[g], C, 0xF, 0xF, false);
                lo[g] = a < lo[g] ? a : lo[g];
                hi[g] = b > hi[g] ? b : hi[g];
            }
        };
        dpp_step(std::integral_constant<int, 0xB1>{});     // quad_perm [1,0,3,2]
        dpp_step(std::integral_constant<int, 0x4E>{});     // quad_perm [2,3,0,1]
        dpp_step(std::integral_constant<int, 0x141>{});    // row_half_mirror
        dpp_step(std::integral_constant<int, 0x140>{});    // row_mirror
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            int a[kNG], b[kNG];
#pragma unroll
            for (int g = 0; g < kNG; ++g) {
                a[g] = __shfl_xor(lo[g], off);
                b[g] = __shfl_xor(hi[g], off);
            }
#pragma unroll
            for (int g = 0; g < kNG; ++g) {
                lo[g] = a[g] < lo[g] ? a[g] : lo[g];
                hi[g] = b[g] > hi[g] ? b[g] : hi[g];
            }
        }
        if (lane < kNG) {
            int l = lo[0], h = hi[0];
#pragma unroll
            for (int g = 1; g < kNG; ++g) {
                l = lane == g ? lo[g] : l;
                h = lane == g ? hi[g] : h;
            }
            atomicMin(&s_rng[lane], l);
            atomicMax(&s_rng[16 + lane], h);
        }
    }
    __syncthreads();
    // ranges stay in LDS and are fetched with asm reads (wave-uniform, once per group per walker): register arrays
    // indexed by the running group land in scratch, and a compiler-visible LDS read inside the loop would be treated
    // as aliasing the in-flight DMA (s_waitcnt vmcnt(0) in front of it)
    const unsigned rng_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) int*)s_rng;
    auto rng = [&](int idx) {
        int v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(rng_base + 4u * (unsigned)idx) : "memory");
        return __builtin_amdgcn_readfirstlane(v);
    };

    // ---- stage walker: (group g, chunk start c, slice sl); groups without any neighbour are skipped
    struct Walk {
        int g, c, hi, sl;
    };
    auto seek = [&](Walk& w) {
        while (w.g < kNG) {
            w.c = rng(w.g);
            w.hi = rng(16 + w.g);
            if (w.c < w.hi) break;
            ++w.g;
        }
        w.sl = 0;
    };
    auto next = [&](Walk& w) {
        if (++w.sl < nsl) return;
        w.sl = 0;
        w.c += S;
        if (w.c >= w.hi) {
            ++w.g;
            seek(w);
        }
    };

    // per-lane constants of the DMA slots: instruction i covers stage rows 8i .. 8i+7, lane = (row, 16 B position)
    const int d_row = lane >> 3, d_pos = lane & 7;
    // ablations (compile-time: -DTT_SP_DEBUG=bits, profiles/r03_sparse_runs_ablation.txt; 0 in the product): 1 = no activation DMA
    // after the first stage, 2 = no weight DMA after the first stage, 4 = no MFMA phase, 8 = no operand split (raw bits as bf16),
    // 16 = every lane reads the zero row (no gather bank conflicts)
    constexpr bool dbg_no_a = (TT_SP_DEBUG & 1) != 0, dbg_no_b = (TT_SP_DEBUG & 2) != 0, dbg_no_mfma = (TT_SP_DEBUG & 4) != 0;
    constexpr bool dbg_no_split = (TT_SP_DEBUG & 8) != 0, dbg_zero_a = (TT_SP_DEBUG & 16) != 0;
    bool first_issue = true;
    // Address diet: everything about a DMA slot that does not change from stage to stage is a per-lane 32-bit element
    // offset computed once (weights: row n, tap t, swizzled chunk; activations: swizzled chunk); per stage a slot costs a
    // clamp + one 24-bit multiply (activations) and one 64-bit add onto a wave-uniform base.  (The first version redid
    // the 64-bit row * stride arithmetic per slot per stage: ~350 instructions between the barrier and the first MFMA.)
    int a_chunk[NIA], a_row[NIA], b_off[NIB];
#pragma unroll
    for (int j = 0; j < NIA; ++j) {
        const int row = 8 * (wave + NW * j) + d_row;
        a_row[j] = row;
        a_chunk[j] = (d_pos ^ ((row >> 1) & 7)) << 2;
    }
#pragma unroll
    for (int j = 0; j < NIB; ++j) {
        const int r = 8 * (wave + NW * j) + d_row;                 // r = t * BN + n
        const int t = r / BN, n = r - t * BN;
        b_off[j] = n * p.K + t * p.Cin + ((d_pos ^ ((n >> 1) & 7)) << 2);     // < 2^31 elements: Cout * K floats
    }
    auto issue = [&](const Walk& w, int buf) {
        const bool skip_a = dbg_no_a && !first_issue, skip_b = dbg_no_b && !first_issue;
        first_issue = false;
        const unsigned sa = lds_base + (unsigned)buf * ABUF, sb = lds_base + 2u * ABUF + (unsigned)buf * BBUF;
        const int nrows = (w.hi - w.c) < S ? (w.hi - w.c) : S;
        const float* abase = in + p.in_coff + w.sl * 32 + (long long)w.c * p.in_cstride;      // wave-uniform
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            const int i = wave_s + NW * j;
            if (i < NA_INSTR && 8 * i < nrows && !skip_a) {            // wave-uniform
                // rows past the range: the last valid line (never read).  row * stride < 2^24 * 2^7 fits 32 bits
                const int row = a_row[j] < nrows ? a_row[j] : nrows - 1;
                const float* sp = abase + (unsigned)(__umul24((unsigned)row, (unsigned)p.in_cstride) + (unsigned)a_chunk[j]);
                __builtin_amdgcn_global_load_lds(sp, (lds_ptr_t)(uintptr_t)(sa + (unsigned)i * 1024u), 16, 0, 0);
            }
        }
        const float* bbase = wgt + (long long)(w.g * kG) * p.Cin + w.sl * 32;                  // wave-uniform
#pragma unroll
        for (int j = 0; j < NIB; ++j) {
            const int i = wave_s + NW * j;
            if (i < NB_INSTR && !skip_b)
                __builtin_amdgcn_global_load_lds(bbase + (unsigned)b_off[j], (lds_ptr_t)(uintptr_t)(sb + (unsigned)i * 1024u),
                                                 16, 0, 0);
        }
    };
    // rulebook entries of this lane's output row for the three taps of group g (asm LDS reads, see `rng`)
    const unsigned ent_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) int*)s_ent +
                              4u * (unsigned)((wm * 32 + (lane & 31)) * KV);
    const bool row_live = m0 + wm * 32 + (lane & 31) < Mlim;
    auto load_ent = [&](int g, int (&e)[kG]) {
#pragma unroll
        for (int t = 0; t < kG; ++t)
            asm volatile("ds_read_b32 %0, %1" : "=v"(e[t]) : "v"(ent_base + 4u * (unsigned)(g * kG + t)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < kG; ++t) {
            asm volatile("" : "+v"(e[t]));
            e[t] = row_live ? e[t] : -1;            // rows beyond the live count: garbage in the rulebook
        }
    };

    f32x16 acc[NCB];
#pragma unroll
    for (int j = 0; j < NCB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto lds_read = [](unsigned addr) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
        return v;
    };
    const unsigned kb = lane >> 5;                                   // K half of this lane's MFMA operands
    unsigned fb_row[NCB], fb_swz[NCB];
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
        const int n = wn * WTN + j * 32 + (lane & 31);
        fb_row[j] = (unsigned)n * kRowB;
        fb_swz[j] = (n >> 1) & 7;
    }

    Walk wc{0, 0, 0, 0};
    seek(wc);
    Walk wi = wc;
    int ent_cur[kG] = {-1, -1, -1};
    int cur_g = -1;
    if (wc.g < kNG) {
        issue(wi, 0);
        next(wi);
    }
    int buf = 0;
    while (wc.g < kNG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");    // stage `buf` is published; everyone is done with the other buffer
        if (wi.g < kNG) {
            issue(wi, buf ^ 1);
            next(wi);
        }
        if (wc.g != cur_g) {
            load_ent(wc.g, ent_cur);
            cur_g = wc.g;
        }
        // ---- compute stage (wc.g, wc.c, wc.sl) from buffer `buf`
        const unsigned sa = lds_base + (unsigned)buf * ABUF, sb = lds_base + 2u * ABUF + (unsigned)buf * BBUF;
        const int nrows = (wc.hi - wc.c) < S ? (wc.hi - wc.c) : S;
        unsigned a_off[kG], a_swz[kG];
        bool any = false;
#pragma unroll
        for (int t = 0; t < kG; ++t) {
            const int s = ent_cur[t] - wc.c;
            const bool ok = ent_cur[t] >= 0 && (unsigned)s < (unsigned)nrows;
            any = any || ok;
            const unsigned ar = (ok && !dbg_zero_a) ? (unsigned)s : (unsigned)S;   // absent (or in another chunk): the zero row
            a_off[t] = sa + ar * kRowB;
            a_swz[t] = (ar >> 1) & 7;
        }
        // no row of this wave has a neighbour of this group in this chunk (chunked ranges, isolated sites): nothing to add
        if (__builtin_amdgcn_ballot_w64(any) != 0ull && !dbg_no_mfma) {
            // 6 sub-steps (tap, k-step); the LDS reads of sub-step i+1 are in flight under the split + MFMAs of sub-step i
            u32x4 ra[2][2], rbh[2][NCB], rbl[2][NCB];
            auto reads = [&](int i, int slot) {
                const int t = i >> 1, ks = i & 1;
                const unsigned a0 = a_off[t] + (((4u * ks + 2u * kb) ^ a_swz[t]) << 4);
                ra[slot][0] = lds_read(a0);
                ra[slot][1] = lds_read(a0 ^ 16u);
#pragma unroll
                for (int j = 0; j < NCB; ++j) {
                    const unsigned b0 = sb + (unsigned)t * (BN * kRowB) + fb_row[j] + (((4u * ks + kb) ^ fb_swz[j]) << 4);
                    rbh[slot][j] = lds_read(b0);
                    rbl[slot][j] = lds_read(b0 ^ 32u);
                }
            };
            reads(0, 0);
#pragma unroll
            for (int i = 0; i < 2 * kG; ++i) {
                const int cur = i & 1;
                if (i + 1 < 2 * kG) {
                    reads(i + 1, cur ^ 1);
                    if constexpr (NCB == 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                u32x4 r0 = ra[cur][0], r1 = ra[cur][1];
                asm volatile("" : "+v"(r0));
                asm volatile("" : "+v"(r1));
                const float x[8] = {__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z),
                                    __uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y),
                                    __uint_as_float(r1.z), __uint_as_float(r1.w)};
                uint32_t h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = pack_bf16x2(x[2 * e], x[2 * e + 1]);                       // round to nearest even
                    const float q0 = x[2 * e] - __uint_as_float(h[e] << 16);           // exact in f32
                    const float q1 = x[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u);
                    l[e] = pack_bf16x2(q0, q1);
                }
                uint4 ah = uint4{h[0], h[1], h[2], h[3]}, al = uint4{l[0], l[1], l[2], l[3]};
                if (dbg_no_split) {
                    ah = __builtin_bit_cast(uint4, r0);
                    al = __builtin_bit_cast(uint4, r1);
                }
                u32x4 bh[NCB], bl[NCB];
#pragma unroll
                for (int j = 0; j < NCB; ++j) {
                    bh[j] = rbh[cur][j];
                    bl[j] = rbl[cur][j];
                    asm volatile("" : "+v"(bh[j]));
                    asm volatile("" : "+v"(bl[j]));
                }
#pragma unroll
                for (int j = 0; j < NCB; ++j) Mfma<uint16_t>::run(al, __builtin_bit_cast(uint4, bh[j]), acc[j]);
#pragma unroll
                for (int j = 0; j < NCB; ++j) Mfma<uint16_t>::run(ah, __builtin_bit_cast(uint4, bl[j]), acc[j]);
#pragma unroll
                for (int j = 0; j < NCB; ++j) Mfma<uint16_t>::run(ah, __builtin_bit_cast(uint4, bh[j]), acc[j]);
            }
        }
        next(wc);
        buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    f32x16 acc2[1][NCB];
#pragma unroll
    for (int j = 0; j < NCB; ++j) acc2[0][j] = acc[j];
    conv_epilogue<float, 1, NCB, 32, WTN>(p, acc2, smem, wave, lane, wm, wn, m0, 0, Mlim);
#endif
}

template <int NCB, int WR, int WC>
static int launch_sp_runs(ConvArgs& a, hipStream_t st) {
    constexpr int TM = WR * 32, BN = WC * NCB * 32, S = TM + 32, NW = WR * WC;
    size_t smem = (size_t)2 * (S + 8) * kRowB + (size_t)2 * kG * BN * kRowB + 128 +
                  (size_t)((TM * kG * kNG * 4 + 1023) / 1024) * 1024;
    const size_t epi = (size_t)NW * 32 * (NCB * 32 + 4) * 4;
    if (smem < epi) smem = epi;
    auto kern = sp_conv_runs_kernel<NCB, WR, WC>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    const int tiles_m = div_up(a.M, TM);
    a.tiles_n = 1;
    a.splits = 1;
    a.ws = nullptr;
    a.m_begin = 0;
    snprintf(g_conv_kernel, sizeof(g_conv_kernel), "sp_conv_runs_kernel<%d, %d, %d>", NCB, WR, WC);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles_m), dim3(NW * 64), smem, st, a, tiles_m);
    return 1;
}

// bf16x3 gathered conv with 27 taps in [kz][ky][kx] order (3x3x3 SubM / strided sparse conv), Cin a multiple of 32,
// Cout 32 / 64 / 128.  `a.weight` = pre-split pair-format weights.  Returns 0 when the shape is not covered.
int try_launch_sp_conv_runs(ConvArgs& a, hipStream_t st) {
    if (!a.gather || a.row_perm || a.KH != 1 || a.KW != kG * kNG) return 0;
    // strided sparse convs (the caller states stride 2): the inputs of a (dz, dy) group sit on every other line, the
    // contiguous range is ~4x the tile and is walked in mostly-empty chunks (measured 0.74 -> 4.1 ms): gather kernel
    if (a.stride != 1) return 0;
    if (a.Cin % 32 != 0 || a.Cin > 128 || a.M < 2048 || a.pixel_shuffle2) return 0;
    if ((a.in_cstride & 3) || (a.in_coff & 3)) return 0;
    if (a.Cout == 32) return launch_sp_runs<1, 8, 1>(a, st);
    if (a.Cout == 64) return launch_sp_runs<2, 8, 1>(a, st);
    if (a.Cout == 128) return launch_sp_runs<2, 4, 2>(a, st);
    return 0;
}

}  // namespace tt
