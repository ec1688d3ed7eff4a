// bf16x3 GEMM for the SHORT-K 1 x 1 layers (K = 128 ... 512): PERSISTENT workgroups that store tile i while they multiply tile i + 1.
//
// Why (tools/conv_trace.py, profiles/r05_conv_tile_trace.txt): on these layers a 256 x 256 tile of the LDS-DMA kernel spends
// 3.5-4.8 us waiting for its first K tile (prologue), 5-24 us in its K loop and 8-22 us in its epilogue -- and the epilogue of a
// residual layer is bound by the LATENCY of the few KiB of residual reads a wave can keep in flight, with every CU of the chip in
// that phase at the same time and the matrix pipes idle.  Neither a phase stagger of the CUs nor deeper residual prefetch inside the
// epilogue helped (profiles/r05_shortk_ab_tiles_stagger.txt, r05_epilogue_residual_prefetch_ab.txt).  What hides a latency is work:
//   * one workgroup per CU walks its tiles (XCD-contiguous per round); the K tiles of consecutive output tiles form ONE stream through
//     the two-stage LDS ring, so only a workgroup's first tile has a prologue;
//   * 256 x 128 output tile, FOUR waves (one per SIMD: 512 registers per lane), each 64 rows x 128 columns: TWO accumulator sets.
//     While set A accumulates tile i + 1, set B (tile i) is drained in pieces behind the MFMAs of tile i + 1's K loop: per K tile a
//     fixed number of passes (LDS-staged 32 x 128 block -> 16 B per lane of one output row: + residual, ReLU, non-temporal store);
//     the residual reads of a K tile's passes are issued one K tile earlier, i.e. they have ~2-3 us of MFMAs to land under;
//   * the staging area (16 KiB per wave) lies beside the ring, so the drain needs no barrier of its own.
// Arithmetic: the X3 body of conv_igemm_glds.hip (same operand split, K order, term order) and conv_epilogue_vec's hot path (same
// scale / shift / per-image shift / residual / ReLU order): results are BIT-IDENTICAL to the 8-wave tile (tools/shortk_ab.py).
// Counted waits: a K tile issues, in this order, the next K tile's DMA (12 pieces per wave), then P residual reads, then -- when it
// drains a tile -- P stores; "the DMA of the next K tile has landed" is therefore vmcnt(P) or vmcnt(2 P).  The residual reads and
// the stores are ordinary C++ memory operations between two compiler fences (the compiler's own waits for the residual registers
// are exact in the fully unrolled tile body); the epilogue's LDS traffic is inline asm, like the fragment reads: hipcc would put a
// vmcnt(0) in front of every LDS read it can see, because the LDS-DMA writes LDS.
// Contract (dispatcher: try_launch_gemm_x3_persist): 1 x 1 / stride 1 / no padding over dense rows, f32 storage, M % 256 == 0,
// Cout % 128 == 0, K in {128, 256, 512}, vector epilogue with f32 output, identity or ReLU, at most one residual, a per-image
// shift only if images are whole 32-row blocks.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "conv_common.h"

namespace tt {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N - 1>) -- indices that are constant EXPRESSIONS (asm immediates)
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

namespace persist {
constexpr int BM = 256, BN = 128, BKB = 128, BK = 32;
constexpr int TM = 2;
constexpr int A_BYTES = BM * BKB, B_BYTES = BN * BKB;          // 32 KiB, 16 KiB per stage
constexpr int OFF_B = 2 * A_BYTES;                              // ring: A stages at 0 / 32 KiB, B stages at 64 / 80 KiB
constexpr int OFF_STG = 2 * A_BYTES + 2 * B_BYTES;              // staging from 96 KiB: 64 KiB whatever the wave count
constexpr int LDS_BYTES = OFF_STG + 64 * 1024;                  // 160 KiB
}  // namespace persist

// Two wave geometries on the 256 x 128 tile (W8 template flag):
//   false: FOUR waves, one per SIMD, 64 rows x 128 columns each (TN = 4: the cheapest LDS traffic per MFMA and 512 registers per lane --
//          but a lone wave's stalls are the matrix pipe's stalls: this body is compiler-scheduled, not hand-placed);
//   true : EIGHT waves as 4 x 2, two per SIMD, 64 x 64 each (TN = 2; 256 registers per lane): the second wave covers the first one's
//          LDS / VMEM waits, like the 8-wave tile of conv_igemm_glds.hip.

template <int NK, bool RES, bool W8>
__global__ __launch_bounds__(W8 ? 512 : 256, 1) void gemm_x3_persist_kernel(const ConvArgs p, int tiles_m, int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace persist;
    static_assert(NK == 4 || NK == 8, "K = 128, 256 (NK = 16 compiles, but its unrolled body leaves the accumulators in scratch)");
    constexpr int NW = W8 ? 8 : 4;            // waves
    constexpr int WNN = W8 ? 2 : 1;           // waves along N; wave (wm, wn) owns rows [64 wm, +64) x columns [WTN wn, +WTN)
    constexpr int WTN = BN / WNN, TN = WTN / 32;
    constexpr int LPR = WTN / 4;              // lanes per staged row (16 B each): 32 or 16
    constexpr int RPP = 64 / LPR;             // rows per pass: 2 or 4
    constexpr int BPASS = 32 / RPP;           // passes per 32-row block: 16 or 8
    constexpr int NPASS = TM * BPASS;         // passes per wave and tile: 32 or 16
    constexpr int STG_BYTES = 32 * WTN * 4;   // staging per wave: one 32 x WTN f32 block (16 or 8 KiB)
    constexpr int NIA = BM * 8 / 64 / NW, NIB = BN * 8 / 64 / NW;   // 1 KiB DMA pieces per wave per K tile: 8 + 4, or 4 + 2
    constexpr int P = NPASS / NK;             // passes per K tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WNN, wn = wave % WNN;
    const int wm_s = wave_s / WNN, wn_s = wave_s % WNN;
    const int T = tiles_m * tiles_n, G = (int)gridDim.x, w = (int)blockIdx.x;
    const float* __restrict__ in = reinterpret_cast<const float*>(p.in);
    const float* __restrict__ wgt = reinterpret_cast<const float*>(p.weight);
    const float* __restrict__ res = reinterpret_cast<const float*>(p.res1);
    float* __restrict__ outp = reinterpret_cast<float*>(p.out);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto swz = [](int row) __attribute__((always_inline)) { return (row >> 1) & 7; };

    // tile r of this workgroup: round r of the launch, XCD-contiguous inside the round (workgroups w % 8 == x share XCD x's L2:
    // they take consecutive tiles = the column tiles of the same activation rows).  Everything about a tile is wave-uniform.
    auto tile_of = [&](int r, int& m0, int& n0) __attribute__((always_inline)) -> bool {
        const int first = r * G;
        int nblk = T - first;
        if (nblk <= 0) return false;
        nblk = nblk < G ? nblk : G;
        if (w >= nblk) return false;
        const int xcd = w & 7, q = nblk >> 3, rr = nblk & 7;
        const int L = first + (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (w >> 3);
        m0 = __builtin_amdgcn_readfirstlane((L / tiles_n) * BM);
        n0 = __builtin_amdgcn_readfirstlane((L % tiles_n) * BN);
        return true;
    };

    // ---- DMA: piece j of this wave = 1 KiB piece (wave + 4 j) of the K tile = rows 8 (wave + 4 j) ... + 7.  The swizzle
    // (row >> 1) & 7 of those rows depends on the piece's parity only, i.e. on the wave: ONE lane offset per operand, the
    // pieces of a wave are a uniform stride apart (32 rows).
    const int a_lane = (wave * 8 + (lane >> 3)) * p.in_cstride + (((lane & 7) ^ swz(wave * 8 + (lane >> 3))) << 2);
    const int b_lane = (wave * 8 + (lane >> 3)) * p.K + (((lane & 7) ^ swz(wave * 8 + (lane >> 3))) << 2);
    auto issue_dma = [&](int m0, int n0, int kt, int stage) __attribute__((always_inline)) {
        const float* ab = in + (long long)m0 * p.in_cstride + p.in_coff + kt * BK + a_lane;
        const float* bb = wgt + (long long)n0 * p.K + kt * BK + b_lane;
        const unsigned sa = lds_base + (unsigned)stage * A_BYTES, sb = lds_base + OFF_B + (unsigned)stage * B_BYTES;
        const long long a_step = (long long)(NW * 8) * p.in_cstride, b_step = (long long)(NW * 8) * p.K;   // wave-uniform
#pragma unroll
        for (int j = 0; j < NIA; ++j)
            __builtin_amdgcn_global_load_lds(ab + j * a_step, (lds_ptr_t)(uintptr_t)(sa + (unsigned)(wave_s + NW * j) * 1024u), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < NIB; ++j)
            __builtin_amdgcn_global_load_lds(bb + j * b_step, (lds_ptr_t)(uintptr_t)(sb + (unsigned)(wave_s + NW * j) * 1024u), 16, 0, 0);
    };

    // ---- fragment offsets inside a stage (conv_igemm_glds.hip, X3).  Row r of a 32-row block has swizzle (r >> 1) & 7 whatever the
    // block, so a block is a constant away (4 KiB); K step 1 flips chunk bit 2 (offset ^ 64); the second half of an A fragment is
    // ^ 16, the lo half of a B fragment ^ 32.
    const unsigned hi = lane >> 5;
    const unsigned fa0 = (wm * 64 + (lane & 31)) * BKB + (((2u * hi) ^ (unsigned)swz(lane & 31)) << 4);
    const unsigned fb0 = (wn * WTN + (lane & 31)) * BKB + ((hi ^ (unsigned)swz(lane & 31)) << 4);
    auto fa_at = [&](int kc, int i) __attribute__((always_inline)) { return (fa0 ^ (kc ? 64u : 0u)) + (unsigned)i * 32u * BKB; };
    auto fb_at = [&](int kc, int j) __attribute__((always_inline)) { return (fb0 ^ (kc ? 64u : 0u)) + (unsigned)j * 32u * BKB; };
    auto lds_read = [](unsigned addr) __attribute__((always_inline)) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
        return v;
    };

    // ---- epilogue state.  Staging: this wave's 32 x WTN f32 block, row-major (a row is read by LPR consecutive lanes).
    const unsigned stg = lds_base + OFF_STG + (unsigned)wave * STG_BYTES;
    unsigned stg_w = stg + (((lane >> 5) * 4) * WTN + (lane & 31)) * 4;      // + ((r & 3) + 8 (r >> 2)) * WTN * 4 + j * 128
    unsigned stg_r = stg + ((lane / LPR) * WTN + (lane % LPR) * 4) * 4;       // + pass * 1024 (a pass = RPP rows of WTN floats = 1 KiB)
    const int col_l = (lane % LPR) * 4, row_l = lane / LPR;
    const int ohw = p.OH * p.OW;
    const bool relu = p.act == TT_ACT_RELU;
    const bool nt_store = (p.flags & 16) != 0;
    struct Tile {                   // a tile and what its drain needs
        int m0, n0;                 // wave-uniform
        float sc[TN], sh[TN];       // folded affine of this lane's four columns
    };
    auto load_affine = [&](Tile& o) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = o.n0 + wn * WTN + j * 32 + (lane & 31);
            o.sc[j] = p.scale ? p.scale[col] : 1.f;
            o.sh[j] = p.shift ? p.shift[col] : 0.f;
        }
    };
    // residual pointer of pass 0 of a tile for this lane (row m0 + 64 wm + row_l, 4 columns from col_l); pass q is RPP q rows
    // further (the 64 rows of a wave are contiguous)
    auto res_base = [&](const Tile& o) __attribute__((always_inline)) {
        return res + (long long)(o.m0 + wm_s * 64 + row_l) * p.res1_cstride + p.res1_coff + o.n0 + wn_s * WTN + col_l;
    };
    const long long res_step = (long long)RPP * p.res1_cstride, out_step = (long long)RPP * p.out_cstride;     // wave-uniform, per pass
    // stage row block b (0, 1) of `acc` (tile o) with the folded affine (+ the per-image shift: the block lies in one image);
    // returns the output pointer of the block's first pass for this lane
    auto stage_block = [&](const Tile& o, f32x16 (&acc)[TM][TN], auto b_t) -> float* {
        constexpr int b = decltype(b_t)::value;
        const int mb = o.m0 + wm_s * 64 + b * 32;                     // wave-uniform: the division below is scalar
        const int img = (p.out_fast && !p.shift_n) ? 0 : mb / ohw;
        float shb[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) shb[j] = o.sh[j];
        if (p.shift_n) {
            const float* sn = p.shift_n + (long long)(img % p.shift_n_mod) * p.Cout + o.n0 + wn * WTN + (lane & 31);
#pragma unroll
            for (int j = 0; j < TN; ++j) shb[j] += sn[j * 32];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's reads of the previous block are done
        static_for<TN>([&](auto j_t) __attribute__((always_inline)) {
            constexpr int j = decltype(j_t)::value;
            static_for<16>([&](auto r_t) __attribute__((always_inline)) {
                constexpr int r = decltype(r_t)::value;
                const float v = acc[b][j][r] * o.sc[j] + shb[j];
                const unsigned base = stg_w;        // (named here: an asm operand alone does not capture in a generic lambda)
                // (the address is ONE register + an immediate: computed addresses are loop-invariant, get hoisted, and spill)
                asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(base), "v"(v), "n"(((r & 3) + 8 * (r >> 2)) * WTN * 4 + j * 128) : "memory");
            });
            __builtin_amdgcn_sched_barrier(0);                        // one column block's 16 values live at a time
        });
        const long long obase = p.out_fast ? 0 : (long long)img * (p.out_nstride - (long long)ohw * p.out_cstride);
        return outp + (long long)(mb + row_l) * p.out_cstride + obase + p.out_coff + o.n0 + wn_s * WTN + col_l;
    };
    // one pass: staged row -> (+ residual) -> ReLU -> 16 B store.  ql: pass inside its row block (0 .. BPASS - 1)
    auto do_pass = [&](float* out_blk, auto ql_t, const float4& rv) __attribute__((always_inline)) {
        constexpr int ql = decltype(ql_t)::value;
        u32x4 t;
        const unsigned base = stg_r;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(base), "n"(ql * 1024) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(t));
        float v[4] = {__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
        if (RES) { v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w; }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        typedef float f4v __attribute__((ext_vector_type(4)));
        float* dst = out_blk + ql * out_step;
        if (nt_store) __builtin_nontemporal_store(f4v{v[0], v[1], v[2], v[3]}, reinterpret_cast<f4v*>(dst));
        else *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    };
    auto load_res = [&](const float* base, int q) __attribute__((always_inline)) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(base + q * res_step));
        return make_float4(t.x, t.y, t.z, t.w);
    };

    float4 rbuf[2][P];                 // residual vectors: [K tile parity][pass of that K tile]
    f32x16 accA[TM][TN], accB[TM][TN];

    // ---- one output tile: NK K tiles of MFMAs into `cur`; if HAS_OLD, tile `o` (accumulators `old`) drains behind them.
    // n_first: VMEM operations the previous K tile issued after ITS next-tile DMA (0 after the prologue, else P or 2 P).
    auto run_tile = [&](auto has_old_t, f32x16 (&cur)[TM][TN], f32x16 (&old)[TM][TN], const Tile& o, const Tile& me, bool has_next,
                        int m0n, int n0n, int n_first) {
        constexpr bool HAS_OLD = decltype(has_old_t)::value;
        const float* res_o = RES ? res_base(o) : nullptr;
        const float* res_me = RES ? res_base(me) : nullptr;
        float* out_blk = nullptr;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) cur[i][j][r] = 0.f;
        static_for<NK>([&](auto kt_t) __attribute__((always_inline)) {
            constexpr int kt = decltype(kt_t)::value;
            // K tile kt of this tile has landed once only the operations issued after its DMA are outstanding
            if (kt == 0) {
                if (n_first == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (n_first == P) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
            } else {
                // the previous K tile of this body: residual reads only when it drains a tile (see below), P stores if it does
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RES && HAS_OLD ? P : 0) + (HAS_OLD ? P : 0)) : "memory");
            }
            asm volatile("s_barrier" ::: "memory");       // publishes it; every wave is done reading the stage the next DMA overwrites
            const int stage = kt & 1;                      // (NK is even: a tile's K tile kt always lies in stage kt & 1)
            if (kt + 1 < NK) issue_dma(me.m0, me.n0, kt + 1, stage ^ 1);
            else if (has_next) issue_dma(m0n, n0n, 0, stage ^ 1);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // drain: a row block is staged when its first pass comes up
            if constexpr (HAS_OLD && (kt * P) % BPASS == 0) out_blk = stage_block(o, old, std::integral_constant<int, (kt * P) / BPASS>{});
            // residual reads of the NEXT K tile's passes (this tile's own first passes when kt is its last K tile).  Every vector
            // read here is consumed (the counted waits above rely on the number of operations issued): without a tile to
            // drain only the last K tile reads
            if (RES && (HAS_OLD || kt + 1 == NK)) {
#pragma unroll
                for (int q = 0; q < P; ++q)
                    rbuf[(kt + 1) & 1][q] = kt + 1 < NK ? load_res(res_o, (kt + 1) * P + q) : load_res(res_me, q);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- bf16x3 K tile: 2 K steps x TN column blocks = 2 TN sub-steps of 6 MFMAs, reads one sub-step ahead
            const unsigned sA = lds_base + (unsigned)stage * A_BYTES, sB = lds_base + OFF_B + (unsigned)stage * B_BYTES;
            u32x4 ra0[TM], ra1[TM], bh[2], bl[2];
            uint4 ah[2][TM], al[2][TM];
            auto split_frag = [&](int i, uint4& hi_out, uint4& lo_out) __attribute__((always_inline)) {
                asm volatile("" : "+v"(ra0[i]));
                asm volatile("" : "+v"(ra1[i]));
                const float x[8] = {__uint_as_float(ra0[i].x), __uint_as_float(ra0[i].y), __uint_as_float(ra0[i].z),
                                    __uint_as_float(ra0[i].w), __uint_as_float(ra1[i].x), __uint_as_float(ra1[i].y),
                                    __uint_as_float(ra1[i].z), __uint_as_float(ra1[i].w)};
                uint32_t h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = pack_bf16x2(x[2 * e], x[2 * e + 1]);
                    const float r0 = x[2 * e] - __uint_as_float(h[e] << 16);
                    const float r1 = x[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u);
                    l[e] = pack_bf16x2(r0, r1);
                }
                hi_out = make_uint4(h[0], h[1], h[2], h[3]);
                lo_out = make_uint4(l[0], l[1], l[2], l[3]);
            };
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ra0[i] = lds_read(sA + fa_at(0, i));
                ra1[i] = lds_read(sA + (fa_at(0, i) ^ 16u));
            }
            bh[0] = lds_read(sB + fb_at(0, 0));
            bl[0] = lds_read(sB + (fb_at(0, 0) ^ 32u));
            constexpr int NS = 2 * TN;
            static_for<NS>([&](auto ss_t) __attribute__((always_inline)) {
                constexpr int ss = decltype(ss_t)::value;
                // (eight waves: ONE set of split fragments -- 16 registers the residual vectors need; the K steps of a wave then
                // serialise on the split, which the SIMD's other wave covers)
                constexpr int kc = ss / TN, j = ss % TN, buf = ss & 1, ab = W8 ? 0 : (kc & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(bh[buf]));
                asm volatile("" : "+v"(bl[buf]));
                if (j == 0) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) split_frag(i, ah[ab][i], al[ab][i]);
                }
                if (j == TN - 1 && kc + 1 < 2) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        ra0[i] = lds_read(sA + fa_at(kc + 1, i));
                        ra1[i] = lds_read(sA + (fa_at(kc + 1, i) ^ 16u));
                    }
                }
                if (ss + 1 < NS) {
                    const int kc2 = (ss + 1) / TN, j2 = (ss + 1) % TN;
                    bh[buf ^ 1] = lds_read(sB + fb_at(kc2, j2));
                    bl[buf ^ 1] = lds_read(sB + (fb_at(kc2, j2) ^ 32u));
                }
                const uint4 bhv = __builtin_bit_cast(uint4, bh[buf]);
                const uint4 blv = __builtin_bit_cast(uint4, bl[buf]);
#pragma unroll
                for (int i = 0; i < TM; ++i) Mfma<uint16_t>::run(al[ab][i], bhv, cur[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i) Mfma<uint16_t>::run(ah[ab][i], blv, cur[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i) Mfma<uint16_t>::run(ah[ab][i], bhv, cur[i][j]);
                // ---- this sub-step's share of the drain: passes [kt P + ss P / 8, kt P + (ss + 1) P / 8) behind the queued MFMAs
                if constexpr (HAS_OLD) {
                    constexpr int q0 = (ss * P) / NS, q1 = ((ss + 1) * P) / NS;
                    static_for<q1 - q0>([&](auto dq_t) __attribute__((always_inline)) {
                        constexpr int q = q0 + decltype(dq_t)::value;
                        do_pass(out_blk, std::integral_constant<int, (kt * P + q) % BPASS>{}, rbuf[kt & 1][q]);
                    });
                }
            });
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- the workgroup's tiles
    Tile t_prev, t_cur;
    if (!tile_of(0, t_cur.m0, t_cur.n0)) return;
    t_prev = t_cur;
    load_affine(t_cur);
    issue_dma(t_cur.m0, t_cur.n0, 0, 0);
    int n_first = 0;
    int r = 0;
    bool first = true;
    // parity: even tiles accumulate in accA, odd ones in accB
    while (true) {
        {
            int m0n = 0, n0n = 0;
            const bool has_next = tile_of(r + 1, m0n, n0n);
            if (first) run_tile(std::false_type{}, accA, accB, t_prev, t_cur, has_next, m0n, n0n, n_first);
            else run_tile(std::true_type{}, accA, accB, t_prev, t_cur, has_next, m0n, n0n, n_first);
            n_first = (RES ? P : 0) + (first ? 0 : P);
            first = false;
            t_prev = t_cur;
            if (!has_next) break;
            t_cur.m0 = m0n; t_cur.n0 = n0n;
            load_affine(t_cur);
            ++r;
        }
        {
            int m0n = 0, n0n = 0;
            const bool has_next = tile_of(r + 1, m0n, n0n);
            run_tile(std::true_type{}, accB, accA, t_prev, t_cur, has_next, m0n, n0n, n_first);
            n_first = (RES ? P : 0) + P;
            t_prev = t_cur;
            if (!has_next) break;
            t_cur.m0 = m0n; t_cur.n0 = n0n;
            load_affine(t_cur);
            ++r;
        }
    }
    // ---- drain of the last tile (tile index r: parity picks the accumulator set): both row blocks, all residual reads of a block
    // up front (there is nothing to overlap with any more)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    auto drain = [&](f32x16 (&acc)[TM][TN]) {
        const float* rb = RES ? res_base(t_prev) : nullptr;
        static_for<TM>([&](auto b_t) __attribute__((always_inline)) {
            constexpr int b = decltype(b_t)::value;
            float4 rv[BPASS];
            if (RES) {
#pragma unroll
                for (int q = 0; q < BPASS; ++q) rv[q] = load_res(rb, b * BPASS + q);
            }
            float* ob = stage_block(t_prev, acc, b_t);
            static_for<BPASS>([&](auto q_t) __attribute__((always_inline)) {
                constexpr int q = decltype(q_t)::value;
                do_pass(ob, q_t, RES ? rv[q] : make_float4(0.f, 0.f, 0.f, 0.f));
            });
        });
    };
    if (r & 1) drain(accB);
    else drain(accA);
#endif
}

// Dispatcher.  Returns 1 if the layer was launched here.
int try_launch_gemm_x3_persist(ConvArgs& a, hipStream_t st) {
    using namespace persist;
    if (a.gather || a.m_dev || a.ws || a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0 || a.m_begin != 0) return 0;
    if (a.M % BM != 0 || a.Cout % BN != 0 || (a.K != 128 && a.K != 256) || a.K != a.Cin) return 0;
    if (a.in_nstride != (long long)a.H * a.W * a.in_cstride || a.in_cstride % 4 != 0 || a.in_coff % 4 != 0) return 0;   // dense rows
    if (!a.vec_epi || a.out_dtype != TT_F32 || a.pixel_shuffle2 || a.res2 || (a.res1 && !a.res_vec)) return 0;
    if (a.act != TT_ACT_NONE && a.act != TT_ACT_RELU) return 0;
    const long long ohw = (long long)a.OH * a.OW;
    if (!a.out_fast && ohw % 32 != 0) return 0;                      // row-linear output inside an image: whole 32-row blocks per image
    if (a.shift_n && ohw % 32 != 0) return 0;
    const int tiles_m = a.M / BM, tiles_n = a.Cout / BN;
    const long long T = (long long)tiles_m * tiles_n;
    if (T < 2LL * kNumCU) return 0;                                  // fewer than two tiles per CU: nothing to overlap
    const int nk = a.K / BK;
    const bool has_res = a.res1 != nullptr;
    static const bool w8 = [] { const char* e = getenv("TT_X3_PERSIST_WAVES"); return e ? atoi(e) == 8 : true; }();   // A/B knob
    void (*kern)(const ConvArgs, int, int) = nullptr;
#define TT_PICK(NK_)                                                                                      \
    kern = w8 ? (has_res ? gemm_x3_persist_kernel<NK_, true, true> : gemm_x3_persist_kernel<NK_, false, true>) \
              : (has_res ? gemm_x3_persist_kernel<NK_, true, false> : gemm_x3_persist_kernel<NK_, false, false>)
    switch (nk) {
        case 4: TT_PICK(4); break;
        default: TT_PICK(8); break;
    }
#undef TT_PICK
    static bool attr_set = false;
    if (!attr_set) {
#define TT_ATTR(NK_, R_)                                                                                                   \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_persist_kernel<NK_, R_, true>),                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);                                      \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_persist_kernel<NK_, R_, false>),                       \
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)
        TT_ATTR(4, true); TT_ATTR(4, false);
        TT_ATTR(8, true); TT_ATTR(8, false);
#undef TT_ATTR
        attr_set = true;
    }
    a.tiles_n = tiles_n;
    a.splits = 1;
    snprintf(g_conv_kernel, sizeof(g_conv_kernel), "gemm_x3_persist_kernel<%d, %s, %s>", nk, has_res ? "true" : "false", w8 ? "8 waves" : "4 waves");
    hipLaunchKernelGGL(kern, dim3((unsigned)kNumCU), dim3(w8 ? 512 : 256), LDS_BYTES, st, a, tiles_m, tiles_n);
    return 1;
}

}  // namespace tt
