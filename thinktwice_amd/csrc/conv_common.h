// Shared definitions of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_igemm_glds.hip).
#pragma once
#include <type_traits>
#include "tt_common.h"

namespace tt {
extern thread_local char g_conv_kernel[96];   // common.cpp: label of the kernel the last conv launch used
extern long long* g_conv_trace;               // common.cpp: tt_conv_set_trace (measurement aid; null in the product)


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct ConvArgs {
    const void* in;
    const void* weight;
    void* out;
    const float* scale;
    const float* shift;
    const float* shift_n;
    const void* res1;
    const void* res2;
    const int* gather;   // GATHER mode: [M][KH*KW] input row per (output row, tap), -1 = none
    const int* m_dev;    // optional device-side row count (rows >= *m_dev are skipped)
    const int* row_perm;         // GATHER, optional: tile slot -> actual output row (rows sorted by tap mask)
    const unsigned* row_mask;    //   "       the sorted masks (bit t = tap t present), 0xFFFFFFFF beyond the live rows
    float* ws;           // split-K: f32 [M][Cout] partial-sum workspace (pre-zeroed), else null
    long long in_nstride, out_nstride;
    int N, H, W, Cin, in_cstride, in_coff;
    int Cout, KH, KW, stride, pad, dil;
    int OH, OW, out_cstride, out_coff;
    int pixel_shuffle2, shift_n_mod;
    int res1_cstride, res1_coff, res2_cstride, res2_coff;
    int act, out_dtype;
    int M, K;            // GEMM sizes
    int cin_fast;        // 1 if Cin % BK == 0 (tap uniform per K tile)
    int out_fast;        // 1 if plain [M][out_cstride] addressing
    int vec_epi;         // 1: LDS-staged epilogue with 16 B stores (channel counts / offsets aligned)
    int res_vec;         // 1: residual chunks are 8/16 B aligned (vector loads)
    int splits;          // split-K factor (gridDim.y)
    int tiles_n;
    int m_begin;         // first output row of this launch (tail-split launches of the LDS-DMA kernel), else 0
    int ws_slices;       // split-K: > 0 = every split stores into its own [M][Cout] slice of ws (ordered finalize)
    int flags;           // bit 4: non-temporal f32 output stores (every launch of the product); bit 5: the activations are pre-split bf16 (hi, lo) pairs (tt_conv_desc.in_pair);
                         // bit 6: write the output in that pair format (tt_conv_desc.out_pair); < 0: split-K query (no launch)
    int res1_up_h, res1_up_w;   // > 0: res1 is a [N][res1_up_h][res1_up_w][..] map read through nearest upsampling (tt_conv_desc)
    float* out2;         // optional second, f32, row-linear copy of the output (tt_conv_desc.out2): [M][out2_cstride] at out2_coff
    int out2_cstride, out2_coff;
    long long* trace;    // measurement aid (tt_conv_set_trace): 4 wall-clock stamps (10 ns ticks) per workgroup of the LDS-DMA kernel
                         // -- entry, first K tile landed, K loop done, epilogue done -- at trace[blockIdx.x * 4]; null in the product
};

template <typename T> struct Mfma;
template <> struct Mfma<float> {
    // one 16 B vector (4 floats) per lane = 4 MFMAs of K=2 (lanes 0-31: k, lanes 32-63: k+4)
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};
template <> struct Mfma<uint16_t> {
    // one 16 B vector (8 bf16) per lane = 1 MFMA of K=16
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                    __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma<f16_t> {
    // one 16 B vector (8 halves) per lane = 1 MFMA of K=16, same rate as bf16
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// Stage one 32 x WTN block of a wave's accumulators into its private LDS region with the folded-BN scale/shift
// already applied.  In the MFMA C/D layout a lane owns ONE output channel per 32-wide column block, so the affine
// costs 2*TN registers here instead of 2*CO per lane after the transposition.
template <int TN, int WTN>
__device__ __forceinline__ void stage_scaled(float* sC, const f32x16 (&acc)[TN], const float (&sc)[TN],
                                             const float (&sh)[TN], int lane) {
    constexpr int LDC = WTN + 4;
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's reads of the previous block are done
#endif
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            sC[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDC + j * 32 + (lane & 31)] = acc[j][r] * sc[j] + sh[j];
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
}

template <int TN, int WTN>
__device__ __forceinline__ void load_scale_shift(const ConvArgs& p, float (&sc)[TN], float (&sh)[TN], int lane, int wn,
                                                 int n0) {
    const int cout_real = p.pixel_shuffle2 ? (p.Cout >> 2) : p.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WTN + j * 32 + (lane & 31);
        const int co = (col < p.Cout) ? col % cout_real : 0;
        sc[j] = p.scale ? p.scale[co] : 1.f;
        sh[j] = p.shift ? p.shift[co] : 0.f;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): settle these loads before the store-heavy passes
#endif
}

// Vector epilogue: each wave stages a 32 x WTN block of its accumulators through its PRIVATE LDS region so that
// every lane ends up with CO (4 f32 / 8 bf16) consecutive channels of one output row = one 16-byte store (full
// 128 B lines).  Everything is compile-time indexed so it stays in registers: an earlier version with runtime CO and
// by-reference lambdas put its small arrays in scratch, and every scratch reload carried an `s_waitcnt vmcnt(0)` that
// also waited for the in-flight global stores (1.4 TB/s ceiling on every memory-bound layer).
// EXT: the round-6 extras -- a second f32 copy of the output (out2), residual 1 read through nearest upsampling (res1_up_*), an f32
// residual beside 16-bit operands (res1_f32).  Compiled into the f32-operand kernels and the h2 kernel only: in the 16-bit tiles with
// 128 x 64 waves they cost the 34 registers that push the epilogue's arrays into scratch (576 B; every scratch reload waits for the
// stores in flight: the bf16 mode's convs went 40.7 -> 63.5 ms before this switch).
template <typename T, int CO, int TM, int TN, int WTM, int WTN, bool EXT>
__device__ __forceinline__ void conv_epilogue_vec(const ConvArgs& p, f32x16 (&acc)[TM][TN], float* sC, int lane, int wm,
                                                  int wn, int m0, int n0, int Mlim) {
    constexpr int LDC = WTN + 4;
    constexpr int cpr = WTN / CO;          // 16 B chunks per row of the wave's block
    constexpr int rpp = 64 / cpr;          // rows per pass
    constexpr int NPASS = 32 / rpp;
    const int cout_real = p.pixel_shuffle2 ? (p.Cout >> 2) : p.Cout;
    const int ohw = p.OH * p.OW;
    const int row_in_pass = lane / cpr;
    const int col_l = (lane % cpr) * CO;
    const int col = n0 + wn * WTN + col_l;
    int co = col, q = 0;
    if (p.pixel_shuffle2) {
        q = col / cout_real;
        co = col - q * cout_real;
    }
    const bool col_ok = col < p.Cout;
    const int act = p.act;
    float sc[TN], sh[TN];
    load_scale_shift<TN, WTN>(p, sc, sh, lane, wn, n0);
    const bool has_sn = p.shift_n != nullptr;
    const bool has_r1 = p.res1 != nullptr, has_r2 = p.res2 != nullptr;
    const bool rvec = p.res_vec != 0;
    struct F8 {
        float4 a, b;
    };
    // residual vectors of CO channels: the operand type T (8 / 4 halves, or f32: one or two float4)
    using RV = typename std::conditional<sizeof(T) == 2 && CO == 8, uint4,
                                         typename std::conditional<sizeof(T) == 2, uint2,
                                                                   typename std::conditional<CO == 8, F8, float4>::type>::type>::type;
    using RVF = typename std::conditional<CO == 8, F8, float4>::type;        // ... of an f32 residual whatever T is (flags bit 7)
    auto add_rvf = [](float (&v)[CO], const RVF& u) {
        if constexpr (CO == 8) {
            v[0] += u.a.x; v[1] += u.a.y; v[2] += u.a.z; v[3] += u.a.w;
            v[4] += u.b.x; v[5] += u.b.y; v[6] += u.b.z; v[7] += u.b.w;
        } else {
            v[0] += u.x; v[1] += u.y; v[2] += u.z; v[3] += u.w;
        }
    };
    auto add_rv = [&](float (&v)[CO], const RV& u) {
        if constexpr (sizeof(T) == 2 && CO == 8) {
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a, b;
                Pair16<T>::unpack(w[e], a, b);
                v[2 * e] += a; v[2 * e + 1] += b;
            }
        } else if constexpr (sizeof(T) == 2) {
            const uint32_t w[2] = {u.x, u.y};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float a, b;
                Pair16<T>::unpack(w[e], a, b);
                v[2 * e] += a; v[2 * e + 1] += b;
            }
        } else {
            add_rvf(v, u);
        }
    };
    // flags bit 7 (tt_conv_desc.res1_f32): residual 1 of a 16-bit-operand layer is an f32 tensor (the PAFPN's f32 sum chains beside
    // its half conv inputs)
    const bool r1_f32 = EXT && sizeof(T) == 2 && (p.flags & 128) != 0;
    const bool fast_act = (act == TT_ACT_NONE || act == TT_ACT_RELU);
    // residual 1 through nearest upsampling (PAFPN top-down path fused into the lateral conv, lss.py:301-305): output pixel
    // (n, oh, ow) reads residual pixel (n, oh * rh / OH, ow * rw / OW) -- F.interpolate(mode='nearest') with an explicit size
    const bool r1_up = EXT && p.res1_up_w > 0;
    auto res1_row = [&](int m) -> long long {
        if (!r1_up) return (long long)m;
        const int n = m / ohw, rem = m - n * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        return ((long long)n * p.res1_up_h + (oh * p.res1_up_h) / p.OH) * p.res1_up_w + (ow * p.res1_up_w) / p.OW;
    };
    auto out_offset = [&](int m, int cc, int& n) -> long long {
        if (p.out_fast) {
            if (has_sn) n = m / ohw;
            return (long long)m * p.out_cstride + p.out_coff + cc;
        }
        n = m / ohw;
        const int rem = m - n * ohw;
        int oh = rem / p.OW, ow = rem - oh * p.OW;
        int OWo = p.OW;
        if (p.pixel_shuffle2) {
            oh = 2 * oh + (q >> 1);
            ow = 2 * ow + (q & 1);
            OWo = 2 * p.OW;
        }
        return (long long)n * p.out_nstride + ((long long)oh * OWo + ow) * p.out_cstride + p.out_coff + cc;
    };
    const bool nt_store = (p.flags & 16) != 0;      // non-temporal f32 output stores (+0.3-0.5 % on the forward, profiles/r04_nt_store.txt)
    auto store_row = [&](long long o, const float (&v)[CO]) {
        if constexpr (CO == 4) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            if (nt_store) __builtin_nontemporal_store(f4v{v[0], v[1], v[2], v[3]}, reinterpret_cast<f4v*>(reinterpret_cast<float*>(p.out) + o));
            else *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + o) = make_float4(v[0], v[1], v[2], v[3]);
        } else if (sizeof(T) == 4 && (p.flags & 64)) {
            // pair-format output (tt_conv_desc.out_pair): the eight channels' bf16 hi halves go to the first 32 B of their 16-channel
            // group (second 16 B for channels 8-15 of the group), the lo halves 32 B further -- hi = rne(v), lo = rne(v - hi), the
            // consumer kernel's own split (conv_igemm_glds.hip split_frag), done once per element here
            uint4 hi, lo;
            uint32_t* hp = reinterpret_cast<uint32_t*>(&hi);
            uint32_t* lp = reinterpret_cast<uint32_t*>(&lo);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t h = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                hp[e] = h;
                lp[e] = pack_bf16x2(v[2 * e] - __uint_as_float(h << 16), v[2 * e + 1] - __uint_as_float(h & 0xffff0000u));
            }
            float* g = reinterpret_cast<float*>(p.out) + (o & ~15ll) + ((o & 8) ? 4 : 0);
            *reinterpret_cast<uint4*>(g) = hi;
            *reinterpret_cast<uint4*>(g + 8) = lo;
        } else {
            // 16-bit output: the storage type of the operands, or (f32 operands: a bf16x3 layer feeding a half-storage stage of
            // the mixed mode) the type out_dtype names
            uint4 pk;
            if constexpr (sizeof(T) == 2) {
                pk.x = Pair16<T>::pack(v[0], v[1]);
                pk.y = Pair16<T>::pack(v[2], v[3]);
                pk.z = Pair16<T>::pack(v[4], v[5]);
                pk.w = Pair16<T>::pack(v[6], v[7]);
            } else if (p.out_dtype == TT_F16) {
                pk.x = pack_f16x2(v[0], v[1]);
                pk.y = pack_f16x2(v[2], v[3]);
                pk.z = pack_f16x2(v[4], v[5]);
                pk.w = pack_f16x2(v[6], v[7]);
            } else {
                pk.x = pack_bf16x2(v[0], v[1]);
                pk.y = pack_bf16x2(v[2], v[3]);
                pk.z = pack_bf16x2(v[4], v[5]);
                pk.w = pack_bf16x2(v[6], v[7]);
            }
            *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.out) + o) = pk;
        }
    };
    // second, f32, row-linear copy (tt_conv_desc.out2)
    float* const out2 = (EXT && p.out2) ? p.out2 + p.out2_coff + co : nullptr;
    auto store_row2 = [&](long long row, const float (&v)[CO]) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        float* q2 = out2 + row * p.out2_cstride;
#pragma unroll
        for (int e = 0; e < CO; e += 4)
            __builtin_nontemporal_store(f4v{v[e], v[e + 1], v[e + 2], v[e + 3]}, reinterpret_cast<f4v*>(q2 + e));
    };
    const int cc = col_ok ? co : 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * WTM + i * 32;
        // a wave's LDS operations execute in program order and the region is private: wave-level ordering suffices
        // per-image shift (ASPP global branch, value_proj camera/level embeddings): when the 32 rows of this block
        // belong to one image -- the rule for every map wider than a few pixels -- it is just another per-column
        // constant and folds into the staging affine; otherwise the rolled path adds it row by row
        const int last_row = (mb + 31 < Mlim ? mb + 31 : Mlim - 1);
        const int img = mb / ohw;
        const bool one_image = (img == last_row / ohw);                  // wave-uniform
        const bool sn_folded = has_sn && one_image;
        const bool sn_loop = has_sn && !sn_folded;
        float shb[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) shb[j] = sh[j];
        if (sn_folded) {
            const int nimg = img % p.shift_n_mod;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cj = n0 + wn * WTN + j * 32 + (lane & 31);
                if (cj < p.Cout) shb[j] += p.shift_n[(long long)nimg * cout_real + cj % cout_real];
            }
        }
        stage_scaled<TN, WTN>(sC, acc[i], sc, shb, lane);
        const bool full_block = (mb + 32 <= Mlim) && (n0 + wn * WTN + WTN <= p.Cout);   // wave-uniform
        // row-linear output addressing: contiguous NHWC, or any image stride as long as the block stays in one image
        // (the decoder's value_proj writes each FPN level into its slice of the concatenated value tensor)
        const bool out_linear = p.out_fast || (one_image && !p.pixel_shuffle2);
        const long long obase = (p.out_fast ? 0 : (long long)img * (p.out_nstride - (long long)ohw * p.out_cstride)) +
                                p.out_coff + co;
        if (fast_act && rvec && out_linear && !sn_loop && !has_r2 && full_block) {
            // Hot path (every large conv of the camera/BEV trunks and the value projections: row-linear output, at most
            // one residual, ReLU or identity, whole 32 x WTN block in range).  The residual reads of ALL passes are issued up front and the
            // passes are branch-free LDS read -> math -> 16 B store, so the stores of consecutive passes stay in flight
            // together.  vmcnt counts stores on gfx950 and is in-order: a load issued between two stores -- or a
            // predicated store, which makes the compiler's count conservative -- serialises the passes on the store
            // latency, which capped these layers at 1.4 TB/s of output.
            auto hot = [&](auto mode_c) {
                constexpr int RM = decltype(mode_c)::value;      // 0: no residual, 1: residual of the operand type, 2: f32 residual
                constexpr bool R1 = RM != 0;
                using R1T = typename std::conditional<RM == 2, RVF, RV>::type;
                using RES = typename std::conditional<RM == 2, float, T>::type;
                // residual reads are batched ahead of the stores, at most PG passes at a time (a 128-wide f32 wave
                // block has 16 passes: all 16 residual vectors at once would not fit beside the accumulators; 32 B vectors: 4)
                constexpr int PG = (sizeof(R1T) > 16 && NPASS > 4) ? 4 : (NPASS > 8 ? 8 : NPASS);
#pragma unroll
                for (int g0 = 0; g0 < NPASS; g0 += PG) {
                    R1T r1[R1 ? PG : 1];
                    int mrow[PG];             // actual output row (sparse tile plan: tile slot -> row_perm[slot])
#pragma unroll
                    for (int q = 0; q < PG; ++q) {
                        const int m = mb + (g0 + q) * rpp + row_in_pass;
                        mrow[q] = p.row_perm ? p.row_perm[m] : m;
                    }
                    if constexpr (R1) {
#pragma unroll
                        for (int q = 0; q < PG; ++q)
                            r1[q] = *reinterpret_cast<const R1T*>(reinterpret_cast<const RES*>(p.res1) +
                                                                 res1_row(mrow[q]) * p.res1_cstride + p.res1_coff + co);
                    }
#pragma unroll
                    for (int q = 0; q < PG; ++q) {
                        const int rl = (g0 + q) * rpp + row_in_pass;
                        const int mr = mrow[q];
                        float v[CO];
#pragma unroll
                        for (int e = 0; e < CO; e += 4) {
                            const float4 t0 = *reinterpret_cast<const float4*>(sC + rl * LDC + col_l + e);
                            v[e] = t0.x; v[e + 1] = t0.y; v[e + 2] = t0.z; v[e + 3] = t0.w;
                        }
                        if constexpr (RM == 2) add_rvf(v, r1[q]);
                        else if constexpr (RM == 1) add_rv(v, r1[q]);
                        if (act == TT_ACT_RELU) {
#pragma unroll
                            for (int e = 0; e < CO; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                        }
                        store_row((long long)mr * p.out_cstride + obase, v);
                        if (out2) store_row2(mr, v);
                    }
                }
            };
            if (!has_r1) hot(std::integral_constant<int, 0>{});
            else if (EXT && sizeof(T) == 2 && r1_f32) {
                if constexpr (EXT && sizeof(T) == 2) hot(std::integral_constant<int, 2>{});
            } else hot(std::integral_constant<int, 1>{});
        } else {
            // Everything else (sigmoid/GELU/softplus, strided or pixel-shuffled outputs, per-image shifts, two
            // residuals: the small and mid-size layers): one pass at a time, rolled
#pragma unroll 1
            for (int pass = 0; pass < NPASS; ++pass) {
                const int rl = pass * rpp + row_in_pass;
                const int m = mb + rl;
                if (m >= Mlim || !col_ok) continue;
                float v[CO];
#pragma unroll
                for (int e = 0; e < CO; e += 4) {
                    const float4 t0 = *reinterpret_cast<const float4*>(sC + rl * LDC + col_l + e);
                    v[e] = t0.x; v[e + 1] = t0.y; v[e + 2] = t0.z; v[e + 3] = t0.w;
                }
                int n = 0;
                const int mo = p.row_perm ? p.row_perm[m] : m;       // sparse tile plan (row-linear output only)
                const long long o = out_offset(mo, co, n);
                if (sn_loop) {
                    const float* sn = p.shift_n + (long long)(n % p.shift_n_mod) * cout_real + co;
#pragma unroll
                    for (int e = 0; e < CO; ++e) v[e] += sn[e];
                }
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    const void* rb = which ? p.res2 : p.res1;
                    if (!rb) continue;
                    const T* rp = reinterpret_cast<const T*>(rb) +
                                  (which ? (long long)mo : res1_row(mo)) * (which ? p.res2_cstride : p.res1_cstride) +
                                  (which ? p.res2_coff : p.res1_coff) + co;
                    if (which == 0 && r1_f32) {
                        const float* rf = reinterpret_cast<const float*>(rb) + res1_row(mo) * p.res1_cstride + p.res1_coff + co;
#pragma unroll
                        for (int e = 0; e < CO; ++e) v[e] += rf[e];
                    } else if (rvec) {
                        add_rv(v, *reinterpret_cast<const RV*>(rp));
                    } else {
#pragma unroll
                        for (int e = 0; e < CO; ++e) v[e] += Elem<T>::ld(rp + e);
                    }
                }
#pragma unroll
                for (int e = 0; e < CO; ++e) v[e] = apply_act(v[e], act);
                store_row(o, v);
                if (out2) store_row2(mo, v);
            }
        }
    }
}

// Fused epilogue shared by both kernels.  C/D map of the 32x32 MFMA: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <typename T, int TM, int TN, int WTM, int WTN, bool EXT = (sizeof(T) == 4)>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[TM][TN], unsigned char* smem,
                                              int wave, int lane, int wm, int wn, int m0, int n0, int Mlim) {
    // ---- epilogue.  C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int cout_real = p.pixel_shuffle2 ? (p.Cout >> 2) : p.Cout;
    const int ohw = p.OH * p.OW;
    if (p.act == 99) return;   // profiling aid (tools/conv_microbench.py): main loop only, no epilogue
    if (p.ws) {   // split-K: raw partial sums; scale/shift/residual/activation run in splitk_finalize_kernel
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WTN + j * 32 + (lane & 31);
            if (col >= p.Cout) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < Mlim) {
                        if (p.ws_slices > 0) p.ws[((long long)blockIdx.y * p.M + m) * p.Cout + col] = acc[i][j][r];
                        else unsafeAtomicAdd(p.ws + (long long)m * p.Cout + col, acc[i][j][r]);
                    }
                }
        }
        return;
    }
    float* sC = reinterpret_cast<float*>(smem) + wave * (32 * (WTN + 4));
    __syncthreads();   // every wave is done with the K-loop tiles before LDS is reused
    if (p.vec_epi) {
        if (p.out_dtype == TT_F32 && !(p.flags & 64))
            conv_epilogue_vec<T, 4, TM, TN, WTM, WTN, EXT>(p, acc, sC, lane, wm, wn, m0, n0, Mlim);
        else
            conv_epilogue_vec<T, 8, TM, TN, WTM, WTN, EXT>(p, acc, sC, lane, wm, wn, m0, n0, Mlim);
        return;
    }
    // Scalar path (channel counts / offsets that are not 16 B friendly: the 3-channel stem input side, heads with
    // odd widths): same LDS staging, then a ROLLED loop over the wave's 32 x WTN block, lanes along channels.
    constexpr int LDC = WTN + 4;
    const int act = p.act;
    float sc[TN], sh[TN];
    load_scale_shift<TN, WTN>(p, sc, sh, lane, wn, n0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        stage_scaled<TN, WTN>(sC, acc[i], sc, sh, lane);
#pragma unroll 1
        for (int idx = lane; idx < 32 * WTN; idx += 64) {
            const int rl = idx / WTN, cl = idx - rl * WTN;
            const int m = m0 + wm * WTM + i * 32 + rl;
            const int col = n0 + wn * WTN + cl;
            if (m >= Mlim || col >= p.Cout) continue;
            int co = col, q = 0;
            if (p.pixel_shuffle2) {
                q = col / cout_real;
                co = col - q * cout_real;
            }
            float v = sC[rl * LDC + cl];
            const int n = m / ohw;
            const int mo = p.row_perm ? p.row_perm[m] : m;           // sparse tile plan (row-linear output only)
            long long o;
            if (p.out_fast) {
                o = (long long)mo * p.out_cstride + p.out_coff + co;
            } else {
                const int rem = m - n * ohw;
                int oh = rem / p.OW, ow = rem - oh * p.OW;
                int OWo = p.OW;
                if (p.pixel_shuffle2) {
                    oh = 2 * oh + (q >> 1);
                    ow = 2 * ow + (q & 1);
                    OWo = 2 * p.OW;
                }
                o = (long long)n * p.out_nstride + ((long long)oh * OWo + ow) * p.out_cstride + p.out_coff + co;
            }
            if (p.shift_n) v += p.shift_n[(n % p.shift_n_mod) * cout_real + co];
            if (p.res1)
                v += Elem<T>::ld(reinterpret_cast<const T*>(p.res1) + (long long)mo * p.res1_cstride + p.res1_coff + co);
            if (p.res2)
                v += Elem<T>::ld(reinterpret_cast<const T*>(p.res2) + (long long)mo * p.res2_cstride + p.res2_coff + co);
            v = apply_act(v, act);
            if (p.out_dtype == TT_F32)
                reinterpret_cast<float*>(p.out)[o] = v;
            else
                store16(p.out, o, v, p.out_dtype);
            if (p.out2) p.out2[(long long)mo * p.out2_cstride + p.out2_coff + co] = v;
        }
    }
}

// glds (LDS-DMA, 3-stage) variant: returns 1 if it took the launch, 0 if the shape is not covered.
int try_launch_conv_glds(ConvArgs& a, int dtype, hipStream_t st);
// bf16x3 variant of the same kernel (f32 storage, pre-split weights in a.weight): same contract.
int try_launch_conv_glds_x3(ConvArgs& a, hipStream_t st);
// split-K form of the 64-wide bf16x3 tile (few rows, long K): workspace slices it would use (0: not eligible) / the tile launch
int conv_glds_x3_splitk_slices(const ConvArgs& a);
int launch_conv_glds_x3_splitk(ConvArgs& a, hipStream_t st);
// bf16x3, 256 x 256 tile as four hand-pipelined 128 x 128 waves (csrc/conv_x3_pipe.hip); `m_tiles_limit` > 0 = tail split
int try_launch_conv_x3_pipe(ConvArgs& a, hipStream_t st, int m_tiles_limit, int bn = 256);
// run-staged sparse 3x3x3 conv (csrc/sp_conv_runs.hip; bf16x3, a.weight = pre-split weights): same contract.
int try_launch_sp_conv_runs(ConvArgs& a, hipStream_t st);
// "h2" arithmetic (csrc/conv_h2.hip): IEEE-half activations x f16 (hi, lo) weight pairs in a.weight, two MFMAs per product.
// Returns 1 if it took the launch, 0 if the shape is outside its contract (dense, Cin % 64 == 0, KH*KW <= 31).
int try_launch_conv_h2(ConvArgs& a, hipStream_t st);
// latency-bound small-M variant (32x32 tile, intra-block split-K): same contract.
int try_launch_conv_small(ConvArgs& a, int dtype, hipStream_t st);

}  // namespace tt
