// Shared definitions of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_igemm_glds.hip).
#pragma once
#include "tt_common.h"

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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



// Fused epilogue shared by both kernels.  C/D map of the 32x32 MFMA: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <typename T, int TM, int TN, int WTM, int WTN>
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
                    if (m < Mlim) unsafeAtomicAdd(p.ws + (long long)m * p.Cout + col, acc[i][j][r]);
                }
        }
        return;
    }
    if (p.vec_epi) {
        // Stage each wave's 32 x WTN accumulator block through LDS (the tile buffers are free after
        // the K loop) so that every lane stores 16 contiguous bytes of one output row: full 128 B
        // lines instead of 64 B half-lines per MFMA register, and residuals are read the same way.
        constexpr int LDC = WTN + 4;
        float* sC = reinterpret_cast<float*>(smem) + wave * (32 * LDC);
        const int CO = (p.out_dtype == TT_F32) ? 4 : 8;           // channels per lane
        const int cpr = WTN / CO;                                  // chunks per row
        const int rpp = 64 / cpr;                                  // rows per pass
        const int row_in_pass = lane / cpr;
        const int col_l = (lane % cpr) * CO;
        const int col = n0 + wn * WTN + col_l;
        int co = col, q = 0;
        if (p.pixel_shuffle2) {
            q = col / cout_real;
            co = col - q * cout_real;
        }
        const bool col_ok = col < p.Cout;
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = (col_ok && e < CO && p.scale) ? p.scale[co + e] : 1.f;
            sh[e] = (col_ok && e < CO && p.shift) ? p.shift[co + e] : 0.f;
        }
        __syncthreads();   // every wave is done with the K-loop tiles before LDS is reused
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // Each wave stages through its PRIVATE region, and a wave's LDS operations execute in program
            // order, so wave-level ordering is enough.  A workgroup barrier here would also wait for the
            // previous block-row's GLOBAL STORES (vmcnt counts stores on gfx950) and serialise the epilogue
            // on store latency.
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sC[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDC + j * 32 + (lane & 31)] = acc[i][j][r];
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            for (int rl = row_in_pass; rl < 32; rl += rpp) {
                const int m = m0 + wm * WTM + i * 32 + rl;
                if (m >= Mlim || !col_ok) continue;
                float v[8];
                const float4 t0 = *reinterpret_cast<const float4*>(sC + rl * LDC + col_l);
                v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w;
                if (CO == 8) {
                    const float4 t1 = *reinterpret_cast<const float4*>(sC + rl * LDC + col_l + 4);
                    v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
                }
                long long o;
                int n = 0;
                if (p.out_fast) {
                    o = (long long)m * p.out_cstride + p.out_coff + co;
                } else {
                    n = m / ohw;
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
                const float* sn = nullptr;
                if (p.shift_n) {
                    if (p.out_fast) n = m / ohw;
                    sn = p.shift_n + (long long)(n % p.shift_n_mod) * cout_real + co;
                }
                const T* r1 = p.res1 ? reinterpret_cast<const T*>(p.res1) + (long long)m * p.res1_cstride + p.res1_coff + co : nullptr;
                const T* r2 = p.res2 ? reinterpret_cast<const T*>(p.res2) + (long long)m * p.res2_cstride + p.res2_coff + co : nullptr;
                float rr[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) rr[e] = 0.f;
                auto add_res = [&](const T* rp) {   // 8 / 16 B vector loads of the residual chunk
                    if (sizeof(T) == 2) {
                        if (CO == 8) {
                            const uint4 u = *reinterpret_cast<const uint4*>(rp);
                            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                rr[2 * e] += __uint_as_float(w[e] << 16);
                                rr[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                            }
                        } else {
                            const uint2 u = *reinterpret_cast<const uint2*>(rp);
                            rr[0] += __uint_as_float(u.x << 16); rr[1] += __uint_as_float(u.x & 0xffff0000u);
                            rr[2] += __uint_as_float(u.y << 16); rr[3] += __uint_as_float(u.y & 0xffff0000u);
                        }
                    } else {
                        const float4 f0 = *reinterpret_cast<const float4*>(rp);
                        rr[0] += f0.x; rr[1] += f0.y; rr[2] += f0.z; rr[3] += f0.w;
                        if (CO == 8) {
                            const float4 f1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(rp) + 4);
                            rr[4] += f1.x; rr[5] += f1.y; rr[6] += f1.z; rr[7] += f1.w;
                        }
                    }
                };
                if (p.res_vec) {
                    if (r1) add_res(r1);
                    if (r2) add_res(r2);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e < CO) {
                            if (r1) rr[e] += Elem<T>::ld(r1 + e);
                            if (r2) rr[e] += Elem<T>::ld(r2 + e);
                        }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (e < CO) {
                        float x = v[e] * sc[e] + sh[e];
                        if (sn) x += sn[e];
                        x += rr[e];
                        v[e] = apply_act(x, p.act);
                    }
                }
                if (CO == 4) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    uint4 pk;
                    pk.x = pack_bf16x2(v[0], v[1]);
                    pk.y = pack_bf16x2(v[2], v[3]);
                    pk.z = pack_bf16x2(v[4], v[5]);
                    pk.w = pack_bf16x2(v[6], v[7]);
                    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.out) + o) = pk;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WTN + j * 32 + (lane & 31);
        if (col >= p.Cout) continue;
        int co = col, q = 0;
        if (p.pixel_shuffle2) {
            q = col / cout_real;
            co = col - q * cout_real;
        }
        const float sc = p.scale ? p.scale[co] : 1.f;
        const float sh = p.shift ? p.shift[co] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= Mlim) continue;
                float v = acc[i][j][r] * sc + sh;
                long long o;
                int n = 0;
                if (p.out_fast) {
                    o = (long long)m * p.out_cstride + p.out_coff + co;
                } else {
                    n = m / ohw;
                    const int rem = m - n * ohw;
                    int oh = rem / p.OW, ow = rem - oh * p.OW;
                    int OWo = p.OW;
                    if (p.pixel_shuffle2) {
                        oh = 2 * oh + (q >> 1);
                        ow = 2 * ow + (q & 1);
                        OWo = 2 * p.OW;
                    }
                    o = (long long)n * p.out_nstride + ((long long)oh * OWo + ow) * p.out_cstride +
                        p.out_coff + co;
                }
                if (p.shift_n) {
                    if (p.out_fast) n = m / ohw;
                    v += p.shift_n[(n % p.shift_n_mod) * cout_real + co];
                }
                if (p.res1)
                    v += Elem<T>::ld(reinterpret_cast<const T*>(p.res1) +
                                     (long long)m * p.res1_cstride + p.res1_coff + co);
                if (p.res2)
                    v += Elem<T>::ld(reinterpret_cast<const T*>(p.res2) +
                                     (long long)m * p.res2_cstride + p.res2_coff + co);
                v = apply_act(v, p.act);
                if (p.out_dtype == TT_F32)
                    reinterpret_cast<float*>(p.out)[o] = v;
                else
                    reinterpret_cast<uint16_t*>(p.out)[o] = f32_to_bf16(v);
            }
        }
    }
}

// glds (LDS-DMA, 3-stage) variant: returns 1 if it took the launch, 0 if the shape is not covered.
int try_launch_conv_glds(ConvArgs& a, int dtype, hipStream_t st);
// latency-bound small-M variant (32x32 tile, intra-block split-K): same contract.
int try_launch_conv_small(ConvArgs& a, int dtype, hipStream_t st);

}  // namespace tt
