// Row-batched MLP chains of the look-and-predict decoder in ONE launch (tt_mlp_chain).
//
// The reference decoder (thinktwice_decoder.py:26-260, multi_scale_deformable_attn_function.py:197-344) is a long
// sequence of nn.Linear layers over a few hundred to a few thousand rows: query_linear -> sampling_offsets /
// attention_weights, ffn.w_1 -> w_2 (+ residual), output_proj, mlp -> traj / ctrl offset heads, the coarse heads.
// Launched one kernel per layer they are ~5 us each of pure latency.  Here a workgroup owns 32 rows and walks the
// whole chain: intermediates never leave LDS, only the outputs a later kernel needs are written to HBM.
//
// Arithmetic: "bf16x3" on the bf16 MFMA (the exact-f32 MFMA is 16x slower): every f32 operand is a bf16 (hi, lo) pair,
// a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi accumulated in f32 (relative error ~1e-5 per dot product).  Weights arrive
// pre-split from the host ("pair format": per 16 K elements 64 B = [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15],
// weights.py::split_pairs_x3); intermediates are split ONCE when a stage writes them to LDS, in the same format, so a
// wave's A fragment is two ds_read_b128 with no conversion; the chain input is split when it is loaded from HBM.
//
// Layout of a stage: out[32 rows][N] = A[32][K] * W^T.  v_mfma_f32_32x32x16_bf16: lane l supplies row / column
// (l & 31) and the 8-element K chunk (l >> 5) of a 16-element K step.  The 8 waves of the workgroup take the 32-column
// blocks of N round-robin, up to 4 accumulator blocks per wave at a time, so one A fragment feeds up to 12 MFMAs.
// Weights stream from L2 straight into the B operand registers (each element is used once per workgroup: staging
// them through LDS would only add a round trip), double-buffered one K step ahead.
#include "conv_common.h"

namespace tt {

constexpr int kChainMaxStages = 8;
constexpr int kChainWaves = 8;
constexpr int kChainNBW = 4;      // accumulator blocks per wave per pass

struct ChainStage {
    const void* w;        // pair-format weights [N padded to 32][Kp]
    const float* bias;    // [N] or null
    const float* res;     // optional residual rows (global f32): v += res[m * res_stride + res_coff + n]
    const float* side;    // optional extra input columns (global f32 [R][side_stride]): v += sum_j side[m][j] * side_w[n][j]
    const float* side_w;  // [N][side_k] f32
    float* out;           // optional global f32 output: out[m * out_stride + out_coff + n]
    int K, Kp, N, act;
    int in_sel;           // -1: the chain input x; s >= 0: the LDS output of stage s
    int res_stride, res_coff, side_stride, side_k;
    int out_stride, out_coff;
    int lds_off;          // byte offset of this stage's LDS output (pair format); -1: not kept
    int lds_stride;       // bytes per row of that buffer (Kp_next * 4 + 16: an odd number of 16 B slots, conflict-free)
};

struct ChainArgs {
    const float* x;
    long long R;
    int x_stride, nstages;
    ChainStage st[kChainMaxStages];
};

__device__ __forceinline__ void split8(const float (&x)[8], uint4& hi, uint4& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = pack_bf16x2(x[2 * e], x[2 * e + 1]);
        const float r0 = x[2 * e] - __uint_as_float(h[e] << 16);
        const float r1 = x[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u);
        l[e] = pack_bf16x2(r0, r1);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

__device__ __forceinline__ void mfma3(const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl, f32x16& c) {
    Mfma<uint16_t>::run(al, bh, c);     // small terms first
    Mfma<uint16_t>::run(ah, bl, c);
    Mfma<uint16_t>::run(ah, bh, c);
}

// store one f32 value as a (hi, lo) bf16 pair at element (row, col) of a pair-format LDS buffer
__device__ __forceinline__ void lds_store_pair(unsigned char* buf, int stride, int row, int col, float v) {
    const uint16_t hi = f32_to_bf16(v);
    const uint16_t lo = f32_to_bf16(v - bf16_to_f32(hi));
    unsigned char* p = buf + (size_t)row * stride + (col >> 4) * 64 + ((col >> 3) & 1) * 16 + (col & 7) * 2;
    *reinterpret_cast<uint16_t*>(p) = hi;
    *reinterpret_cast<uint16_t*>(p + 32) = lo;
}

__global__ __launch_bounds__(kChainWaves * 64) void mlp_chain_kernel(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long m0 = (long long)blockIdx.x * 32;
    const long long m = m0 + r;
    const bool row_ok = m < a.R;

    for (int s = 0; s < a.nstages; ++s) {
        const ChainStage& S = a.st[s];
        const int NB = (S.N + 31) >> 5;
        const int nsteps = S.Kp >> 4;
        const bool from_x = S.in_sel < 0;
        const unsigned char* abuf = from_x ? nullptr : smem + a.st[S.in_sel].lds_off;
        const int astride = from_x ? 0 : a.st[S.in_sel].lds_stride;
        const float* xrow = a.x + (row_ok ? m : 0) * a.x_stride;

        for (int nb0 = wave; nb0 < NB; nb0 += kChainWaves * kChainNBW) {
            // this pass: blocks nb0, nb0 + 8, nb0 + 16, nb0 + 24 (those < NB)
            const unsigned char* bptr[kChainNBW];
            bool live[kChainNBW];
#pragma unroll
            for (int q = 0; q < kChainNBW; ++q) {
                const int nb = nb0 + q * kChainWaves;
                live[q] = nb < NB;
                bptr[q] = reinterpret_cast<const unsigned char*>(S.w) +
                          ((size_t)((live[q] ? nb : nb0) * 32 + r) * S.Kp) * 4 + h * 16;
            }
            f32x16 acc[kChainNBW];
#pragma unroll
            for (int q = 0; q < kChainNBW; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;

            auto load_a = [&](int ks, uint4& ah, uint4& al) {
                if (from_x) {
                    const int k0 = ks * 16 + h * 8;
                    float v[8];
                    if (row_ok && k0 + 8 <= S.K) {
                        const float4 t0 = *reinterpret_cast<const float4*>(xrow + k0);
                        const float4 t1 = *reinterpret_cast<const float4*>(xrow + k0 + 4);
                        v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w;
                        v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (row_ok && k0 + e < S.K) ? xrow[k0 + e] : 0.f;
                    }
                    split8(v, ah, al);
                } else {
                    const unsigned char* p = abuf + (size_t)r * astride + ks * 64 + h * 16;
                    ah = *reinterpret_cast<const uint4*>(p);
                    al = *reinterpret_cast<const uint4*>(p + 32);
                }
            };
            auto load_b = [&](int ks, uint4 (&bh)[kChainNBW], uint4 (&bl)[kChainNBW]) {
#pragma unroll
                for (int q = 0; q < kChainNBW; ++q) {
                    if (live[q]) {
                        const unsigned char* p = bptr[q] + (size_t)ks * 64;
                        bh[q] = *reinterpret_cast<const uint4*>(p);
                        bl[q] = *reinterpret_cast<const uint4*>(p + 32);
                    }
                }
            };

            uint4 ah[2], al[2], bh[2][kChainNBW], bl[2][kChainNBW];
            load_a(0, ah[0], al[0]);
            load_b(0, bh[0], bl[0]);
            for (int ks = 0; ks < nsteps; ks += 2) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int k = ks + half;
                    if (k < nsteps) {
                        if (k + 1 < nsteps) {
                            load_a(k + 1, ah[half ^ 1], al[half ^ 1]);
                            load_b(k + 1, bh[half ^ 1], bl[half ^ 1]);
                        }
#pragma unroll
                        for (int q = 0; q < kChainNBW; ++q)
                            if (live[q]) mfma3(ah[half], al[half], bh[half][q], bl[half][q], acc[q]);
                    }
                }
            }

            // epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
#pragma unroll
            for (int q = 0; q < kChainNBW; ++q) {
                if (!live[q]) continue;
                const int n = (nb0 + q * kChainWaves) * 32 + r;
                const bool n_ok = n < S.N;
                const float bias = (S.bias && n_ok) ? S.bias[n] : 0.f;
                float sw[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) sw[j] = (S.side && n_ok && j < S.side_k) ? S.side_w[n * S.side_k + j] : 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rr = (i & 3) + 8 * (i >> 2) + 4 * h;
                    const long long mr = m0 + rr;
                    const bool ok = n_ok && mr < a.R;
                    float v = acc[q][i] + bias;
                    if (S.side && ok) {
                        const float* sp = S.side + mr * S.side_stride;
                        for (int j = 0; j < S.side_k; ++j) v += sp[j] * sw[j];
                    }
                    if (S.res && ok) v += S.res[mr * S.res_stride + S.res_coff + n];
                    v = apply_act(v, S.act);
                    if (!ok) v = 0.f;
                    if (S.out && ok) S.out[mr * S.out_stride + S.out_coff + n] = v;
                    if (S.lds_off >= 0) lds_store_pair(smem + S.lds_off, S.lds_stride, rr, n, v);
                }
            }
        }
        __syncthreads();   // this stage's LDS output is complete (and its input buffer free) before the next stage
    }
}

}  // namespace tt

using namespace tt;

extern "C" int tt_mlp_chain(const float* x, long long R, int x_stride, int nstages, const tt_chain_stage* st,
                            void* stream) {
    TT_REQUIRE(x && st && R > 0 && nstages >= 1 && nstages <= kChainMaxStages, "tt_mlp_chain: bad arguments");
    TT_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && x_stride % 4 == 0, "tt_mlp_chain: x must be 16 B aligned rows");
    ChainArgs a;
    a.x = x; a.R = R; a.x_stride = x_stride; a.nstages = nstages;
    // LDS plan: an output is kept while a LATER stage reads it; buffers are placed first-fit over the live ranges
    int last_use[kChainMaxStages];
    for (int s = 0; s < nstages; ++s) last_use[s] = -1;
    for (int s = 0; s < nstages; ++s) {
        TT_REQUIRE(st[s].in_sel < s, "tt_mlp_chain: stage %d reads stage %d", s, st[s].in_sel);
        if (st[s].in_sel >= 0) last_use[st[s].in_sel] = s;
    }
    size_t off[kChainMaxStages], len[kChainMaxStages];
    size_t total = 0;
    for (int s = 0; s < nstages; ++s) {
        const tt_chain_stage& d = st[s];
        TT_REQUIRE(d.w && d.K > 0 && d.N > 0 && d.Kp % 16 == 0 && d.Kp >= d.K && d.K % 4 == 0,
                   "tt_mlp_chain: stage %d: K=%d Kp=%d N=%d", s, d.K, d.Kp, d.N);
        TT_REQUIRE((reinterpret_cast<uintptr_t>(d.w) & 15) == 0, "tt_mlp_chain: stage %d weights unaligned", s);
        TT_REQUIRE(d.side_k >= 0 && d.side_k <= 8 && (!d.side || d.side_w), "tt_mlp_chain: stage %d side input", s);
        TT_REQUIRE(d.in_sel >= 0 || x_stride >= d.K, "tt_mlp_chain: stage %d: x rows shorter than K", s);
        TT_REQUIRE(d.in_sel < 0 || st[d.in_sel].N == d.K, "tt_mlp_chain: stage %d: K=%d but stage %d has N=%d", s, d.K,
                   d.in_sel, d.in_sel >= 0 ? st[d.in_sel].N : 0);
        ChainStage& S = a.st[s];
        S.w = d.w; S.bias = d.bias; S.res = d.res; S.side = d.side; S.side_w = d.side_w; S.out = d.out;
        S.K = d.K; S.Kp = d.Kp; S.N = d.N; S.act = d.act; S.in_sel = d.in_sel;
        S.res_stride = d.res_stride; S.res_coff = d.res_coff; S.side_stride = d.side_stride; S.side_k = d.side_k;
        S.out_stride = d.out_stride; S.out_coff = d.out_coff;
        S.lds_off = -1; S.lds_stride = 0;
        off[s] = 0; len[s] = 0;
        if (last_use[s] >= 0) {
            const int np = (d.N + 15) / 16 * 16;          // the consumer's Kp
            S.lds_stride = np * 4 + 16;
            len[s] = (size_t)32 * S.lds_stride;
            // first fit against the buffers still live at stage s (those with last_use >= s, i.e. read at or after s)
            size_t pos = 0;
            bool moved = true;
            while (moved) {
                moved = false;
                for (int t = 0; t < s; ++t) {
                    if (len[t] == 0 || last_use[t] < s) continue;
                    if (pos < off[t] + len[t] && off[t] < pos + len[s]) {
                        pos = (off[t] + len[t] + 15) / 16 * 16;
                        moved = true;
                    }
                }
            }
            off[s] = pos;
            S.lds_off = (int)pos;
            if (pos + len[s] > total) total = pos + len[s];
        }
    }
    for (int s = 0; s < nstages; ++s)
        if (a.st[s].in_sel >= 0)
            TT_REQUIRE(a.st[a.st[s].in_sel].lds_stride == a.st[s].Kp * 4 + 16, "tt_mlp_chain: stage %d Kp mismatch", s);
    TT_REQUIRE(total <= 160 * 1024, "tt_mlp_chain: intermediates need %zu B of LDS (> 160 KiB)", total);
    if (total < 16) total = 16;
    static size_t attr = 0;
    if (total > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        attr = 160 * 1024;
    }
    const unsigned blocks = (unsigned)((R + 31) / 32);
    hipLaunchKernelGGL(mlp_chain_kernel, dim3(blocks), dim3(kChainWaves * 64), total, (hipStream_t)stream, a);
    return check_launch("tt_mlp_chain");
}
