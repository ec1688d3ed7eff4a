// Backward of the channel-last convolution (training step): weight gradient.
//
//   dW[co][kh][kw][ci] = sum over output pixels m = (n, oh, ow) of  dY[m][co] * X[n][oh*s - p + kh*d][ow*s - p + kw*d][ci]
//
// (torch.nn.functional.conv2d backward w.r.t. the weight; the reference gets it from autograd through cuDNN.)
//
// A GEMM whose K dimension is the pixel index, i.e. the ROW index of both channel-last operands.  The exact-f32 MFMA
// v_mfma_f32_32x32x2_f32 takes ONE f32 per lane per operand -- A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]
// -- which is exactly a coalesced 128 B row segment of 32 consecutive channels of pixel k for both dY (i = output channel) and X
// (j = input channel): the operands go from global memory straight into the MFMA operand registers, no transpose, no LDS, no
// precision split.  It runs at the f32 MFMA rate (157 TF peak, 1/16 of bf16): gradients are exact f32 products with f32
// accumulation, the first correct form of the training step; a bf16x3 variant would need an LDS transpose of both operands.
//
// Work split: a workgroup of 4 waves owns one (64 output channels) x (64 input channels) tile of ONE filter tap over a
// contiguous range of output rows (n, oh); its waves take the rows round-robin and their accumulators are added through LDS.
// gridDim.y row ranges ("splits") write their partial tiles to a workspace that a second kernel adds in split order:
// deterministic, no atomics.
#include "conv_common.h"

namespace tt {

struct WgradArgs {
    const float* x;        // [N][H][W][x_cstride], channels [x_coff, x_coff + Cin)
    const float* dy;       // [N][OH][OW][dy_cstride], channels [dy_coff, dy_coff + Cout)
    float* ws;             // [splits][Cout][KH*KW][cin_p] partial sums
    int N, H, W, Cin, x_cstride, x_coff;
    int OH, OW, Cout, dy_cstride, dy_coff;
    int KH, KW, stride, pad, dil, cin_p;
    int ci_tiles, rows_per_split;
    int xcd_tiles;         // > 0: number of real tiles of an XCD-remapped grid (see the kernel); 0: identity mapping
};

// v if ok else 0, as a MULTIPLY on an always-executed load of a clamped (valid, finite) element: a select or a bit mask is
// folded back into select(ok, load, 0), which the code generator turns into a divergent branch around the load with a wait
// inside -- serialising the loads of a group.  Used by the wide kernel, whose software pipeline needs its loads in flight under
// the MFMAs (measured on the 64 x 64 kernel: the conditional loads are FASTER there, 55 vs 36 TF/s, so it keeps them)
__device__ __forceinline__ float masked(float v, bool ok) { return v * (ok ? 1.f : 0.f); }

// f32 x 8 -> bf16 (hi, lo) operand pair: hi = bf16(x) round-to-nearest-even, lo = bf16(x - hi)
__device__ __forceinline__ void split8(const float (&x)[8], uint4& hi, uint4& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = pack_bf16x2(x[2 * e], x[2 * e + 1]);
        const float q0 = x[2 * e] - __uint_as_float(h[e] << 16);           // exact in f32
        const float q1 = x[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u);
        l[e] = pack_bf16x2(q0, q1);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

constexpr int kWgUnroll = 4;       // pixel pairs in flight per wave (each: 2 + 2 dword loads, 4 MFMAs)

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    __shared__ float red[3][64 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, k = lane >> 5;                 // channel within a 32-block, pixel of the pair
    const int taps = a.KH * a.KW;
    int t = blockIdx.x;
    if (a.xcd_tiles) {      // workgroup i runs on XCD i % 8: give every XCD a contiguous run of tiles, so that the taps of one
        t = (t & 7) * (gridDim.x >> 3) + (t >> 3);      // (co, ci) tile -- which read the same dy rows and shifted x rows --
        if (t >= a.xcd_tiles) return;                   // share that XCD's L2 (the grid is padded to a multiple of 8)
    }
    const int tap = t % taps;
    t /= taps;
    const int ci0 = (t % a.ci_tiles) * 64, co0 = (t / a.ci_tiles) * 64;
    const int kh = tap / a.KW, kw = tap % a.KW;
    const int rows = a.N * a.OH;
    const int r_begin = blockIdx.y * a.rows_per_split;
    const int r_end = min(rows, r_begin + a.rows_per_split);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // channel validity of this lane's two co / ci blocks (zero operands outside the tensor's channels)
    const bool co_ok[2] = {co0 + c < a.Cout, co0 + 32 + c < a.Cout};
    const bool ci_ok[2] = {ci0 + c < a.Cin, ci0 + 32 + c < a.Cin};
    const int npairs = (a.OW + 1) >> 1;
    for (int r = r_begin + wave; r < r_end; r += 4) {
        const int n = r / a.OH, oh = r - n * a.OH;
        const int ih = oh * a.stride - a.pad + kh * a.dil;
        if (ih < 0 || ih >= a.H) continue;                   // this tap reads padding on the whole row (wave-uniform)
        const float* dyrow = a.dy + ((long long)r * a.OW) * a.dy_cstride + a.dy_coff + co0 + c;
        const float* xrow = a.x + (((long long)n * a.H + ih) * a.W) * a.x_cstride + a.x_coff + ci0 + c;
        for (int p0 = 0; p0 < npairs; p0 += kWgUnroll) {
            float av[kWgUnroll][2], bv[kWgUnroll][2];
#pragma unroll
            for (int u = 0; u < kWgUnroll; ++u) {
                const int ow = 2 * (p0 + u) + k;
                const int iw = ow * a.stride - a.pad + kw * a.dil;
                const bool m_ok = ow < a.OW;
                const bool x_ok = m_ok && iw >= 0 && iw < a.W;
                const float* dp = dyrow + (long long)(m_ok ? ow : 0) * a.dy_cstride;
                const float* xp = xrow + (long long)(x_ok ? iw : 0) * a.x_cstride;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    av[u][b] = (m_ok && co_ok[b]) ? dp[32 * b] : 0.f;
                    bv[u][b] = (x_ok && ci_ok[b]) ? xp[32 * b] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < kWgUnroll; ++u)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
        }
    }

    // add the four waves' tiles (fixed order: wave 0 + 1 + 2 + 3), then store this split's partial tile
    // C/D map of the 32x32 MFMA: col = lane & 31 (input channel), row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) (output channel)
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * k;
                    red[wave - 1][row * 64 + 32 * j + c] = acc[i][j][e];
                }
    }
    __syncthreads();
    if (wave > 0) return;
    float* ws = a.ws + (long long)blockIdx.y * a.Cout * taps * a.cin_p;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * k;
                const int col = 32 * j + c;
                float v = acc[i][j][e];
#pragma unroll
                for (int w = 0; w < 3; ++w) v += red[w][row * 64 + col];
                if (co0 + row < a.Cout && ci0 + col < a.cin_p)
                    ws[((long long)(co0 + row) * taps + tap) * a.cin_p + ci0 + col] = (ci0 + col < a.Cin) ? v : 0.f;
            }
}

// Wide variant for layers with >= 128 channels on a side: every WAVE owns a (32 BI) x (32 BJ) tile of dW[.][tap][.] (up to
// 128 x 128 = 256 accumulator registers) over its share of the rows, so a pixel pair costs BI + BJ dword loads per BI * BJ
// MFMAs (0.5 per MFMA at 4 x 4, against 1.0 in the 64 x 64 kernel above: that kernel is bound by its operand loads, not by the
// f32 MFMA pipe).  No cross-wave reduction in the workgroup: each wave stores its partial tile as its own split slice
// (blockIdx.y * 4 + wave) and the ordered reduce kernel adds them -- still deterministic.
template <int BI, int BJ>
__global__ __launch_bounds__(256) void conv_wgrad_wide_kernel(const WgradArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, k = lane >> 5;
    const int taps = a.KH * a.KW;
    int t = blockIdx.x;
    const int tap = t % taps;
    t /= taps;
    const int ci0 = (t % a.ci_tiles) * (32 * BJ), co0 = (t / a.ci_tiles) * (32 * BI);
    const int kh = tap / a.KW, kw = tap % a.KW;
    const int rows = a.N * a.OH;
    const int r_begin = blockIdx.y * a.rows_per_split;
    const int r_end = min(rows, r_begin + a.rows_per_split);

    f32x16 acc[BI][BJ];
#pragma unroll
    for (int i = 0; i < BI; ++i)
#pragma unroll
        for (int j = 0; j < BJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bool co_ok[BI], ci_ok[BJ];
    int cco[BI], cci[BJ];                                    // clamped channels: unconditional loads, masked afterwards
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        co_ok[i] = co0 + 32 * i + c < a.Cout;
        cco[i] = min(co0 + 32 * i + c, a.Cout - 1);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        ci_ok[j] = ci0 + 32 * j + c < a.Cin;
        cci[j] = min(ci0 + 32 * j + c, a.Cin - 1);
    }
    const int npairs = (a.OW + 1) >> 1;
    // pixel pairs in flight (each: BI + BJ loads, BI * BJ MFMAs): more of them where a pair is little MFMA work
    constexpr int U = BI * BJ >= 8 ? 2 : (BI * BJ >= 4 ? 4 : 8);
    for (int r = r_begin + wave; r < r_end; r += 4) {
        const int n = r / a.OH, oh = r - n * a.OH;
        const int ih = oh * a.stride - a.pad + kh * a.dil;
        if (ih < 0 || ih >= a.H) continue;
        const float* dyrow = a.dy + ((long long)r * a.OW) * a.dy_cstride + a.dy_coff;
        const float* xrow = a.x + (((long long)n * a.H + ih) * a.W) * a.x_cstride + a.x_coff;
        // software pipeline: the operands of the next U pairs are in flight while the MFMAs of the current ones run (one wave
        // per SIMD at this accumulator size: nothing else hides the load latency)
        auto load_pairs = [&](int p0, float (&av)[U][BI], float (&bv)[U][BJ]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ow = 2 * (p0 + u) + k;
                const int iw = ow * a.stride - a.pad + kw * a.dil;
                const bool m_ok = ow < a.OW;
                const bool x_ok = m_ok && iw >= 0 && iw < a.W;
                const float* dp = dyrow + (long long)(m_ok ? ow : 0) * a.dy_cstride;
                const float* xp = xrow + (long long)(x_ok ? iw : 0) * a.x_cstride;
#pragma unroll
                for (int i = 0; i < BI; ++i) {
                    av[u][i] = masked(dp[cco[i]], m_ok && co_ok[i]);
                }
#pragma unroll
                for (int j = 0; j < BJ; ++j) {
                    bv[u][j] = masked(xp[cci[j]], x_ok && ci_ok[j]);
                }
            }
        };
        float av[U][BI], bv[U][BJ], an[U][BI], bn[U][BJ];
        load_pairs(0, av, bv);
        for (int p0 = 0; p0 < npairs; p0 += U) {
            load_pairs(p0 + U, an, bn);                      // (past the row: every pair is masked to zero operands)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < BI; ++i)
#pragma unroll
                    for (int j = 0; j < BJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int i = 0; i < BI; ++i) av[u][i] = an[u][i];
#pragma unroll
                for (int j = 0; j < BJ; ++j) bv[u][j] = bn[u][j];
            }
        }
    }
    float* ws = a.ws + ((long long)blockIdx.y * 4 + wave) * a.Cout * taps * a.cin_p;
#pragma unroll
    for (int i = 0; i < BI; ++i)
#pragma unroll
        for (int j = 0; j < BJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * k;
                const int col = 32 * j + c;
                if (co0 + row < a.Cout && ci0 + col < a.cin_p)
                    ws[((long long)(co0 + row) * taps + tap) * a.cin_p + ci0 + col] = (ci0 + col < a.Cin) ? acc[i][j][e] : 0.f;
            }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-staged bf16x3 weight gradient for layers with >= 64 channels on both sides (round 3).
//
// dW[co][tap][ci] = sum over pixels p of dy[p][co] * x[p + tap][ci] is a GEMM whose contraction index (the pixel) is the SLOW
// index of both operands in memory (channel-last rows).  The f32 MFMA form above feeds one pixel pair per instruction
// straight from global memory (K = 2 in 64 cycles: 53 TF/s over an iteration).  v_mfma_f32_32x32x16_bf16 contracts 16 pixels in
// 32 cycles, but a lane must then hold 8 CONSECUTIVE PIXELS of one channel -- a transposed read.  Here a workgroup (four
// waves, 2 x 2, each a (32 BI) x (32 BJ) tile of dW[.][tap][.]) stages, per 32 output pixels of an image row, the dy rows
// [32][64 BI channels] and the tap-shifted x rows [32][64 BJ] in LDS as they lie in memory (1 KiB LDS-DMA pieces, one or two
// pixels each, padding / stride / dilation / channel tails by per-lane source selection with a zero page; double buffered)
// and every lane gathers its operand with 8 ds_read_b32 down a column (lanes = consecutive channels: conflict-free), splits
// it into bf16 (hi, lo) and issues dy_lo*x_hi + dy_hi*x_lo + dy_hi*x_hi (the forward's bf16x3 arithmetic, ~2^-17 per
// product, f32 accumulation).  Per 16 pixels a wave does 8 (BI + BJ) LDS reads and 3 BI BJ MFMAs: 64 + 48 at 128 x 128.
// One partial tile per (workgroup, row split), added by the ordered reduce kernel: deterministic.
// ---------------------------------------------------------------------------------------------------------------------
template <int OFF>
__device__ __forceinline__ float lds_rd_f32(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int ROWB>
__device__ __forceinline__ void lds_col8(unsigned addr, float (&v)[8]) {      // 8 rows of one column, ROWB bytes apart
    v[0] = lds_rd_f32<0 * ROWB>(addr); v[1] = lds_rd_f32<1 * ROWB>(addr);
    v[2] = lds_rd_f32<2 * ROWB>(addr); v[3] = lds_rd_f32<3 * ROWB>(addr);
    v[4] = lds_rd_f32<4 * ROWB>(addr); v[5] = lds_rd_f32<5 * ROWB>(addr);
    v[6] = lds_rd_f32<6 * ROWB>(addr); v[7] = lds_rd_f32<7 * ROWB>(addr);
}

template <int BI, int BJ>
__global__ __launch_bounds__(256, 1) void conv_wgrad_lds_kernel(const WgradArgs a, const float* __restrict__ zp) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr int TA = 64 * BI, TB = 64 * BJ;             // channels per staged row = the workgroup's tile of dW
    constexpr int PX = 32;                                // output pixels per stage (two 16-pixel MFMA steps)
    constexpr int RA = TA * 4, RB = TB * 4;               // row bytes
    constexpr int ABYTES = PX * RA, BBYTES = PX * RB, STAGE = ABYTES + BBYTES;
    constexpr int NIA = ABYTES / 1024 / 4, NIB = BBYTES / 1024 / 4;     // 1 KiB DMA pieces per wave and stage
    constexpr int LPA = RA / 16, LPB = RB / 16;           // lanes per pixel row inside a piece (64: one pixel, 32: two)
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wy = wave >> 1, wx = wave & 1;
    const int c = lane & 31, kg = lane >> 5;
    const int taps = a.KH * a.KW;
    int t = blockIdx.x;
    const int tap = t % taps;
    t /= taps;
    const int ci0 = (t % a.ci_tiles) * TB, co0 = (t / a.ci_tiles) * TA;
    const int kh = tap / a.KW, kw = tap % a.KW;
    const int rows = a.N * a.OH;
    const int r_begin = blockIdx.y * a.rows_per_split;
    const int r_end = min(rows, r_begin + a.rows_per_split);
    const int segs = (a.OW + PX - 1) / PX;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)wsm;

    f32x16 acc[BI][BJ];
#pragma unroll
    for (int i = 0; i < BI; ++i)
#pragma unroll
        for (int j = 0; j < BJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- stage walker: (row r, 32-pixel segment); rows whose tap row falls outside the image are skipped
    struct It {
        int r, seg;
    };
    auto row_ih = [&](int r) {
        const int n = r / a.OH, oh = r - n * a.OH;
        return oh * a.stride - a.pad + kh * a.dil;
    };
    auto seek = [&](It& it) {
        while (it.r < r_end) {
            const int ih = row_ih(it.r);
            if (ih >= 0 && ih < a.H) break;
            ++it.r;
        }
        it.seg = 0;
    };
    auto next = [&](It& it) {
        if (++it.seg < segs) return;
        ++it.r;
        seek(it);
    };
    // per-lane constants of the DMA pieces: piece i of an operand covers 64 / LP pixels, lane = (pixel in piece, 16 B slot)
    const int pa_px = lane / LPA, pa_ch = (lane % LPA) * 4;
    const int pb_px = lane / LPB, pb_ch = (lane % LPB) * 4;
    const bool a_ch_ok = co0 + pa_ch < a.Cout;            // (Cout, Cin multiples of 4: a 16 B slot is all in or all out)
    const bool b_ch_ok = ci0 + pb_ch < a.Cin;
    auto issue = [&](const It& it, int buf) {
        const int n = it.r / a.OH;
        const int ih = row_ih(it.r);
        const float* dyrow = a.dy + ((long long)it.r * a.OW) * a.dy_cstride + a.dy_coff + co0 + pa_ch;
        const float* xrow = a.x + (((long long)n * a.H + ih) * a.W) * a.x_cstride + a.x_coff + ci0 + pb_ch;
        const unsigned sa = lds_base + (unsigned)buf * STAGE, sb = sa + ABYTES;
        const int ow0 = it.seg * PX;
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            const int i = wave_s + 4 * j;                                  // piece
            const int ow = ow0 + i * (64 / LPA) + pa_px;
            const float* src = (ow < a.OW && a_ch_ok) ? dyrow + (long long)ow * a.dy_cstride : zp;
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(uintptr_t)(sa + (unsigned)i * 1024u), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NIB; ++j) {
            const int i = wave_s + 4 * j;
            const int ow = ow0 + i * (64 / LPB) + pb_px;
            const int iw = ow * a.stride - a.pad + kw * a.dil;
            const float* src = (ow < a.OW && iw >= 0 && iw < a.W && b_ch_ok) ? xrow + (long long)iw * a.x_cstride : zp;
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(uintptr_t)(sb + (unsigned)i * 1024u), 16, 0, 0);
        }
    };
    // fragment columns of this lane: row block i of dy / column block j of x, pixel group kg (8 pixels) of a 16-pixel step
    unsigned fa[BI], fb[BJ];
#pragma unroll
    for (int i = 0; i < BI; ++i) fa[i] = lds_base + (unsigned)(kg * 8 * RA + (wy * 32 * BI + 32 * i + c) * 4);
#pragma unroll
    for (int j = 0; j < BJ; ++j) fb[j] = lds_base + (unsigned)(ABYTES + kg * 8 * RB + (wx * 32 * BJ + 32 * j + c) * 4);

    It ic{r_begin, 0};
    seek(ic);
    It ii = ic;
    if (ic.r < r_end) {
        issue(ii, 0);
        next(ii);
    }
    int buf = 0;
    while (ic.r < r_end) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");        // stage `buf` is published; everyone is done with the other buffer
        if (ii.r < r_end) {
            issue(ii, buf ^ 1);
            next(ii);
        }
        const unsigned boff = (unsigned)buf * STAGE;
#pragma unroll
        for (int st = 0; st < PX / 16; ++st) {
            uint4 ah[BI], al[BI], bh[BJ], bl[BJ];
            float ra[BI][8], rb[BJ][8];
#pragma unroll
            for (int i = 0; i < BI; ++i) lds_col8<RA>(fa[i] + boff + (unsigned)(st * 16 * RA), ra[i]);
#pragma unroll
            for (int j = 0; j < BJ; ++j) lds_col8<RB>(fb[j] + boff + (unsigned)(st * 16 * RB), rb[j]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < BI; ++i) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(ra[i][e]));
                split8(ra[i], ah[i], al[i]);
            }
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(rb[j][e]));
                split8(rb[j], bh[j], bl[j]);
            }
            // term-major: consecutive MFMAs write different accumulators; small terms first
#pragma unroll
            for (int i = 0; i < BI; ++i)
#pragma unroll
                for (int j = 0; j < BJ; ++j) Mfma<uint16_t>::run(al[i], bh[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < BI; ++i)
#pragma unroll
                for (int j = 0; j < BJ; ++j) Mfma<uint16_t>::run(ah[i], bl[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < BI; ++i)
#pragma unroll
                for (int j = 0; j < BJ; ++j) Mfma<uint16_t>::run(ah[i], bh[j], acc[i][j]);
        }
        next(ic);
        buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // C/D map of the 32x32 MFMA: col = lane & 31 (input channel), row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) (output channel)
    float* ws = a.ws + (long long)blockIdx.y * a.Cout * taps * a.cin_p;
#pragma unroll
    for (int i = 0; i < BI; ++i)
#pragma unroll
        for (int j = 0; j < BJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = co0 + wy * 32 * BI + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * kg;
                const int col = ci0 + wx * 32 * BJ + 32 * j + c;
                if (row < a.Cout && col < a.cin_p)
                    ws[((long long)row * taps + tap) * a.cin_p + col] = (col < a.Cin) ? acc[i][j][e] : 0.f;
            }
#endif
}

// dw[i] = (accumulate ? dw[i] : 0) + sum over the partial slices ws[s][i], in a FIXED order: 32 elements per workgroup, eight
// thread groups walk every 8th slice, then the eight group sums are added in index order (deterministic; with up to a few
// thousand slices of a small tile a single serial loop per element was the slowest part of the launch)
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ ws, long long n, int splits,
                                                                int accumulate, float* __restrict__ dw) {
    __shared__ float sh[8][33];
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long long i = (long long)blockIdx.x * 32 + e;
    float v = 0.f;
    if (i < n)
        for (int s = g; s < splits; s += 8) v += ws[(long long)s * n + i];
    sh[g][e] = v;
    __syncthreads();
    if (g == 0 && i < n) {
        v = accumulate ? dw[i] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += sh[k][e];
        dw[i] = v;
    }
}

// Weight gradient of the gathered (sparse) convolution  y[m] = sum_t W_t x[nbr[m][t]]  (spconv SubMConv3d / SparseConv3d
// as tt_conv2d_fwd runs them: rulebook rows, -1 = no input):  dW[co][t][ci] = sum_m dconv[m][co] * x[nbr[m][t]][ci].
// Same structure as conv_wgrad_kernel: the pixel pair of an MFMA step is two consecutive output rows, whose input rows come
// from the rulebook (one index load per lane half and step, then the same coalesced 128 B channel segments).
struct GatherWgradArgs {
    const float* x; const float* dy; const int* nbr; const int* m_dev; float* ws;
    long long M;
    int Cin, x_cstride, Cout, dy_cstride, taps, cin_p, ci_tiles;
    long long pairs_per_split;
    int xcd_tiles;
    int skip_empty;        // per-wave-tile kernel: skip row pairs without an input at the tap (TT_GATHER_WGRAD_SKIP=0: never)
};

__global__ __launch_bounds__(256) void gather_wgrad_kernel(const GatherWgradArgs a) {
    __shared__ float red[3][64 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, k = lane >> 5;
    int t = blockIdx.x;
    if (a.xcd_tiles) {      // XCD-contiguous tile order, as in conv_wgrad_kernel
        t = (t & 7) * (gridDim.x >> 3) + (t >> 3);
        if (t >= a.xcd_tiles) return;
    }
    const int tap = t % a.taps;
    t /= a.taps;
    const int ci0 = (t % a.ci_tiles) * 64, co0 = (t / a.ci_tiles) * 64;
    const long long Mlive = a.m_dev ? min(a.M, (long long)*a.m_dev) : a.M;
    const long long npairs = (Mlive + 1) >> 1;
    // the LIVE rows (device count) are divided over the splits -- not the capacity M, which would leave most splits idle
    const long long pps = (npairs + gridDim.y - 1) / gridDim.y;
    const long long p_begin = (long long)blockIdx.y * pps;
    const long long p_end = min(npairs, p_begin + pps);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const bool co_ok[2] = {co0 + c < a.Cout, co0 + 32 + c < a.Cout};
    const bool ci_ok[2] = {ci0 + c < a.Cin, ci0 + 32 + c < a.Cin};
    for (long long p0 = p_begin + (long long)wave * kWgUnroll; p0 < p_end; p0 += 4 * kWgUnroll) {
        float av[kWgUnroll][2], bv[kWgUnroll][2];
#pragma unroll
        for (int u = 0; u < kWgUnroll; ++u) {
            const long long m = 2 * (p0 + u) + k;
            const bool m_ok = (p0 + u) < p_end && m < Mlive;
            const int j = m_ok ? a.nbr[m * a.taps + tap] : -1;
            const float* dp = a.dy + (m_ok ? m : 0) * a.dy_cstride + co0 + c;
            const float* xp = a.x + (long long)(j >= 0 ? j : 0) * a.x_cstride + ci0 + c;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                av[u][b] = (m_ok && co_ok[b]) ? dp[32 * b] : 0.f;
                bv[u][b] = (j >= 0 && ci_ok[b]) ? xp[32 * b] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < kWgUnroll; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * k;
                    red[wave - 1][row * 64 + 32 * j + c] = acc[i][j][e];
                }
    }
    __syncthreads();
    if (wave > 0) return;
    float* ws = a.ws + (long long)blockIdx.y * a.Cout * a.taps * a.cin_p;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * k;
                const int col = 32 * j + c;
                float v = acc[i][j][e];
#pragma unroll
                for (int w = 0; w < 3; ++w) v += red[w][row * 64 + col];
                if (co0 + row < a.Cout && ci0 + col < a.cin_p)
                    ws[((long long)(co0 + row) * a.taps + tap) * a.cin_p + ci0 + col] = (ci0 + col < a.Cin) ? v : 0.f;
            }
}

// Transposed rulebook of a strided sparse convolution: inv[j][t] = the output row m with nbr[m][t] == j (at most one: the
// output coordinate is determined by the input coordinate and the tap), -1 otherwise.  `inv` must be pre-filled with -1.
__global__ __launch_bounds__(256) void sp_inverse_rulebook_kernel(const int* __restrict__ nbr, const int* __restrict__ m_dev,
                                                                  long long M, int taps, int* __restrict__ inv) {
    const long long Mlive = min(M, (long long)*m_dev);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= Mlive * taps) return;
    const int j = nbr[i];
    if (j >= 0) inv[(long long)j * taps + (int)(i % taps)] = (int)(i / taps);
}

// backward of sp_to_dense: grows[r][c] += gdense[b][y][x][c * D + z] for the live rows
__global__ __launch_bounds__(256) void sp_from_dense_kernel(const float* __restrict__ gdense, const int* __restrict__ coords,
                                                            const int* __restrict__ rows_n, long long max_rows, int C, int D,
                                                            int H, int W, float* __restrict__ grows) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long r = t / C;
    const int c = (int)(t % C);
    if (r >= *rows_n || r >= max_rows) return;
    const int b = coords[r * 4], z = coords[r * 4 + 1], y = coords[r * 4 + 2], x = coords[r * 4 + 3];
    grows[r * C + c] += gdense[(((long long)b * H + y) * W + x) * ((long long)C * D) + (long long)c * D + z];
}

// Backward of the fused conv epilogue  y = act(scale[c] * conv + shift[c] + res1 + res2)  (folded BatchNorm affine /
// bias, residual adds, activation; conv_common.h's forward epilogue).  From dy and the SAVED OUTPUT y:
//   g      = dy * act'(pre)           act' from y: ReLU y > 0; sigmoid y (1 - y); none 1 (GELU / softplus need pre: refused)
//   dconv  = g * scale[c]             -> the dy of tt_conv2d_wgrad / conv2d_dgrad
//   dres   = g                        -> gradient of every residual input (optional output)
//   dshift[c] = sum_m g,   dscale[c] = sum_m g * conv = sum_m g * (pre - shift[c] - res) / scale[c]
// with pre recovered from y (ReLU: pre = y wherever g != 0; sigmoid: logit(y); none: y).  The per-channel sums go through
// per-workgroup partials added in index order (deterministic).  BatchNorm parameters follow on the host side from
// scale = gamma / sigma, shift = beta - mu * scale:  dgamma = (dscale - mu * dshift) / sigma,  dbeta = dshift.
constexpr int kEpiBlocks = 512;

struct EpiBwdArgs {
    const float* dy; const float* y; const float* res1; const float* res2;
    const float* scale; const float* shift;
    float* dconv; float* dres; float* dres2; float* partial;     // partial: [kEpiBlocks][2][C]
    const int* m_dev;      // sparse layers: device count of live rows (rows beyond it are not touched), or null
    const float* pre;      // optional dense [M][C] pre-activation (scale*conv + shift + res): any activation
    const float* zraw;     // optional dense [M][C] raw convolution output: dscale = sum g * zraw (no division by scale)
    long long M;
    int C, dy_cstride, dy_coff, y_cstride, y_coff, r1_cstride, r1_coff, r2_cstride, r2_coff;
    int dconv_cstride, dconv_coff, dres_cstride, dres_coff, dres2_cstride, dres2_coff, dres_accumulate, act;
};

__global__ __launch_bounds__(256) void conv_epilogue_bwd_kernel(const EpiBwdArgs a) {
    __shared__ float red[2][256];
    const int tid = threadIdx.x;
    // thread (ty, tx): channel tx, tx + TX, ...; rows ty, ty + TY, ... of this workgroup's row range
    const int TX = a.C < 256 ? a.C : 256, TY = 256 / TX;
    const int tx = tid % TX, ty = tid / TX;
    const long long Mlive = a.m_dev ? min(a.M, (long long)*a.m_dev) : a.M;
    const long long rows_per = (a.M + gridDim.x - 1) / gridDim.x;
    const long long r0 = (long long)blockIdx.x * rows_per, r1 = min(Mlive, r0 + rows_per);
    for (int c0 = 0; c0 < a.C; c0 += TX) {             // uniform trip count: the loop body holds barriers
        const int c = c0 + tx;
        const bool c_ok = c < a.C;
        const float sc = (c_ok && a.scale) ? a.scale[c] : 1.f, sh = (c_ok && a.shift) ? a.shift[c] : 0.f;
        float s_g = 0.f, s_gx = 0.f;
        if (ty < TY && c_ok) {
            // one element: activation derivative, per-channel sums (in row order), the gradient stores
            auto element = [&](long long m, float yv, float g, float pre_in, float r, float zr) {
                float pre = yv;
                if (a.pre) {            // activation derivative from the recomputed pre-activation
                    pre = pre_in;
                    if (a.act == TT_ACT_RELU) g = pre > 0.f ? g : 0.f;
                    else if (a.act == TT_ACT_SIGMOID) g *= yv * (1.f - yv);
                    else if (a.act == TT_ACT_GELU)
                        g *= 0.5f * (1.f + erff(pre * 0.70710678118654752440f)) +
                             pre * 0.39894228040143267794f * expf(-0.5f * pre * pre);
                    else if (a.act == TT_ACT_SOFTPLUS) g *= pre > 20.f ? 1.f : 1.f / (1.f + expf(-pre));
                    else if (a.act == TT_ACT_SOFTPLUS_CLAMP)
                        g *= yv > 1e-3f ? (pre > 20.f ? 1.f : 1.f / (1.f + expf(-pre))) : 0.f;
                } else if (a.act == TT_ACT_RELU) {
                    g = yv > 0.f ? g : 0.f;
                } else if (a.act == TT_ACT_SIGMOID) {
                    g *= yv * (1.f - yv);
                    pre = logf(yv / (1.f - yv));
                }
                s_g += g;
                // dscale: from the raw convolution output when the caller recomputed it (layers with a vanishing or zero
                // BatchNorm gamma, sigmoid epilogues: the reconstruction below is 0/0 or inf there); else reconstructed from
                // the saved output, a zero scale contributing nothing instead of NaN
                if (a.zraw) s_gx += g * zr;
                else if (a.scale) s_gx += (sc != 0.f && g != 0.f) ? g * ((pre - sh - r) / sc) : 0.f;
                a.dconv[m * a.dconv_cstride + a.dconv_coff + c] = g * sc;
                if (a.dres) {
                    float* d = a.dres + m * a.dres_cstride + a.dres_coff + c;
                    *d = a.dres_accumulate ? *d + g : g;
                }
                if (a.dres2) {
                    float* d = a.dres2 + m * a.dres2_cstride + a.dres2_coff + c;
                    *d = a.dres_accumulate ? *d + g : g;
                }
            };
            auto fetch = [&](long long m, float& yv, float& g, float& pre_in, float& r, float& zr) {
                yv = a.y[m * a.y_cstride + a.y_coff + c];
                g = a.dy[m * a.dy_cstride + a.dy_coff + c];
                pre_in = a.pre ? a.pre[m * a.C + c] : 0.f;
                zr = a.zraw ? a.zraw[m * a.C + c] : 0.f;
                r = 0.f;
                if (a.res1) r += a.res1[m * a.r1_cstride + a.r1_coff + c];
                if (a.res2) r += a.res2[m * a.r2_cstride + a.r2_coff + c];
            };
            // four rows per trip with all their loads issued before the first use (a thread walks up to thousands of rows:
            // one row at a time leaves the loop bound by load latency); the elements are still consumed in row order, so
            // the sums are the same as a one-row loop's
            long long m = r0 + ty;
            for (; m + 3LL * TY < r1; m += 4LL * TY) {
                float yv[4], g[4], pv[4], rv[4], zv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) fetch(m + (long long)u * TY, yv[u], g[u], pv[u], rv[u], zv[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) element(m + (long long)u * TY, yv[u], g[u], pv[u], rv[u], zv[u]);
            }
            for (; m < r1; m += TY) {
                float yv, g, pv, rv, zv;
                fetch(m, yv, g, pv, rv, zv);
                element(m, yv, g, pv, rv, zv);
            }
        }
        // add the TY row lanes of this channel (fixed order)
        __syncthreads();
        red[0][tid] = s_g;
        red[1][tid] = s_gx;
        __syncthreads();
        if (ty == 0 && c_ok) {
            float t0 = 0.f, t1 = 0.f;
            for (int q = 0; q < TY; ++q) {
                t0 += red[0][q * TX + tx];
                t1 += red[1][q * TX + tx];
            }
            a.partial[((long long)blockIdx.x * 2 + 0) * a.C + c] = t0;
            a.partial[((long long)blockIdx.x * 2 + 1) * a.C + c] = t1;
        }
    }
}

// Per-wave-tile form of gather_wgrad_kernel (see conv_wgrad_wide_kernel): 32 channels a side for the 16 / 32-channel levels of
// the sparse encoder (a 64-wide tile multiplies mostly zeros there), 128 for its 128-channel levels.  Every wave walks its
// share of the live row pairs and stores its partial tile as its own slice (blockIdx.y * 4 + wave) of the ordered reduction.
template <int BI, int BJ>
__global__ __launch_bounds__(256) void gather_wgrad_wide_kernel(const GatherWgradArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, k = lane >> 5;
    int t = blockIdx.x;
    const int tap = t % a.taps;
    t /= a.taps;
    const int ci0 = (t % a.ci_tiles) * (32 * BJ), co0 = (t / a.ci_tiles) * (32 * BI);
    const long long Mlive = a.m_dev ? min(a.M, (long long)*a.m_dev) : a.M;
    const long long npairs = (Mlive + 1) >> 1;
    const long long pps = (npairs + gridDim.y - 1) / gridDim.y;
    const long long p_begin = (long long)blockIdx.y * pps;
    const long long p_end = min(npairs, p_begin + pps);
    f32x16 acc[BI][BJ];
#pragma unroll
    for (int i = 0; i < BI; ++i)
#pragma unroll
        for (int j = 0; j < BJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bool co_ok[BI], ci_ok[BJ];
    int cco[BI], cci[BJ];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        co_ok[i] = co0 + 32 * i + c < a.Cout;
        cco[i] = min(co0 + 32 * i + c, a.Cout - 1);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        ci_ok[j] = ci0 + 32 * j + c < a.Cin;
        cci[j] = min(ci0 + 32 * j + c, a.Cin - 1);
    }
    constexpr int U = BI * BJ >= 8 ? 2 : 4;                  // row pairs in flight per wave
    for (long long p0 = p_begin + (long long)wave * U; p0 < p_end; p0 += 4 * U) {
        float av[U][BI], bv[U][BJ];
        int jr[U];
        long long mc[U];
        bool m_ok[U], any[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                      // the rulebook entries of the U row pairs (U loads in flight)
            const long long m = 2 * (p0 + u) + k;
            m_ok[u] = (p0 + u) < p_end && m < Mlive;
            mc[u] = m_ok[u] ? m : 0;
            jr[u] = a.nbr[mc[u] * a.taps + tap];
        }
        // A row pair neither of whose rows has an input at this tap contributes nothing: skip its loads and MFMAs (wave-
        // uniform test).  With the ~6.5 of 27 taps a LiDAR voxel has, more than half of the pairs go.
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool x_ok = m_ok[u] && jr[u] >= 0;
            any[u] = !a.skip_empty || __builtin_amdgcn_ballot_w64(x_ok) != 0;
            if (any[u]) {
                const float* dp = a.dy + mc[u] * a.dy_cstride;
                const float* xp = a.x + (long long)(x_ok ? jr[u] : 0) * a.x_cstride;
#pragma unroll
                for (int i = 0; i < BI; ++i) av[u][i] = masked(dp[cco[i]], m_ok[u] && co_ok[i]);
#pragma unroll
                for (int j = 0; j < BJ; ++j) bv[u][j] = masked(xp[cci[j]], x_ok && ci_ok[j]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (any[u]) {
#pragma unroll
                for (int i = 0; i < BI; ++i)
#pragma unroll
                    for (int j = 0; j < BJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
            }
    }
    float* ws = a.ws + ((long long)blockIdx.y * 4 + wave) * a.Cout * a.taps * a.cin_p;
#pragma unroll
    for (int i = 0; i < BI; ++i)
#pragma unroll
        for (int j = 0; j < BJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * k;
                const int col = 32 * j + c;
                if (co0 + row < a.Cout && ci0 + col < a.cin_p)
                    ws[((long long)(co0 + row) * a.taps + tap) * a.cin_p + ci0 + col] = (ci0 + col < a.Cin) ? acc[i][j][e] : 0.f;
            }
}

// per-channel sums of the workgroups' partials [blocks][2][C], in a fixed order (deterministic): 16 channels per workgroup,
// 16 thread groups each walk every 16th partial row, then the 16 group sums are added in index order
__global__ __launch_bounds__(256) void conv_epilogue_bwd_finish_kernel(const float* __restrict__ partial, int blocks, int C,
                                                                       int accumulate, float* __restrict__ dscale,
                                                                       float* __restrict__ dshift) {
    __shared__ float sh[2][16][17];
    const int cx = threadIdx.x & 15, by = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    float g = 0.f, gx = 0.f;
    if (c < C)
        for (int b = by; b < blocks; b += 16) {
            g += partial[((long long)b * 2 + 0) * C + c];
            gx += partial[((long long)b * 2 + 1) * C + c];
        }
    sh[0][by][cx] = g;
    sh[1][by][cx] = gx;
    __syncthreads();
    if (by == 0 && c < C) {
        g = 0.f;
        gx = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            g += sh[0][k][cx];
            gx += sh[1][k][cx];
        }
        if (dshift) dshift[c] = (accumulate ? dshift[c] : 0.f) + g;
        if (dscale) dscale[c] = (accumulate ? dscale[c] : 0.f) + gx;
    }
}

// TT_WGRAD_XCD=1: XCD-contiguous workgroup -> tile order.  Measured (MI355X, 3x3 layers of the camera trunk at batch 8): no gain
// where the grid is large (256 -> 256: 65.0 vs 63.8 TF/s) and a loss where it is small (64 -> 64: 36.3 vs 55.1 TF/s; a
// training iteration 1701 vs 1368 ms), so the default is the identity mapping.
// tile of one wave in 32-channel blocks per side: 4 (128 channels) where the side has >= 128, 1 where it has <= 32 (the
// segmentation / depth heads, the stem's 3 input channels: a 64-wide tile would multiply mostly zeros), else 2.  2 x 2 is the
// 64 x 64 workgroup-tile kernel, everything else the per-wave-tile kernel.  TT_WGRAD_WIDE=0: always 2 x 2.
static void wgrad_blocks(int Cout, int Cin, int* bi, int* bj) {
    *bi = Cout >= 128 ? 4 : (Cout <= 32 ? 1 : 2);
    *bj = Cin >= 128 ? 4 : (Cin <= 32 ? 1 : 2);
}

static int wgrad_splits(int N, int OH, int Cout, int Cin, int taps) {
    int bi, bj;
    wgrad_blocks(Cout, Cin, &bi, &bj);
    const long long tiles = (long long)div_up(Cout, 32 * bi) * div_up(Cin, 32 * bj) * taps;
    // aim at >= 4 workgroups per CU, 2 for the 128-wide wave tiles (one wave per SIMD each, and every split costs four
    // partial slices)
    long long s = ((bi * bj >= 8 ? 2LL : 4LL) * kNumCU + tiles - 1) / tiles;
    const int rows = N * OH;
    if (s > rows / 4) s = rows / 4;                          // every wave of a workgroup gets at least one row
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    return (int)s;
}

static const float* wgrad_zero_page() {      // source of the LDS-DMA pieces that lie outside the image / the channel window
    static void* z = nullptr;
    if (!z) {
        if (hipMalloc(&z, 256) != hipSuccess) return nullptr;
        (void)hipMemset(z, 0, 256);
    }
    return (const float*)z;
}

// partial-sum slices in the workspace: one per split, x 4 in the wide kernel (one per wave)
static int wgrad_slices(int N, int OH, int Cout, int Cin, int taps) {
    int bi, bj;
    wgrad_blocks(Cout, Cin, &bi, &bj);
    return wgrad_splits(N, OH, Cout, Cin, taps) * ((bi == 2 && bj == 2) ? 1 : 4);
}

}  // namespace tt

using namespace tt;

extern "C" long long tt_conv2d_wgrad_workspace_bytes(int N, int OH, int Cout, int Cin, int cin_pad, int KH, int KW) {
    return (long long)wgrad_slices(N, OH, Cout, Cin, KH * KW) * Cout * KH * KW * cin_pad * 4;
}

static int wgrad_run(const float* x, int N, int H, int W, int Cin, int x_cstride, int x_coff, const float* dy,
                     int OH, int OW, int Cout, int dy_cstride, int dy_coff, int KH, int KW, int stride, int pad,
                     int dil, int cin_pad, int accumulate, float* dw, void* workspace, long long workspace_bytes,
                     void* stream, bool x3) {
    TT_REQUIRE(x && dy && dw && workspace && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && OH > 0 && OW > 0 &&
                   KH > 0 && KW > 0 && stride > 0 && dil > 0 && cin_pad >= Cin,
               "tt_conv2d_wgrad: bad argument");
    TT_REQUIRE(x_cstride >= x_coff + Cin && dy_cstride >= dy_coff + Cout, "tt_conv2d_wgrad: channel window outside the row");
    const int taps = KH * KW;
    const int splits = wgrad_splits(N, OH, Cout, Cin, taps);
    TT_REQUIRE(workspace_bytes >= tt_conv2d_wgrad_workspace_bytes(N, OH, Cout, Cin, cin_pad, KH, KW),
               "tt_conv2d_wgrad: workspace too small");
    WgradArgs a;
    a.x = x; a.dy = dy; a.ws = (float*)workspace;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.x_cstride = x_cstride; a.x_coff = x_coff;
    a.OH = OH; a.OW = OW; a.Cout = Cout; a.dy_cstride = dy_cstride; a.dy_coff = dy_coff;
    a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.dil = dil; a.cin_p = cin_pad;
    hipStream_t st = (hipStream_t)stream;
    // >= 64 channels on both sides: the LDS-staged bf16x3 kernel (iteration 962 -> 820 ms against the f32-MFMA wave-tile form)
    if (x3 && Cout >= 64 && Cin >= 64 && Cout % 4 == 0 && Cin % 4 == 0 && x_cstride % 4 == 0 && x_coff % 4 == 0 &&
        dy_cstride % 4 == 0 && dy_coff % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0) {
        const float* zp = wgrad_zero_page();
        // the kernel walks image rows in 32-pixel segments: a 1x1 / stride-1 / unpadded layer (linear layers over rows: OW = 1)
        // is the same sum over ANY regrouping of its pixels, so short rows are merged into pseudo-rows of >= 128 pixels;
        // other layers with rows shorter than 16 pixels (the 1 x 9 grouped deformable-conv GEMM) keep the f32 kernels
        WgradArgs al = a;
        if (KH == 1 && KW == 1 && stride == 1 && pad == 0 && OW < 128) {
            const long long prow = (long long)N * OH;
            long long m = (128 + OW - 1) / OW;
            while (m < prow && prow % m) ++m;
            if (m <= prow && prow % m == 0) {
                al.N = 1;
                al.OW = al.W = (int)(OW * m);
                al.OH = al.H = (int)(prow / m);
            }
        }
        const int cap = wgrad_slices(N, OH, Cout, Cin, taps);         // partial-sum slices the caller's workspace holds
        if (zp && al.OW >= 16) {
            WgradArgs& a = al;      // (shadows the caller's view for this launch)
            const int N = a.N, OH = a.OH;
            const int BI = Cout >= 256 ? 4 : (Cout >= 128 ? 2 : 1), BJ = Cin >= 256 ? 4 : (Cin >= 128 ? 2 : 1);
            a.ci_tiles = div_up(cin_pad, 64 * BJ);
            const long long tiles = (long long)div_up(Cout, 64 * BI) * a.ci_tiles * taps;
            const int rows = N * OH;
            const long long per_cu = BI * BJ >= 16 ? 2 : (BI * BJ >= 4 ? 3 : 4);     // resident workgroups (LDS) x ~1.5 rounds
            long long sp = (per_cu * kNumCU + tiles - 1) / tiles;
            if (sp > cap) sp = cap;
            if (sp > rows) sp = rows;
            if (sp < 1) sp = 1;
            a.rows_per_split = div_up(rows, (int)sp);
            const int nsplit = div_up(rows, a.rows_per_split);
            a.xcd_tiles = 0;
            const dim3 grid((unsigned)tiles, (unsigned)nsplit);
            const size_t smem = (size_t)2 * 32 * (64 * BI + 64 * BJ) * 4;
#define TT_WGL(BI_, BJ_)                                                                                                  \
    do {                                                                                                                  \
        static bool attr = false;                                                                                         \
        if (!attr) {                                                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_lds_kernel<BI_, BJ_>),                     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                             \
            attr = true;                                                                                                  \
        }                                                                                                                 \
        hipLaunchKernelGGL((conv_wgrad_lds_kernel<BI_, BJ_>), grid, dim3(256), smem, st, a, zp);                          \
    } while (0)
            switch (BI * 8 + BJ) {
                case 4 * 8 + 4: TT_WGL(4, 4); break;
                case 4 * 8 + 2: TT_WGL(4, 2); break;
                case 4 * 8 + 1: TT_WGL(4, 1); break;
                case 2 * 8 + 4: TT_WGL(2, 4); break;
                case 2 * 8 + 2: TT_WGL(2, 2); break;
                case 2 * 8 + 1: TT_WGL(2, 1); break;
                case 1 * 8 + 4: TT_WGL(1, 4); break;
                case 1 * 8 + 2: TT_WGL(1, 2); break;
                default: TT_WGL(1, 1); break;
            }
#undef TT_WGL
            const long long n = (long long)Cout * taps * cin_pad;
            hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)div_up(n, 32)), dim3(256), 0, st, (const float*)workspace,
                               n, nsplit, accumulate, dw);
            return check_launch("tt_conv2d_wgrad");
        }
    }
    int bi, bj;
    wgrad_blocks(Cout, Cin, &bi, &bj);
    a.ci_tiles = div_up(cin_pad, 32 * bj);
    a.rows_per_split = div_up(N * OH, splits);
    const unsigned tiles = (unsigned)(div_up(Cout, 32 * bi) * a.ci_tiles * taps);
    int slices = splits;
    if (bi == 2 && bj == 2) {
        // (an XCD-contiguous tile order measured slower here: 36 vs 55 TF/s on 64 -> 64; identity mapping)
        a.xcd_tiles = 0;
        hipLaunchKernelGGL(conv_wgrad_kernel, dim3(tiles, (unsigned)splits), dim3(256), 0, st, a);
    } else {
        a.xcd_tiles = 0;
        slices = splits * 4;
        const dim3 grid(tiles, (unsigned)splits);
        switch (bi * 8 + bj) {
#define TT_WG(BI_, BJ_) \
    case BI_ * 8 + BJ_: hipLaunchKernelGGL((conv_wgrad_wide_kernel<BI_, BJ_>), grid, dim3(256), 0, st, a); break;
            TT_WG(4, 4) TT_WG(4, 2) TT_WG(2, 4) TT_WG(4, 1) TT_WG(1, 4) TT_WG(2, 1) TT_WG(1, 2) TT_WG(1, 1)
#undef TT_WG
            default: TT_REQUIRE(false, "tt_conv2d_wgrad: no kernel for wave tile %d x %d", bi, bj);
        }
    }
    const long long n = (long long)Cout * taps * cin_pad;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)div_up(n, 32)), dim3(256), 0, st, (const float*)workspace, n,
                       slices, accumulate, dw);
    return check_launch("tt_conv2d_wgrad");
}

extern "C" int tt_conv2d_wgrad(const float* x, int N, int H, int W, int Cin, int x_cstride, int x_coff, const float* dy,
                               int OH, int OW, int Cout, int dy_cstride, int dy_coff, int KH, int KW, int stride, int pad,
                               int dil, int cin_pad, int accumulate, float* dw, void* workspace, long long workspace_bytes,
                               void* stream) {
    return wgrad_run(x, N, H, W, Cin, x_cstride, x_coff, dy, OH, OW, Cout, dy_cstride, dy_coff, KH, KW, stride, pad, dil,
                     cin_pad, accumulate, dw, workspace, workspace_bytes, stream, false);
}

extern "C" int tt_conv2d_wgrad_x3(const float* x, int N, int H, int W, int Cin, int x_cstride, int x_coff, const float* dy,
                                  int OH, int OW, int Cout, int dy_cstride, int dy_coff, int KH, int KW, int stride, int pad,
                                  int dil, int cin_pad, int accumulate, float* dw, void* workspace,
                                  long long workspace_bytes, void* stream) {
    return wgrad_run(x, N, H, W, Cin, x_cstride, x_coff, dy, OH, OW, Cout, dy_cstride, dy_coff, KH, KW, stride, pad, dil,
                     cin_pad, accumulate, dw, workspace, workspace_bytes, stream, true);
}

extern "C" long long tt_conv_epilogue_bwd_workspace_bytes(int C) { return (long long)kEpiBlocks * 2 * C * 4; }

extern "C" int tt_conv_epilogue_bwd(const float* dy, int dy_cstride, int dy_coff, const float* y, int y_cstride, int y_coff,
                                    const float* res1, int res1_cstride, int res1_coff, const float* res2, int res2_cstride,
                                    int res2_coff, const float* scale, const float* shift, long long M, int C, int act,
                                    float* dconv, int dconv_cstride, int dconv_coff, float* dres, int dres_cstride,
                                    int dres_coff, float* dres2, int dres2_cstride, int dres2_coff, int dres_accumulate,
                                    float* dscale, float* dshift, int accumulate, const int* m_dev_or_null,
                                    const float* pre_or_null, const float* conv_raw_or_null, void* workspace,
                                    long long workspace_bytes, void* stream) {
    TT_REQUIRE(dy && y && dconv && workspace && M > 0 && C > 0, "tt_conv_epilogue_bwd: bad argument");
    TT_REQUIRE(pre_or_null || act == TT_ACT_NONE || act == TT_ACT_RELU || act == TT_ACT_SIGMOID,
               "tt_conv_epilogue_bwd: activation %d needs the pre-activation (pass the recomputed scale*conv+shift+res)", act);
    TT_REQUIRE(workspace_bytes >= tt_conv_epilogue_bwd_workspace_bytes(C), "tt_conv_epilogue_bwd: workspace too small");
    EpiBwdArgs a;
    a.dy = dy; a.y = y; a.res1 = res1; a.res2 = res2; a.scale = scale; a.shift = shift;
    a.dconv = dconv; a.dres = dres; a.partial = (float*)workspace;
    a.M = M; a.C = C; a.dy_cstride = dy_cstride; a.dy_coff = dy_coff; a.y_cstride = y_cstride; a.y_coff = y_coff;
    a.r1_cstride = res1_cstride; a.r1_coff = res1_coff; a.r2_cstride = res2_cstride; a.r2_coff = res2_coff;
    a.dconv_cstride = dconv_cstride; a.dconv_coff = dconv_coff; a.dres_cstride = dres_cstride; a.dres_coff = dres_coff;
    a.dres2 = dres2; a.dres2_cstride = dres2_cstride; a.dres2_coff = dres2_coff; a.dres_accumulate = dres_accumulate;
    a.act = act; a.m_dev = m_dev_or_null; a.pre = pre_or_null; a.zraw = conv_raw_or_null;
    const int blocks = (int)(M < kEpiBlocks ? M : kEpiBlocks);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv_epilogue_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    if (dscale || dshift)
        hipLaunchKernelGGL(conv_epilogue_bwd_finish_kernel, dim3((unsigned)div_up(C, 16)), dim3(256), 0, st,
                           (const float*)workspace, blocks, C, accumulate, dscale, dshift);
    return check_launch("tt_conv_epilogue_bwd");
}

// splits of the live rows (and, in the per-wave-tile kernel, four partial slices per split)
static int gather_wgrad_splits(long long M, int Cout, int Cin, int cin_pad, int taps, int* slices) {
    int bi, bj;
    wgrad_blocks(Cout, Cin, &bi, &bj);
    const long long tiles = (long long)div_up(Cout, 32 * bi) * div_up(cin_pad, 32 * bj) * taps;
    long long s = ((bi * bj >= 8 ? 2LL : 4LL) * kNumCU + tiles - 1) / tiles;
    if (s > (M + 63) / 64) s = (M + 63) / 64;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    *slices = (int)s * ((bi == 2 && bj == 2) ? 1 : 4);
    return (int)s;
}

extern "C" long long tt_gather_conv_wgrad_workspace_bytes(long long M, int Cout, int Cin, int cin_pad, int taps) {
    int slices;
    gather_wgrad_splits(M, Cout, Cin, cin_pad, taps, &slices);
    return (long long)slices * Cout * taps * cin_pad * 4;
}

extern "C" int tt_gather_conv_wgrad(const float* x, int x_cstride, int Cin, const int* nbr, const int* m_dev, long long M,
                                    int taps, const float* dy, int dy_cstride, int Cout, int cin_pad, int accumulate,
                                    float* dw, void* workspace, long long workspace_bytes, void* stream) {
    TT_REQUIRE(x && nbr && dy && dw && workspace && M > 0 && taps > 0 && Cin > 0 && Cout > 0 && cin_pad >= Cin,
               "tt_gather_conv_wgrad: bad argument");
    const long long need = tt_gather_conv_wgrad_workspace_bytes(M, Cout, Cin, cin_pad, taps);
    TT_REQUIRE(workspace_bytes >= need, "tt_gather_conv_wgrad: workspace too small");
    const long long n = (long long)Cout * taps * cin_pad;
    int slices, bi, bj;
    const int splits = gather_wgrad_splits(M, Cout, Cin, cin_pad, taps, &slices);
    wgrad_blocks(Cout, Cin, &bi, &bj);
    GatherWgradArgs a;
    a.x = x; a.dy = dy; a.nbr = nbr; a.m_dev = m_dev; a.ws = (float*)workspace;
    a.M = M; a.Cin = Cin; a.x_cstride = x_cstride; a.Cout = Cout; a.dy_cstride = dy_cstride; a.taps = taps;
    a.cin_p = cin_pad; a.ci_tiles = div_up(cin_pad, 32 * bj);
    a.pairs_per_split = div_up(div_up(M, 2), (long long)splits);
    a.skip_empty = 0;
    hipStream_t st = (hipStream_t)stream;
    const unsigned tiles = (unsigned)(div_up(Cout, 32 * bi) * a.ci_tiles * taps);
    if (bi == 2 && bj == 2) {
        // (an XCD-contiguous tile order measured slower here: 36 vs 55 TF/s on 64 -> 64; identity mapping)
        a.xcd_tiles = 0;
        hipLaunchKernelGGL(gather_wgrad_kernel, dim3(tiles, (unsigned)splits), dim3(256), 0, st, a);
    } else {
        a.xcd_tiles = 0;
        a.skip_empty = 1;
        const dim3 grid(tiles, (unsigned)splits);
        switch (bi * 8 + bj) {
#define TT_GW(BI_, BJ_) \
    case BI_ * 8 + BJ_: hipLaunchKernelGGL((gather_wgrad_wide_kernel<BI_, BJ_>), grid, dim3(256), 0, st, a); break;
            TT_GW(4, 4) TT_GW(4, 2) TT_GW(2, 4) TT_GW(4, 1) TT_GW(1, 4) TT_GW(2, 1) TT_GW(1, 2) TT_GW(1, 1)
#undef TT_GW
            default: TT_REQUIRE(false, "tt_gather_conv_wgrad: no kernel for wave tile %d x %d", bi, bj);
        }
    }
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)div_up(n, 32)), dim3(256), 0, st, (const float*)workspace, n,
                       slices, accumulate, dw);
    return check_launch("tt_gather_conv_wgrad");
}

extern "C" int tt_sp_inverse_rulebook(const int* nbr, const int* m_dev, long long M, int taps, int* inv_prefilled_minus1,
                                      void* stream) {
    TT_REQUIRE(nbr && m_dev && inv_prefilled_minus1 && M > 0 && taps > 0, "tt_sp_inverse_rulebook: bad argument");
    hipLaunchKernelGGL(sp_inverse_rulebook_kernel, dim3((unsigned)div_up(M * taps, 256)), dim3(256), 0, (hipStream_t)stream,
                       nbr, m_dev, M, taps, inv_prefilled_minus1);
    return check_launch("tt_sp_inverse_rulebook");
}

extern "C" int tt_sp_from_dense(const float* gdense, const int* coords, const int* num_rows, long long max_rows, int C, int D,
                                int H, int W, float* grows, void* stream) {
    TT_REQUIRE(gdense && coords && num_rows && grows && max_rows > 0 && C > 0, "tt_sp_from_dense: bad argument");
    hipLaunchKernelGGL(sp_from_dense_kernel, dim3((unsigned)div_up(max_rows * C, 256)), dim3(256), 0, (hipStream_t)stream,
                       gdense, coords, num_rows, max_rows, C, D, H, W, grows);
    return check_launch("tt_sp_from_dense");
}
