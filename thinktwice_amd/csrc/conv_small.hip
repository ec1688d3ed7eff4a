// Implicit-GEMM convolution / linear for LATENCY-BOUND shapes (M <= a few thousand rows): the conv-GRU cell,
// the BEV-update convs and every row-batched decoder MLP (M = B or 4B).  With so few rows the large-tile
// kernels occupy a handful of CUs and each wave walks the whole K loop alone; what sets the time is the
// dependent chain global load -> LDS -> barrier -> MFMA per K tile, not bandwidth or FLOPs.
//
// Here a workgroup owns ONE 32x32 output tile and its 4 waves split K between them (intra-block split-K).
// The MFMA operand layout needs no transposition for this tile shape: lane l supplies row (l & 31) and the
// 16-byte K chunk (l >> 5) of a 32-byte K step for BOTH operands (A = im2col row of the activation, B = weight
// row, each [row][K] with K contiguous), so every lane loads its operands straight from global memory into the
// registers the MFMA reads -- no LDS staging, no barrier in the K loop, and all loads of a step group are
// independent: a wave issues GROUP steps of loads back to back (the next group before the current group's
// MFMAs), so a whole GRU conv (K = 288..360) is one or two memory round trips.  The four partial accumulators
// are reduced through LDS and all 256 threads run the fused epilogue.  No atomics, no second launch.
#include "conv_common.h"

namespace tt {

template <typename T>
__global__ __launch_bounds__(256) void conv_small_kernel(const ConvArgs p, int tiles_n) {
    constexpr int VEC = Elem<T>::kVec;                // elements per 16 B chunk (4 f32 / 8 bf16)
    constexpr int STEP = 2 * VEC;                     // K elements per step (32 B per row)
    constexpr int GROUP = 6;                          // steps in flight per wave (x2 with the prefetched group)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int m0 = tile_m * 32, n0 = tile_n * 32;
    const int Mlim = p.M;
    const T* __restrict__ in = reinterpret_cast<const T*>(p.in);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(p.weight);

    // this lane's output row (A operand) and output channel (B operand)
    const int row = lane & 31;
    const int m = m0 + row;
    const bool a_ok = m < Mlim;
    const int mm = a_ok ? m : 0;
    const int n_img = mm / (p.OH * p.OW);
    const int r_img = mm - n_img * (p.OH * p.OW);
    const int oh = r_img / p.OW, ow = r_img - oh * p.OW;
    const int h0 = oh * p.stride - p.pad, w0 = ow * p.stride - p.pad;
    const T* a_base = in + (long long)n_img * p.in_nstride + p.in_coff;
    const bool b_ok = (n0 + row) < p.Cout;
    const T* b_base = wgt + (long long)(b_ok ? n0 + row : 0) * p.K;

    // K position of this lane's chunk in the wave's current step, kept as (kh, kw, ci) incrementally:
    // wave w owns steps w, w+4, w+8, ...; consecutive owned steps are 4*STEP elements apart
    const int nsteps = (p.K + STEP - 1) / STEP;
    int k = wave * STEP + (lane >> 5) * VEC;
    int tap = k / p.Cin;
    int ci = k - tap * p.Cin;
    int kh = tap / p.KW, kw = tap - kh * p.KW;

    auto load_step = [&](uint4& ra, uint4& rb, bool live) {
        const int ih = h0 + kh * p.dil, iw = w0 + kw * p.dil;
        const bool ok = live && a_ok && (k < p.K) && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        const T* src = a_base + ((long long)ih * p.W + iw) * p.in_cstride + ci;
        ra = ok ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
        rb = (live && b_ok && k < p.K) ? *reinterpret_cast<const uint4*>(b_base + k) : make_uint4(0, 0, 0, 0);
        // advance to this wave's next step
        k += 4 * STEP;
        ci += 4 * STEP;
        while (ci >= p.Cin) {
            ci -= p.Cin;
            if (++kw == p.KW) {
                kw = 0;
                ++kh;
            }
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int my_steps = (nsteps - wave + 3) / 4;
    uint4 ra[2][GROUP], rb[2][GROUP];
#pragma unroll
    for (int g = 0; g < GROUP; ++g) load_step(ra[0][g], rb[0][g], g < my_steps);
    for (int s0 = 0; s0 < my_steps; s0 += 2 * GROUP) {
        // two groups per trip so the register double buffer is statically indexed
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int base = s0 + half * GROUP;
            if (base < my_steps) {
                if (base + GROUP < my_steps) {
#pragma unroll
                    for (int g = 0; g < GROUP; ++g)
                        load_step(ra[half ^ 1][g], rb[half ^ 1][g], base + GROUP + g < my_steps);
                }
#pragma unroll
                for (int g = 0; g < GROUP; ++g) Mfma<T>::run(ra[half][g], rb[half][g], acc);   // dead steps add zeros
            }
        }
    }

    // reduce the 4 partial accumulators through LDS (4 x 32 x 33 floats)
    float* red = reinterpret_cast<float*>(smem);
    constexpr int LDR = 33;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        red[wave * 32 * LDR + ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDR + (lane & 31)] = acc[r];
    __syncthreads();
    // fused epilogue, 256 threads over the 32 x 32 tile (4 elements each)
    const int cout_real = p.pixel_shuffle2 ? (p.Cout >> 2) : p.Cout;
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = tid + 256 * e;
        const int rl = idx >> 5, cl = idx & 31;
        const int mo = m0 + rl, col = n0 + cl;
        if (mo >= Mlim || col >= p.Cout) continue;
        float v = red[rl * LDR + cl] + red[32 * LDR + rl * LDR + cl] + red[2 * 32 * LDR + rl * LDR + cl] +
                  red[3 * 32 * LDR + rl * LDR + cl];
        int co = col, q = 0;
        if (p.pixel_shuffle2) { q = col / cout_real; co = col - q * cout_real; }
        const int n = mo / ohw;
        const int rem = mo - n * ohw;
        int oho = rem / p.OW, owo = rem - oho * p.OW, OWo = p.OW;
        if (p.pixel_shuffle2) { oho = 2 * oho + (q >> 1); owo = 2 * owo + (q & 1); OWo = 2 * p.OW; }
        const long long o = (long long)n * p.out_nstride + ((long long)oho * OWo + owo) * p.out_cstride + p.out_coff + co;
        v = v * (p.scale ? p.scale[co] : 1.f) + (p.shift ? p.shift[co] : 0.f);
        if (p.shift_n) v += p.shift_n[(n % p.shift_n_mod) * cout_real + co];
        if (p.res1) v += Elem<T>::ld(reinterpret_cast<const T*>(p.res1) + (long long)mo * p.res1_cstride + p.res1_coff + co);
        if (p.res2) v += Elem<T>::ld(reinterpret_cast<const T*>(p.res2) + (long long)mo * p.res2_cstride + p.res2_coff + co);
        v = apply_act(v, p.act);
        if (p.out_dtype == TT_F32) reinterpret_cast<float*>(p.out)[o] = v;
        else store16(p.out, o, v, p.out_dtype);
    }
}

int try_launch_conv_small(ConvArgs& a, int dtype, hipStream_t st) {
    if (a.gather || a.m_dev || a.M > 4096) return 0;
    const int tiles_m = div_up(a.M, 32), tiles_n = div_up(a.Cout, 32);
    if ((long long)tiles_m * tiles_n > 4096) return 0;
    a.ws = nullptr;
    a.splits = 1;
    const size_t smem = (size_t)4 * 32 * 33 * 4;
    if (dtype == TT_F32)
        hipLaunchKernelGGL(conv_small_kernel<float>, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), smem, st, a, tiles_n);
    else if (dtype == TT_F16)
        hipLaunchKernelGGL(conv_small_kernel<f16_t>, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), smem, st, a, tiles_n);
    else
        hipLaunchKernelGGL(conv_small_kernel<uint16_t>, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), smem, st, a,
                           tiles_n);
    return 1;
}

}  // namespace tt
