// Implicit-GEMM convolution / linear for LATENCY-BOUND shapes (M <= a few thousand rows): the conv-GRU cell,
// the BEV-update convs and every row-batched decoder MLP (M = B or 4B).  With so few rows the large-tile
// kernels occupy a handful of CUs and each wave walks the whole K loop alone -- on the f32 MFMA pipe
// (64 cycles per 32x32x2 instruction) that chain, not bandwidth, sets the time (23-30 us per GRU conv).
// Here a workgroup owns ONE 32x32 output tile and its 4 waves split the K tiles between them (intra-block
// split-K), each wave staging its own A/B slices through a private LDS region; the four partial accumulators
// are reduced through LDS and wave 0 runs the fused epilogue.  ~4x shorter dependent chain, ~4x more
// workgroups than the 128-row tiles, no atomics, no second launch.
#include "conv_common.h"

namespace tt {

template <typename T>
__global__ __launch_bounds__(256) void conv_small_kernel(const ConvArgs p, int tiles_n) {
    constexpr int VEC = Elem<T>::kVec;
    constexpr int BKB = (sizeof(T) == 4) ? 64 : 128;
    constexpr int BK = BKB / (int)sizeof(T);
    constexpr int VPR = BKB / 16;
    constexpr int ROWB = BKB + 16;
    constexpr int NV = 32 * VPR / 64;                 // 16 B vectors per lane per operand per tile (2 or 4)
    constexpr int WAVE_LDS = 2 * 64 * ROWB;           // [2 buffers][A 32 rows | B 32 rows]

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* my = smem + wave * WAVE_LDS;

    const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int m0 = tile_m * 32, n0 = tile_n * 32;
    const int Mlim = p.M;
    const T* __restrict__ in = reinterpret_cast<const T*>(p.in);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(p.weight);

    int a_row[NV], a_vc[NV], a_h0[NV], a_w0[NV];
    long long a_base[NV];
    bool a_ok[NV], b_ok[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = lane + 64 * i;
        a_row[i] = idx / VPR;
        a_vc[i] = idx % VPR;
        const int m = m0 + a_row[i];
        a_ok[i] = m < Mlim;
        const int mm = a_ok[i] ? m : 0;
        const int n = mm / (p.OH * p.OW);
        const int r = mm - n * (p.OH * p.OW);
        const int oh = r / p.OW, ow = r - oh * p.OW;
        a_h0[i] = oh * p.stride - p.pad;
        a_w0[i] = ow * p.stride - p.pad;
        a_base[i] = (long long)n * p.in_nstride + p.in_coff;
        b_ok[i] = (n0 + a_row[i]) < p.Cout;
    }
    const int nk = (p.K + BK - 1) / BK;
    const int my_tiles = (nk - wave + 3) / 4;          // this wave owns K tiles wave, wave+4, ...
    const int max_tiles = (nk + 3) / 4;                // uniform trip count (barriers)
    uint4 ra[NV], rb[NV];

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int k = k0 + a_vc[i] * VEC;
            const int tap = k / p.Cin;
            const int ci = k - tap * p.Cin;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const int ih = a_h0[i] + kh * p.dil, iw = a_w0[i] + kw * p.dil;
            const bool ok = a_ok[i] && (k < p.K) && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            ra[i] = ok ? *reinterpret_cast<const uint4*>(in + a_base[i] + ((long long)ih * p.W + iw) * p.in_cstride + ci)
                       : make_uint4(0, 0, 0, 0);
            rb[i] = (b_ok[i] && k < p.K)
                        ? *reinterpret_cast<const uint4*>(wgt + (long long)(n0 + a_row[i]) * p.K + k)
                        : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            *reinterpret_cast<uint4*>(my + buf * 64 * ROWB + a_row[i] * ROWB + a_vc[i] * 16) = ra[i];
            *reinterpret_cast<uint4*>(my + buf * 64 * ROWB + (32 + a_row[i]) * ROWB + a_vc[i] * 16) = rb[i];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    if (my_tiles > 0) {
        load_tile(wave);
        store_tile(0);
    }
    __syncthreads();
    const int frag = (lane & 31) * ROWB + (lane >> 5) * 16;
    for (int t = 0; t < max_tiles; ++t) {
        const bool live = t < my_tiles;
        const bool more = (t + 1) < my_tiles;
        if (more) load_tile(wave + 4 * (t + 1));
        if (live) {
            const unsigned char* tA = my + (t & 1) * 64 * ROWB + frag;
            const unsigned char* tB = tA + 32 * ROWB;
#pragma unroll
            for (int kc = 0; kc < BKB / 32; ++kc) {
                const uint4 fa = *reinterpret_cast<const uint4*>(tA + kc * 32);
                const uint4 fb = *reinterpret_cast<const uint4*>(tB + kc * 32);
                Mfma<T>::run(fa, fb, acc);
            }
        }
        if (more) store_tile((t + 1) & 1);
        __syncthreads();
    }
    // reduce the 4 partial accumulators through LDS (reuse the staging area: 4 x 32 x 33 floats)
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
        const int m = m0 + rl, col = n0 + cl;
        if (m >= Mlim || col >= p.Cout) continue;
        float v = red[rl * LDR + cl] + red[32 * LDR + rl * LDR + cl] + red[2 * 32 * LDR + rl * LDR + cl] +
                  red[3 * 32 * LDR + rl * LDR + cl];
        int co = col, q = 0;
        if (p.pixel_shuffle2) { q = col / cout_real; co = col - q * cout_real; }
        const int n = m / ohw;
        const int rem = m - n * ohw;
        int oh = rem / p.OW, ow = rem - oh * p.OW, OWo = p.OW;
        if (p.pixel_shuffle2) { oh = 2 * oh + (q >> 1); ow = 2 * ow + (q & 1); OWo = 2 * p.OW; }
        const long long o = (long long)n * p.out_nstride + ((long long)oh * OWo + ow) * p.out_cstride + p.out_coff + co;
        v = v * (p.scale ? p.scale[co] : 1.f) + (p.shift ? p.shift[co] : 0.f);
        if (p.shift_n) v += p.shift_n[(n % p.shift_n_mod) * cout_real + co];
        if (p.res1) v += Elem<T>::ld(reinterpret_cast<const T*>(p.res1) + (long long)m * p.res1_cstride + p.res1_coff + co);
        if (p.res2) v += Elem<T>::ld(reinterpret_cast<const T*>(p.res2) + (long long)m * p.res2_cstride + p.res2_coff + co);
        v = apply_act(v, p.act);
        if (p.out_dtype == TT_F32) reinterpret_cast<float*>(p.out)[o] = v;
        else reinterpret_cast<uint16_t*>(p.out)[o] = f32_to_bf16(v);
    }
}

int try_launch_conv_small(ConvArgs& a, int dtype, hipStream_t st) {
    if (a.gather || a.m_dev || a.M > 4096) return 0;
    const int tiles_m = div_up(a.M, 32), tiles_n = div_up(a.Cout, 32);
    if ((long long)tiles_m * tiles_n > 4096) return 0;
    a.ws = nullptr;
    a.splits = 1;
    const int rowb = (dtype == TT_F32 ? 64 : 128) + 16;
    size_t smem = (size_t)4 * 2 * 64 * rowb;
    const size_t red = (size_t)4 * 32 * 33 * 4;
    if (smem < red) smem = red;
    if (dtype == TT_F32)
        hipLaunchKernelGGL(conv_small_kernel<float>, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), smem, st, a, tiles_n);
    else {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_small_kernel<uint16_t>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            attr = true;
        }
        hipLaunchKernelGGL(conv_small_kernel<uint16_t>, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), smem, st, a,
                           tiles_n);
    }
    return 1;
}

}  // namespace tt
