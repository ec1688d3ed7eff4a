// Implicit-GEMM convolution / linear on gfx950 MFMA (wave64), channel-last activations.
//
//   C[M = N*OH*OW, Cout] = A[M, K = KH*KW*Cin] (gathered on the fly) x W[Cout, K]^T
//
// Replaces the cuDNN / cuBLAS calls behind every nn.Conv2d / nn.Linear / ConvTranspose2d(k2,s2)
// of the reference forward: ResNet-50 + PAFPN (mmdet), DepthNet / UNet /
// seg->feature / merge convs (backbones/lss.py:161-282,409-439), SECOND / SECONDFPN,
// BEV fusion neck (encoder_decoder_framework.py:81-138) and the decoder's 1x1 / 3x3 convs
// and row-batched linears (dense_heads/).
//
// Tiling (MI355X-first, not a warp-tile port):
//   * 256 threads = 4 waves per workgroup, block tile BM x BN, K-tile = 64 B (f32) / 128 B
//     (bf16) of contiguous K per row, both operands K-contiguous in LDS with a 16 B row pad
//     (row stride 80 / 144 B => the four 16-lane groups of ds_read_b128 are conflict-free).
//   * each wave owns TM x TN tiles of 32x32; f32 mode uses v_mfma_f32_32x32x2_f32 (exact
//     f32, 157 TF peak), bf16 mode v_mfma_f32_32x32x16_bf16 (2.5 PF peak), f32 accumulate.
//     One ds_read_b128 per operand feeds 4 (f32) or 1 (bf16) MFMA per k-chunk.
//   * global->register->LDS double buffering: tile k+1 is fetched (16 B per lane, coalesced
//     along Cin) while tile k is on the matrix pipe; one barrier per K-tile.
//   * fused epilogue: BN/bias scale+shift, per-image shift, up to two residuals, activation,
//     channel-offset / strided output (concat-free), ConvTranspose k2s2 pixel shuffle.
#include <stdlib.h>

#include "conv_common.h"

namespace tt {

// KX = K-tile widening factor (1 => 64 B (f32) / 128 B (bf16) of K per row per tile).  A 4x variant was measured
// on the M ~ 3.5k GRU / decoder layers and did not help: those layers are bound by the f32 MFMA issue rate of a
// single 32x32 tile per wave (profiles/README.md), not by the K-loop round trips.
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool GATHER, int KX = 1>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
    constexpr int VEC = Elem<T>::kVec;               // elements per 16 B
    constexpr int BKB = ((sizeof(T) == 4) ? 64 : 128) * KX; // K bytes per row per tile
    constexpr int BK = BKB / (int)sizeof(T);
    constexpr int VPR = BKB / 16;                    // 16 B vectors per row
    constexpr int ROWB = BKB + 16;                   // padded LDS row stride (bytes)
    constexpr int NVA = (BM * VPR + 255) / 256;
    constexpr int NVB = (BN * VPR + 255) / 256;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nbuf = (p.K > BK) ? 2 : 1;               // single K tile: no double buffer (more blocks/CU)
    unsigned char* sA = smem;                          // [nbuf][BM][ROWB]
    unsigned char* sB = smem + nbuf * BM * ROWB;       // [nbuf][BN][ROWB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int tile_n = blockIdx.x % p.tiles_n;
    const int tile_m = blockIdx.x / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    int Mlim = p.M;
    if (p.m_dev) {
        const int md = *p.m_dev;
        Mlim = md < Mlim ? md : Mlim;
    }
    if (m0 >= Mlim) return;   // block-uniform: before any barrier

    const T* __restrict__ in = reinterpret_cast<const T*>(p.in);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(p.weight);

    // ---- per-thread gather rows (fixed over the K loop)
    int a_row[NVA], a_vc[NVA], a_h0[NVA], a_w0[NVA];
    long long a_base[NVA];
    bool a_ok[NVA];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int idx = tid + 256 * i;
        a_row[i] = idx / VPR;
        a_vc[i] = idx % VPR;
        const int m = m0 + a_row[i];
        a_ok[i] = (m < Mlim) && (idx < BM * VPR);
        const int mm = a_ok[i] ? m : 0;
        if (GATHER) {
            a_h0[i] = 0;
            a_w0[i] = 0;
            a_base[i] = (long long)mm * (p.KH * p.KW);      // row of the gather table
        } else {
            const int n = mm / (p.OH * p.OW);
            const int r = mm - n * (p.OH * p.OW);
            const int oh = r / p.OW, ow = r - oh * p.OW;
            a_h0[i] = oh * p.stride - p.pad;
            a_w0[i] = ow * p.stride - p.pad;
            a_base[i] = (long long)n * p.in_nstride + p.in_coff;
        }
    }
    int b_row[NVB], b_vc[NVB];
    bool b_ok[NVB];
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
        const int idx = tid + 256 * i;
        b_row[i] = idx / VPR;
        b_vc[i] = idx % VPR;
        b_ok[i] = ((n0 + b_row[i]) < p.Cout) && (idx < BN * VPR);
    }

    int jn[NVA];   // GATHER + cin_fast: rulebook entries of the NEXT K tile, fetched one tile ahead
#pragma unroll
    for (int i = 0; i < NVA; ++i) jn[i] = -1;
    const int nk_all = (p.K + BK - 1) / BK;
    int kt0 = 0, nk = nk_all;
    if (p.splits > 1) {       // this block owns K tiles [kt0, nk)
        const int per = (nk_all + p.splits - 1) / p.splits;
        kt0 = blockIdx.y * per;
        nk = min(nk_all, kt0 + per);
        if (kt0 >= nk) return;
    }
    uint4 ra[NVA], rb[NVB];

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        int tap_u = 0, ci_u = 0, kh_u = 0, kw_u = 0;
        if (p.cin_fast) {
            tap_u = k0 / p.Cin;
            ci_u = k0 - tap_u * p.Cin;
            kh_u = tap_u / p.KW;
            kw_u = tap_u - kh_u * p.KW;
        }
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            int kh, kw, ci;
            const int k = k0 + a_vc[i] * VEC;
            if (p.cin_fast) {
                kh = kh_u; kw = kw_u; ci = ci_u + a_vc[i] * VEC;
            } else {
                const int tap = k / p.Cin;
                ci = k - tap * p.Cin;
                kh = tap / p.KW;
                kw = tap - kh * p.KW;
            }
            if (GATHER) {
                int j = -1;
                if (p.cin_fast) {
                    // the entry for this tile was fetched while the previous tile was in flight, so the
                    // feature-row load below does not sit behind a dependent index load
                    j = jn[i];
                    const int kn = k0 + BK;
                    jn[i] = (a_ok[i] && kn < p.K) ? p.gather[a_base[i] + kn / p.Cin] : -1;
                } else if (a_ok[i] && (k < p.K)) {
                    j = p.gather[a_base[i] + kh * p.KW + kw];
                }
                if (j >= 0) {
                    const T* src = in + (long long)j * p.in_cstride + p.in_coff + ci;
                    ra[i] = *reinterpret_cast<const uint4*>(src);
                } else {
                    ra[i] = make_uint4(0, 0, 0, 0);
                }
            } else {
                const int ih = a_h0[i] + kh * p.dil;
                const int iw = a_w0[i] + kw * p.dil;
                const bool ok = a_ok[i] && (k < p.K) && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                if (ok) {
                    const T* src = in + a_base[i] + ((long long)ih * p.W + iw) * p.in_cstride + ci;
                    ra[i] = *reinterpret_cast<const uint4*>(src);
                } else {
                    ra[i] = make_uint4(0, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int k = k0 + b_vc[i] * VEC;
            if (b_ok[i] && k < p.K) {
                rb[i] = *reinterpret_cast<const uint4*>(wgt + (long long)(n0 + b_row[i]) * p.K + k);
            } else {
                rb[i] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NVA; ++i)
            if (tid + 256 * i < BM * VPR)
                *reinterpret_cast<uint4*>(sA + (buf * BM + a_row[i]) * ROWB + a_vc[i] * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < NVB; ++i)
            if (tid + 256 * i < BN * VPR)
                *reinterpret_cast<uint4*>(sB + (buf * BN + b_row[i]) * ROWB + b_vc[i] * 16) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (GATHER && p.cin_fast) {
#pragma unroll
        for (int i = 0; i < NVA; ++i)
            jn[i] = (a_ok[i] && kt0 * BK < p.K) ? p.gather[a_base[i] + (kt0 * BK) / p.Cin] : -1;
    }
    load_tile(kt0);
    store_tile(0);
    __syncthreads();

    const int frag_off = (lane & 31) * ROWB + (lane >> 5) * 16;
    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned char* tA = sA + (buf * BM + wm * WTM) * ROWB + frag_off;
        const unsigned char* tB = sB + (buf * BN + wn * WTN) * ROWB + frag_off;
#pragma unroll
        for (int kc = 0; kc < BKB / 32; ++kc) {
            uint4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const uint4*>(tA + i * 32 * ROWB + kc * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const uint4*>(tB + j * 32 * ROWB + kc * 32);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mfma<T>::run(fa[i], fb[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    conv_epilogue<T, TM, TN, WTM, WTN>(p, acc, smem, wave, lane, wm, wn, m0, n0, Mlim);
}

// split-K epilogue: out = act(scale * ws + shift + shift_n + res1 + res2), same addressing as the fused one
template <typename T>
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const ConvArgs p) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)p.M * p.Cout) return;
    const int m = (int)(t / p.Cout), col = (int)(t % p.Cout);
    const int cout_real = p.pixel_shuffle2 ? (p.Cout >> 2) : p.Cout;
    int co = col, q = 0;
    if (p.pixel_shuffle2) { q = col / cout_real; co = col - q * cout_real; }
    const int ohw = p.OH * p.OW;
    const int n = m / ohw;
    const int rem = m - n * ohw;
    int oh = rem / p.OW, ow = rem - oh * p.OW, OWo = p.OW;
    if (p.pixel_shuffle2) { oh = 2 * oh + (q >> 1); ow = 2 * ow + (q & 1); OWo = 2 * p.OW; }
    const long long o = (long long)n * p.out_nstride + ((long long)oh * OWo + ow) * p.out_cstride + p.out_coff + co;
    float acc = p.ws[t];
    for (int sl = 1; sl < p.ws_slices; ++sl) acc += p.ws[(long long)sl * p.M * p.Cout + t];     // the K splits, in index order
    float v = acc * (p.scale ? p.scale[co] : 1.f) + (p.shift ? p.shift[co] : 0.f);
    if (p.shift_n) v += p.shift_n[(n % p.shift_n_mod) * cout_real + co];
    if (p.res1) v += Elem<T>::ld(reinterpret_cast<const T*>(p.res1) + (long long)m * p.res1_cstride + p.res1_coff + co);
    if (p.res2) v += Elem<T>::ld(reinterpret_cast<const T*>(p.res2) + (long long)m * p.res2_cstride + p.res2_coff + co);
    v = apply_act(v, p.act);
    if (p.out_dtype == TT_F32) reinterpret_cast<float*>(p.out)[o] = v;
    else store16(p.out, o, v, p.out_dtype);
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool GATHER, int KX = 1>
static int launch_conv(ConvArgs& a, hipStream_t st) {
    constexpr int BKB = ((sizeof(T) == 4) ? 64 : 128) * KX;
    constexpr int BK = BKB / (int)sizeof(T);
    constexpr int ROWB = BKB + 16;
    const int tiles_m = div_up(a.M, BM);
    a.tiles_n = div_up(a.Cout, BN);
    a.cin_fast = (a.Cin % BK == 0) ? 1 : 0;
    constexpr int WTN_ = BN / WAVES_N;
    size_t smem = (size_t)((a.K > BK) ? 2 : 1) * (BM + BN) * ROWB;
    const size_t epi = (size_t)4 * 32 * (WTN_ + 4) * 4;            // LDS-staged epilogue
    if (smem < epi) smem = epi;
    auto kern = conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, GATHER, KX>;
    static bool attr_set = false;
    if (!attr_set) {   // once per instantiation, sized for the largest (double-buffered) request
        size_t mx = (size_t)2 * (BM + BN) * ROWB;
        if (mx < epi) mx = epi;
        if (mx > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)mx);
        attr_set = true;
    }
    const int tiles = tiles_m * a.tiles_n;
    const int nk = div_up(a.K, BK);
    a.splits = 1;
    if ((a.ws || a.flags < 0) && !a.m_dev && tiles < 128 && nk >= 8) {
        int sp = div_up(512, tiles);
        if (sp > nk / 2) sp = nk / 2;
        if (sp > 64) sp = 64;
        a.splits = sp < 1 ? 1 : sp;
    }
    if (a.flags < 0) return a.splits > 1 ? a.splits : 0;       // query (tt_conv2d_splitk_slices): no launch
    if (a.splits <= 1) a.ws = nullptr;
    if (a.ws && a.ws_slices > 0) {
        // ordered form: the non-empty splits (the K tiles are dealt in runs of ceil(nk / splits)) store into their own slices
        const int per = div_up(nk, a.splits);
        const int eff = div_up(nk, per);
        TT_REQUIRE(a.ws_slices >= eff, "tt_conv2d_fwd: split-K workspace holds %d slices, %d needed", a.ws_slices, eff);
        a.ws_slices = eff;
    }
    snprintf(g_conv_kernel, sizeof(g_conv_kernel), "conv_igemm_kernel<%s, %d, %d>%s", sizeof(T) == 4 ? "float" : "16-bit",
             BM, BN, a.ws ? " split-K" : "");
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)a.splits), dim3(256), smem, st, a);
    if (a.ws) {
        const long long tot = (long long)a.M * a.Cout;
        hipLaunchKernelGGL(splitk_finalize_kernel<T>, dim3((unsigned)div_up(tot, 256)), dim3(256), 0, st, a);
    }
    return check_launch("tt_conv2d_fwd");
}

template <typename T, bool GATHER>
static int dispatch_conv2(ConvArgs& a, hipStream_t st) {
    if (a.Cout > 64) return launch_conv<T, 128, 128, 2, 2, GATHER>(a, st);
    if (a.Cout > 32) return launch_conv<T, 128, 64, 2, 2, GATHER>(a, st);
    return launch_conv<T, 128, 32, 4, 1, GATHER>(a, st);
}

template <typename T>
static int dispatch_conv(ConvArgs& a, hipStream_t st) {
    return a.gather ? dispatch_conv2<T, true>(a, st) : dispatch_conv2<T, false>(a, st);
}

}  // namespace tt

using namespace tt;

static int conv2d_run(const tt_conv_desc* d, void* stream, bool query);

extern "C" int tt_conv2d_fwd(const tt_conv_desc* d, void* stream) { return conv2d_run(d, stream, false); }

extern "C" int tt_conv2d_splitk_slices(const tt_conv_desc* d) {
    const int n = conv2d_run(d, nullptr, true);
    return n > 0 ? n : 0;
}

static int conv2d_run(const tt_conv_desc* d, void* stream, bool query) {
    TT_REQUIRE(d && d->in && d->weight && d->out, "tt_conv2d_fwd: null pointer");
    TT_REQUIRE(d->dtype == TT_F32 || d->dtype == TT_BF16 || d->dtype == TT_F16, "tt_conv2d_fwd: bad dtype %d", d->dtype);
    TT_REQUIRE(d->out_dtype == TT_F32 || d->out_dtype == d->dtype ||
                   (d->dtype == TT_F32 && (d->out_dtype == TT_F16 || d->out_dtype == TT_BF16) && !d->res2),
               "tt_conv2d_fwd: out_dtype must be TT_F32, the operand dtype, or (f32 operands, at most res1) a 16-bit type "
               "(got %d for dtype %d)", d->out_dtype, d->dtype);
    TT_REQUIRE(!d->res1_f32 || (d->weight_h2 && d->res1),
               "tt_conv2d_fwd: res1_f32 goes with an h2 layer's res1 (the other 16-bit kernels read residuals of their operand type)");
    TT_REQUIRE(d->dtype == TT_F32 || d->weight_h2 || (!d->out2 && d->res1_up_w <= 0),
               "tt_conv2d_fwd: out2 / res1_up_* exist in the f32-operand kernels and the h2 kernel only");
    TT_REQUIRE(!d->weight_h2 || (d->dtype == TT_F16 && !d->gather_idx && !d->splitk_ws && !d->pixel_shuffle2),
               "tt_conv2d_fwd: weight_h2 goes with dense TT_F16 operands (no split-K workspace, no pixel shuffle)");
    TT_REQUIRE(!d->out2 || (!d->splitk_ws && !d->pixel_shuffle2 && !d->gather_idx && d->out2_cstride % 4 == 0 &&
                            d->out2_coff % 4 == 0 && (reinterpret_cast<uintptr_t>(d->out2) & 15) == 0),
               "tt_conv2d_fwd: out2 needs a dense row-linear layer and 16-byte aligned rows");
    const int vec = d->dtype == TT_F32 ? 4 : 8;
    TT_REQUIRE(d->Cin > 0 && d->Cin % vec == 0 && d->in_cstride % vec == 0 && d->in_coff % vec == 0,
               "tt_conv2d_fwd: Cin=%d in_cstride=%d in_coff=%d must be multiples of %d (pad channels)",
               d->Cin, d->in_cstride, d->in_coff, vec);
    TT_REQUIRE((reinterpret_cast<uintptr_t>(d->in) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(d->weight) & 15) == 0,
               "tt_conv2d_fwd: in/weight must be 16-byte aligned");
    TT_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 &&
                   d->stride > 0 && d->dil > 0 && d->OH > 0 && d->OW > 0,
               "tt_conv2d_fwd: bad geometry");
    TT_REQUIRE(!d->pixel_shuffle2 || d->Cout % 4 == 0, "tt_conv2d_fwd: pixel_shuffle2 needs Cout%%4==0");
    ConvArgs a;
    a.in = d->in; a.weight = d->weight; a.out = d->out;
    a.scale = d->scale; a.shift = d->shift; a.shift_n = d->shift_n;
    a.res1 = d->res1; a.res2 = d->res2;
    a.gather = d->gather_idx; a.m_dev = d->m_dev;
    a.row_perm = d->gather_idx ? d->row_perm : nullptr;
    a.row_mask = d->gather_idx ? d->row_mask : nullptr;
    TT_REQUIRE(!a.row_perm == !a.row_mask, "tt_conv2d_fwd: row_perm and row_mask come together");
    a.ws = d->splitk_ws;
    a.ws_slices = d->splitk_ws ? d->splitk_slices : 0;
    TT_REQUIRE(!d->gather_idx || (d->H == 1 && d->W == 1 && d->OH == 1 && d->OW == 1 && d->KH == 1 &&
                                  !d->pixel_shuffle2),
               "tt_conv2d_fwd: gather mode wants H=W=OH=OW=KH=1, KW=taps");
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.in_cstride = d->in_cstride; a.in_coff = d->in_coff;
    a.in_nstride = d->in_nstride ? d->in_nstride : (long long)d->H * d->W * d->in_cstride;
    a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.OH = d->OH; a.OW = d->OW; a.out_cstride = d->out_cstride; a.out_coff = d->out_coff;
    const long long def_on = (long long)d->OH * d->OW * d->out_cstride * (d->pixel_shuffle2 ? 4 : 1);
    a.out_nstride = d->out_nstride ? d->out_nstride : def_on;
    a.pixel_shuffle2 = d->pixel_shuffle2;
    a.shift_n_mod = d->shift_n_mod > 0 ? d->shift_n_mod : 1;
    a.res1_cstride = d->res1_cstride; a.res1_coff = d->res1_coff;
    a.res2_cstride = d->res2_cstride; a.res2_coff = d->res2_coff;
    a.act = d->act; a.out_dtype = d->out_dtype;
    a.out2 = d->out2; a.out2_cstride = d->out2_cstride; a.out2_coff = d->out2_coff;
    a.res1_up_h = d->res1_up_h; a.res1_up_w = d->res1_up_w;
    TT_REQUIRE((d->res1_up_h > 0) == (d->res1_up_w > 0) && (d->res1_up_w <= 0 || (d->res1 && !d->gather_idx && !d->splitk_ws &&
                                                                                 !d->pixel_shuffle2)),
               "tt_conv2d_fwd: res1_up_h / res1_up_w come together, with a dense layer's res1");
    a.M = d->N * d->OH * d->OW;
    a.K = d->KH * d->KW * d->Cin;
    a.out_fast = (!d->pixel_shuffle2 && a.out_nstride == (long long)d->OH * d->OW * d->out_cstride) ? 1 : 0;
    TT_REQUIRE(!(d->pixel_shuffle2 && (d->res1 || d->res2)),
               "tt_conv2d_fwd: residuals are not supported with pixel_shuffle2");
    a.tiles_n = 1; a.cin_fast = 0; a.m_begin = 0;
    a.trace = g_conv_trace;
    // f32 outputs are written once and read by a LATER launch: non-temporal stores keep them from displacing the tile operands in
    // L2 / MALL (+0.3-0.5 % on the forward, profiles/r04_nt_store.txt)
    a.flags = 16;
    if (d->in_pair || d->out_pair) {
        TT_REQUIRE(d->dtype == TT_F32 && d->out_dtype == TT_F32 && d->weight_x3 && !d->gather_idx && !d->splitk_ws && !d->pixel_shuffle2,
                   "tt_conv2d_fwd: in_pair / out_pair go with dense bf16x3 layers (f32 containers, weight_x3, no split-K workspace)");
        TT_REQUIRE(!d->out_pair || (!d->res1 && !d->res2 && !d->shift_n && d->Cout % 16 == 0 && d->out_cstride % 16 == 0 &&
                                    d->out_coff % 16 == 0 && (reinterpret_cast<uintptr_t>(d->out) & 63) == 0),
                   "tt_conv2d_fwd: out_pair needs Cout / out_cstride / out_coff multiples of 16, a 64-byte aligned output and no "
                   "residual / per-image shift");
        if (d->in_pair) a.flags |= 32;
        if (d->out_pair) a.flags |= 64;
    }
    if (d->res1_f32) a.flags |= 128;
    if (query) a.flags = -1;       // launch_conv returns the split count instead of launching
    {
        const int co_vec = (d->out_dtype == TT_F32 && !d->out_pair) ? 4 : 8;
        const int osz = d->out_dtype == TT_F32 ? 4 : 2;
        const int cr = d->pixel_shuffle2 ? d->Cout / 4 : d->Cout;
        bool ok = (cr % co_vec == 0) && (d->out_cstride % co_vec == 0) && (d->out_coff % co_vec == 0) &&
                  (a.out_nstride % co_vec == 0) && ((reinterpret_cast<uintptr_t>(d->out) & 15) == 0);
        (void)osz;
        a.vec_epi = ok ? 1 : 0;
        auto res_ok = [&](const void* r, int cs, int co_) {
            return !r || ((cs % co_vec == 0) && (co_ % co_vec == 0) && ((reinterpret_cast<uintptr_t>(r) & 15) == 0));
        };
        a.res_vec = (res_ok(d->res1, d->res1_cstride, d->res1_coff) && res_ok(d->res2, d->res2_cstride, d->res2_coff)) ? 1 : 0;
    }
    hipStream_t st = (hipStream_t)stream;
    if (query) {
        if (d->in_pair || d->out_pair || d->weight_h2 || d->res1_up_w > 0) return 0;
        a.ws = nullptr;
        a.row_perm = nullptr;
        a.row_mask = nullptr;
        if (d->weight_x3 && d->dtype == TT_F32 && a.K % 16 == 0) {      // few rows, long K, bf16x3 operand: the 64-wide x3 tile, K split
            const int n = conv_glds_x3_splitk_slices(a);
            if (n > 0) return n;
        }
        if (d->dtype == TT_F32) return dispatch_conv<float>(a, st);
        if (d->dtype == TT_F16) return dispatch_conv<f16_t>(a, st);
        return dispatch_conv<uint16_t>(a, st);
    }
    if (d->weight_h2) {
        // half storage x (hi, lo) weights: the only kernel with this arithmetic -- a shape outside its contract is an error, not a
        // silent change of precision
        TT_REQUIRE((reinterpret_cast<uintptr_t>(d->weight_h2) & 15) == 0, "tt_conv2d_fwd: weight_h2 must be 16-byte aligned");
        ConvArgs ah = a;
        ah.weight = d->weight_h2;
        TT_REQUIRE(try_launch_conv_h2(ah, st), "tt_conv2d_fwd: weight_h2 layer outside the h2 kernel's contract (Cin=%d KH*KW=%d)",
                   d->Cin, d->KH * d->KW);
        return check_launch("tt_conv2d_fwd(h2)");
    }
    TT_REQUIRE(!d->out2 || a.vec_epi, "tt_conv2d_fwd: out2 needs the vector epilogue (aligned channel counts)");
    TT_REQUIRE(d->res1_up_w <= 0 || a.vec_epi, "tt_conv2d_fwd: an upsampled res1 needs the vector epilogue (aligned channel counts)");
    TT_REQUIRE(!d->res1_f32 || (a.vec_epi && a.res_vec), "tt_conv2d_fwd: an f32 res1 needs the vector epilogue and aligned residual rows");
    TT_REQUIRE(!(d->dtype == TT_F32 && d->out_dtype != TT_F32 && d->res1) || (a.vec_epi && a.res_vec),
               "tt_conv2d_fwd: a 16-bit output of an f32 layer with a residual needs the vector epilogue");
    if (d->in_pair || d->out_pair) {
        TT_REQUIRE(a.vec_epi && (reinterpret_cast<uintptr_t>(d->weight_x3) & 15) == 0 && a.K % 16 == 0,
                   "tt_conv2d_fwd: in_pair / out_pair need the vector epilogue and a 16-byte aligned weight_x3");
        ConvArgs ax = a;
        ax.weight = d->weight_x3;
        TT_REQUIRE(try_launch_conv_glds_x3(ax, st), "tt_conv2d_fwd: pair-format layer outside the LDS-DMA bf16x3 kernel's contract "
                   "(M=%d Cin=%d Cout=%d)", a.M, d->Cin, d->Cout);
        return check_launch("tt_conv2d_fwd(glds x3, pair)");
    }
    if (!d->splitk_ws && !d->out2 && d->res1_up_w <= 0 && !d->res1_f32 && try_launch_conv_small(a, d->dtype, st)) {
        snprintf(g_conv_kernel, sizeof(g_conv_kernel), "conv_small_kernel");
        return check_launch("tt_conv2d_fwd(small)");
    }
    if (d->splitk_ws && d->weight_x3 && d->dtype == TT_F32 && a.K % 16 == 0 &&
        (reinterpret_cast<uintptr_t>(d->weight_x3) & 15) == 0) {
        ConvArgs ax = a;
        ax.weight = d->weight_x3;
        if (launch_conv_glds_x3_splitk(ax, st)) {
            const long long tot = (long long)ax.M * ax.Cout;
            hipLaunchKernelGGL(splitk_finalize_kernel<float>, dim3((unsigned)div_up(tot, 256)), dim3(256), 0, st, ax);
            return check_launch("tt_conv2d_fwd(glds x3 split-K)");
        }
    }
    if (!d->splitk_ws && d->weight_x3 && d->dtype == TT_F32) {
        TT_REQUIRE((reinterpret_cast<uintptr_t>(d->weight_x3) & 15) == 0 && a.K % 16 == 0,
                   "tt_conv2d_fwd: weight_x3 needs 16-byte alignment and K %% 16 == 0 (K = %d)", a.K);
        ConvArgs ax = a;
        ax.weight = d->weight_x3;
        if (try_launch_conv_glds_x3(ax, st)) return check_launch("tt_conv2d_fwd(glds x3)");
    }
    if (!d->splitk_ws && try_launch_conv_glds(a, d->dtype, st)) return check_launch("tt_conv2d_fwd(glds)");
    a.row_perm = nullptr;      // the tile plan is an LDS-DMA-kernel feature: the other kernels walk rows and taps in
    a.row_mask = nullptr;      // natural order (same result)
    if (d->dtype == TT_F32) return dispatch_conv<float>(a, st);
    if (d->dtype == TT_F16) return dispatch_conv<f16_t>(a, st);
    return dispatch_conv<uint16_t>(a, st);
}
