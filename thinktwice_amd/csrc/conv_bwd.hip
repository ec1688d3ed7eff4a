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
};

constexpr int kWgUnroll = 4;       // pixel pairs in flight per wave (each: 2 + 2 dword loads, 4 MFMAs)

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    __shared__ float red[3][64 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, k = lane >> 5;                 // channel within a 32-block, pixel of the pair
    const int taps = a.KH * a.KW;
    int t = blockIdx.x;
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

// dw[i] = (accumulate ? dw[i] : 0) + sum over splits (in order) of ws[s][i]
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ ws, long long n, int splits,
                                                                int accumulate, float* __restrict__ dw) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = accumulate ? dw[i] : 0.f;
    for (int s = 0; s < splits; ++s) v += ws[(long long)s * n + i];
    dw[i] = v;
}

static int wgrad_splits(int N, int OH, int Cout, int Cin, int taps) {
    const long long tiles = (long long)div_up(Cout, 64) * div_up(Cin, 64) * taps;
    long long s = (4LL * kNumCU + tiles - 1) / tiles;        // aim at >= 4 workgroups per CU
    const int rows = N * OH;
    if (s > rows / 4) s = rows / 4;                          // every wave of a workgroup gets at least one row
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    return (int)s;
}

}  // namespace tt

using namespace tt;

extern "C" long long tt_conv2d_wgrad_workspace_bytes(int N, int OH, int Cout, int Cin, int cin_pad, int KH, int KW) {
    return (long long)wgrad_splits(N, OH, Cout, Cin, KH * KW) * Cout * KH * KW * cin_pad * 4;
}

extern "C" int tt_conv2d_wgrad(const float* x, int N, int H, int W, int Cin, int x_cstride, int x_coff, const float* dy,
                               int OH, int OW, int Cout, int dy_cstride, int dy_coff, int KH, int KW, int stride, int pad,
                               int dil, int cin_pad, int accumulate, float* dw, void* workspace, long long workspace_bytes,
                               void* stream) {
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
    a.ci_tiles = div_up(cin_pad, 64);
    a.rows_per_split = div_up(N * OH, splits);
    hipStream_t st = (hipStream_t)stream;
    const unsigned tiles = (unsigned)(div_up(Cout, 64) * a.ci_tiles * taps);
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(tiles, (unsigned)splits), dim3(256), 0, st, a);
    const long long n = (long long)Cout * taps * cin_pad;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, st, (const float*)workspace, n,
                       splits, accumulate, dw);
    return check_launch("tt_conv2d_wgrad");
}
