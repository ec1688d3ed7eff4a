// Deformable-conv (v1) column builder: mmcv DeformConv2dPack call site backbones/lss.py:189-197.
// cols[pix][tap][c] = bilinear(x, (y + i - pad + dy_tap, x + j - pad + dx_tap)), zero outside
// (mmcv deformable_im2col_bilinear semantics; offset channel order [dy_0,dx_0,dy_1,dx_1,...],
// deform_groups = 1, stride 1, 3x3, dilation 1).  The grouped 3x3 GEMM that follows runs on
// tt_conv2d_fwd over the columns (one launch per group, K = 9 * C/groups).
#include "tt_common.h"

namespace tt {

template <typename T>
__global__ __launch_bounds__(256) void deform_im2col_kernel(const T* __restrict__ x, const float* __restrict__ off,
                                                            T* __restrict__ cols, int N, int H, int W, int C,
                                                            int off_cstride, int pad) {
    constexpr int V = Elem<T>::kVec;
    const int cv = C / V;
    const long long total = (long long)N * H * W * 9 * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long long r = i / cv;
        const int tap = (int)(r % 9);
        const long long pix = r / 9;
        const int xw = (int)(pix % W);
        const int yh = (int)((pix / W) % H);
        const long long n = pix / ((long long)H * W);
        const float dy = off[pix * off_cstride + 2 * tap];
        const float dx = off[pix * off_cstride + 2 * tap + 1];
        const float py = (float)(yh + tap / 3 - pad) + dy;
        const float px = (float)(xw + tap % 3 - pad) + dx;
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        if (py > -1.f && py < (float)H && px > -1.f && px < (float)W) {
            const int y0 = (int)floorf(py), x0 = (int)floorf(px);
            const float ly = py - (float)y0, lx = px - (float)x0;
            const float hy = 1.f - ly, hx = 1.f - lx;
            const T* base = x + n * H * W * C + c * V;
            auto add = [&](int yy, int xx, float wgt) {
                if (yy < 0 || yy > H - 1 || xx < 0 || xx > W - 1) return;
                const T* p = base + ((long long)yy * W + xx) * C;
#pragma unroll
                for (int k = 0; k < V; ++k) acc[k] += wgt * Elem<T>::ld(p + k);
            };
            add(y0, x0, hy * hx);
            add(y0, x0 + 1, hy * lx);
            add(y0 + 1, x0, ly * hx);
            add(y0 + 1, x0 + 1, ly * lx);
        }
        T* dst = cols + (pix * 9 + tap) * C + c * V;
#pragma unroll
        for (int k = 0; k < V; ++k) Elem<T>::st(dst + k, acc[k]);
    }
}

}  // namespace tt

using namespace tt;

extern "C" int tt_deform_im2col3x3(const void* x, const float* offsets, void* cols, int N, int H, int W, int C,
                                   int off_cstride, int pad, int dtype, void* stream) {
    TT_REQUIRE(x && offsets && cols, "tt_deform_im2col3x3: null");
    const int vec = dtype == TT_F32 ? 4 : 8;
    TT_REQUIRE(C % vec == 0 && off_cstride >= 18, "tt_deform_im2col3x3: bad C/off_cstride");
    const long long total = (long long)N * H * W * 9 * (C / vec);
    long long blocks = (total + 255) / 256;
    if (blocks > 256LL * 32) blocks = 256LL * 32;
    if (dtype == TT_F32)
        hipLaunchKernelGGL(deform_im2col_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, offsets, (float*)cols, N, H, W, C, off_cstride, pad);
    else if (dtype == TT_F16)
        hipLaunchKernelGGL(deform_im2col_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const f16_t*)x, offsets, (f16_t*)cols, N, H, W, C, off_cstride, pad);
    else
        hipLaunchKernelGGL(deform_im2col_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)x, offsets, (uint16_t*)cols, N, H, W, C, off_cstride, pad);
    return check_launch("tt_deform_im2col3x3");
}
