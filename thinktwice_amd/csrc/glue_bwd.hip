// Backward of the HBM-bound glue kernels of the camera trunk.  f32, channel-last, gather form (every output
// element is written by one thread: deterministic, no atomics); gradients are ACCUMULATED into their destination, the
// buffers are zeroed once per step by the tape (thinktwice_amd/autodiff.py).
#include "tt_common.h"

namespace tt {

// F.max_pool2d(x, 3, 2, 1) backward.  torch keeps the FIRST maximum of a window in (kh, kw) scan order (`val > maxval`),
// which matters here: post-ReLU maps are full of exact ties at 0.
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ dx, int N, int H, int W, int C, int OH,
                                                               int OW) {
    const long long total = (long long)N * H * W * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int iw = (int)(r % W); r /= W;
        const int ih = (int)(r % H);
        const long long n = r / H;
        const float* xn = x + n * H * W * C + c;
        float g = 0.f;
        // windows containing (ih, iw): oh with 2*oh - 1 <= ih <= 2*oh + 1
        for (int oh = (ih) / 2; oh <= (ih + 1) / 2; ++oh) {
            if (oh >= OH) continue;
            for (int ow = (iw) / 2; ow <= (iw + 1) / 2; ++ow) {
                if (ow >= OW) continue;
                float m = -INFINITY;
                int arg = -1;
                for (int dh = 0; dh < 3; ++dh) {
                    const int yy = oh * 2 - 1 + dh;
                    if (yy < 0 || yy >= H) continue;
                    for (int dw = 0; dw < 3; ++dw) {
                        const int xx = ow * 2 - 1 + dw;
                        if (xx < 0 || xx >= W) continue;
                        const float v = xn[((long long)yy * W + xx) * C];
                        if (v > m || arg < 0) {
                            m = v;
                            arg = yy * W + xx;
                        }
                    }
                }
                if (arg == ih * W + iw) g += dy[((n * OH + oh) * OW + ow) * C + c];
            }
        }
        dx[i] += g;
    }
}

__global__ __launch_bounds__(256) void upsample_nearest_add_bwd_kernel(const float* __restrict__ ddst, float* __restrict__ dsrc,
                                                                       int N, int H, int W, int C, int h, int w) {
    const long long total = (long long)N * h * w * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int sx = (int)(r % w); r /= w;
        const int sy = (int)(r % h);
        const long long n = r / h;
        // dst pixels y with floor(y * h / H) == sy:  y in [ceil(sy * H / h), ceil((sy + 1) * H / h))
        const int y0 = (int)(((long long)sy * H + h - 1) / h), y1 = (int)(((long long)(sy + 1) * H + h - 1) / h);
        const int x0 = (int)(((long long)sx * W + w - 1) / w), x1 = (int)(((long long)(sx + 1) * W + w - 1) / w);
        float g = 0.f;
        for (int y = y0; y < y1 && y < H; ++y)
            for (int xx = x0; xx < x1 && xx < W; ++xx) g += ddst[((n * H + y) * W + xx) * C + c];
        dsrc[i] += g;
    }
}

}  // namespace tt

using namespace tt;

static unsigned bwd_grid(long long total) {
    const long long b = (total + 255) / 256;
    return (unsigned)(b > 65536 ? 65536 : (b < 1 ? 1 : b));
}

extern "C" int tt_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    TT_REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0, "tt_maxpool3x3s2_bwd: bad argument");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(bwd_grid((long long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream,
                       x, dy, dx, N, H, W, C, OH, OW);
    return check_launch("tt_maxpool3x3s2_bwd");
}

extern "C" int tt_upsample_nearest_add_bwd(const float* ddst, float* dsrc, int N, int H, int W, int C, int h, int w,
                                           void* stream) {
    TT_REQUIRE(ddst && dsrc && N > 0 && H >= h && W >= w && h > 0 && w > 0 && C > 0, "tt_upsample_nearest_add_bwd: bad argument");
    hipLaunchKernelGGL(upsample_nearest_add_bwd_kernel, dim3(bwd_grid((long long)N * h * w * C)), dim3(256), 0,
                       (hipStream_t)stream, ddst, dsrc, N, H, W, C, h, w);
    return check_launch("tt_upsample_nearest_add_bwd");
}
