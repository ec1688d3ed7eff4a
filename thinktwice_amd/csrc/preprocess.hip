// Fused camera preprocessing: raw uint8 900x1600x3 frames -> network input.
// Replaces, per tick, the CPU chain of the reference data pipeline
//   IDAImageTransform.__call__   undistort = F.grid_sample(img, map_grid, align_corners=False)      transform.py:283-286
//   img_transform                T.Resize((504, 896)) bilinear + crop rows 56:504 (eval IDA)          transform.py:346-356
//   ImageTransformMulti          .div(255) + Normalize(mean, std)                                     transform.py:144,163
// in ONE gather kernel: every output pixel blends 2x2 taps of the (virtual) undistorted image, each of
// which is a bilinear read of the raw frame through the undistortion map (zero padding), so the
// 17.3 MB uint8 frame set is read once and the normalised tensor is written once, channel-last.
#include "tt_common.h"

namespace tt {

struct PreArgs {
    int NI, H, W;            // raw frames
    int RH, RW;              // resized size (504, 896)
    int crop_y, crop_x;      // crop origin in the resized image
    int OH, OW, Cp;          // output size and padded channels
    float mean[3], inv_std[3];
};

__device__ __forceinline__ float raw_at(const uint8_t* __restrict__ img, int H, int W, int y, int x, int c) {
    return (y >= 0 && y < H && x >= 0 && x < W) ? (float)img[((long long)y * W + x) * 3 + c] : 0.f;
}

// undistorted(Y, X, c) = grid_sample(raw, map)(Y, X): bilinear at (mapx - 0.5, mapy - 0.5), zeros outside
__device__ __forceinline__ void undist_px(const uint8_t* __restrict__ img, const float* __restrict__ mapx,
                                          const float* __restrict__ mapy, int H, int W, int Y, int X, float out[3]) {
    const float px = mapx[(long long)Y * W + X] - 0.5f;   // ((mapx-W/2)/(W/2) + 1) * W / 2 - 0.5
    const float py = mapy[(long long)Y * W + X] - 0.5f;
    const float fx = floorf(px), fy = floorf(py);
    const int x0 = (int)fx, y0 = (int)fy;
    const float lx = px - fx, ly = py - fy;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v00 = raw_at(img, H, W, y0, x0, c), v01 = raw_at(img, H, W, y0, x0 + 1, c);
        const float v10 = raw_at(img, H, W, y0 + 1, x0, c), v11 = raw_at(img, H, W, y0 + 1, x0 + 1, c);
        out[c] = v00 * (1.f - lx) * (1.f - ly) + v01 * lx * (1.f - ly) + v10 * (1.f - lx) * ly + v11 * lx * ly;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(PreArgs a, const uint8_t* __restrict__ raw,
                                                         const float* __restrict__ mapx,
                                                         const float* __restrict__ mapy, T* __restrict__ out,
                                                         float* __restrict__ out_nchw) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a.NI * a.OH * a.OW;
    if (t >= total) return;
    const int ox = (int)(t % a.OW);
    const int oy = (int)((t / a.OW) % a.OH);
    const int n = (int)(t / ((long long)a.OW * a.OH));
    // F.interpolate(bilinear, align_corners=False): src = (dst + 0.5) * in/out - 0.5, clamped at 0
    const float sy = fmaxf(((float)(oy + a.crop_y) + 0.5f) * ((float)a.H / (float)a.RH) - 0.5f, 0.f);
    const float sx = fmaxf(((float)(ox + a.crop_x) + 0.5f) * ((float)a.W / (float)a.RW) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, a.H - 1), x1 = min(x0 + 1, a.W - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const uint8_t* img = raw + (long long)n * a.H * a.W * 3;
    float p00[3], p01[3], p10[3], p11[3];
    undist_px(img, mapx, mapy, a.H, a.W, y0, x0, p00);
    undist_px(img, mapx, mapy, a.H, a.W, y0, x1, p01);
    undist_px(img, mapx, mapy, a.H, a.W, y1, x0, p10);
    undist_px(img, mapx, mapy, a.H, a.W, y1, x1, p11);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = (1.f - lx) * p00[c] + lx * p01[c];
        const float bot = (1.f - lx) * p10[c] + lx * p11[c];
        float v = (1.f - ly) * top + ly * bot;
        v = (v / 255.f - a.mean[c]) * a.inv_std[c];
        if (out) Elem<T>::st(out + t * a.Cp + c, v);
        if (out_nchw) out_nchw[(((long long)n * 3 + c) * a.OH + oy) * a.OW + ox] = v;
    }
    if (out)
        for (int c = 3; c < a.Cp; ++c) Elem<T>::st(out + t * a.Cp + c, 0.f);
}

// LiDAR half-sweep merge of the agent tick (leaderboard/team_code/thinktwice_agent.py:340-352): the simulator runs at
// 20 Hz, the LiDAR at 10 Hz, so every tick delivers a 180-degree half sweep; the previous half sweep is moved into the
// current ego frame (rigid planar transform, rows of `mat` = first three rows of inv(T_now) @ T_prev), the two halves
// are concatenated [previous | current] and the sensor height (z += 2.5) is added to every point.  Points are
// (x, y, z, intensity) f32.
__global__ void lidar_merge_kernel(const float* __restrict__ prev, int n_prev, const float* __restrict__ now, int n_now,
                                   float m00, float m01, float m02, float m03, float m10, float m11, float m12, float m13,
                                   float m20, float m21, float m22, float m23, float z_shift, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_prev + n_now) return;
    float4 p;
    if (i < n_prev) {
        const float4 q = reinterpret_cast<const float4*>(prev)[i];
        p.x = m00 * q.x + m01 * q.y + m02 * q.z + m03;
        p.y = m10 * q.x + m11 * q.y + m12 * q.z + m13;
        p.z = m20 * q.x + m21 * q.y + m22 * q.z + m23;
        p.w = q.w;
    } else {
        p = reinterpret_cast<const float4*>(now)[i - n_prev];
    }
    p.z += z_shift;
    reinterpret_cast<float4*>(out)[i] = p;
}

}  // namespace tt

using namespace tt;

extern "C" int tt_preprocess_images(const uint8_t* raw_hwc, int num_images, int H, int W, const float* mapx,
                                    const float* mapy, int resized_h, int resized_w, int crop_y, int crop_x,
                                    int out_h, int out_w, const float* mean3, const float* std3, void* out_nhwc,
                                    int out_channels_padded, int out_dtype, float* out_nchw_or_null, void* stream) {
    TT_REQUIRE(raw_hwc && mapx && mapy && mean3 && std3 && (out_nhwc || out_nchw_or_null), "tt_preprocess_images: null");
    TT_REQUIRE(out_channels_padded >= 3, "tt_preprocess_images: need >= 3 output channels");
    PreArgs a;
    a.NI = num_images; a.H = H; a.W = W; a.RH = resized_h; a.RW = resized_w; a.crop_y = crop_y; a.crop_x = crop_x;
    a.OH = out_h; a.OW = out_w; a.Cp = out_channels_padded;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.inv_std[c] = 1.f / std3[c]; }
    const long long total = (long long)num_images * out_h * out_w;
    const dim3 grid((unsigned)div_up(total, 256));
    if (out_dtype == TT_F32)
        hipLaunchKernelGGL(preprocess_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, a, raw_hwc, mapx, mapy,
                           (float*)out_nhwc, out_nchw_or_null);
    else if (out_dtype == TT_BF16)
        hipLaunchKernelGGL(preprocess_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, a, raw_hwc, mapx, mapy,
                           (uint16_t*)out_nhwc, out_nchw_or_null);
    else if (out_dtype == TT_F16)
        hipLaunchKernelGGL(preprocess_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, a, raw_hwc, mapx, mapy,
                           (f16_t*)out_nhwc, out_nchw_or_null);
    else
        TT_REQUIRE(false, "tt_preprocess_images: bad dtype");
    return check_launch("tt_preprocess_images");
}

extern "C" int tt_lidar_merge_half_sweeps(const float* prev_xyzi, int n_prev, const float* now_xyzi, int n_now,
                                          const float* rel_transform_3x4, float z_shift, float* out_xyzi, void* stream) {
    TT_REQUIRE(rel_transform_3x4 && n_now >= 0 && n_prev >= 0 && (n_prev == 0 || prev_xyzi) && (n_now == 0 || now_xyzi),
               "tt_lidar_merge_half_sweeps: bad arguments");
    const int n = n_prev + n_now;
    if (n == 0) return 0;
    TT_REQUIRE(out_xyzi, "tt_lidar_merge_half_sweeps: null output");
    const float* m = rel_transform_3x4;
    hipLaunchKernelGGL(lidar_merge_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, prev_xyzi,
                       n_prev, now_xyzi, n_now, m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11],
                       z_shift, out_xyzi);
    return check_launch("tt_lidar_merge_half_sweeps");
}
