// Backward of the look module's gather / sample / reduce kernels (csrc/look_module.hip).  f32 only; the waypoint
// and control inputs of a refinement layer are DETACHED in the reference (thinktwice_decoder.py:429-430), so nothing flows
// into them or into the projected reference points.  Scatters into shared rows (FPN-side maps, value projections,
// embeddings, per-sample vectors) use f32 atomics.
#include "tt_common.h"

namespace tt {

constexpr int kBwdCams = 4;
constexpr int kBwdQ = 120;

struct BwdLevels {
    float* p[4];
    int H[4], W[4];
};

// scatter g * (bilinear corner weights) into channel c of a channel-last map; align_corners=False, zero padding
__device__ __forceinline__ void bilinear_scatter(float* __restrict__ map, int H, int W, int cstride, int c, float nx, float ny,
                                                 float g) {
    const float x = nx * (float)W - 0.5f, y = ny * (float)H - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const int x0 = (int)fx, y0 = (int)fy;
    const float lx = x - fx, ly = y - fy;
    if (y0 >= 0 && y0 < H) {
        if (x0 >= 0 && x0 < W) unsafeAtomicAdd(map + ((long long)y0 * W + x0) * cstride + c, (1.f - ly) * (1.f - lx) * g);
        if (x0 + 1 >= 0 && x0 + 1 < W) unsafeAtomicAdd(map + ((long long)y0 * W + x0 + 1) * cstride + c, (1.f - ly) * lx * g);
    }
    if (y0 + 1 >= 0 && y0 + 1 < H) {
        if (x0 >= 0 && x0 < W) unsafeAtomicAdd(map + ((long long)(y0 + 1) * W + x0) * cstride + c, ly * (1.f - lx) * g);
        if (x0 + 1 >= 0 && x0 + 1 < W) unsafeAtomicAdd(map + ((long long)(y0 + 1) * W + x0 + 1) * cstride + c, ly * lx * g);
    }
}

// look_gather_query backward: one workgroup per (b, cam, slot) row of the query matrix [ctrl4 | xyz3 | emb128 | meas128 |
// flat256 | samp 1024 (channel-major, level-minor)].
__global__ __launch_bounds__(256) void look_gather_query_bwd_kernel(const int* __restrict__ query_of_slot,
                                                                    const float* __restrict__ ref_packed,
                                                                    const float* __restrict__ dout, int row_stride,
                                                                    float* __restrict__ dtemporal, float* __restrict__ dstat,
                                                                    float* __restrict__ dmeas, float* __restrict__ dflat,
                                                                    BwdLevels maps) {
    const long long row = blockIdx.x;
    const int bc = (int)(row / kBwdQ), b = bc / kBwdCams;
    const int q = query_of_slot[row];
    if (q < 0) return;
    const float* g = dout + row * row_stride;
    const int t = threadIdx.x, pt = q / 15;
    if (t < 128) {
        unsafeAtomicAdd((pt < 4 ? dtemporal + pt * 128 : dstat + (pt - 4) * 128) + t, g[7 + t]);
        unsafeAtomicAdd(dmeas + b * 128 + t, g[135 + t]);
    }
    unsafeAtomicAdd(dflat + b * 256 + t, g[263 + t]);
    const float rx = ref_packed[row * 2 + 0], ry = ref_packed[row * 2 + 1];
#pragma unroll
    for (int l = 0; l < 4; ++l)
        bilinear_scatter(maps.p[l] + (long long)bc * maps.H[l] * maps.W[l] * 256, maps.H[l], maps.W[l], 256, t, rx, ry,
                         g[519 + t * 4 + l]);
}

// msda_sample backward: one workgroup per query row, thread = head * 32 + channel.  With w = softmax(logits of the head),
// s_i = bilinear sample i of the head's value channels:  out = sum_i w_i s_i
//   dvalue  += g w_i (corner weights)                      (atomics)
//   dw_i     = sum_channels g s_i,  dlogit_i = w_i (dw_i - sum_j w_j dw_j)
//   doffset_i.x = sum_channels g w_i [(v01 - v00)(1 - ly) + (v11 - v10) ly]   (d px / d off.x = 1 because off is in pixels)
__global__ __launch_bounds__(256) void msda_sample_bwd_kernel(const float* __restrict__ value, const float* __restrict__ offsets,
                                                              const float* __restrict__ logits, const float* __restrict__ ref,
                                                              const float* __restrict__ dout, BwdLevels lv, int S, int vcs,
                                                              int vco, float* __restrict__ dvalue,
                                                              float* __restrict__ doffsets, float* __restrict__ dlogits) {
    const long long row = blockIdx.x;
    const int bc = (int)(row / kBwdQ);
    const int t = threadIdx.x, head = t >> 5, lane32 = t & 31;
    const float* lg = logits + row * 256 + head * 32;
    float mx = -INFINITY;
    for (int i = 0; i < 32; ++i) mx = fmaxf(mx, lg[i]);
    float den = 0.f;
    for (int i = 0; i < 32; ++i) den += expf(lg[i] - mx);
    const float rx = ref[row * 2 + 0], ry = ref[row * 2 + 1];
    const float* of = offsets + row * 512 + head * 64;
    const float g = dout[row * 256 + t];
    float my_dw = 0.f, wdw = 0.f;          // lane i of the head keeps dw_i; wdw = sum_j w_j dw_j
    long long start = 0;
    for (int l = 0; l < 4; ++l) {
        const int H = lv.H[l], W = lv.W[l];
        const float* map = value + ((long long)bc * S + start) * vcs + vco;
        float* dmap = dvalue + ((long long)bc * S + start) * vcs + vco;
        for (int p = 0; p < 8; ++p) {
            const int i = l * 8 + p;
            const float w = expf(lg[i] - mx) / den;
            const float nx = rx + of[i * 2 + 0] / (float)W, ny = ry + of[i * 2 + 1] / (float)H;
            const float x = nx * (float)W - 0.5f, y = ny * (float)H - 0.5f;
            const float fx = floorf(x), fy = floorf(y);
            const int x0 = (int)fx, y0 = (int)fy;
            const float lx = x - fx, ly = y - fy;
            const bool r0 = y0 >= 0 && y0 < H, r1 = y0 + 1 >= 0 && y0 + 1 < H;
            const bool c0 = x0 >= 0 && x0 < W, c1 = x0 + 1 >= 0 && x0 + 1 < W;
            const float v00 = (r0 && c0) ? map[((long long)y0 * W + x0) * vcs + t] : 0.f;
            const float v01 = (r0 && c1) ? map[((long long)y0 * W + x0 + 1) * vcs + t] : 0.f;
            const float v10 = (r1 && c0) ? map[((long long)(y0 + 1) * W + x0) * vcs + t] : 0.f;
            const float v11 = (r1 && c1) ? map[((long long)(y0 + 1) * W + x0 + 1) * vcs + t] : 0.f;
            const float s = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
            float dws = g * s;
            float dox = g * w * ((v01 - v00) * (1.f - ly) + (v11 - v10) * ly);
            float doy = g * w * ((v10 - v00) * (1.f - lx) + (v11 - v01) * lx);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {              // over the head's 32 channels (half a wave)
                dws += __shfl_xor(dws, o);
                dox += __shfl_xor(dox, o);
                doy += __shfl_xor(doy, o);
            }
            if (lane32 == i) my_dw = dws;
            wdw += w * dws;
            if (lane32 == 0) {
                doffsets[row * 512 + head * 64 + i * 2 + 0] += dox;
                doffsets[row * 512 + head * 64 + i * 2 + 1] += doy;
            }
            const float gw = g * w;
            if (r0 && c0) unsafeAtomicAdd(dmap + ((long long)y0 * W + x0) * vcs + t, gw * (1.f - ly) * (1.f - lx));
            if (r0 && c1) unsafeAtomicAdd(dmap + ((long long)y0 * W + x0 + 1) * vcs + t, gw * (1.f - ly) * lx);
            if (r1 && c0) unsafeAtomicAdd(dmap + ((long long)(y0 + 1) * W + x0) * vcs + t, gw * ly * (1.f - lx));
            if (r1 && c1) unsafeAtomicAdd(dmap + ((long long)(y0 + 1) * W + x0 + 1) * vcs + t, gw * ly * lx);
        }
        start += (long long)H * W;
    }
    const float wi = expf(lg[lane32] - mx) / den;
    dlogits[row * 256 + head * 32 + lane32] += wi * (my_dw - wdw);
}

// sca_reduce backward: out[bc][c] = sum_{k = B .. min(max_len, 120) - 1} x[bc][k][c] / B
__global__ __launch_bounds__(256) void sca_reduce_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ max_len, int B,
                                                             float* __restrict__ dx) {
    const int bc = blockIdx.x, c = threadIdx.x;
    const int ml = min(*max_len, kBwdQ);
    const float g = dout[(long long)bc * 256 + c] / (float)B;
    for (int k = B; k < ml; ++k) dx[((long long)bc * kBwdQ + k) * 256 + c] += g;
}

}  // namespace tt

using namespace tt;

static void fill_bwd_levels(BwdLevels& m, float* const* maps, const int* hw) {
    for (int l = 0; l < 4; ++l) {
        m.p[l] = maps ? maps[l] : nullptr;
        m.H[l] = hw[2 * l];
        m.W[l] = hw[2 * l + 1];
    }
}

extern "C" int tt_look_gather_query_bwd(int B, const int* query_of_slot, const float* ref_packed, const float* dout,
                                        int row_stride, float* dtemporal, float* dstatic, float* dmeas, float* dflat,
                                        float* const* dlevel_maps, const int* level_hw, void* stream) {
    TT_REQUIRE(query_of_slot && ref_packed && dout && dtemporal && dstatic && dmeas && dflat && dlevel_maps && level_hw,
               "tt_look_gather_query_bwd: null");
    BwdLevels m;
    fill_bwd_levels(m, dlevel_maps, level_hw);
    hipLaunchKernelGGL(look_gather_query_bwd_kernel, dim3((unsigned)(B * kBwdCams * kBwdQ)), dim3(256), 0, (hipStream_t)stream,
                       query_of_slot, ref_packed, dout, row_stride, dtemporal, dstatic, dmeas, dflat, m);
    return check_launch("tt_look_gather_query_bwd");
}

extern "C" int tt_msda_sample_bwd(int B, const float* value, int value_cstride, int value_coff, const float* offsets,
                                  const float* logits, const float* ref_packed, const int* level_hw, const float* dout,
                                  float* dvalue, float* doffsets, float* dlogits, void* stream) {
    TT_REQUIRE(value && offsets && logits && ref_packed && level_hw && dout && dvalue && doffsets && dlogits,
               "tt_msda_sample_bwd: null");
    BwdLevels m;
    fill_bwd_levels(m, nullptr, level_hw);
    int S = 0;
    for (int l = 0; l < 4; ++l) S += m.H[l] * m.W[l];
    hipLaunchKernelGGL(msda_sample_bwd_kernel, dim3((unsigned)(B * kBwdCams * kBwdQ)), dim3(256), 0, (hipStream_t)stream, value,
                       offsets, logits, ref_packed, dout, m, S, value_cstride, value_coff, dvalue, doffsets, dlogits);
    return check_launch("tt_msda_sample_bwd");
}

extern "C" int tt_sca_reduce_bwd(int B, const float* dout, const int* max_len, float* dx, void* stream) {
    TT_REQUIRE(dout && max_len && dx && B > 0, "tt_sca_reduce_bwd: null");
    hipLaunchKernelGGL(sca_reduce_bwd_kernel, dim3((unsigned)(B * kBwdCams)), dim3(256), 0, (hipStream_t)stream, dout, max_len, B,
                       dx);
    return check_launch("tt_sca_reduce_bwd");
}
