// "Look" module kernels of the ThinkTwice decoder (dense_heads/thinktwice_decoder.py:88-187,
// dense_heads/multi_scale_deformable_attn_function.py:279-344,423-526).
//
// The reference builds per-(sample, camera) query lists with B*4 host-synchronising nonzero()
// calls per layer and pads them to a data-dependent max_len.  Here everything stays on the device:
//   look_project_pack : project the 120 3-D look points into the 4 cameras, in-image mask, stable
//                       left-packing per (sample, camera) by wave ballot, max_len by atomicMax.
//   look_gather_query : assemble the 1543-wide query rows (519 base + 4 levels x 256 bilinear
//                       samples of the FPN maps) for all 120 slots (padded slots = zero rows).
//   msda_sample       : multi-scale deformable attention core (softmax over 32 (level, point) logits,
//                       bilinear sampling with zero padding, align_corners=False).
//   sca_reduce        : the reference's batch-coupled "mask & average" (first B slots zeroed, / B,
//                       sum over slots < max_len) -- reproduced as is.
#include "tt_common.h"

namespace tt {

constexpr int kQ = 120;     // 8 BEV points x 15 heights
constexpr int kCams = 4;

__device__ __forceinline__ float dot4(const float* m, float a, float b, float c, float d) {
    float s = m[0] * a;
    s = s + m[1] * b;
    s = s + m[2] * c;
    s = s + m[3] * d;
    return s;
}

// one wave per (b, cam).  wp (B,4,2) f32; outputs: ref_packed (B,4,120,2), query_of_slot (B,4,120) int,
// count (B,4) int, max_len (1) int (atomicMax; must be zeroed before launch).
__global__ __launch_bounds__(64) void look_project_pack_kernel(
    const float* __restrict__ wp, const float* __restrict__ lidar2img, const float* __restrict__ ida,
    float img_h, float img_w, float* __restrict__ ref_packed, int* __restrict__ query_of_slot,
    int* __restrict__ count, int* __restrict__ max_len, float* __restrict__ pts3d_out) {
    const int bc = blockIdx.x;
    const int b = bc / kCams;
    const int lane = threadIdx.x;
    const float* L = lidar2img + (long long)bc * 16;
    const float* A = ida + (long long)bc * 16;
    int base = 0;
    for (int half = 0; half < 2; ++half) {
        const int q = half * 64 + lane;
        bool ok = false;
        float rx = 0.f, ry = 0.f;
        if (q < kQ) {
            const int pt = q / 15, zi = q % 15;
            float X, Y;
            if (pt < 4) {
                X = wp[(b * 4 + pt) * 2 + 0];
                Y = wp[(b * 4 + pt) * 2 + 1];
            } else {   // static look points (DEC:157)
                const float sx[4] = {5.f, 0.f, 0.f, -5.f};
                const float sy[4] = {0.f, -5.f, 5.f, 0.f};
                X = sx[pt - 4];
                Y = sy[pt - 4];
            }
            const float Z = (float)(-4.0 + (double)zi * (14.0 / 14.0));   // linspace(-4, 10, 15)
            if (pts3d_out && (bc % kCams) == 0) {
                pts3d_out[(b * kQ + q) * 3 + 0] = X;
                pts3d_out[(b * kQ + q) * 3 + 1] = Y;
                pts3d_out[(b * kQ + q) * 3 + 2] = Z;
            }
            const float cx = dot4(L + 0, X, Y, Z, 1.f);
            const float cy = dot4(L + 4, X, Y, Z, 1.f);
            const float cz = dot4(L + 8, X, Y, Z, 1.f);
            const float cw = dot4(L + 12, X, Y, Z, 1.f);
            const float eps = 1e-5f;
            const float zz = fmaxf(cz, eps);
            const float ux = cx / zz, uy = cy / zz;
            const float ix = dot4(A + 0, ux, uy, cz, cw);
            const float iy = dot4(A + 4, ux, uy, cz, cw);
            const float iz = dot4(A + 8, ux, uy, cz, cw);
            rx = ix / img_w;
            ry = iy / img_h;
            ok = (iz > eps) && (ry > 0.f) && (ry < 1.f) && (rx < 1.f) && (rx > 0.f);
        }
        const unsigned long long m = __ballot(ok);
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (ok) {
            query_of_slot[(long long)bc * kQ + pos] = q;
            ref_packed[((long long)bc * kQ + pos) * 2 + 0] = rx;
            ref_packed[((long long)bc * kQ + pos) * 2 + 1] = ry;
        }
        base += __popcll(m);
    }
    // padded slots
    for (int s = base + lane; s < kQ; s += 64) {
        query_of_slot[(long long)bc * kQ + s] = -1;
        ref_packed[((long long)bc * kQ + s) * 2 + 0] = 0.f;
        ref_packed[((long long)bc * kQ + s) * 2 + 1] = 0.f;
    }
    if (lane == 0) {
        count[bc] = base;
        atomicMax(max_len, base);
    }
}

// bilinear sample of a channel-last map at normalised (x, y), align_corners=False, zero padding
template <typename T>
__device__ __forceinline__ float bilinear_cl(const T* __restrict__ map, int H, int W, int cstride, int c,
                                             float nx, float ny) {
    const float x = nx * (float)W - 0.5f, y = ny * (float)H - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const int x0 = (int)fx, y0 = (int)fy;
    const float lx = x - fx, ly = y - fy;
    float v = 0.f;
    if (y0 >= 0 && y0 < H) {
        if (x0 >= 0 && x0 < W) v += (1.f - ly) * (1.f - lx) * Elem<T>::ld(map + ((long long)y0 * W + x0) * cstride + c);
        if (x0 + 1 >= 0 && x0 + 1 < W) v += (1.f - ly) * lx * Elem<T>::ld(map + ((long long)y0 * W + x0 + 1) * cstride + c);
    }
    if (y0 + 1 >= 0 && y0 + 1 < H) {
        if (x0 >= 0 && x0 < W) v += ly * (1.f - lx) * Elem<T>::ld(map + ((long long)(y0 + 1) * W + x0) * cstride + c);
        if (x0 + 1 >= 0 && x0 + 1 < W) v += ly * lx * Elem<T>::ld(map + ((long long)(y0 + 1) * W + x0 + 1) * cstride + c);
    }
    return v;
}

struct LevelMaps {
    const void* p[4];
    int H[4], W[4];
};

// one block (256 threads) per (b, cam, slot): query row = [ctrl4 | xyz3 | emb128 | meas128 | flat256 | samp 1024]
template <typename T>
__global__ __launch_bounds__(256) void look_gather_query_kernel(
    const int* __restrict__ query_of_slot, const float* __restrict__ ref_packed, const float* __restrict__ wp,
    const float* __restrict__ ctrl_sp, const float* __restrict__ temporal, const float* __restrict__ stat,
    const float* __restrict__ meas, const float* __restrict__ flat, LevelMaps maps, float* __restrict__ out,
    int row_stride) {
    const long long row = blockIdx.x;               // (b*4+cam)*120 + slot
    const int bc = (int)(row / kQ);
    const int b = bc / kCams;
    const int q = query_of_slot[row];
    float* o = out + row * row_stride;
    const int t = threadIdx.x;
    if (q < 0) {
        for (int i = t; i < row_stride; i += 256) o[i] = 0.f;
        return;
    }
    const int pt = q / 15, zi = q % 15;
    if (t < 4) o[t] = (pt < 4) ? ctrl_sp[(b * 4 + pt) * 4 + t] : 0.f;
    if (t == 4) {
        const float sx[4] = {5.f, 0.f, 0.f, -5.f};
        const float sy[4] = {0.f, -5.f, 5.f, 0.f};
        o[4] = (pt < 4) ? wp[(b * 4 + pt) * 2 + 0] : sx[pt - 4];
        o[5] = (pt < 4) ? wp[(b * 4 + pt) * 2 + 1] : sy[pt - 4];
        o[6] = (float)(-4.0 + (double)zi);
    }
    if (t < 128) {
        o[7 + t] = (pt < 4) ? temporal[pt * 128 + t] : stat[(pt - 4) * 128 + t];
        o[135 + t] = meas[b * 128 + t];
    }
    o[263 + t] = flat[b * 256 + t];
    const float rx = ref_packed[row * 2 + 0], ry = ref_packed[row * 2 + 1];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const T* map = reinterpret_cast<const T*>(maps.p[l]) + (long long)bc * maps.H[l] * maps.W[l] * 256;
        o[519 + t * 4 + l] = bilinear_cl<T>(map, maps.H[l], maps.W[l], 256, t, rx, ry);
    }
    if (t < row_stride - 1543) o[1543 + t] = 0.f;
}

// one block per (bc, slot): thread = head*32 + channel.  value [B*4][S][256] (T), offsets f32 [R][512]
// ((head, level, point, xy)), logits f32 [R][256] ((head, level*point)), ref f32 [R][2] -> out f32 [R][256]
template <typename T>
__global__ __launch_bounds__(256) void msda_sample_kernel(const T* __restrict__ value,
                                                          const float* __restrict__ offsets,
                                                          const float* __restrict__ logits,
                                                          const float* __restrict__ ref, LevelMaps lv,
                                                          int S, int vcs, int vco, float* __restrict__ out) {
    const long long row = blockIdx.x;
    const int bc = (int)(row / kQ);
    const int t = threadIdx.x, head = t >> 5;
    const float* lg = logits + row * 256 + head * 32;
    float mx = -INFINITY;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) mx = fmaxf(mx, lg[i]);
    float den = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) den += expf(lg[i] - mx);
    const float rx = ref[row * 2 + 0], ry = ref[row * 2 + 1];
    const float* of = offsets + row * 512 + head * 64;
    float acc = 0.f;
    long long start = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const int H = lv.H[l], W = lv.W[l];
        const T* map = value + ((long long)bc * S + start) * vcs + vco;   // 256 of vcs channels per position
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float w = expf(lg[l * 8 + p] - mx) / den;
            const float nx = rx + of[(l * 8 + p) * 2 + 0] / (float)W;
            const float ny = ry + of[(l * 8 + p) * 2 + 1] / (float)H;
            acc += w * bilinear_cl<T>(map, H, W, vcs, t, nx, ny);
        }
        start += (long long)H * W;
    }
    out[row * 256 + t] = acc;
}

// out[b, cam*256 + c] = (1/B) * sum_{s = B}^{max_len-1} x[(b*4+cam)*120 + s, c]
__global__ __launch_bounds__(256) void sca_reduce_kernel(const float* __restrict__ x, const int* __restrict__ max_len,
                                                         int B, float* __restrict__ out) {
    const int bc = blockIdx.x, c = threadIdx.x;
    const int ml = min(*max_len, kQ);
    float s = 0.f;
    for (int k = B; k < ml; ++k) s += x[((long long)bc * kQ + k) * 256 + c] / (float)B;
    out[(long long)bc * 256 + c] = s;
}

// ------------------------------------------------------------------------------------------------------------------
// Fused row producers of the composite decoder: the same gathers / reductions as above with the nn.LayerNorm that the
// reference applies next (query_linear.0, ffn.norm, output_proj.0, mlp.0) folded in, so that each feeds tt_mlp_chain
// directly (one launch instead of producer + tt_layernorm_rows).
__device__ __forceinline__ float block_sum256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();                                   // previous use of `red` is over
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// look_gather_query + LayerNorm(1543) (thinktwice_decoder.py:131-150 query_linear.0).  `ctrl` is the RAW control
// parameters when raw_ctrl != 0 (softplus applied here, DEC:239), else already softplus'ed.
template <typename T>
__global__ __launch_bounds__(256) void look_query_ln_kernel(
    const int* __restrict__ query_of_slot, const float* __restrict__ ref_packed, const float* __restrict__ wp,
    const float* __restrict__ ctrl, int raw_ctrl, const float* __restrict__ temporal, const float* __restrict__ stat,
    const float* __restrict__ meas, const float* __restrict__ flat, LevelMaps maps, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float* __restrict__ out, int row_stride) {
    __shared__ float srow[1544];
    __shared__ float red[4];
    const long long row = blockIdx.x;               // (b*4+cam)*120 + slot
    const int bc = (int)(row / kQ);
    const int b = bc / kCams;
    const int q = query_of_slot[row];
    const int t = threadIdx.x;
    if (q < 0) {
        for (int i = t; i < 1543; i += 256) srow[i] = 0.f;
    } else {
        const int pt = q / 15, zi = q % 15;
        if (t < 4) {
            float c = (pt < 4) ? ctrl[(b * 4 + pt) * 4 + t] : 0.f;
            if (raw_ctrl && pt < 4) c = apply_act(c, TT_ACT_SOFTPLUS);
            srow[t] = c;
        }
        if (t == 4) {
            const float sx[4] = {5.f, 0.f, 0.f, -5.f};
            const float sy[4] = {0.f, -5.f, 5.f, 0.f};
            srow[4] = (pt < 4) ? wp[(b * 4 + pt) * 2 + 0] : sx[pt - 4];
            srow[5] = (pt < 4) ? wp[(b * 4 + pt) * 2 + 1] : sy[pt - 4];
            srow[6] = (float)(-4.0 + (double)zi);
        }
        if (t < 128) {
            srow[7 + t] = (pt < 4) ? temporal[pt * 128 + t] : stat[(pt - 4) * 128 + t];
            srow[135 + t] = meas[b * 128 + t];
        }
        srow[263 + t] = flat[b * 256 + t];
        const float rx = ref_packed[row * 2 + 0], ry = ref_packed[row * 2 + 1];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const T* map = reinterpret_cast<const T*>(maps.p[l]) + (long long)bc * maps.H[l] * maps.W[l] * 256;
            srow[519 + t * 4 + l] = bilinear_cl<T>(map, maps.H[l], maps.W[l], 256, t, rx, ry);
        }
    }
    __syncthreads();
    float s = 0.f;
    for (int i = t; i < 1543; i += 256) s += srow[i];
    const float mean = block_sum256(s, red) / 1543.f;
    float qq = 0.f;
    for (int i = t; i < 1543; i += 256) {
        const float d = srow[i] - mean;
        qq += d * d;
    }
    const float rstd = 1.f / sqrtf(block_sum256(qq, red) / 1543.f + eps);
    float* o = out + row * row_stride;
    for (int i = t; i < row_stride; i += 256) o[i] = (i < 1543) ? (srow[i] - mean) * rstd * gamma[i] + beta[i] : 0.f;
}

// msda_sample + LayerNorm(256) (ffn.norm, multi_scale_deformable_attn_function.py:262): out = raw attention rows (the
// residual of the ffn), out_ln = their LayerNorm (the ffn input)
template <typename T>
__global__ __launch_bounds__(256) void msda_sample_ln_kernel(const T* __restrict__ value,
                                                             const float* __restrict__ offsets,
                                                             const float* __restrict__ logits,
                                                             const float* __restrict__ ref, LevelMaps lv, int S, int vcs,
                                                             int vco, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             float* __restrict__ out, float* __restrict__ out_ln,
                                                             const int* __restrict__ max_len) {
    __shared__ float red[4];
    const long long row = blockIdx.x;
    const int bc = (int)(row / kQ);
    // slots at or beyond the longest camera list are never read (tt_sca_reduce_ln sums k < max_len): their rows -- random 128 B
    // reads of the value maps, the kernel's whole cost -- are skipped and left unwritten
    if (max_len && (int)(row - (long long)bc * kQ) >= *max_len) return;
    const int t = threadIdx.x, head = t >> 5;
    const float* lg = logits + row * 256 + head * 32;
    float mx = -INFINITY;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) mx = fmaxf(mx, lg[i]);
    float den = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) den += expf(lg[i] - mx);
    const float rx = ref[row * 2 + 0], ry = ref[row * 2 + 1];
    const float* of = offsets + row * 512 + head * 64;
    float acc = 0.f;
    long long start = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const int H = lv.H[l], W = lv.W[l];
        const T* map = value + ((long long)bc * S + start) * vcs + vco;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float w = expf(lg[l * 8 + p] - mx) / den;
            const float nx = rx + of[(l * 8 + p) * 2 + 0] / (float)W;
            const float ny = ry + of[(l * 8 + p) * 2 + 1] / (float)H;
            acc += w * bilinear_cl<T>(map, H, W, vcs, t, nx, ny);
        }
        start += (long long)H * W;
    }
    out[row * 256 + t] = acc;
    const float mean = block_sum256(acc, red) / 256.f;
    const float d = acc - mean;
    const float rstd = 1.f / sqrtf(block_sum256(d * d, red) / 256.f + eps);
    out_ln[row * 256 + t] = d * rstd * gamma[t] + beta[t];
}

// "Sample first, project after" (round 6): the same attention rows WITHOUT the value tensor.  value_proj is linear and the
// bilinear sample with zero padding is linear in the map, so for head h
//     sum_s w_s * sample(W_h x + b_h + e_{level, cam}, loc_s)  =  W_h . z_h  +  sum_level beta_{h, level} * (b_h + e_{level, cam}),
//     z_h = sum_s w_s * sample(x, loc_s)  (256 raw FPN channels),   beta_{h, level} = sum_{s in level} w_s * (in-bounds corner weight)
// (multi_scale_deformable_attn_function.py:474 projects all 33,320 positions x 4 cameras x B of every layer -- 5.5 GB of f32 at
// B = 8 of which the sampler reads a few per cent; here a row gathers the raw 1 KiB FPN rows of its 8 x 32 x 4 corners and applies
// the head's 32 x 256 slice once).  One block (four groups of 256 threads) per (b, cam, slot); phase 0: thread = (head, sample) -> softmax weight, corner
// positions and weights into LDS; phase 1: thread = input channel, eight accumulators z_h[k] over the 1024 corner rows (coalesced
// 1 KiB loads); phase 2: thread = output channel, a 256-long dot product with the transposed weight (coalesced) + the bias terms;
// then the LayerNorm of msda_sample_ln_kernel.  Exact f32 FMAs.
__global__ __launch_bounds__(1024) void msda_sample_proj_ln_kernel(LevelMaps lv, const float* __restrict__ offsets,
                                                                   const float* __restrict__ logits,
                                                                   const float* __restrict__ ref, const float* __restrict__ wvT,
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ vshift,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float eps,
                                                                   float* __restrict__ out, float* __restrict__ out_ln,
                                                                   const int* __restrict__ max_len) {
    // 1024 threads = 4 groups of 256: a row's 1 MiB of corner rows is latency-bound on ONE workgroup (184 rows per tick at batch 1),
    // so the heads (phase 1) and the K range of the projection (phase 2) are dealt over four groups
    __shared__ float red[16];
    __shared__ int cidx[8 * 32 * 4];          // corner position inside its level map, -1 = outside (zero padding)
    __shared__ float cwt[8 * 32 * 4];         // attention weight x bilinear corner weight
    __shared__ float sbeta[8 * 4];
    __shared__ float z[8 * 256];
    __shared__ float part[4 * 256];
    const long long row = blockIdx.x;
    const int bc = (int)(row / kQ);
    if (max_len && (int)(row - (long long)bc * kQ) >= *max_len) return;          // block-uniform: slots nobody reads
    const int tid = threadIdx.x, grp = tid >> 8, t = tid & 255;
    if (grp == 0) {                                                              // phase 0: thread = (head, sample)
        const int head = t >> 5, i = t & 31, l = i >> 3;
        const float lg = logits[row * 256 + t];
        float mx = lg;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));      // (xor < 32: stays inside the head's 32 lanes)
        const float e = expf(lg - mx);
        float den = e;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) den += __shfl_xor(den, o);
        const float w = e / den;
        const int H = lv.H[l], W = lv.W[l];
        const float nx = ref[row * 2 + 0] + offsets[row * 512 + t * 2 + 0] / (float)W;
        const float ny = ref[row * 2 + 1] + offsets[row * 512 + t * 2 + 1] / (float)H;
        const float x = nx * (float)W - 0.5f, y = ny * (float)H - 0.5f;
        const float fx = floorf(x), fy = floorf(y);
        const int x0 = (int)fx, y0 = (int)fy;
        const float lx = x - fx, ly = y - fy;
        float inb = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int xx = x0 + (c & 1), yy = y0 + (c >> 1);
            const bool ok = xx >= 0 && xx < W && yy >= 0 && yy < H;
            const float cw = ((c >> 1) ? ly : 1.f - ly) * ((c & 1) ? lx : 1.f - lx);
            cidx[t * 4 + c] = ok ? yy * W + xx : -1;
            cwt[t * 4 + c] = w * cw;
            inb += ok ? w * cw : 0.f;
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) inb += __shfl_xor(inb, o);               // the level's 8 points
        if ((i & 7) == 0) sbeta[head * 4 + l] = inb;
    }
    __syncthreads();
    // ---- phase 1: z[h][t] over the head's 32 samples x 4 corners; thread = raw FPN channel, group g takes heads 2 g, 2 g + 1
    const float* base[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        base[q] = reinterpret_cast<const float*>(lv.p[q]) + (long long)bc * lv.H[q] * lv.W[q] * 256 + t;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int h = 2 * grp + hh;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll 8
            for (int sidx = 0; sidx < 32; ++sidx) {                              // 8 points x 4 corners of level q
                const int e = (h * 32 + q * 8 + (sidx >> 2)) * 4 + (sidx & 3);
                const int ci = cidx[e];
                const float v = ci >= 0 ? base[q][(long long)ci * 256] : 0.f;
                acc += cwt[e] * v;
            }
        }
        z[h * 256 + t] = acc;
    }
    __syncthreads();
    // ---- phase 2: output channel t (head = t >> 5); group g sums K in [64 g, 64 g + 64), group 0 adds the partial sums in order
    {
        const float* zh = z + (t >> 5) * 256 + 64 * grp;
        const float* wk = wvT + (long long)(64 * grp) * 256 + t;
        float pa = 0.f;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) pa += wk[k * 256] * zh[k];
        part[grp * 256 + t] = pa;
    }
    __syncthreads();
    float acc = 0.f;
    if (grp == 0) {
        const int head = t >> 5;
        acc = (part[t] + part[256 + t]) + (part[512 + t] + part[768 + t]);
        const int cam = bc & 3;
        const float b0 = bias[t];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += sbeta[head * 4 + q] * (b0 + vshift[(q * 4 + cam) * 256 + t]);
        out[row * 256 + t] = acc;
    }
    // LayerNorm over the 256 outputs (groups 1-3 contribute zeros to the block sums)
    auto block_sum = [&](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);                            // group 0's four waves
    };
    const float mean = block_sum(grp == 0 ? acc : 0.f) / 256.f;
    const float d = acc - mean;
    const float rstd = 1.f / sqrtf(block_sum(grp == 0 ? d * d : 0.f) / 256.f + eps);
    if (grp == 0) out_ln[row * 256 + t] = d * rstd * gamma[t] + beta[t];
}

// sca_reduce + LayerNorm(1024) (output_proj.0): one block of 1024 threads per sample, thread = (camera, channel)
__global__ __launch_bounds__(1024) void sca_reduce_ln_kernel(const float* __restrict__ x, const int* __restrict__ max_len,
                                                             int B, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             float* __restrict__ out) {
    __shared__ float red[16];
    const int b = blockIdx.x, cam = threadIdx.x >> 8, c = threadIdx.x & 255;
    const int ml = min(*max_len, kQ);
    float a = 0.f;
    for (int k = B; k < ml; ++k) a += x[((long long)(b * kCams + cam) * kQ + k) * 256 + c] / (float)B;
    auto block_sum = [&](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        return t;
    };
    const float mean = block_sum(a) / 1024.f;
    const float d = a - mean;
    const float rstd = 1.f / sqrtf(block_sum(d * d) / 1024.f + eps);
    out[(long long)b * 1024 + cam * 256 + c] = d * rstd * gamma[cam * 256 + c] + beta[cam * 256 + c];
}

// mlp.0 input of a refinement layer (thinktwice_decoder.py:247-250): row (b, t) = LayerNorm(cat([future flat (b,t) 256 |
// look (b) 256 | zeros 256 (LiDAR look, DEC:186) | temporal (t) 128 | measurement (b) 128]))
__global__ __launch_bounds__(256) void dec_merge_in_kernel(const float* __restrict__ fflat, const float* __restrict__ look,
                                                           const float* __restrict__ temporal,
                                                           const float* __restrict__ meas, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           float* __restrict__ out) {
    __shared__ float red[4];
    const int row = blockIdx.x, b = row >> 2, t = row & 3, c = threadIdx.x;
    float v[4];
    v[0] = fflat[(long long)row * 256 + c];
    v[1] = look[(long long)b * 256 + c];
    v[2] = 0.f;
    v[3] = c < 128 ? temporal[t * 128 + c] : meas[(long long)b * 128 + c - 128];
    const float mean = block_sum256((v[0] + v[1]) + (v[2] + v[3]), red) / 1024.f;
    float qq = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) qq += (v[k] - mean) * (v[k] - mean);
    const float rstd = 1.f / sqrtf(block_sum256(qq, red) / 1024.f + eps);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        out[(long long)row * 1024 + k * 256 + c] = (v[k] - mean) * rstd * gamma[k * 256 + c] + beta[k * 256 + c];
}

}  // namespace tt

using namespace tt;

extern "C" int tt_look_project_pack(int B, const float* wp, const float* lidar2img, const float* ida_mat,
                                    float img_h, float img_w, float* ref_packed, int* query_of_slot,
                                    int* count, int* max_len, void* stream) {
    TT_REQUIRE(wp && lidar2img && ida_mat && ref_packed && query_of_slot && count && max_len,
               "tt_look_project_pack: null");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(max_len, 0, sizeof(int), st) != hipSuccess) {
        set_error("tt_look_project_pack: memset failed");
        return -2;
    }
    hipLaunchKernelGGL(look_project_pack_kernel, dim3(B * kCams), dim3(64), 0, st, wp, lidar2img, ida_mat, img_h,
                       img_w, ref_packed, query_of_slot, count, max_len, (float*)nullptr);
    return check_launch("tt_look_project_pack");
}

static int fill_levels(LevelMaps& m, const void* const* maps, const int* hw) {
    for (int l = 0; l < 4; ++l) {
        m.p[l] = maps ? maps[l] : nullptr;
        m.H[l] = hw[2 * l];
        m.W[l] = hw[2 * l + 1];
    }
    return 0;
}

extern "C" int tt_look_gather_query(int B, const int* query_of_slot, const float* ref_packed, const float* wp,
                                    const float* ctrl_softplus, const float* temporal_embedding,
                                    const float* static_embedding, const float* measurement_feat,
                                    const float* flattened_feat, const void* const* level_maps,
                                    const int* level_hw, int maps_dtype, float* out, int row_stride, void* stream) {
    TT_REQUIRE(query_of_slot && ref_packed && wp && ctrl_softplus && level_maps && level_hw && out,
               "tt_look_gather_query: null");
    TT_REQUIRE(row_stride >= 1543 && row_stride <= 1543 + 256, "tt_look_gather_query: row_stride %d", row_stride);
    LevelMaps m;
    fill_levels(m, level_maps, level_hw);
    const unsigned rows_n = (unsigned)(B * kCams * kQ);
    hipStream_t st = (hipStream_t)stream;
    if (maps_dtype == TT_F32)
        hipLaunchKernelGGL(look_gather_query_kernel<float>, dim3(rows_n), dim3(256), 0, st, query_of_slot, ref_packed,
                           wp, ctrl_softplus, temporal_embedding, static_embedding, measurement_feat,
                           flattened_feat, m, out, row_stride);
    else if (maps_dtype == TT_F16)
        hipLaunchKernelGGL(look_gather_query_kernel<f16_t>, dim3(rows_n), dim3(256), 0, st, query_of_slot,
                           ref_packed, wp, ctrl_softplus, temporal_embedding, static_embedding, measurement_feat,
                           flattened_feat, m, out, row_stride);
    else
        hipLaunchKernelGGL(look_gather_query_kernel<uint16_t>, dim3(rows_n), dim3(256), 0, st, query_of_slot,
                           ref_packed, wp, ctrl_softplus, temporal_embedding, static_embedding, measurement_feat,
                           flattened_feat, m, out, row_stride);
    return check_launch("tt_look_gather_query");
}

extern "C" int tt_msda_sample_strided(int B, const void* value, int value_dtype, int value_cstride, int value_coff,
                                      const float* offsets, const float* logits, const float* ref_packed,
                                      const int* level_hw, float* out, void* stream) {
    TT_REQUIRE(value && offsets && logits && ref_packed && level_hw && out, "tt_msda_sample: null");
    TT_REQUIRE(value_cstride >= 256 && value_coff >= 0 && value_coff + 256 <= value_cstride,
               "tt_msda_sample: channel window [%d, %d) outside a %d-channel row", value_coff, value_coff + 256,
               value_cstride);
    LevelMaps m;
    fill_levels(m, nullptr, level_hw);
    int S = 0;
    for (int l = 0; l < 4; ++l) S += m.H[l] * m.W[l];
    const unsigned rows_n = (unsigned)(B * kCams * kQ);
    hipStream_t st = (hipStream_t)stream;
    if (value_dtype == TT_F32)
        hipLaunchKernelGGL(msda_sample_kernel<float>, dim3(rows_n), dim3(256), 0, st, (const float*)value, offsets,
                           logits, ref_packed, m, S, value_cstride, value_coff, out);
    else if (value_dtype == TT_F16)
        hipLaunchKernelGGL(msda_sample_kernel<f16_t>, dim3(rows_n), dim3(256), 0, st, (const f16_t*)value,
                           offsets, logits, ref_packed, m, S, value_cstride, value_coff, out);
    else
        hipLaunchKernelGGL(msda_sample_kernel<uint16_t>, dim3(rows_n), dim3(256), 0, st, (const uint16_t*)value,
                           offsets, logits, ref_packed, m, S, value_cstride, value_coff, out);
    return check_launch("tt_msda_sample");
}

extern "C" int tt_msda_sample(int B, const void* value, int value_dtype, const float* offsets, const float* logits,
                              const float* ref_packed, const int* level_hw, float* out, void* stream) {
    return tt_msda_sample_strided(B, value, value_dtype, 256, 0, offsets, logits, ref_packed, level_hw, out, stream);
}

extern "C" int tt_sca_reduce(int B, const float* x, const int* max_len, float* out, void* stream) {
    TT_REQUIRE(x && max_len && out, "tt_sca_reduce: null");
    hipLaunchKernelGGL(sca_reduce_kernel, dim3(B * kCams), dim3(256), 0, (hipStream_t)stream, x, max_len, B, out);
    return check_launch("tt_sca_reduce");
}

extern "C" int tt_look_query_ln(int B, const int* query_of_slot, const float* ref_packed, const float* wp,
                                const float* ctrl, int raw_ctrl, const float* temporal_embedding,
                                const float* static_embedding, const float* measurement_feat, const float* flattened_feat,
                                const void* const* level_maps, const int* level_hw, int maps_dtype, const float* gamma,
                                const float* beta, float eps, float* out, int row_stride, void* stream) {
    TT_REQUIRE(query_of_slot && ref_packed && wp && ctrl && level_maps && level_hw && gamma && beta && out,
               "tt_look_query_ln: null");
    TT_REQUIRE(row_stride >= 1543 && row_stride <= 1543 + 256, "tt_look_query_ln: row_stride %d", row_stride);
    LevelMaps m;
    fill_levels(m, level_maps, level_hw);
    const unsigned rows_n = (unsigned)(B * kCams * kQ);
    hipStream_t st = (hipStream_t)stream;
#define QLN(T)                                                                                                        \
    hipLaunchKernelGGL(look_query_ln_kernel<T>, dim3(rows_n), dim3(256), 0, st, query_of_slot, ref_packed, wp, ctrl,  \
                       raw_ctrl, temporal_embedding, static_embedding, measurement_feat, flattened_feat, m, gamma,   \
                       beta, eps, out, row_stride)
    if (maps_dtype == TT_F32) QLN(float);
    else if (maps_dtype == TT_F16) QLN(f16_t);
    else QLN(uint16_t);
#undef QLN
    return check_launch("tt_look_query_ln");
}

extern "C" int tt_msda_sample_ln(int B, const void* value, int value_dtype, int value_cstride, int value_coff,
                                 const float* offsets, const float* logits, const float* ref_packed, const int* level_hw,
                                 const float* gamma, const float* beta, float eps, float* out, float* out_ln,
                                 const int* max_len_or_null, void* stream) {
    TT_REQUIRE(value && offsets && logits && ref_packed && level_hw && gamma && beta && out && out_ln,
               "tt_msda_sample_ln: null");
    TT_REQUIRE(value_cstride >= 256 && value_coff >= 0 && value_coff + 256 <= value_cstride,
               "tt_msda_sample_ln: channel window outside the row");
    LevelMaps m;
    fill_levels(m, nullptr, level_hw);
    int S = 0;
    for (int l = 0; l < 4; ++l) S += m.H[l] * m.W[l];
    const unsigned rows_n = (unsigned)(B * kCams * kQ);
    hipStream_t st = (hipStream_t)stream;
#define MLN(T)                                                                                                     \
    hipLaunchKernelGGL(msda_sample_ln_kernel<T>, dim3(rows_n), dim3(256), 0, st, (const T*)value, offsets, logits, \
                       ref_packed, m, S, value_cstride, value_coff, gamma, beta, eps, out, out_ln, max_len_or_null)
    if (value_dtype == TT_F32) MLN(float);
    else if (value_dtype == TT_F16) MLN(f16_t);
    else MLN(uint16_t);
#undef MLN
    return check_launch("tt_msda_sample_ln");
}

extern "C" int tt_msda_sample_proj_ln(int B, const void* const* level_maps, const int* level_hw, const float* offsets,
                                      const float* logits, const float* ref_packed, const float* wvT, const float* bias,
                                      const float* vshift, const float* gamma, const float* beta, float eps, float* out,
                                      float* out_ln, const int* max_len_or_null, void* stream) {
    TT_REQUIRE(level_maps && level_hw && offsets && logits && ref_packed && wvT && bias && vshift && gamma && beta && out && out_ln,
               "tt_msda_sample_proj_ln: null");
    LevelMaps m;
    fill_levels(m, level_maps, level_hw);
    const long long rows_n = (long long)B * kCams * kQ;
    hipLaunchKernelGGL(msda_sample_proj_ln_kernel, dim3((unsigned)rows_n), dim3(1024), 0, (hipStream_t)stream, m, offsets, logits,
                       ref_packed, wvT, bias, vshift, gamma, beta, eps, out, out_ln, max_len_or_null);
    return check_launch("tt_msda_sample_proj_ln");
}

extern "C" int tt_sca_reduce_ln(int B, const float* x, const int* max_len, const float* gamma, const float* beta,
                                float eps, float* out, void* stream) {
    TT_REQUIRE(x && max_len && gamma && beta && out, "tt_sca_reduce_ln: null");
    hipLaunchKernelGGL(sca_reduce_ln_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, max_len, B, gamma, beta, eps,
                       out);
    return check_launch("tt_sca_reduce_ln");
}

extern "C" int tt_dec_merge_in(int B, const float* fflat, const float* look, const float* temporal_embedding,
                               const float* measurement_feat, const float* gamma, const float* beta, float eps,
                               float* out, void* stream) {
    TT_REQUIRE(fflat && look && temporal_embedding && measurement_feat && gamma && beta && out, "tt_dec_merge_in: null");
    hipLaunchKernelGGL(dec_merge_in_kernel, dim3(B * 4), dim3(256), 0, (hipStream_t)stream, fflat, look,
                       temporal_embedding, measurement_feat, gamma, beta, eps, out);
    return check_launch("tt_dec_merge_in");
}
