// Loss reductions of the training forward: every term of ThinkTwiceDecoder.loss
// (thinktwice_decoder.py:536-619), the focal segmentation loss (utils.py:31-47, encoder_decoder_framework.py:172-176)
// and the depth BCE (encoder_decoder_framework.py:179-190, 441-481) as device reductions.
//
// All of them are HBM-trivial (the largest operand, the 12-class seg logits of 32 images, is ~150 MB at batch 8; the
// rest are KBs), so the design goal is determinism and one pass: a fixed grid of kLossBlocks workgroups accumulates
// in f64, every workgroup writes its partial, and the LAST workgroup to arrive (ticket counter in the workspace) adds
// the partials in index order and applies the term's closing formula -- one launch per term, bit-reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tt_common.h"

namespace tt {

constexpr int kLossBlocks = 256;
constexpr int kLossThreads = 256;
constexpr int kLossMaxOut = 8;                      // partial sums per workgroup (columns / auxiliary counts)

struct LossWs {
    double partial[kLossBlocks][kLossMaxOut];
    unsigned ticket;
};

__device__ __forceinline__ double block_sum(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < kLossThreads / 64; ++w) t += sh[w];
    return t;                                       // valid in thread 0
}

// Every workgroup calls this with its NV partial sums (thread 0 holds them).  Returns true in thread 0 of the last
// workgroup, with `total[]` = the sums over all workgroups in index order; the ticket is reset for the next launch.
template <int NV>
__device__ __forceinline__ bool finish(LossWs* ws, const double (&mine)[NV], double (&total)[NV]) {
    __shared__ bool last;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) ws->partial[blockIdx.x][v] = mine[v];
        __threadfence();
        const unsigned t = atomicAdd(&ws->ticket, 1u);
        last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (!last || threadIdx.x != 0) return false;
    __threadfence();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double s = 0.0;
        for (unsigned b = 0; b < gridDim.x; ++b) s += reinterpret_cast<volatile double*>(&ws->partial[b][v])[0];
        total[v] = s;
    }
    ws->ticket = 0u;
    return true;
}

__device__ __forceinline__ float smooth_l1(float d) {          // torch smooth_l1_loss, beta = 1
    const float a = fabsf(d);
    return a < 1.f ? 0.5f * d * d : a - 0.5f;
}

// pred [n_outer][repeat][inner] against target [n_outer][inner] (null: zeros).  reduce: out[0] = scale * mean of
// min(sl1, clamp_max); otherwise out[i] = scale * sl1 elementwise.
__global__ __launch_bounds__(kLossThreads) void loss_smooth_l1_kernel(const float* __restrict__ pred,
                                                                      const float* __restrict__ target, long long total,
                                                                      long long rep_inner, long long inner,
                                                                      float clamp_max, float scale, int reduce,
                                                                      float* __restrict__ out, LossWs* ws) {
    __shared__ double sh[kLossThreads / 64];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * kLossThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kLossThreads) {
        float t = 0.f;
        if (target) {
            const long long n = i / rep_inner, e = i % inner;
            t = target[n * inner + e];
        }
        float l = smooth_l1(pred[i] - t);
        if (clamp_max > 0.f) l = fminf(l, clamp_max);
        if (reduce) acc += (double)l;
        else out[i] = l * scale;
    }
    if (!reduce) return;
    const double mine[1] = {block_sum(acc, sh)};
    double tot[1];
    if (finish<1>(ws, mine, tot)) out[0] = (float)(tot[0] / (double)total * (double)scale);
}

// digamma for x > 0: upward recurrence to x >= 8, then the asymptotic series (f64: |error| < 1e-12)
__device__ __forceinline__ double digamma_pos(double x) {
    double r = 0.0;
    while (x < 8.0) {
        r -= 1.0 / x;
        x += 1.0;
    }
    const double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x - f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0 - f / 132.0))));
}

// KL( Beta(pa, pb) || Beta(qa, qb) ), torch.distributions.kl._kl_beta_beta
__device__ __forceinline__ double kl_beta(double pa, double pb, double qa, double qb) {
    const double sp = pa + pb, sq = qa + qb;
    const double t1 = lgamma(qa) + lgamma(qb) + lgamma(sp);
    const double t2 = lgamma(pa) + lgamma(pb) + lgamma(sq);
    return t1 - t2 + (pa - qa) * digamma_pos(pa) + (pb - qb) * digamma_pos(pb) + (sq - sp) * digamma_pos(sp);
}

// target (p) [n_outer][inner], prediction (q) [n_outer][repeat][inner]: out[0] = scale * mean KL(p || q)
__global__ __launch_bounds__(kLossThreads) void loss_beta_kl_kernel(const float* __restrict__ p_alpha,
                                                                    const float* __restrict__ p_beta,
                                                                    const float* __restrict__ q_alpha,
                                                                    const float* __restrict__ q_beta, long long total,
                                                                    long long rep_inner, long long inner, float scale,
                                                                    float* __restrict__ out, LossWs* ws) {
    __shared__ double sh[kLossThreads / 64];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * kLossThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kLossThreads) {
        const long long n = i / rep_inner, e = i % inner;
        acc += kl_beta(p_alpha[n * inner + e], p_beta[n * inner + e], q_alpha[i], q_beta[i]);
    }
    const double mine[1] = {block_sum(acc, sh)};
    double tot[1];
    if (finish<1>(ws, mine, tot)) out[0] = (float)(tot[0] / (double)total * (double)scale);
}

// ThinkTwiceDecoder._get_action_beta (thinktwice_decoder.py:622-637): mode of the Beta with its edge cases, in [-1, 1]
__device__ __forceinline__ float beta_mode(float a, float b) {
    float x;
    if (a > 1.f && b > 1.f) x = (a - 1.f) / (a + b - 2.f);
    else if (a <= 1.f && b > 1.f) x = 0.f;
    else if (a > 1.f && b <= 1.f) x = 1.f;
    else x = a / fmaxf(a + b, 1e-5f);
    return x * 2.f - 1.f;
}

// out[c] = mean over rows of |pred[r][c] - target[r][c]|, cols <= kLossMaxOut; rows of pred are pred_row_stride apart.
// beta: the operands are Beta parameters (pred, pred_b) / (target, target_b) and the compared values are their modes.
__global__ __launch_bounds__(kLossThreads) void loss_l1_cols_kernel(const float* __restrict__ pred,
                                                                    const float* __restrict__ pred_b,
                                                                    long long pred_row_stride,
                                                                    const float* __restrict__ target,
                                                                    const float* __restrict__ target_b, long long rows,
                                                                    int cols, float* __restrict__ out, LossWs* ws) {
    __shared__ double sh[kLossThreads / 64];
    double acc[kLossMaxOut];
#pragma unroll
    for (int c = 0; c < kLossMaxOut; ++c) acc[c] = 0.0;
    for (long long r = (long long)blockIdx.x * kLossThreads + threadIdx.x; r < rows; r += (long long)gridDim.x * kLossThreads) {
#pragma unroll
        for (int c = 0; c < kLossMaxOut; ++c)
            if (c < cols) {
                float p = pred[r * pred_row_stride + c], t = target[r * cols + c];
                if (pred_b) {
                    p = beta_mode(p, pred_b[r * pred_row_stride + c]);
                    t = beta_mode(t, target_b[r * cols + c]);
                }
                acc[c] += (double)fabsf(p - t);
            }
    }
    double mine[kLossMaxOut], tot[kLossMaxOut];
#pragma unroll
    for (int c = 0; c < kLossMaxOut; ++c) mine[c] = block_sum(acc[c], sh);
    if (finish<kLossMaxOut>(ws, mine, tot))
        for (int c = 0; c < cols; ++c) out[c] = (float)(tot[c] / (double)rows);
}

// Focal loss on the MEAN cross entropy (utils.py:31-47: alpha 0.5, gamma 2, x10 at the call site EDF:176) of channel-
// last logits [BN][h][w][row_stride >= C] against labels [BN][H][W] (float class ids, 255 = ignore) sampled with the
// nearest rule of torchvision Resize / F.interpolate (source pixel = dst * factor).
__global__ __launch_bounds__(kLossThreads) void loss_seg_focal_kernel(const float* __restrict__ logits, int row_stride,
                                                                      int C, const float* __restrict__ labels, int BN,
                                                                      int H, int W, int factor, float* __restrict__ out,
                                                                      float* __restrict__ aux, LossWs* ws) {
    __shared__ double sh[kLossThreads / 64];
    const int h = H / factor, w = W / factor;
    const long long total = (long long)BN * h * w;
    double acc = 0.0, cnt = 0.0;
    for (long long i = (long long)blockIdx.x * kLossThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kLossThreads) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const long long bn = i / ((long long)w * h);
        const long long lab = (long long)labels[(bn * H + (long long)y * factor) * W + (long long)x * factor];
        if (lab == 255) continue;
        const float* v = logits + i * row_stride;
        float m = v[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, v[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(v[c] - m);
        acc += (double)(m + logf(s) - v[lab]);
        cnt += 1.0;
    }
    double mine[2], tot[2];
    mine[0] = block_sum(acc, sh);
    mine[1] = block_sum(cnt, sh);
    if (finish<2>(ws, mine, tot)) {
        const float logpt = -(float)(tot[0] / tot[1]);
        const float pt = expf(logpt);
        out[0] = -((1.f - pt) * (1.f - pt)) * 0.5f * logpt * 10.f;
        if (aux) {          // for the backward: mean cross entropy and the number of contributing pixels
            aux[0] = -logpt;
            aux[1] = (float)tot[1];
        }
    }
}

// d(focal seg loss) / d(logits): with u = mean CE, L = 5 (1 - e^-u)^2 u, dL/du = 5 [2 (1 - e^-u) e^-u u + (1 - e^-u)^2];
// d u / d logit[pixel][c] = (softmax_c - [c == label]) / count for the contributing pixels, 0 elsewhere (and on the row's
// padding channels).  `upstream`: device scalar d(total)/d(seg_loss) or NULL (= 1).
__global__ __launch_bounds__(kLossThreads) void loss_seg_focal_bwd_kernel(const float* __restrict__ logits, int row_stride,
                                                                          int C, const float* __restrict__ labels, int BN,
                                                                          int H, int W, int factor,
                                                                          const float* __restrict__ aux,
                                                                          const float* __restrict__ upstream,
                                                                          float* __restrict__ dlogits) {
    const int h = H / factor, w = W / factor;
    const long long total = (long long)BN * h * w;
    const float u = aux[0], cnt = aux[1];
    const float e = expf(-u);
    const float coef = (upstream ? upstream[0] : 1.f) * 5.f * (2.f * (1.f - e) * e * u + (1.f - e) * (1.f - e)) / cnt;
    for (long long i = (long long)blockIdx.x * kLossThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kLossThreads) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const long long bn = i / ((long long)w * h);
        const long long lab = (long long)labels[(bn * H + (long long)y * factor) * W + (long long)x * factor];
        const float* v = logits + i * row_stride;
        float* d = dlogits + i * row_stride;
        if (lab == 255) {
            for (int c = 0; c < row_stride; ++c) d[c] = 0.f;
            continue;
        }
        float m = v[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, v[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(v[c] - m);
        const float inv = 1.f / s;
        for (int c = 0; c < row_stride; ++c)
            d[c] = c < C ? coef * (expf(v[c] - m) * inv - (c == lab ? 1.f : 0.f)) : 0.f;
    }
}

// Depth BCE (EDF:179-190 + get_downsampled_gt_depth EDF:441-481): per factor x factor cell the nearest return (0 = no
// return), its depth bin as a one-hot over D bins, BCE-with-logits summed over the bins of the foreground cells,
// divided by max(1, #foreground).  logits channel-last [BN][h][w][row_stride >= D].
__global__ __launch_bounds__(kLossThreads) void loss_depth_bce_kernel(const float* __restrict__ logits, int row_stride,
                                                                      int D, const float* __restrict__ gt, int BN, int H,
                                                                      int W, int factor, float d0, float dstep,
                                                                      float* __restrict__ out, LossWs* ws) {
    __shared__ double sh[kLossThreads / 64];
    const int h = H / factor, w = W / factor;
    const long long total = (long long)BN * h * w;
    double acc = 0.0, cnt = 0.0;
    for (long long i = (long long)blockIdx.x * kLossThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kLossThreads) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const long long bn = i / ((long long)w * h);
        float g = 1e5f;
        for (int dy = 0; dy < factor; ++dy) {
            const float* row = gt + (bn * H + (long long)y * factor + dy) * W + (long long)x * factor;
            for (int dx = 0; dx < factor; ++dx) {
                const float t = row[dx];
                g = fminf(g, t == 0.f ? 1e5f : t);
            }
        }
        g = (g - (d0 - dstep)) / dstep;
        if (!(g < (float)(D + 1) && g >= 0.f)) g = 0.f;
        const int bin = (int)g;                     // 0 = background, 1..D = depth bin + 1
        if (bin < 1) continue;
        const float* v = logits + i * row_stride;
        float s = 0.f;
        for (int d = 0; d < D; ++d) {
            const float xl = v[d];
            s += fmaxf(xl, 0.f) - (d == bin - 1 ? xl : 0.f) + log1pf(expf(-fabsf(xl)));
        }
        acc += (double)s;
        cnt += 1.0;
    }
    double mine[2], tot[2];
    mine[0] = block_sum(acc, sh);
    mine[1] = block_sum(cnt, sh);
    if (finish<2>(ws, mine, tot)) {
        out[0] = (float)(tot[0] / fmax(1.0, tot[1]));
        out[1] = (float)fmax(1.0, tot[1]);          // the divisor, for the backward
    }
}

// d(depth BCE)/d(logits): (sigmoid(x_d) - [d == bin - 1]) / max(1, #foreground) on the foreground cells, 0 elsewhere
__global__ __launch_bounds__(kLossThreads) void loss_depth_bce_bwd_kernel(const float* __restrict__ logits, int row_stride,
                                                                          int D, const float* __restrict__ gt, int BN, int H,
                                                                          int W, int factor, float d0, float dstep,
                                                                          const float* __restrict__ aux,
                                                                          const float* __restrict__ upstream,
                                                                          float* __restrict__ dlogits) {
    const int h = H / factor, w = W / factor;
    const long long total = (long long)BN * h * w;
    const float coef = (upstream ? upstream[0] : 1.f) / aux[0];
    for (long long i = (long long)blockIdx.x * kLossThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kLossThreads) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const long long bn = i / ((long long)w * h);
        float g = 1e5f;
        for (int dy = 0; dy < factor; ++dy) {
            const float* row = gt + (bn * H + (long long)y * factor + dy) * W + (long long)x * factor;
            for (int dx = 0; dx < factor; ++dx) {
                const float t = row[dx];
                g = fminf(g, t == 0.f ? 1e5f : t);
            }
        }
        g = (g - (d0 - dstep)) / dstep;
        if (!(g < (float)(D + 1) && g >= 0.f)) g = 0.f;
        const int bin = (int)g;
        const float* v = logits + i * row_stride;
        float* d = dlogits + i * row_stride;
        for (int c = 0; c < row_stride; ++c)
            d[c] = (bin >= 1 && c < D) ? coef * (1.f / (1.f + expf(-v[c])) - (c == bin - 1 ? 1.f : 0.f)) : 0.f;
    }
}

// ---- gradients of the mean-reduced terms w.r.t. the prediction (dense, same layout as `pred`): elementwise, no reduction
// d/dpred [weight * mean(min(sl1(pred - t), clamp_max))] = weight/total * clip(d, -1, 1), zero where the clamp is active
__global__ __launch_bounds__(kLossThreads) void loss_smooth_l1_bwd_kernel(const float* __restrict__ pred,
                                                                          const float* __restrict__ target, long long total,
                                                                          long long rep_inner, long long inner,
                                                                          float clamp_max, float weight,
                                                                          float* __restrict__ dpred) {
    for (long long i = (long long)blockIdx.x * kLossThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kLossThreads) {
        float t = 0.f;
        if (target) {
            const long long n = i / rep_inner, e = i % inner;
            t = target[n * inner + e];
        }
        const float d = pred[i] - t;
        float g = fminf(fmaxf(d, -1.f), 1.f);
        if (clamp_max > 0.f && smooth_l1(d) > clamp_max) g = 0.f;
        dpred[i] = g * weight;
    }
}

// d KL(p || q) / d q_alpha = psi(q_alpha) - psi(q_alpha + q_beta) - psi(p_alpha) + psi(p_alpha + p_beta)   (same with beta)
__global__ __launch_bounds__(kLossThreads) void loss_beta_kl_bwd_kernel(const float* __restrict__ p_alpha,
                                                                        const float* __restrict__ p_beta,
                                                                        const float* __restrict__ q_alpha,
                                                                        const float* __restrict__ q_beta, long long total,
                                                                        long long rep_inner, long long inner, float weight,
                                                                        float* __restrict__ dq_alpha,
                                                                        float* __restrict__ dq_beta) {
    for (long long i = (long long)blockIdx.x * kLossThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kLossThreads) {
        const long long n = i / rep_inner, e = i % inner;
        const double pa = p_alpha[n * inner + e], pb = p_beta[n * inner + e], qa = q_alpha[i], qb = q_beta[i];
        const double common = digamma_pos(pa + pb) - digamma_pos(qa + qb);
        dq_alpha[i] = (float)((digamma_pos(qa) - digamma_pos(pa) + common) * (double)weight);
        dq_beta[i] = (float)((digamma_pos(qb) - digamma_pos(pb) + common) * (double)weight);
    }
}

}  // namespace tt

using namespace tt;

extern "C" long long tt_loss_workspace_bytes(void) { return (long long)sizeof(LossWs); }

static int loss_grid(long long total) {
    const long long b = (total + kLossThreads - 1) / kLossThreads;
    return (int)(b < 1 ? 1 : (b > kLossBlocks ? kLossBlocks : b));
}

extern "C" int tt_loss_smooth_l1(const float* pred, const float* target, long long n_outer, int repeat, long long inner,
                                 float clamp_max, float scale, int reduce, float* out, void* workspace, void* stream) {
    TT_REQUIRE(pred && out && workspace && n_outer > 0 && repeat > 0 && inner > 0, "tt_loss_smooth_l1: bad argument");
    const long long total = n_outer * repeat * inner;
    hipLaunchKernelGGL(loss_smooth_l1_kernel, dim3(loss_grid(total)), dim3(kLossThreads), 0, (hipStream_t)stream, pred,
                       target, total, (long long)repeat * inner, inner, clamp_max, scale, reduce, out, (LossWs*)workspace);
    return check_launch("tt_loss_smooth_l1");
}

extern "C" int tt_loss_beta_kl(const float* target_alpha, const float* target_beta, const float* pred_alpha,
                               const float* pred_beta, long long n_outer, int repeat, long long inner, float scale,
                               float* out, void* workspace, void* stream) {
    TT_REQUIRE(target_alpha && target_beta && pred_alpha && pred_beta && out && workspace && n_outer > 0 && repeat > 0 &&
                   inner > 0, "tt_loss_beta_kl: bad argument");
    const long long total = n_outer * repeat * inner;
    hipLaunchKernelGGL(loss_beta_kl_kernel, dim3(loss_grid(total)), dim3(kLossThreads), 0, (hipStream_t)stream,
                       target_alpha, target_beta, pred_alpha, pred_beta, total, (long long)repeat * inner, inner, scale,
                       out, (LossWs*)workspace);
    return check_launch("tt_loss_beta_kl");
}

extern "C" int tt_loss_l1_cols(const float* pred, const float* pred_beta, long long pred_row_stride, const float* target,
                               const float* target_beta, long long rows, int cols, float* out, void* workspace,
                               void* stream) {
    TT_REQUIRE(pred && target && out && workspace && rows > 0 && cols > 0 && cols <= kLossMaxOut &&
                   (!pred_beta == !target_beta), "tt_loss_l1_cols: bad argument");
    hipLaunchKernelGGL(loss_l1_cols_kernel, dim3(loss_grid(rows)), dim3(kLossThreads), 0, (hipStream_t)stream, pred,
                       pred_beta, pred_row_stride, target, target_beta, rows, cols, out, (LossWs*)workspace);
    return check_launch("tt_loss_l1_cols");
}

extern "C" int tt_loss_seg_focal_bwd(const float* logits_cl, int row_stride, int num_classes, const float* labels, int BN,
                                     int H, int W, int factor, const float* aux, const float* upstream_or_null,
                                     float* dlogits_cl, void* stream) {
    TT_REQUIRE(logits_cl && labels && aux && dlogits_cl && BN > 0 && factor > 0 && H >= factor && W >= factor &&
                   num_classes > 0 && row_stride >= num_classes, "tt_loss_seg_focal_bwd: bad argument");
    const long long total = (long long)BN * (H / factor) * (W / factor);
    hipLaunchKernelGGL(loss_seg_focal_bwd_kernel, dim3(loss_grid(total)), dim3(kLossThreads), 0, (hipStream_t)stream,
                       logits_cl, row_stride, num_classes, labels, BN, H, W, factor, aux, upstream_or_null, dlogits_cl);
    return check_launch("tt_loss_seg_focal_bwd");
}

extern "C" int tt_loss_seg_focal(const float* logits_cl, int row_stride, int num_classes, const float* labels, int BN,
                                 int H, int W, int factor, float* out, void* workspace, void* stream) {
    TT_REQUIRE(logits_cl && labels && out && workspace && BN > 0 && factor > 0 && H >= factor && W >= factor &&
                   num_classes > 0 && row_stride >= num_classes, "tt_loss_seg_focal: bad argument");
    const long long total = (long long)BN * (H / factor) * (W / factor);
    hipLaunchKernelGGL(loss_seg_focal_kernel, dim3(loss_grid(total)), dim3(kLossThreads), 0, (hipStream_t)stream,
                       logits_cl, row_stride, num_classes, labels, BN, H, W, factor, out, out + 1, (LossWs*)workspace);
    return check_launch("tt_loss_seg_focal");
}

extern "C" int tt_loss_depth_bce_bwd(const float* logits_cl, int row_stride, int D, const float* gt_depth, int BN, int H,
                                     int W, int factor, float d_lo, float d_step, const float* aux,
                                     const float* upstream_or_null, float* dlogits_cl, void* stream) {
    TT_REQUIRE(logits_cl && gt_depth && aux && dlogits_cl && BN > 0 && factor > 0 && H >= factor && W >= factor && D > 0 &&
                   row_stride >= D && d_step > 0.f, "tt_loss_depth_bce_bwd: bad argument");
    const long long total = (long long)BN * (H / factor) * (W / factor);
    hipLaunchKernelGGL(loss_depth_bce_bwd_kernel, dim3(loss_grid(total)), dim3(kLossThreads), 0, (hipStream_t)stream,
                       logits_cl, row_stride, D, gt_depth, BN, H, W, factor, d_lo, d_step, aux, upstream_or_null, dlogits_cl);
    return check_launch("tt_loss_depth_bce_bwd");
}

extern "C" int tt_loss_depth_bce(const float* logits_cl, int row_stride, int D, const float* gt_depth, int BN, int H,
                                 int W, int factor, float d_lo, float d_step, float* out, void* workspace, void* stream) {
    TT_REQUIRE(logits_cl && gt_depth && out && workspace && BN > 0 && factor > 0 && H >= factor && W >= factor && D > 0 &&
                   row_stride >= D && d_step > 0.f, "tt_loss_depth_bce: bad argument");
    const long long total = (long long)BN * (H / factor) * (W / factor);
    hipLaunchKernelGGL(loss_depth_bce_kernel, dim3(loss_grid(total)), dim3(kLossThreads), 0, (hipStream_t)stream,
                       logits_cl, row_stride, D, gt_depth, BN, H, W, factor, d_lo, d_step, out, (LossWs*)workspace);
    return check_launch("tt_loss_depth_bce");
}

extern "C" int tt_loss_smooth_l1_bwd(const float* pred, const float* target, long long n_outer, int repeat, long long inner,
                                     float clamp_max, float scale, float* dpred, void* stream) {
    TT_REQUIRE(pred && dpred && n_outer > 0 && repeat > 0 && inner > 0, "tt_loss_smooth_l1_bwd: bad argument");
    const long long total = n_outer * repeat * inner;
    hipLaunchKernelGGL(loss_smooth_l1_bwd_kernel, dim3(loss_grid(total)), dim3(kLossThreads), 0, (hipStream_t)stream, pred,
                       target, total, (long long)repeat * inner, inner, clamp_max, (float)((double)scale / (double)total), dpred);
    return check_launch("tt_loss_smooth_l1_bwd");
}

extern "C" int tt_loss_beta_kl_bwd(const float* target_alpha, const float* target_beta, const float* pred_alpha,
                                   const float* pred_beta, long long n_outer, int repeat, long long inner, float scale,
                                   float* dpred_alpha, float* dpred_beta, void* stream) {
    TT_REQUIRE(target_alpha && target_beta && pred_alpha && pred_beta && dpred_alpha && dpred_beta && n_outer > 0 &&
                   repeat > 0 && inner > 0, "tt_loss_beta_kl_bwd: bad argument");
    const long long total = n_outer * repeat * inner;
    hipLaunchKernelGGL(loss_beta_kl_bwd_kernel, dim3(loss_grid(total)), dim3(kLossThreads), 0, (hipStream_t)stream,
                       target_alpha, target_beta, pred_alpha, pred_beta, total, (long long)repeat * inner, inner,
                       (float)((double)scale / (double)total), dpred_alpha, dpred_beta);
    return check_launch("tt_loss_beta_kl_bwd");
}
