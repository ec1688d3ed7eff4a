// Train-mode BatchNorm (batch statistics) over channel-last rows, forward and backward.
//
// Reference: every nn.BatchNorm2d / BatchNorm1d of the model under `model.train()` -- the reference trains with
// norm_eval=False and SyncBN=True (configs/thinktwice.py:39,146; apis/mmdet_train.py:86-87 converts them to
// torch.nn.SyncBatchNorm), i.e.  y = gamma * (z - mean_B) / sqrt(var_B + eps) + beta  with the biased variance of the
// (global) batch, running statistics updated with `momentum` and the unbiased variance.  The eval-mode path folds BatchNorm
// into the conv epilogue (csrc/conv_common.h); in train mode the convolution writes its raw output z (bias included) and
// these kernels do the rest:
//   tt_bn_stats       per-channel sum / sum of squares / row count, per row GROUP (the camera trunk runs its T sweeps as
//                     one batch of T equal image groups, each normalised with its own statistics like the reference's
//                     per-sweep passes, lss.py:690-717); f64 accumulation, per-workgroup partials added in index order
//   (host)            SyncBN: ONE all-reduce of the [groups][2C + 2] statistics block (thinktwice_amd/ops.py)
//   tt_bn_finalize    mean / invstd / scale = gamma * invstd / shift = beta per group, running-statistics update
//   tt_bn_apply       y = act((z - mean) * scale + shift + res1 + res2) into a channel window of the consumer's buffer
//   tt_bn_bwd_reduce  g = dy * act'(y) (written over dy), residual gradients, per-channel sum g and sum g * xhat
//   (host)            SyncBN: one all-reduce of the [groups][2C] sums
//   tt_bn_bwd_apply   dz = scale * (g - sum_g / n - xhat * sum_gxhat / n)
// dgamma = sum g * xhat and dbeta = sum g are the LOCAL sums (the gradient all-reduce averages them like DDP does).
#include <initializer_list>

#include "tt_common.h"

namespace tt {

constexpr int kBnBlocks = 512;     // most workgroups per row group in the reductions (= stride of the partial sums)

struct BnRows {
    long long M;          // allocated rows
    const int* m_dev;     // sparse layers: device count of live rows (groups == 1), else null
    int C, groups;
    int nb;               // workgroups per row group of this launch (<= kBnBlocks; >= 256 rows each where the group has them)
};

static BnRows bn_rows(long long M, const int* m_dev, int C, int groups) {
    const long long per_group = M / groups;
    long long nb = (per_group + 255) / 256;
    nb = nb < 1 ? 1 : (nb > kBnBlocks ? kBnBlocks : nb);
    return BnRows{M, m_dev, C, groups, (int)nb};
}

// 16-byte path of the four streaming kernels: every tensor window starts on a 16 B boundary and is a whole number of
// 4-channel pieces wide
static bool bn_vec_ok(int C, std::initializer_list<const void*> ptrs, std::initializer_list<int> strides_offsets) {
    if (C % 4) return false;
    for (const void* q : ptrs) if (q && (reinterpret_cast<uintptr_t>(q) & 15)) return false;
    for (int v : strides_offsets) if (v % 4) return false;
    return true;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ void group_range(const BnRows& r, int g, int blk, long long& r0, long long& r1) {
    const long long Mlive = r.m_dev ? min(r.M, (long long)*r.m_dev) : r.M;
    const long long per_group = r.m_dev ? Mlive : r.M / r.groups;
    const long long g0 = (long long)g * per_group;
    const long long rows_per = (per_group + r.nb - 1) / r.nb;
    r0 = g0 + (long long)blk * rows_per;
    r1 = min(g0 + per_group, r0 + rows_per);
}

// partial[(g * kBnBlocks + blk) * 2 * C + {0, C} + c]
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ z, int z_cstride, int z_coff, BnRows r,
                                                       double* __restrict__ partial) {
    __shared__ double red[2][256];
    const int tid = threadIdx.x, g = blockIdx.y, blk = blockIdx.x;
    const int TX = r.C < 256 ? r.C : 256, TY = 256 / TX;
    const int tx = tid % TX, ty = tid / TX;
    long long r0, r1;
    group_range(r, g, blk, r0, r1);
    for (int c0 = 0; c0 < r.C; c0 += TX) {
        const int c = c0 + tx;
        double s = 0.0, ss = 0.0;
        if (ty < TY && c < r.C) {
            long long m = r0 + ty;
            for (; m + 3LL * TY < r1; m += 4LL * TY) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = z[(m + (long long)u * TY) * z_cstride + z_coff + c];
#pragma unroll
                for (int u = 0; u < 4; ++u) { s += (double)v[u]; ss += (double)v[u] * (double)v[u]; }
            }
            for (; m < r1; m += TY) {
                const float v = z[m * z_cstride + z_coff + c];
                s += (double)v;
                ss += (double)v * (double)v;
            }
        }
        __syncthreads();
        red[0][tid] = s;
        red[1][tid] = ss;
        __syncthreads();
        if (ty == 0 && c < r.C) {
            double t0 = 0.0, t1 = 0.0;
            for (int q = 0; q < TY; ++q) { t0 += red[0][q * TX + tx]; t1 += red[1][q * TX + tx]; }
            double* p = partial + ((long long)g * kBnBlocks + blk) * 2 * r.C;
            p[c] = t0;
            p[r.C + c] = t1;
        }
    }
}

// 16 B form of bn_stats_kernel: a thread owns 4 consecutive channels, TY rows of the block's range are in flight at a time,
// two rows per thread per trip; f64 accumulators, LDS reduction over the row lanes in index order (deterministic)
__global__ __launch_bounds__(256) void bn_stats_vec_kernel(const float* __restrict__ z, int z_cstride, int z_coff, BnRows r,
                                                           double* __restrict__ partial) {
    __shared__ double red[8][256];
    const int tid = threadIdx.x, g = blockIdx.y, blk = blockIdx.x;
    const int C = r.C, C4 = C >> 2;
    const int TX = C4 < 256 ? C4 : 256, TY = 256 / TX;
    const int tx = tid % TX, ty = tid / TX;
    long long r0, r1;
    group_range(r, g, blk, r0, r1);
    for (int q0 = 0; q0 < C4; q0 += TX) {
        const int q = q0 + tx, c = q * 4;
        double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
        if (ty < TY && q < C4) {
            const float* zp = z + z_coff + c;
            for (long long m = r0 + ty; m < r1; m += 4LL * TY) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long mm = m + (long long)u * TY;
                    v[u] = mm < r1 ? ld4(zp + mm * z_cstride) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s[0] += (double)v[u].x; ss[0] += (double)v[u].x * (double)v[u].x;
                    s[1] += (double)v[u].y; ss[1] += (double)v[u].y * (double)v[u].y;
                    s[2] += (double)v[u].z; ss[2] += (double)v[u].z * (double)v[u].z;
                    s[3] += (double)v[u].w; ss[3] += (double)v[u].w * (double)v[u].w;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[k][tid] = s[k]; red[4 + k][tid] = ss[k]; }
        __syncthreads();
        if (ty == 0 && q < C4) {
            double* p = partial + ((long long)g * kBnBlocks + blk) * 2 * C;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double t0 = 0.0, t1 = 0.0;
                for (int j = 0; j < TY; ++j) { t0 += red[k][j * TX + tx]; t1 += red[4 + k][j * TX + tx]; }
                p[c + k] = t0;
                p[C + c + k] = t1;
            }
        }
    }
}

// out[g][0..C) = sum of partial[.][0], out[g][C..2C) = sum of partial[.][1]; with `count`: out[g][2C] = rows.
// A workgroup takes 32 columns; its eight 32-thread slices each sum every eighth partial row (four independent chains per thread),
// the slices are added in index order: a fixed summation order, nb / 32 dependent double adds per thread instead of nb (the
// one-thread-per-column loop over up to 512 strided rows took 47 us per call, 388 calls per training iteration).
constexpr int kFinishSlices = 8;
__global__ __launch_bounds__(256) void bn_finish_kernel(const double* __restrict__ partial, BnRows r, int out_stride,
                                                        int with_count, double* __restrict__ out) {
    __shared__ double sh[kFinishSlices][32];
    const int lc = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + lc, g = blockIdx.y;
    const int C2 = 2 * r.C;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < C2) {
        const double* p = partial + (long long)g * kBnBlocks * C2 + c;      // [blk][{sum, sum of squares}][C]: column c of row blk
        int b = slice;
        for (; b + 3 * kFinishSlices < r.nb; b += 4 * kFinishSlices) {
            s0 += p[(long long)b * C2];
            s1 += p[(long long)(b + kFinishSlices) * C2];
            s2 += p[(long long)(b + 2 * kFinishSlices) * C2];
            s3 += p[(long long)(b + 3 * kFinishSlices) * C2];
        }
        for (; b < r.nb; b += kFinishSlices) s0 += p[(long long)b * C2];
    }
    sh[slice][lc] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && c < C2) {
        double s = sh[0][lc];
#pragma unroll
        for (int q = 1; q < kFinishSlices; ++q) s += sh[q][lc];
        out[(long long)g * out_stride + c] = s;
    }
    if (with_count && blockIdx.x == 0 && threadIdx.x == 0) {
        const long long Mlive = r.m_dev ? min(r.M, (long long)*r.m_dev) : r.M;
        out[(long long)g * out_stride + 2 * r.C] = (double)(r.m_dev ? Mlive : r.M / r.groups);
        out[(long long)g * out_stride + 2 * r.C + 1] = 0.0;
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, int C, int groups, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* running_mean,
                                   float* running_var, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean, float* __restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int stride = 2 * C + 2;
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
    // group 0 is the KEY sweep (sweep-major image order): the reference runs the key frame first, then the older sweeps
    // (lss.py:689-714), so its running statistics see the groups in index order
    for (int g = 0; g < groups; ++g) {
        const double n = stats[(long long)g * stride + 2 * C];
        const double mu = n > 0 ? stats[(long long)g * stride + c] / n : 0.0;
        double var = n > 0 ? stats[(long long)g * stride + C + c] / n - mu * mu : 0.0;
        var = var > 0.0 ? var : 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = (gamma ? gamma[c] : 1.f) * is;
        scale[g * C + c] = sc;
        shift[g * C + c] = beta ? beta[c] : 0.f;      // NOT folded with the mean: tt_bn_apply centres first (see there)
        mean[g * C + c] = (float)mu;
        invstd[g * C + c] = is;
        // a non-finite forward (overflow / NaN in this channel) must not reach the running statistics: the optimizer skips
        // the update of such an iteration on the device (tt_grad_norm_clip's NaN factor), and the buffers stay those of the
        // last good step too
        const bool finite = mu == mu && var == var && fabs(mu) <= 3.0e38 && var <= 3.0e38;
        if (n > 0 && finite) {
            rm = (1.f - momentum) * rm + momentum * (float)mu;
            rv = (1.f - momentum) * rv + momentum * (float)(n > 1 ? var * n / (n - 1.0) : var);
        }
    }
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
}

struct BnApplyArgs {
    const float* z; const float* scale; const float* shift; const float* mean; const float* res1; const float* res2; float* out;
    BnRows r;
    int z_cstride, z_coff, r1_cstride, r1_coff, r2_cstride, r2_coff, out_cstride, out_coff, act;
};

__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApplyArgs a) {
    const long long Mlive = a.r.m_dev ? min(a.r.M, (long long)*a.r.m_dev) : a.r.M;
    const long long per_group = a.r.m_dev ? (Mlive > 0 ? Mlive : 1) : a.r.M / a.r.groups;
    const long long total = Mlive * a.r.C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / a.r.C;
        const int c = (int)(i - m * a.r.C);
        const int g = (int)(m / per_group);
        // centred form (z - mean) * (gamma * invstd) + beta, like torch: the folded z * scale + (beta - mean * scale) loses
        // ulp(mean * scale) -- 0.008 on the (near-)constant camera-parameter columns of DepthNet's BatchNorm1d, whose
        // variance is ~0 and invstd = 1 / sqrt(eps) = 316 (measured: 1.4e-3 on the camera BEV before this)
        float v = (a.z[m * a.z_cstride + a.z_coff + c] - a.mean[g * a.r.C + c]) * a.scale[g * a.r.C + c] + a.shift[g * a.r.C + c];
        if (a.res1) v += a.res1[m * a.r1_cstride + a.r1_coff + c];
        if (a.res2) v += a.res2[m * a.r2_cstride + a.r2_coff + c];
        if (a.act == TT_ACT_RELU) v = v > 0.f ? v : 0.f;
        a.out[m * a.out_cstride + a.out_coff + c] = v;
    }
}

// 16 B form: thread = (4 channels, row lane); a block walks rows with a grid stride, no per-element division
__global__ __launch_bounds__(256) void bn_apply_vec_kernel(const BnApplyArgs a) {
    const int tid = threadIdx.x;
    const int C = a.r.C, C4 = C >> 2;
    const int TX = C4 < 256 ? C4 : 256, TY = 256 / TX;
    const int tx = tid % TX, ty = tid / TX;
    const long long Mlive = a.r.m_dev ? min(a.r.M, (long long)*a.r.m_dev) : a.r.M;
    const long long per_group = a.r.m_dev ? (Mlive > 0 ? Mlive : 1) : a.r.M / a.r.groups;
    if (ty >= TY) return;
    for (long long m = (long long)blockIdx.x * TY + ty; m < Mlive; m += (long long)gridDim.x * TY) {
        const int g = (int)(m / per_group);
        for (int q = tx; q < C4; q += TX) {
            const int c = q * 4;
            const float4 z = ld4(a.z + m * a.z_cstride + a.z_coff + c);
            const float4 mu = ld4(a.mean + g * C + c), sc = ld4(a.scale + g * C + c), sh = ld4(a.shift + g * C + c);
            // centred form, see bn_apply_kernel
            float4 v = make_float4((z.x - mu.x) * sc.x + sh.x, (z.y - mu.y) * sc.y + sh.y, (z.z - mu.z) * sc.z + sh.z,
                                   (z.w - mu.w) * sc.w + sh.w);
            if (a.res1) {
                const float4 t = ld4(a.res1 + m * a.r1_cstride + a.r1_coff + c);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            if (a.res2) {
                const float4 t = ld4(a.res2 + m * a.r2_cstride + a.r2_coff + c);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            if (a.act == TT_ACT_RELU) {
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            }
            st4(a.out + m * a.out_cstride + a.out_coff + c, v);
        }
    }
}

struct BnBwdArgs {
    float* dy; const float* y; const float* z; const float* mean; const float* invstd;
    float* dres1; float* dres2; double* partial;
    BnRows r;
    int dy_cstride, dy_coff, y_cstride, y_coff, z_cstride, z_coff, d1_cstride, d1_coff, d2_cstride, d2_coff, act;
};

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBwdArgs a) {
    __shared__ double red[2][256];
    const int tid = threadIdx.x, g = blockIdx.y, blk = blockIdx.x;
    const int C = a.r.C;
    const int TX = C < 256 ? C : 256, TY = 256 / TX;
    const int tx = tid % TX, ty = tid / TX;
    long long r0, r1;
    group_range(a.r, g, blk, r0, r1);
    for (int c0 = 0; c0 < C; c0 += TX) {
        const int c = c0 + tx;
        double s = 0.0, sx = 0.0;
        if (ty < TY && c < C) {
            const float mu = a.mean[g * C + c], is = a.invstd[g * C + c];
            for (long long m = r0 + ty; m < r1; m += TY) {
                float gv = a.dy[m * a.dy_cstride + a.dy_coff + c];
                if (a.act == TT_ACT_RELU && !(a.y[m * a.y_cstride + a.y_coff + c] > 0.f)) gv = 0.f;
                a.dy[m * a.dy_cstride + a.dy_coff + c] = gv;
                if (a.dres1) a.dres1[m * a.d1_cstride + a.d1_coff + c] += gv;
                if (a.dres2) a.dres2[m * a.d2_cstride + a.d2_coff + c] += gv;
                const float xh = (a.z[m * a.z_cstride + a.z_coff + c] - mu) * is;
                s += (double)gv;
                sx += (double)gv * (double)xh;
            }
        }
        __syncthreads();
        red[0][tid] = s;
        red[1][tid] = sx;
        __syncthreads();
        if (ty == 0 && c < C) {
            double t0 = 0.0, t1 = 0.0;
            for (int q = 0; q < TY; ++q) { t0 += red[0][q * TX + tx]; t1 += red[1][q * TX + tx]; }
            double* p = a.partial + ((long long)g * kBnBlocks + blk) * 2 * C;
            p[c] = t0;
            p[C + c] = t1;
        }
    }
}

struct BnBwdApplyArgs {
    const float* g; const float* z; const float* scale; const float* mean; const float* invstd;
    const double* sums;    // [groups][2C] (global under SyncBN)
    const double* stats;   // [groups][2C + 2]: the forward's statistics block, [2C] = (global) row count
    float* dz;
    BnRows r;
    int g_cstride, g_coff, z_cstride, z_coff, dz_cstride, dz_coff;
};

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdApplyArgs a) {
    const int C = a.r.C;
    const long long Mlive = a.r.m_dev ? min(a.r.M, (long long)*a.r.m_dev) : a.r.M;
    const long long per_group = a.r.m_dev ? (Mlive > 0 ? Mlive : 1) : a.r.M / a.r.groups;
    const long long total = Mlive * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C;
        const int c = (int)(i - m * C);
        const int g = (int)(m / per_group);
        const double n = a.stats[(long long)g * (2 * C + 2) + 2 * C];
        const float inv_n = n > 0 ? (float)(1.0 / n) : 0.f;
        const float sg = (float)a.sums[(long long)g * 2 * C + c] * inv_n, sgx = (float)a.sums[(long long)g * 2 * C + C + c] * inv_n;
        const float xh = (a.z[m * a.z_cstride + a.z_coff + c] - a.mean[g * C + c]) * a.invstd[g * C + c];
        const float gv = a.g[m * a.g_cstride + a.g_coff + c];
        a.dz[m * a.dz_cstride + a.dz_coff + c] = a.scale[g * C + c] * (gv - sg - xh * sgx);
    }
}

// 16 B forms of the two backward kernels (same sums in the same order per block as the scalar forms' structure: f64
// accumulators, row lanes reduced through LDS in index order)
__global__ __launch_bounds__(256) void bn_bwd_reduce_vec_kernel(const BnBwdArgs a) {
    __shared__ double red[8][256];
    const int tid = threadIdx.x, g = blockIdx.y, blk = blockIdx.x;
    const int C = a.r.C, C4 = C >> 2;
    const int TX = C4 < 256 ? C4 : 256, TY = 256 / TX;
    const int tx = tid % TX, ty = tid / TX;
    const bool relu = a.act == TT_ACT_RELU;
    long long r0, r1;
    group_range(a.r, g, blk, r0, r1);
    for (int q0 = 0; q0 < C4; q0 += TX) {
        const int q = q0 + tx, c = q * 4;
        double s[4] = {0.0, 0.0, 0.0, 0.0}, sx[4] = {0.0, 0.0, 0.0, 0.0};
        if (ty < TY && q < C4) {
            const float4 mu = ld4(a.mean + g * C + c), is = ld4(a.invstd + g * C + c);
            for (long long m = r0 + ty; m < r1; m += 2LL * TY) {
                float4 gv[2], yv[2], zv[2];
                bool ok[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const long long mm = m + (long long)u * TY;
                    ok[u] = mm < r1;
                    gv[u] = yv[u] = zv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok[u]) {
                        gv[u] = ld4(a.dy + mm * a.dy_cstride + a.dy_coff + c);
                        zv[u] = ld4(a.z + mm * a.z_cstride + a.z_coff + c);
                        if (relu) yv[u] = ld4(a.y + mm * a.y_cstride + a.y_coff + c);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (!ok[u]) continue;
                    const long long mm = m + (long long)u * TY;
                    float4 gg = gv[u];
                    if (relu) {
                        gg.x = yv[u].x > 0.f ? gg.x : 0.f; gg.y = yv[u].y > 0.f ? gg.y : 0.f;
                        gg.z = yv[u].z > 0.f ? gg.z : 0.f; gg.w = yv[u].w > 0.f ? gg.w : 0.f;
                        st4(a.dy + mm * a.dy_cstride + a.dy_coff + c, gg);
                    }
                    if (a.dres1) {
                        float* d = a.dres1 + mm * a.d1_cstride + a.d1_coff + c;
                        float4 t = ld4(d);
                        t.x += gg.x; t.y += gg.y; t.z += gg.z; t.w += gg.w;
                        st4(d, t);
                    }
                    if (a.dres2) {
                        float* d = a.dres2 + mm * a.d2_cstride + a.d2_coff + c;
                        float4 t = ld4(d);
                        t.x += gg.x; t.y += gg.y; t.z += gg.z; t.w += gg.w;
                        st4(d, t);
                    }
                    const float x0 = (zv[u].x - mu.x) * is.x, x1 = (zv[u].y - mu.y) * is.y;
                    const float x2 = (zv[u].z - mu.z) * is.z, x3 = (zv[u].w - mu.w) * is.w;
                    s[0] += (double)gg.x; sx[0] += (double)gg.x * (double)x0;
                    s[1] += (double)gg.y; sx[1] += (double)gg.y * (double)x1;
                    s[2] += (double)gg.z; sx[2] += (double)gg.z * (double)x2;
                    s[3] += (double)gg.w; sx[3] += (double)gg.w * (double)x3;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[k][tid] = s[k]; red[4 + k][tid] = sx[k]; }
        __syncthreads();
        if (ty == 0 && q < C4) {
            double* p = a.partial + ((long long)g * kBnBlocks + blk) * 2 * C;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double t0 = 0.0, t1 = 0.0;
                for (int j = 0; j < TY; ++j) { t0 += red[k][j * TX + tx]; t1 += red[4 + k][j * TX + tx]; }
                p[c + k] = t0;
                p[C + c + k] = t1;
            }
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(const BnBwdApplyArgs a) {
    const int tid = threadIdx.x;
    const int C = a.r.C, C4 = C >> 2;
    const int TX = C4 < 256 ? C4 : 256, TY = 256 / TX;
    const int tx = tid % TX, ty = tid / TX;
    const long long Mlive = a.r.m_dev ? min(a.r.M, (long long)*a.r.m_dev) : a.r.M;
    const long long per_group = a.r.m_dev ? (Mlive > 0 ? Mlive : 1) : a.r.M / a.r.groups;
    if (ty >= TY) return;
    for (long long m = (long long)blockIdx.x * TY + ty; m < Mlive; m += (long long)gridDim.x * TY) {
        const int g = (int)(m / per_group);
        const double n = a.stats[(long long)g * (2 * C + 2) + 2 * C];
        const float inv_n = n > 0 ? (float)(1.0 / n) : 0.f;
        for (int q = tx; q < C4; q += TX) {
            const int c = q * 4;
            const double* sp = a.sums + (long long)g * 2 * C;
            const float4 zz = ld4(a.z + m * a.z_cstride + a.z_coff + c), gv = ld4(a.g + m * a.g_cstride + a.g_coff + c);
            const float4 mu = ld4(a.mean + g * C + c), is = ld4(a.invstd + g * C + c), sc = ld4(a.scale + g * C + c);
            float4 o;
            {
                const float sg = (float)sp[c] * inv_n, sgx = (float)sp[C + c] * inv_n;
                o.x = sc.x * (gv.x - sg - (zz.x - mu.x) * is.x * sgx);
            }
            {
                const float sg = (float)sp[c + 1] * inv_n, sgx = (float)sp[C + c + 1] * inv_n;
                o.y = sc.y * (gv.y - sg - (zz.y - mu.y) * is.y * sgx);
            }
            {
                const float sg = (float)sp[c + 2] * inv_n, sgx = (float)sp[C + c + 2] * inv_n;
                o.z = sc.z * (gv.z - sg - (zz.z - mu.z) * is.z * sgx);
            }
            {
                const float sg = (float)sp[c + 3] * inv_n, sgx = (float)sp[C + c + 3] * inv_n;
                o.w = sc.w * (gv.w - sg - (zz.w - mu.w) * is.w * sgx);
            }
            st4(a.dz + m * a.dz_cstride + a.dz_coff + c, o);
        }
    }
}

// nn.Dropout in train mode.  Keep decision: splitmix64(seed + element index) uniform in [0, 1) >= p
__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, long long n, float p,
                                   unsigned long long seed, const uint8_t* __restrict__ mask_in, uint8_t* __restrict__ mask_out) {
    const float inv = 1.f / (1.f - p);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        bool keep;
        if (mask_in) {
            keep = mask_in[i] != 0;
        } else {
            unsigned long long zz = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
            zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull;
            zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull;
            zz ^= zz >> 31;
            keep = (float)(zz >> 40) * (1.f / 16777216.f) >= p;
        }
        if (mask_out) mask_out[i] = keep ? 1 : 0;
        out[i] = keep ? x[i] * inv : 0.f;
    }
}

__global__ void dropout_bwd_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ mask, float* __restrict__ dx,
                                   long long n, float p) {
    const float inv = 1.f / (1.f - p);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dx[i] += mask[i] ? dout[i] * inv : 0.f;
}

static int check_rows(long long M, int C, int groups, const int* m_dev, const char* who) {
    TT_REQUIRE(M > 0 && C > 0 && groups >= 1, "%s: bad sizes (M=%lld C=%d groups=%d)", who, M, C, groups);
    TT_REQUIRE(!m_dev || groups == 1, "%s: a device row count goes with one group", who);
    TT_REQUIRE(m_dev || M % groups == 0, "%s: %lld rows do not split into %d equal groups", who, M, groups);
    return 0;
}

}  // namespace tt

using namespace tt;

extern "C" long long tt_bn_workspace_bytes(int C, int groups) {
    return (long long)groups * kBnBlocks * 2 * C * (long long)sizeof(double);
}

extern "C" int tt_bn_stats(const float* z, long long M, int C, int z_cstride, int z_coff, const int* m_dev, int groups,
                           double* stats, void* workspace, long long workspace_bytes, void* stream) {
    TT_REQUIRE(z && stats && workspace, "tt_bn_stats: null");
    if (int rc = check_rows(M, C, groups, m_dev, "tt_bn_stats")) return rc;
    TT_REQUIRE(workspace_bytes >= tt_bn_workspace_bytes(C, groups), "tt_bn_stats: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const BnRows r = bn_rows(M, m_dev, C, groups);
    if (bn_vec_ok(C, {z}, {z_cstride, z_coff}))
        hipLaunchKernelGGL(bn_stats_vec_kernel, dim3(r.nb, groups), dim3(256), 0, st, z, z_cstride, z_coff, r, (double*)workspace);
    else
        hipLaunchKernelGGL(bn_stats_kernel, dim3(r.nb, groups), dim3(256), 0, st, z, z_cstride, z_coff, r, (double*)workspace);
    hipLaunchKernelGGL(bn_finish_kernel, dim3(div_up(2 * C, 32), groups), dim3(256), 0, st, (const double*)workspace, r,
                       2 * C + 2, 1, stats);
    return check_launch("tt_bn_stats");
}

extern "C" int tt_bn_finalize(const double* stats, int C, int groups, const float* gamma, const float* beta, float eps,
                              float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* mean,
                              float* invstd, void* stream) {
    TT_REQUIRE(stats && scale && shift && mean && invstd && C > 0 && groups >= 1, "tt_bn_finalize: null / bad sizes");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(div_up(C, 256)), dim3(256), 0, (hipStream_t)stream, stats, C, groups, gamma,
                       beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd);
    return check_launch("tt_bn_finalize");
}

extern "C" int tt_bn_apply(const float* z, long long M, int C, int z_cstride, int z_coff, const int* m_dev, int groups,
                           const float* scale, const float* shift, const float* mean, const float* res1, int r1_cstride, int r1_coff,
                           const float* res2, int r2_cstride, int r2_coff, int act, float* out, int out_cstride, int out_coff,
                           void* stream) {
    TT_REQUIRE(z && scale && shift && mean && out, "tt_bn_apply: null");
    if (int rc = check_rows(M, C, groups, m_dev, "tt_bn_apply")) return rc;
    TT_REQUIRE(act == TT_ACT_NONE || act == TT_ACT_RELU, "tt_bn_apply: activation %d after a train-mode BatchNorm", act);
    BnApplyArgs a{z, scale, shift, mean, res1, res2, out, bn_rows(M, m_dev, C, groups), z_cstride, z_coff, r1_cstride, r1_coff,
                  r2_cstride, r2_coff, out_cstride, out_coff, act};
    const long long total = M * C;
    if (bn_vec_ok(C, {z, res1, res2, out, scale, shift, mean},
                  {z_cstride, z_coff, res1 ? r1_cstride : 0, res1 ? r1_coff : 0, res2 ? r2_cstride : 0, res2 ? r2_coff : 0,
                   out_cstride, out_coff})) {
        const int ty = 256 / (C / 4 < 256 ? C / 4 : 256);
        const int blocks = (int)min((long long)kNumCU * 8, (M + ty - 1) / ty);
        hipLaunchKernelGGL(bn_apply_vec_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
        return check_launch("tt_bn_apply");
    }
    const int blocks = (int)min((long long)kNumCU * 16, (total + 255) / 256);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("tt_bn_apply");
}

extern "C" int tt_bn_bwd_reduce(float* dy, int dy_cstride, int dy_coff, const float* y, int y_cstride, int y_coff,
                                const float* z, int z_cstride, int z_coff, long long M, int C, const int* m_dev, int groups,
                                const float* mean, const float* invstd, int act, float* dres1, int d1_cstride, int d1_coff,
                                float* dres2, int d2_cstride, int d2_coff, double* sums, void* workspace,
                                long long workspace_bytes, void* stream) {
    TT_REQUIRE(dy && y && z && mean && invstd && sums && workspace, "tt_bn_bwd_reduce: null");
    if (int rc = check_rows(M, C, groups, m_dev, "tt_bn_bwd_reduce")) return rc;
    TT_REQUIRE(act == TT_ACT_NONE || act == TT_ACT_RELU, "tt_bn_bwd_reduce: activation %d", act);
    TT_REQUIRE(workspace_bytes >= tt_bn_workspace_bytes(C, groups), "tt_bn_bwd_reduce: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    BnBwdArgs a{dy, y, z, mean, invstd, dres1, dres2, (double*)workspace, bn_rows(M, m_dev, C, groups), dy_cstride, dy_coff,
                y_cstride, y_coff, z_cstride, z_coff, d1_cstride, d1_coff, d2_cstride, d2_coff, act};
    if (bn_vec_ok(C, {dy, y, z, dres1, dres2, mean, invstd},
                  {dy_cstride, dy_coff, y_cstride, y_coff, z_cstride, z_coff, dres1 ? d1_cstride : 0, dres1 ? d1_coff : 0,
                   dres2 ? d2_cstride : 0, dres2 ? d2_coff : 0}))
        hipLaunchKernelGGL(bn_bwd_reduce_vec_kernel, dim3(a.r.nb, groups), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(a.r.nb, groups), dim3(256), 0, st, a);
    hipLaunchKernelGGL(bn_finish_kernel, dim3(div_up(2 * C, 32), groups), dim3(256), 0, st, (const double*)workspace, a.r,
                       2 * C, 0, sums);
    return check_launch("tt_bn_bwd_reduce");
}

extern "C" int tt_bn_bwd_apply(const float* g, int g_cstride, int g_coff, const float* z, int z_cstride, int z_coff,
                               long long M, int C, const int* m_dev, int groups, const double* sums, const double* stats,
                               const float* scale, const float* mean, const float* invstd, float* dz, int dz_cstride,
                               int dz_coff, void* stream) {
    TT_REQUIRE(g && z && sums && stats && scale && mean && invstd && dz, "tt_bn_bwd_apply: null");
    if (int rc = check_rows(M, C, groups, m_dev, "tt_bn_bwd_apply")) return rc;
    BnBwdApplyArgs a{g, z, scale, mean, invstd, sums, stats, dz, bn_rows(M, m_dev, C, groups), g_cstride, g_coff, z_cstride,
                     z_coff, dz_cstride, dz_coff};
    const long long total = M * C;
    if (bn_vec_ok(C, {g, z, dz, scale, mean, invstd}, {g_cstride, g_coff, z_cstride, z_coff, dz_cstride, dz_coff})) {
        const int ty = 256 / (C / 4 < 256 ? C / 4 : 256);
        const int blocks = (int)min((long long)kNumCU * 8, (M + ty - 1) / ty);
        hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
        return check_launch("tt_bn_bwd_apply");
    }
    const int blocks = (int)min((long long)kNumCU * 16, (total + 255) / 256);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("tt_bn_bwd_apply");
}

extern "C" int tt_dropout_fwd(const float* x, float* out, long long n, float p, unsigned long long seed,
                              const uint8_t* mask_in, uint8_t* mask_out, void* stream) {
    TT_REQUIRE(x && out && n > 0 && p >= 0.f && p < 1.f, "tt_dropout_fwd: null / bad p");
    const int blocks = (int)min((long long)kNumCU * 16, (n + 255) / 256);
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, n, p, seed, mask_in,
                       mask_out);
    return check_launch("tt_dropout_fwd");
}

extern "C" int tt_dropout_bwd(const float* dout, const uint8_t* mask, float* dx, long long n, float p, void* stream) {
    TT_REQUIRE(dout && mask && dx && n > 0 && p >= 0.f && p < 1.f, "tt_dropout_bwd: null / bad p");
    const int blocks = (int)min((long long)kNumCU * 16, (n + 255) / 256);
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, mask, dx, n, p);
    return check_launch("tt_dropout_bwd");
}
