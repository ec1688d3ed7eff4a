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

// the same, four channels per thread (C % 4 == 0, 16 B aligned tensors): the window scans are 16 B loads
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   float* __restrict__ dx, int N, int H, int W, int C, int OH,
                                                                   int OW) {
    const int C4 = C >> 2;
    const long long total = (long long)N * H * W * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        long long r = i / C4;
        const int iw = (int)(r % W); r /= W;
        const int ih = (int)(r % H);
        const long long n = r / H;
        const float* xn = x + n * H * W * C + c;
        const int me = ih * W + iw;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        for (int oh = ih / 2; oh <= (ih + 1) / 2; ++oh) {
            if (oh >= OH) continue;
            for (int ow = iw / 2; ow <= (iw + 1) / 2; ++ow) {
                if (ow >= OW) continue;
                float m[4];
                int arg[4] = {-1, -1, -1, -1};
                for (int dh = 0; dh < 3; ++dh) {
                    const int yy = oh * 2 - 1 + dh;
                    if (yy < 0 || yy >= H) continue;
                    for (int dw = 0; dw < 3; ++dw) {
                        const int xx = ow * 2 - 1 + dw;
                        if (xx < 0 || xx >= W) continue;
                        const float4 v4 = *reinterpret_cast<const float4*>(xn + ((long long)yy * W + xx) * C);
                        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (arg[e] < 0 || v[e] > m[e]) {
                                m[e] = v[e];
                                arg[e] = yy * W + xx;
                            }
                    }
                }
                const float4 d4 = *reinterpret_cast<const float4*>(dy + ((n * OH + oh) * OW + ow) * C + c);
                const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (arg[e] == me) g[e] += d[e];
            }
        }
        float4* o = reinterpret_cast<float4*>(dx + i * 4);
        float4 cur = *o;
        cur.x += g[0]; cur.y += g[1]; cur.z += g[2]; cur.w += g[3];
        *o = cur;
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

// F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True) backward: dx[y][x] += sum over the output pixels
// whose 2x2 footprint contains (y, x) of weight * dy, with the weights recomputed exactly as the forward computes them.
__global__ __launch_bounds__(256) void bilinear_up2_ac_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N,
                                                                  int H, int W, int C) {
    const int OH = 2 * H, OW = 2 * W;
    const float sh = (OH > 1) ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sw = (OW > 1) ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const long long total = (long long)N * H * W * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const long long n = r / H;
        // outputs whose source coordinate lies in (y - 1, y + 1): a conservative index range, exact weights inside
        const int oy0 = sh > 0.f ? max(0, (int)floorf((y - 1) / sh) - 1) : 0;
        const int oy1 = sh > 0.f ? min(OH - 1, (int)ceilf((y + 1) / sh) + 1) : OH - 1;
        const int ox0 = sw > 0.f ? max(0, (int)floorf((x - 1) / sw) - 1) : 0;
        const int ox1 = sw > 0.f ? min(OW - 1, (int)ceilf((x + 1) / sw) + 1) : OW - 1;
        float g = 0.f;
        for (int oy = oy0; oy <= oy1; ++oy) {
            const float fy = sh * oy;
            const int y0 = (int)fy, y1 = min(y0 + 1, H - 1);
            const float ly = fy - y0;
            const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox0; ox <= ox1; ++ox) {
                const float fx = sw * ox;
                const int x0 = (int)fx, x1 = min(x0 + 1, W - 1);
                const float lx = fx - x0;
                const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
                if (wx != 0.f) g += wy * wx * dy[((n * OH + oy) * OW + ox) * C + c];
            }
        }
        dx[i] += g;
    }
}

// out = x * sigmoid(gate[n][c])  (SELayer, lss.py:158) backward:  dx += dy * s,  dgate[n][c] += s (1 - s) sum_hw dy * x.
// One workgroup per (image, 64-channel slab): 4 pixel partitions x 64 channels, the partitions added through LDS.
// With `out` (the saved output) the forward was  out = relu(x * s + res): dy is masked by out > 0 first, and `dres` (optional)
// receives the masked gradient of the residual input.
__global__ __launch_bounds__(256) void channel_gate_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                               const float* __restrict__ dy, float* __restrict__ dx,
                                                               float* __restrict__ dgate, int HW, int C,
                                                               const float* __restrict__ out, float* __restrict__ dres) {
    __shared__ float red[4][64];
    const int n = blockIdx.y, cl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f, acc = 0.f;
    if (c < C) {
        s = 1.f / (1.f + expf(-gate[(long long)n * C + c]));
        const long long base = (long long)n * HW * C + c;
        for (int p = part; p < HW; p += 4) {
            float g = dy[base + (long long)p * C];
            if (out && !(out[base + (long long)p * C] > 0.f)) g = 0.f;
            acc += g * x[base + (long long)p * C];
            dx[base + (long long)p * C] += g * s;
            if (dres) dres[base + (long long)p * C] += g;
        }
    }
    red[part][cl] = acc;
    __syncthreads();
    if (part == 0 && c < C)
        dgate[(long long)n * C + c] += s * (1.f - s) * (red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

// SEModule pooling 0.5 * mean + 0.5 * amax (code/utils.py:91-92) backward: the mean half spreads dpool / HW, the amax half
// goes to the maxima of the (image, channel) plane -- split EVENLY among ties, as torch.amax does (post-ReLU planes that are
// all zero have HW ties).  One workgroup per (image, 64-channel slab).
__global__ __launch_bounds__(256) void spatial_meanmax_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dpool,
                                                                  float* __restrict__ dx, int HW, int C) {
    __shared__ float red[4][64];
    __shared__ float cnt[4][64];
    const int n = blockIdx.y, cl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long long base = (long long)n * HW * C + c;
    float m = -INFINITY;
    if (c < C)
        for (int p = part; p < HW; p += 4) m = fmaxf(m, x[base + (long long)p * C]);
    red[part][cl] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0][cl], red[1][cl]), fmaxf(red[2][cl], red[3][cl]));
    float k = 0.f;
    if (c < C)
        for (int p = part; p < HW; p += 4) k += (x[base + (long long)p * C] == m) ? 1.f : 0.f;
    cnt[part][cl] = k;
    __syncthreads();
    k = cnt[0][cl] + cnt[1][cl] + cnt[2][cl] + cnt[3][cl];
    if (c >= C) return;
    const float g = dpool[(long long)n * C + c];
    const float gm = 0.5f * g / (float)HW, gx = 0.5f * g / k;
    for (int p = part; p < HW; p += 4) dx[base + (long long)p * C] += gm + ((x[base + (long long)p * C] == m) ? gx : 0.f);
}

// mean over the pixels of an image (AdaptiveAvgPool2d(1)) backward: dx[n][p][coff + c] += dpool[n][c] / HW
__global__ __launch_bounds__(256) void spatial_mean_bwd_kernel(const float* __restrict__ dpool, float* __restrict__ dx, int N,
                                                               int HW, int C, int cstride, int coff) {
    const long long total = (long long)N * HW * C;
    const float inv = 1.f / (float)HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long np = i / C;
        dx[np * cstride + coff + c] += dpool[(np / HW) * C + c] * inv;
    }
}

// Deformable im2col (mmcv DCN v1, 3x3, deform_groups = 1) backward.  Forward (csrc/dcn.hip): cols[pix][tap][c] = bilinear
// sample of x[n] at (y + tap/3 - pad + off[2 tap], x + tap%3 - pad + off[2 tap + 1]), zero outside.  From gcols:
//   gx   += corner weight * gcols  (scattered with f32 atomics: a sample point can land anywhere)
//   goff[pix][2 tap] = sum_c gcols * d(sample)/d(py),  goff[pix][2 tap + 1] = ... d(px)
// One wave per (pixel, tap): lanes stride the channels, the two offset gradients are wave reductions.
__global__ __launch_bounds__(256) void deform_im2col_bwd_kernel(const float* __restrict__ x, const float* __restrict__ off,
                                                                const float* __restrict__ gcols, float* __restrict__ gx,
                                                                float* __restrict__ goff, int N, int H, int W, int C,
                                                                int off_cstride, int pad) {
    const int lane = threadIdx.x & 63;
    const long long total = (long long)N * H * W * 9;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6; i < total; i += ((long long)gridDim.x * 256) >> 6) {
        const int tap = (int)(i % 9);
        const long long pix = i / 9;
        const int xw = (int)(pix % W), yh = (int)((pix / W) % H);
        const long long n = pix / ((long long)H * W);
        const float py = (float)(yh + tap / 3 - pad) + off[pix * off_cstride + 2 * tap];
        const float px = (float)(xw + tap % 3 - pad) + off[pix * off_cstride + 2 * tap + 1];
        float gy_ = 0.f, gx_ = 0.f;
        if (py > -1.f && py < (float)H && px > -1.f && px < (float)W) {
            const int y0 = (int)floorf(py), x0 = (int)floorf(px);
            const float ly = py - (float)y0, lx = px - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
            const bool v00 = y0 >= 0 && x0 >= 0, v01 = y0 >= 0 && x0 + 1 <= W - 1;
            const bool v10 = y0 + 1 <= H - 1 && x0 >= 0, v11 = y0 + 1 <= H - 1 && x0 + 1 <= W - 1;
            const long long b = n * H * W;
            const float* g = gcols + (pix * 9 + tap) * C;
            for (int c = lane; c < C; c += 64) {
                const float gv = g[c];
                const float a00 = v00 ? x[(b + (long long)y0 * W + x0) * C + c] : 0.f;
                const float a01 = v01 ? x[(b + (long long)y0 * W + x0 + 1) * C + c] : 0.f;
                const float a10 = v10 ? x[(b + (long long)(y0 + 1) * W + x0) * C + c] : 0.f;
                const float a11 = v11 ? x[(b + (long long)(y0 + 1) * W + x0 + 1) * C + c] : 0.f;
                gy_ += gv * (hx * (a10 - a00) + lx * (a11 - a01));
                gx_ += gv * (hy * (a01 - a00) + ly * (a11 - a10));
                if (v00) unsafeAtomicAdd(gx + (b + (long long)y0 * W + x0) * C + c, gv * hy * hx);
                if (v01) unsafeAtomicAdd(gx + (b + (long long)y0 * W + x0 + 1) * C + c, gv * hy * lx);
                if (v10) unsafeAtomicAdd(gx + (b + (long long)(y0 + 1) * W + x0) * C + c, gv * ly * hx);
                if (v11) unsafeAtomicAdd(gx + (b + (long long)(y0 + 1) * W + x0 + 1) * C + c, gv * ly * lx);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            gy_ += __shfl_xor(gy_, o);
            gx_ += __shfl_xor(gx_, o);
        }
        if (lane == 0) {
            goff[pix * off_cstride + 2 * tap] += gy_;
            goff[pix * off_cstride + 2 * tap + 1] += gx_;
        }
    }
}

// tt_ew backward: out = act(v), v = a + b | (1 - b) a | (1 - g) a + g b | a  (ops 0..3).  gv = dout * act'(.) from the saved
// output (none / ReLU / sigmoid / softplus [clamped]; GELU needs the pre-activation and is refused); every requested input gradient is
// accumulated in its own row-strided channel window.
struct EwBwdArgs {
    const float* a; const float* b; const float* g; const float* out; const float* dout;
    float* da; float* db; float* dg;
    long long R;
    int C, as, aco, bs, bco, gs, gco, os, oco, ds, dco, das, daco, dbs, dbco, dgs, dgco, op, act;
};

__global__ __launch_bounds__(256) void ew_bwd_kernel(const EwBwdArgs p) {
    const long long total = p.R * p.C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % p.C);
        const long long r = i / p.C;
        float gv = p.dout[r * p.ds + p.dco + c];
        if (p.act != TT_ACT_NONE) {
            const float o = p.out[r * p.os + p.oco + c];
            if (p.act == TT_ACT_RELU) gv = o > 0.f ? gv : 0.f;
            else if (p.act == TT_ACT_SIGMOID) gv *= o * (1.f - o);
            else if (p.act == TT_ACT_SOFTPLUS_CLAMP) gv = o > 1e-3f ? gv * (1.f - expf(-o)) : 0.f;   // clamped: no gradient
            else gv *= 1.f - expf(-o);                           // softplus: sigmoid(pre) = 1 - exp(-out)
        }
        const float av = p.a[r * p.as + p.aco + c];
        const float bv = p.b ? p.b[r * p.bs + p.bco + c] : 0.f;
        const float gg = p.g ? p.g[r * p.gs + p.gco + c] : 0.f;
        float ga, gb = 0.f, gd = 0.f;
        if (p.op == 0) { ga = gv; gb = gv; }
        else if (p.op == 1) { ga = (1.f - bv) * gv; gb = -av * gv; }
        else if (p.op == 2) { ga = (1.f - gg) * gv; gb = gg * gv; gd = (bv - av) * gv; }
        else ga = gv;
        if (p.da) p.da[r * p.das + p.daco + c] += ga;
        if (p.db) p.db[r * p.dbs + p.dbco + c] += gb;
        if (p.dg) p.dg[r * p.dgs + p.dgco + c] += gd;
    }
}

// LayerNorm over the first D columns of each row (tt_layernorm_rows) backward, one wave per row:
//   xh = (x - mean) * rstd,  dx += rstd * (g*gamma - mean_D(g*gamma) - xh * mean_D(g*gamma*xh)),
//   dgamma[i] += sum_rows g * xh,  dbeta[i] += sum_rows g   (per-workgroup partials [blocks][2][D], added in order by the
//   finish kernel: deterministic).
__global__ __launch_bounds__(256) void layernorm_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ dout, float* __restrict__ dx,
                                                                 float* __restrict__ partial, long long R, int D, int xs,
                                                                 int os, int dxs, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* pg = partial + (long long)blockIdx.x * 2 * D;
    for (int i = threadIdx.x; i < 2 * D; i += 256) pg[i] = 0.f;
    __syncthreads();
    // the 4 waves of the workgroup take rows blockIdx.x * 4 * RPW + ...; partial sums through global atomics WITHIN the block
    // would be non-deterministic, so each wave walks the block's rows in turn instead: wave w handles column slice, all rows
    const long long rows_per = (R + gridDim.x - 1) / gridDim.x;
    const long long r0 = (long long)blockIdx.x * rows_per, r1 = min(R, r0 + rows_per);
    for (long long row = r0; row < r1; ++row) {
        const float* xr = x + row * xs;
        const float* gr = dout + row * os;
        // statistics (every wave recomputes them: rows are short)
        float s = 0.f;
        for (int i = lane; i < D; i += 64) s += xr[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)D;
        float q = 0.f, a = 0.f, b = 0.f;
        for (int i = lane; i < D; i += 64) {
            const float d = xr[i] - mean;
            q += d * d;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.f / sqrtf(q / (float)D + eps);
        for (int i = lane; i < D; i += 64) {
            const float gg = gr[i] * gamma[i], xh = (xr[i] - mean) * rstd;
            a += gg;
            b += gg * xh;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o);
            b += __shfl_xor(b, o);
        }
        a /= (float)D;
        b /= (float)D;
        // wave w owns the columns i with (i / 64) % 4 == w: dx and the column sums are written by exactly one lane
        for (int i = lane + 64 * wave; i < D; i += 256) {
            const float xh = (xr[i] - mean) * rstd, g = gr[i];
            dx[row * dxs + i] += rstd * (g * gamma[i] - a - xh * b);
            pg[i] += g * xh;
            pg[D + i] += g;
        }
    }
}

__global__ __launch_bounds__(256) void layernorm_rows_bwd_finish_kernel(const float* __restrict__ partial, int blocks, int D,
                                                                        float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D) return;
    float a = 0.f, b = 0.f;
    for (int k = 0; k < blocks; ++k) {
        a += partial[(long long)k * 2 * D + i];
        b += partial[(long long)k * 2 * D + D + i];
    }
    dgamma[i] += a;
    dbeta[i] += b;
}

// tt_concat_rows backward for ONE piece: dsrc[row][c] += sum over the output rows r that read it of dout[r][coff + c]
// (row = (r / div) % mod, or r / div when mod == 0).  R is small (batch x 4 time steps): every thread walks the rows.
__global__ __launch_bounds__(256) void concat_piece_bwd_kernel(const float* __restrict__ dout, int out_stride, int coff,
                                                               long long R, int C, int div, int mod, float* __restrict__ dsrc,
                                                               int src_stride, int src_rows) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= src_rows * C) return;
    const int row = i / C, c = i % C;
    float s = 0.f;
    for (long long r = 0; r < R; ++r) {
        const long long q = r / div;
        if ((mod ? q % mod : q) == row) s += dout[r * out_stride + coff + c];
    }
    dsrc[(long long)row * src_stride + c] += s;
}

// tt_broadcast_rows backward: dv[n][c] += sum over the HW pixels of image n of dout[n][p][coff + c]
__global__ __launch_bounds__(256) void broadcast_rows_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dv, int N,
                                                                 int HW, int C, int cstride, int coff, int v_stride) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i % C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += dout[((long long)n * HW + p) * cstride + coff + c];
    dv[(long long)n * v_stride + c] += s;
}

// Lift-splat (softmax over the depth bins (x) context, scattered into the BEV cells; lss.py:583-632 + the voxel pooling
// op) backward, fused like the forward: for pixel (image, h, w) with probabilities p_d = softmax(logits)_d and the cell
// c(d) of its d-th frustum point,
//   dctx[c]    = sum_d p_d * gbev[cell(d)][c]
//   dp_d       = sum_c ctx[c] * gbev[cell(d)][c]
//   dlogits_d  = p_d * (dp_d - sum_d' p_d' dp_d')            (softmax)
// One wave per pixel; lanes hold 4 channels each (C <= 256), the per-bin dot products are wave reductions.  Every output is
// written by exactly one wave: deterministic, and dctx / dlogits are ACCUMULATED into the gradient buffers.
__global__ __launch_bounds__(256) void lift_splat_bwd_kernel(int B, int ncam, int D, int fH, int fW, int C, int X, int Y, int Z,
                                                             const float* __restrict__ logits, const float* __restrict__ ctx,
                                                             const int32_t* __restrict__ geom, const float* __restrict__ gbev,
                                                             int g_cstride, int g_coff, float* __restrict__ dlogits,
                                                             float* __restrict__ dctx) {
    __shared__ float prob[4][128], dprob[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long npix = (long long)B * ncam * fH * fW;
    const long long per_cam = (long long)D * fH * fW;
    for (long long pix = (long long)blockIdx.x * 4 + wave; pix < npix; pix += (long long)gridDim.x * 4) {
        const int w = (int)(pix % fW), h = (int)((pix / fW) % fH);
        const long long bc = pix / ((long long)fH * fW);
        const int b = (int)(bc / ncam);
        const float* lg = logits + pix * D;
        const float l0 = (lane < D) ? lg[lane] : -INFINITY, l1 = (lane + 64 < D) ? lg[lane + 64] : -INFINITY;
        float m = fmaxf(l0, l1);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        const float e0 = (lane < D) ? expf(l0 - m) : 0.f, e1 = (lane + 64 < D) ? expf(l1 - m) : 0.f;
        float ssum = e0 + e1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o);
        const float inv = 1.f / ssum;
        if (lane < D) prob[wave][lane] = e0 * inv;
        if (lane + 64 < D) prob[wave][lane + 64] = e1 * inv;
        const float4 cv = (lane * 4 < C) ? *reinterpret_cast<const float4*>(ctx + pix * C + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int32_t* g = geom + bc * per_cam * 3;
        for (int d = 0; d < D; ++d) {
            const int32_t* q = g + ((long long)d * fH * fW + (long long)h * fW + w) * 3;
            const int x = q[0], y = q[1], z = q[2];
            float dot = 0.f;
            if (x >= 0 && x < X && y >= 0 && y < Y && z >= 0 && z < Z) {          // wave-uniform
                const float* row = gbev + (((long long)b * Y + y) * X + x) * g_cstride + g_coff;
                const float4 gv = (lane * 4 < C) ? *reinterpret_cast<const float4*>(row + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float p = prob[wave][d];
                acc.x += p * gv.x; acc.y += p * gv.y; acc.z += p * gv.z; acc.w += p * gv.w;
                dot = cv.x * gv.x + cv.y * gv.y + cv.z * gv.z + cv.w * gv.w;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
            }
            if (lane == 0) dprob[wave][d] = dot;
        }
        if (lane * 4 < C) {
            float4* dc = reinterpret_cast<float4*>(dctx + pix * C + lane * 4);
            float4 t = *dc;
            t.x += acc.x; t.y += acc.y; t.z += acc.z; t.w += acc.w;
            *dc = t;
        }
        // softmax backward over the D bins
        const float p0 = (lane < D) ? prob[wave][lane] : 0.f, p1 = (lane + 64 < D) ? prob[wave][lane + 64] : 0.f;
        const float q0 = (lane < D) ? dprob[wave][lane] : 0.f, q1 = (lane + 64 < D) ? dprob[wave][lane + 64] : 0.f;
        float s = p0 * q0 + p1 * q1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane < D) dlogits[pix * D + lane] += p0 * (q0 - s);
        if (lane + 64 < D) dlogits[pix * D + lane + 64] += p1 * (q1 - s);
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
    if (C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0)
        hipLaunchKernelGGL(maxpool3x3s2_bwd_vec_kernel, dim3(bwd_grid((long long)N * H * W * (C / 4))), dim3(256), 0,
                           (hipStream_t)stream, x, dy, dx, N, H, W, C, OH, OW);
    else
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

extern "C" int tt_bilinear_up2_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    TT_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0, "tt_bilinear_up2_bwd: bad argument");
    hipLaunchKernelGGL(bilinear_up2_ac_bwd_kernel, dim3(bwd_grid((long long)N * H * W * C)), dim3(256), 0,
                       (hipStream_t)stream, dy, dx, N, H, W, C);
    return check_launch("tt_bilinear_up2_bwd");
}

extern "C" int tt_channel_gate_bwd(const float* x, const float* gate, const float* dy, float* dx, float* dgate, int N, int HW,
                                   int C, const float* out_relu_or_null, float* dres_or_null, void* stream) {
    TT_REQUIRE(x && gate && dy && dx && dgate && N > 0 && HW > 0 && C > 0, "tt_channel_gate_bwd: bad argument");
    hipLaunchKernelGGL(channel_gate_bwd_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)N), dim3(256), 0, (hipStream_t)stream,
                       x, gate, dy, dx, dgate, HW, C, out_relu_or_null, dres_or_null);
    return check_launch("tt_channel_gate_bwd");
}

extern "C" int tt_spatial_meanmax_bwd(const float* x, const float* dpool, float* dx, int N, int HW, int C, void* stream) {
    TT_REQUIRE(x && dpool && dx && N > 0 && HW > 0 && C > 0, "tt_spatial_meanmax_bwd: bad argument");
    hipLaunchKernelGGL(spatial_meanmax_bwd_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)N), dim3(256), 0,
                       (hipStream_t)stream, x, dpool, dx, HW, C);
    return check_launch("tt_spatial_meanmax_bwd");
}

extern "C" int tt_spatial_mean_bwd(const float* dpool, float* dx, int N, int HW, int C, int cstride, int coff, void* stream) {
    TT_REQUIRE(dpool && dx && N > 0 && HW > 0 && C > 0 && cstride >= coff + C, "tt_spatial_mean_bwd: bad argument");
    hipLaunchKernelGGL(spatial_mean_bwd_kernel, dim3(bwd_grid((long long)N * HW * C)), dim3(256), 0, (hipStream_t)stream, dpool,
                       dx, N, HW, C, cstride, coff);
    return check_launch("tt_spatial_mean_bwd");
}

extern "C" int tt_deform_im2col3x3_bwd(const float* x, const float* offsets, const float* gcols, float* gx, float* goffsets,
                                       int N, int H, int W, int C, int off_cstride, int pad, void* stream) {
    TT_REQUIRE(x && offsets && gcols && gx && goffsets && off_cstride >= 18, "tt_deform_im2col3x3_bwd: bad argument");
    hipLaunchKernelGGL(deform_im2col_bwd_kernel, dim3(bwd_grid((long long)N * H * W * 9 * 64)), dim3(256), 0,
                       (hipStream_t)stream, x, offsets, gcols, gx, goffsets, N, H, W, C, off_cstride, pad);
    return check_launch("tt_deform_im2col3x3_bwd");
}

extern "C" int tt_lift_splat_bwd(int batch_size, int num_cams, int D, int fH, int fW, int C, int num_voxel_x, int num_voxel_y,
                                 int num_voxel_z, const float* depth_logits, const float* context, const int32_t* geom_xyz,
                                 const float* grad_out, int out_cstride, int out_coff, float* grad_depth_logits,
                                 float* grad_context, void* stream) {
    TT_REQUIRE(depth_logits && context && geom_xyz && grad_out && grad_depth_logits && grad_context,
               "tt_lift_splat_bwd: null pointer");
    TT_REQUIRE(D <= 128 && C % 4 == 0 && C <= 256 && out_cstride % 4 == 0 && out_coff % 4 == 0,
               "tt_lift_splat_bwd: D <= 128, C <= 256, 16-byte channel windows");
    const long long npix = (long long)batch_size * num_cams * fH * fW;
    long long blocks = (npix + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(lift_splat_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, batch_size, num_cams,
                       D, fH, fW, C, num_voxel_x, num_voxel_y, num_voxel_z, depth_logits, context, geom_xyz, grad_out,
                       out_cstride, out_coff, grad_depth_logits, grad_context);
    return check_launch("tt_lift_splat_bwd");
}

extern "C" int tt_ew_bwd(int op, int act, long long R, int C, const float* a, int a_stride, int a_coff, const float* b,
                         int b_stride, int b_coff, const float* g, int g_stride, int g_coff, const float* out, int o_stride,
                         int o_coff, const float* dout, int d_stride, int d_coff, float* da, int da_stride, int da_coff,
                         float* db, int db_stride, int db_coff, float* dg, int dg_stride, int dg_coff, void* stream) {
    TT_REQUIRE(a && dout && R > 0 && C > 0 && op >= 0 && op <= 3, "tt_ew_bwd: bad argument");
    TT_REQUIRE(act == TT_ACT_NONE || ((act == TT_ACT_RELU || act == TT_ACT_SIGMOID || act == TT_ACT_SOFTPLUS || act == TT_ACT_SOFTPLUS_CLAMP) && out),
               "tt_ew_bwd: activation %d needs the pre-activation (or the saved output is missing)", act);
    TT_REQUIRE(op == 3 || b, "tt_ew_bwd: op %d needs b", op);
    TT_REQUIRE(op != 2 || g, "tt_ew_bwd: op 2 needs g");
    EwBwdArgs p;
    p.a = a; p.b = b; p.g = g; p.out = out; p.dout = dout; p.da = da; p.db = db; p.dg = dg;
    p.R = R; p.C = C; p.as = a_stride; p.aco = a_coff; p.bs = b_stride; p.bco = b_coff; p.gs = g_stride; p.gco = g_coff;
    p.os = o_stride; p.oco = o_coff; p.ds = d_stride; p.dco = d_coff; p.das = da_stride; p.daco = da_coff;
    p.dbs = db_stride; p.dbco = db_coff; p.dgs = dg_stride; p.dgco = dg_coff; p.op = op; p.act = act;
    hipLaunchKernelGGL(ew_bwd_kernel, dim3(bwd_grid(R * C)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("tt_ew_bwd");
}

extern "C" int tt_broadcast_rows_bwd(const float* dout, float* dv, int N, int HW, int C, int cstride, int coff, int v_stride,
                                     void* stream) {
    TT_REQUIRE(dout && dv && N > 0 && HW > 0 && C > 0, "tt_broadcast_rows_bwd: bad argument");
    hipLaunchKernelGGL(broadcast_rows_bwd_kernel, dim3((unsigned)((N * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout,
                       dv, N, HW, C, cstride, coff, v_stride);
    return check_launch("tt_broadcast_rows_bwd");
}

extern "C" long long tt_layernorm_rows_bwd_workspace_bytes(long long R, int D) {
    const long long blocks = R < 256 ? R : 256;
    return blocks * 2 * D * 4;
}

extern "C" int tt_layernorm_rows_bwd(const float* x, const float* gamma, const float* dout, float* dx, float* dgamma,
                                     float* dbeta, long long R, int D, int x_stride, int dout_stride, int dx_stride, float eps,
                                     void* workspace, long long workspace_bytes, void* stream) {
    TT_REQUIRE(x && gamma && dout && dx && dgamma && dbeta && workspace && R > 0 && D > 0, "tt_layernorm_rows_bwd: bad argument");
    TT_REQUIRE(workspace_bytes >= tt_layernorm_rows_bwd_workspace_bytes(R, D), "tt_layernorm_rows_bwd: workspace too small");
    const int blocks = (int)(R < 256 ? R : 256);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(layernorm_rows_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, dout, dx,
                       (float*)workspace, R, D, x_stride, dout_stride, dx_stride, eps);
    hipLaunchKernelGGL(layernorm_rows_bwd_finish_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, st,
                       (const float*)workspace, blocks, D, dgamma, dbeta);
    return check_launch("tt_layernorm_rows_bwd");
}

extern "C" int tt_concat_piece_bwd(const float* dout, int out_stride, int coff, long long R, int C, int div, int mod,
                                   float* dsrc, int src_stride, int src_rows, void* stream) {
    TT_REQUIRE(dout && dsrc && R > 0 && C > 0 && div >= 1 && mod >= 0 && src_rows > 0, "tt_concat_piece_bwd: bad argument");
    hipLaunchKernelGGL(concat_piece_bwd_kernel, dim3((unsigned)((src_rows * C + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, dout, out_stride, coff, R, C, div, mod, dsrc, src_stride, src_rows);
    return check_launch("tt_concat_piece_bwd");
}
