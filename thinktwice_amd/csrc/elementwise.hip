// HBM-bound glue kernels of the forward path (channel-last, 16-byte vectors per lane).
// Each replaces a torch elementwise / pooling / resize call of the reference forward; the
// citing comment names the call site.  T = float or uint16_t (bf16 storage); math in f32.
#include "tt_common.h"

namespace tt {

template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <typename T16> struct Vec16 {     // bf16 (uint16_t) or IEEE half (f16_t) storage, 8 elements per 16 B
    static constexpr int N = 8;
    float v[8];
    __device__ __forceinline__ void load(const T16* p) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) Pair16<T16>::unpack(w[i], v[2 * i], v[2 * i + 1]);
    }
    __device__ __forceinline__ void store(T16* p) const {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = Pair16<T16>::pack(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <> struct Vec<uint16_t> : Vec16<uint16_t> {};
template <> struct Vec<f16_t> : Vec16<f16_t> {};

#define TT_GRID_STRIDE(i, n) \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

static inline unsigned grid_for(long long n) {
    long long b = (n + 255) / 256;
    if (b > 256LL * 32) b = 256LL * 32;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ---- NCHW f32 image -> channel-last, channels zero-padded to Cp (LSS.get_cam_feats input, lss.py:517-519)
template <typename T>
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict__ in, T* __restrict__ out, long long NHW,
                                        int HW, int C, int Cp) {
    TT_GRID_STRIDE(i, NHW) {
        const long long n = i / HW;
        const int p = (int)(i - n * HW);
        for (int c = 0; c < Cp; ++c) {
            const float v = (c < C) ? in[(n * C + c) * HW + p] : 0.f;
            Elem<T>::st(out + i * Cp + c, v);
        }
    }
}

// ---- same, into the interior of a spatially padded channel-last buffer [N][Hp][Wp][Cp] at (top, left): the
// row-run stem conv (lss.py) reads 8-pixel runs that overhang the image, so the zero border is physical
template <typename T>
__global__ void nchw_to_nhwc_border_kernel(const float* __restrict__ in, T* __restrict__ out, long long NHW,
                                           int H, int W, int C, int Cp, int Hp, int Wp, int top, int left) {
    const int HW = H * W;
    TT_GRID_STRIDE(i, NHW) {
        const long long n = i / HW;
        const int p = (int)(i - n * HW);
        const int h = p / W, w = p - h * W;
        T* o = out + (((long long)n * Hp + h + top) * Wp + w + left) * Cp;
        for (int c = 0; c < Cp; ++c) Elem<T>::st(o + c, (c < C) ? in[(n * C + c) * HW + p] : 0.f);
    }
}

// ---- channel-last -> NCHW f32 (outputs handed back in the reference's layout)
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, long long total,
                                    int HW, int C, int cstride, int coff) {
    TT_GRID_STRIDE(i, total) {          // i over [n][c][p]
        const int p = (int)(i % HW);
        const long long nc = i / HW;
        const int c = (int)(nc % C);
        const long long n = nc / C;
        out[i] = Elem<T>::ld(in + (n * HW + p) * cstride + coff + c);
    }
}

// ---- F.max_pool2d(x, 3, 2, 1)  (mmdet ResNet stem)
template <typename T>
__global__ void maxpool3x3s2_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W,
                                    int C, int OH, int OW) {
    constexpr int V = Vec<T>::N;
    const int cv = C / V;
    const long long total = (long long)N * OH * OW * cv;
    TT_GRID_STRIDE(i, total) {
        const int c = (int)(i % cv);
        long long r = i / cv;
        const int ow = (int)(r % OW); r /= OW;
        const int oh = (int)(r % OH);
        const long long n = r / OH;
        Vec<T> m;
#pragma unroll
        for (int k = 0; k < V; ++k) m.v[k] = -INFINITY;
        for (int dh = 0; dh < 3; ++dh) {
            const int ih = oh * 2 - 1 + dh;
            if (ih < 0 || ih >= H) continue;
            for (int dw = 0; dw < 3; ++dw) {
                const int iw = ow * 2 - 1 + dw;
                if (iw < 0 || iw >= W) continue;
                Vec<T> x;
                x.load(in + ((n * H + ih) * W + iw) * C + c * V);
#pragma unroll
                for (int k = 0; k < V; ++k) m.v[k] = fmaxf(m.v[k], x.v[k]);
            }
        }
        m.store(out + i * V);
    }
}

// ---- dst += nearest_upsample(src)  (PAFPN top-down path, lss.py:301-305)
template <typename T>
__global__ void upsample_nearest_add_kernel(T* __restrict__ dst, const T* __restrict__ src, int N, int H,
                                            int W, int C, int h, int w) {
    constexpr int V = Vec<T>::N;
    const int cv = C / V;
    const long long total = (long long)N * H * W * cv;
    TT_GRID_STRIDE(i, total) {
        const int c = (int)(i % cv);
        long long r = i / cv;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const long long n = r / H;
        const int sy = (int)(((long long)y * h) / H), sx = (int)(((long long)x * w) / W);
        Vec<T> a, b;
        a.load(dst + i * V);
        b.load(src + ((n * h + sy) * w + sx) * C + c * V);
#pragma unroll
        for (int k = 0; k < V; ++k) a.v[k] += b.v[k];
        a.store(dst + i * V);
    }
}

// ---- bilinear x2, align_corners=True (UNet.unet_layer0 nn.Upsample, lss.py:267)
template <typename T>
__global__ void bilinear_up2_ac_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W,
                                       int C) {
    constexpr int V = Vec<T>::N;
    const int cv = C / V, OH = 2 * H, OW = 2 * W;
    const float sh = (OH > 1) ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sw = (OW > 1) ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const long long total = (long long)N * OH * OW * cv;
    TT_GRID_STRIDE(i, total) {
        const int c = (int)(i % cv);
        long long r = i / cv;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const long long n = r / OH;
        const float fy = sh * oy, fx = sw * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = fy - y0, lx = fx - x0;
        Vec<T> a, b, cc, d, o;
        const T* base = in + n * H * W * C + c * V;
        a.load(base + ((long long)y0 * W + x0) * C);
        b.load(base + ((long long)y0 * W + x1) * C);
        cc.load(base + ((long long)y1 * W + x0) * C);
        d.load(base + ((long long)y1 * W + x1) * C);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            // same operation order as ATen's upsample_bilinear2d: ((1-ly)*((1-lx)*a + lx*b) + ly*(...))
            const float top = (1.f - lx) * a.v[k] + lx * b.v[k];
            const float bot = (1.f - lx) * cc.v[k] + lx * d.v[k];
            o.v[k] = (1.f - ly) * top + ly * bot;
        }
        o.store(out + i * V);
    }
}

// ---- the same, f32 -> bf16x3 PAIR format (tt_conv_desc.in_pair): a thread owns 8 channels and writes their bf16 hi halves into
// the first 32 B of the 16-channel group (second 16 B for channels 8-15), the lo halves 32 B further -- hi = rne(v),
// lo = rne(v - hi): the split the consuming convolution would otherwise redo for every tap and column tile
__global__ void bilinear_up2_ac_pair_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W, int C) {
    const int cv = C / 8, OH = 2 * H, OW = 2 * W;
    const float sh = (OH > 1) ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sw = (OW > 1) ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const long long total = (long long)N * OH * OW * cv;
    TT_GRID_STRIDE(i, total) {
        const int c = (int)(i % cv);
        long long r = i / cv;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const long long n = r / OH;
        const float fy = sh * oy, fx = sw * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = fy - y0, lx = fx - x0;
        const float* base = in + n * H * W * C + c * 8;
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            Vec<float> a, b, cc, d;
            a.load(base + ((long long)y0 * W + x0) * C + 4 * h);
            b.load(base + ((long long)y0 * W + x1) * C + 4 * h);
            cc.load(base + ((long long)y1 * W + x0) * C + 4 * h);
            d.load(base + ((long long)y1 * W + x1) * C + 4 * h);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float top = (1.f - lx) * a.v[k] + lx * b.v[k];
                const float bot = (1.f - lx) * cc.v[k] + lx * d.v[k];
                v[4 * h + k] = (1.f - ly) * top + ly * bot;
            }
        }
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
            lo[e] = pack_bf16x2(v[2 * e] - __uint_as_float(hi[e] << 16), v[2 * e + 1] - __uint_as_float(hi[e] & 0xffff0000u));
        }
        float* g = out + (i / cv) * C + (c >> 1) * 16 + (c & 1) * 4;        // pixel row, 16-channel group, 8-channel half
        *reinterpret_cast<uint4*>(g) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(g + 8) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}

// ---- per-(image, channel) spatial reductions: mode 0 = mean (AdaptiveAvgPool2d(1), lss.py:80),
//      mode 1 = 0.5*mean + 0.5*max (SEModule pooling, code/utils.py:91-92).  One block per image
//      and 64-channel slab; out f32 [N, C].
template <typename T>
__global__ __launch_bounds__(256) void spatial_pool_kernel(const T* __restrict__ in, float* __restrict__ out,
                                                           int HW, int C, int cstride, int coff, int mode) {
    const int n = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;  // 4 row-partitions
    float s = 0.f, m = -INFINITY;
    if (c < C) {
        const T* base = in + (long long)n * HW * cstride + coff + c;
        // eight rows in flight per thread (the loop is one dependent 256 B load per trip otherwise: latency-bound)
        int p = part;
        for (; p + 28 < HW; p += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = Elem<T>::ld(base + (long long)(p + 4 * u) * cstride);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s += v[u];
                m = fmaxf(m, v[u]);
            }
        }
        for (; p < HW; p += 4) {
            const float v = Elem<T>::ld(base + (long long)p * cstride);
            s += v;
            m = fmaxf(m, v);
        }
    }
    __shared__ float ss[4][64], sm[4][64];
    ss[part][threadIdx.x & 63] = s;
    sm[part][threadIdx.x & 63] = m;
    __syncthreads();
    if (part == 0 && c < C) {
        const int l = threadIdx.x & 63;
        const float st = ss[0][l] + ss[1][l] + ss[2][l] + ss[3][l];
        const float mt = fmaxf(fmaxf(sm[0][l], sm[1][l]), fmaxf(sm[2][l], sm[3][l]));
        const float mean = st / (float)HW;
        out[(long long)n * C + c] = (mode == 0) ? mean : 0.5f * mean + 0.5f * mt;
    }
}

// ---- out = act2( x * act1(gate[n, c]) + res )   SELayer (lss.py:158) / SEModule + residual (utils.py:96,117-119)
template <typename T>
__global__ void channel_gate_kernel(const T* __restrict__ x, const float* __restrict__ gate,
                                    const T* __restrict__ res, T* __restrict__ out, long long NHW, int HW,
                                    int C, int gate_act, int out_act) {
    constexpr int V = Vec<T>::N;
    const int cv = C / V;
    const long long total = NHW * cv;
    TT_GRID_STRIDE(i, total) {
        const int c = (int)(i % cv);
        const long long pix = i / cv;
        const long long n = pix / HW;
        Vec<T> a, r;
        a.load(x + i * V);
        if (res) r.load(res + i * V);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float g = apply_act(gate[n * C + c * V + k], gate_act);
            float v = a.v[k] * g;
            if (res) v += r.v[k];
            a.v[k] = apply_act(v, out_act);
        }
        a.store(out + i * V);
    }
}

// ---- rows: out[r, c] = act(x[r, c] * scale[c] + shift[c])  (BatchNorm1d eval: lss.py:232, EDF:134)
template <typename T>
__global__ void affine_rows_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                   const float* __restrict__ shift, T* __restrict__ out, long long R, int C,
                                   int xs, int os, int act) {
    const long long total = R * C;
    TT_GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const long long r = i / C;
        float v = Elem<T>::ld(x + r * xs + c);
        v = v * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
        Elem<T>::st(out + r * os + c, apply_act(v, act));
    }
}

// ---- LayerNorm over the last dim, one wave per row (MSDA:252,201,262; DEC:197)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const T* __restrict__ x,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, T* __restrict__ out,
                                                             long long R, int D, int xs, int os, float eps) {
    const long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const T* xr = x + row * xs;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += Elem<T>::ld(xr + i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)D;
    float q = 0.f;
    for (int i = lane; i < D; i += 64) {
        const float d = Elem<T>::ld(xr + i) - mean;
        q += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.f / sqrtf(q / (float)D + eps);
    T* orow = out + row * os;
    for (int i = lane; i < D; i += 64) {
        const float v = (Elem<T>::ld(xr + i) - mean) * rstd * gamma[i] + beta[i];
        Elem<T>::st(orow + i, v);
    }
}

// ---- generic strided 2-D copy with optional channel offsets (concat assembly; rot90(flip) of BEV maps
//      EDF:241,246 when rot_flip != 0: out[i][j] = in[H-1-j][W-1-i], square maps)
template <typename TI, typename TO>
__global__ void copy_nhwc_kernel(const TI* __restrict__ in, TO* __restrict__ out, int N, int H, int W, int C,
                                 int ics, int ico, int ocs, int oco, int rot_flip) {
    const long long total = (long long)N * H * W * C;
    TT_GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const long long n = r / H;
        int sy = y, sx = x;
        if (rot_flip) { sy = H - 1 - x; sx = W - 1 - y; }
        const float v = Elem<TI>::ld(in + ((n * H + sy) * W + sx) * ics + ico + c);
        Elem<TO>::st(out + ((n * H + y) * W + x) * ocs + oco + c, v);
    }
}

// ---- broadcast rows over a spatial map: out[n, p, oco + c] = v[n, c]  (DEC:257 all_future_feat repeat)
template <typename T>
__global__ void broadcast_rows_kernel(const T* __restrict__ v, T* __restrict__ out, int N, int HW, int C,
                                      int vs, int ocs, int oco) {
    const long long total = (long long)N * HW * C;
    TT_GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const long long np = i / C;
        const long long n = np / HW;
        out[np * ocs + oco + c] = v[n * vs + c];
    }
}

// ---- binary / ternary elementwise ops used by the conv-GRU cell and residual updates (DHU:93-106):
//   op 0: out = a + b                 op 1: out = (1 - b) * a            (reset gate applied to state)
//   op 2: out = (1 - g) * a + g * b   (GRU blend, g = update gate)       op 3: out = act(a)
template <typename T>
__global__ void ew_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ g,
                          T* __restrict__ out, long long R, int C, int as, int aco, int bs, int bco, int gs,
                          int gco, int os, int oco, int op, int act) {
    const long long total = R * C;
    TT_GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const long long r = i / C;
        const float av = Elem<T>::ld(a + r * as + aco + c);
        float v;
        if (op == 0) {
            v = av + Elem<T>::ld(b + r * bs + bco + c);
        } else if (op == 1) {
            v = (1.f - Elem<T>::ld(b + r * bs + bco + c)) * av;
        } else if (op == 2) {
            const float gv = Elem<T>::ld(g + r * gs + gco + c);
            v = (1.f - gv) * av + gv * Elem<T>::ld(b + r * bs + bco + c);
        } else {
            v = av;
        }
        Elem<T>::st(out + r * os + oco + c, apply_act(v, act));
    }
}

// ---- torch.cat([...], -1) of row-batched f32 pieces with per-piece row mapping, in ONE launch.
// Piece s fills out[r, coff_s : coff_s + C_s] from src_s[((r / div_s) % mod_s), :C_s] (mod_s = 0: no modulo;
// src_s = null: zeros): a plain copy (div 1), a per-sample vector broadcast over the 4 time steps (div 4) or a
// per-time-step embedding (mod 4).  The decoder assembles its MLP inputs from 2-14 such pieces per layer
// (thinktwice_decoder.py:236-260); one launch each instead of one per piece.
struct CatSeg {
    const float* src;
    int stride, C, coff, div, mod;
};
struct CatArgs {
    CatSeg seg[8];
    int nseg;
};

__global__ void concat_rows_kernel(CatArgs a, float* __restrict__ out, long long R, int out_stride, int total_c) {
    const long long total = R * total_c;
    TT_GRID_STRIDE(i, total) {
        const long long r = i / total_c;
        int c = (int)(i - r * total_c);
        // pieces are given in output-column order and tile [first coff, first coff + total_c)
        int s = 0;
        while (s + 1 < a.nseg && c >= a.seg[s].C) {
            c -= a.seg[s].C;
            ++s;
        }
        const CatSeg sg = a.seg[s];
        float v = 0.f;
        if (sg.src) {
            long long sr = r / sg.div;
            if (sg.mod > 0) sr %= sg.mod;
            v = sg.src[sr * sg.stride + c];
        }
        out[r * out_stride + sg.coff + c] = v;
    }
}

}  // namespace tt

using namespace tt;

#define TT_DISPATCH(dtype, CALL)                              \
    do {                                                      \
        if ((dtype) == TT_F32) { using T = float; CALL; }     \
        else if ((dtype) == TT_BF16) { using T = uint16_t; CALL; } \
        else if ((dtype) == TT_F16) { using T = f16_t; CALL; }    \
        else { TT_REQUIRE(false, "bad dtype %d", (int)(dtype)); }  \
    } while (0)

extern "C" int tt_nchw_to_nhwc_pad(const float* in, void* out, int N, int C, int H, int W, int Cp,
                                   int out_dtype, void* stream) {
    TT_REQUIRE(in && out && Cp >= C, "tt_nchw_to_nhwc_pad: bad args");
    const long long NHW = (long long)N * H * W;
    TT_DISPATCH(out_dtype, hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<T>, dim3(grid_for(NHW)), dim3(256), 0,
                                              (hipStream_t)stream, in, (T*)out, NHW, H * W, C, Cp));
    return check_launch("tt_nchw_to_nhwc_pad");
}

extern "C" int tt_nchw_to_nhwc_border(const float* in, void* out, int N, int C, int H, int W, int Cp, int Hp, int Wp,
                                      int top, int left, int out_dtype, void* stream) {
    TT_REQUIRE(in && out && Cp >= C && top >= 0 && left >= 0 && Hp >= H + top && Wp >= W + left,
               "tt_nchw_to_nhwc_border: bad args");
    const long long NHW = (long long)N * H * W;
    TT_DISPATCH(out_dtype, hipLaunchKernelGGL(nchw_to_nhwc_border_kernel<T>, dim3(grid_for(NHW)), dim3(256), 0,
                                              (hipStream_t)stream, in, (T*)out, NHW, H, W, C, Cp, Hp, Wp, top, left));
    return check_launch("tt_nchw_to_nhwc_border");
}

extern "C" int tt_nhwc_to_nchw(const void* in, float* out, int N, int C, int H, int W, int cstride, int coff,
                               int in_dtype, void* stream) {
    TT_REQUIRE(in && out, "tt_nhwc_to_nchw: null");
    const long long total = (long long)N * C * H * W;
    TT_DISPATCH(in_dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(grid_for(total)), dim3(256), 0,
                                             (hipStream_t)stream, (const T*)in, out, total, H * W, C, cstride, coff));
    return check_launch("tt_nhwc_to_nchw");
}

static inline int vec_of(int dtype) { return dtype == TT_F32 ? 4 : 8; }

extern "C" int tt_maxpool3x3s2(const void* in, void* out, int N, int H, int W, int C, int dtype, void* stream) {
    TT_REQUIRE(in && out && C % vec_of(dtype) == 0, "tt_maxpool3x3s2: C=%d must be a multiple of %d", C, vec_of(dtype));
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * OH * OW * (C / vec_of(dtype));
    TT_DISPATCH(dtype, hipLaunchKernelGGL(maxpool3x3s2_kernel<T>, dim3(grid_for(total)), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)in, (T*)out, N, H, W, C, OH, OW));
    return check_launch("tt_maxpool3x3s2");
}

extern "C" int tt_upsample_nearest_add(void* dst, const void* src, int N, int H, int W, int C, int h, int w,
                                       int dtype, void* stream) {
    TT_REQUIRE(dst && src && C % vec_of(dtype) == 0, "tt_upsample_nearest_add: bad args");
    const long long total = (long long)N * H * W * (C / vec_of(dtype));
    TT_DISPATCH(dtype, hipLaunchKernelGGL(upsample_nearest_add_kernel<T>, dim3(grid_for(total)), dim3(256), 0,
                                          (hipStream_t)stream, (T*)dst, (const T*)src, N, H, W, C, h, w));
    return check_launch("tt_upsample_nearest_add");
}

extern "C" int tt_bilinear_up2(const void* in, void* out, int N, int H, int W, int C, int dtype, void* stream) {
    TT_REQUIRE(in && out && C % vec_of(dtype) == 0, "tt_bilinear_up2: bad args");
    const long long total = (long long)N * 4 * H * W * (C / vec_of(dtype));
    TT_DISPATCH(dtype, hipLaunchKernelGGL(bilinear_up2_ac_kernel<T>, dim3(grid_for(total)), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)in, (T*)out, N, H, W, C));
    return check_launch("tt_bilinear_up2");
}

extern "C" int tt_bilinear_up2_pair(const float* in, float* out, int N, int H, int W, int C, void* stream) {
    TT_REQUIRE(in && out && C % 16 == 0 && (reinterpret_cast<uintptr_t>(out) & 63) == 0, "tt_bilinear_up2_pair: C %% 16, 64 B aligned out");
    const long long total = (long long)N * 4 * H * W * (C / 8);
    hipLaunchKernelGGL(bilinear_up2_ac_pair_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, out, N, H, W, C);
    return check_launch("tt_bilinear_up2_pair");
}

extern "C" int tt_spatial_pool(const void* in, float* out, int N, int HW, int C, int cstride, int coff, int mode,
                               int dtype, void* stream) {
    TT_REQUIRE(in && out && N > 0 && HW > 0, "tt_spatial_pool: bad args");
    TT_DISPATCH(dtype, hipLaunchKernelGGL(spatial_pool_kernel<T>, dim3((C + 63) / 64, N), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)in, out, HW, C, cstride, coff, mode));
    return check_launch("tt_spatial_pool");
}

extern "C" int tt_channel_gate(const void* x, const float* gate, const void* res, void* out, int N, int HW, int C,
                               int gate_act, int out_act, int dtype, void* stream) {
    TT_REQUIRE(x && gate && out && C % vec_of(dtype) == 0, "tt_channel_gate: bad args");
    const long long NHW = (long long)N * HW;
    const long long total = NHW * (C / vec_of(dtype));
    TT_DISPATCH(dtype, hipLaunchKernelGGL(channel_gate_kernel<T>, dim3(grid_for(total)), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)x, gate, (const T*)res, (T*)out, NHW, HW,
                                          C, gate_act, out_act));
    return check_launch("tt_channel_gate");
}

extern "C" int tt_affine_rows(const void* x, const float* scale, const float* shift, void* out, long long R, int C,
                              int xs, int os, int act, int dtype, void* stream) {
    TT_REQUIRE(x && out, "tt_affine_rows: null");
    TT_DISPATCH(dtype, hipLaunchKernelGGL(affine_rows_kernel<T>, dim3(grid_for(R * C)), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)x, scale, shift, (T*)out, R, C, xs, os, act));
    return check_launch("tt_affine_rows");
}

extern "C" int tt_layernorm_rows(const void* x, const float* gamma, const float* beta, void* out, long long R,
                                 int D, int xs, int os, float eps, int dtype, void* stream) {
    TT_REQUIRE(x && gamma && beta && out, "tt_layernorm_rows: null");
    TT_DISPATCH(dtype, hipLaunchKernelGGL(layernorm_rows_kernel<T>, dim3((unsigned)div_up(R, 4)), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)x, gamma, beta, (T*)out, R, D, xs, os, eps));
    return check_launch("tt_layernorm_rows");
}

extern "C" int tt_copy_nhwc(const void* in, void* out, int N, int H, int W, int C, int in_cstride, int in_coff,
                            int out_cstride, int out_coff, int rot_flip, int in_dtype, int out_dtype, void* stream) {
    TT_REQUIRE(in && out, "tt_copy_nhwc: null");
    TT_REQUIRE(!rot_flip || H == W, "tt_copy_nhwc: rot_flip needs a square map");
    const long long total = (long long)N * H * W * C;
    hipStream_t st = (hipStream_t)stream;
#define CP(TI, TO)                                                                                       \
    hipLaunchKernelGGL((copy_nhwc_kernel<TI, TO>), dim3(grid_for(total)), dim3(256), 0, st, (const TI*)in, \
                       (TO*)out, N, H, W, C, in_cstride, in_coff, out_cstride, out_coff, rot_flip)
    if (in_dtype == TT_F32 && out_dtype == TT_F32) CP(float, float);
    else if (in_dtype == TT_F32 && out_dtype == TT_BF16) CP(float, uint16_t);
    else if (in_dtype == TT_BF16 && out_dtype == TT_F32) CP(uint16_t, float);
    else if (in_dtype == TT_BF16 && out_dtype == TT_BF16) CP(uint16_t, uint16_t);
    else if (in_dtype == TT_F32 && out_dtype == TT_F16) CP(float, f16_t);
    else if (in_dtype == TT_F16 && out_dtype == TT_F32) CP(f16_t, float);
    else if (in_dtype == TT_F16 && out_dtype == TT_F16) CP(f16_t, f16_t);
    else TT_REQUIRE(false, "tt_copy_nhwc: bad dtypes");
#undef CP
    return check_launch("tt_copy_nhwc");
}

extern "C" int tt_broadcast_rows(const void* v, void* out, int N, int HW, int C, int v_stride, int out_cstride,
                                 int out_coff, int dtype, void* stream) {
    TT_REQUIRE(v && out, "tt_broadcast_rows: null");
    const long long total = (long long)N * HW * C;
    TT_DISPATCH(dtype, hipLaunchKernelGGL(broadcast_rows_kernel<T>, dim3(grid_for(total)), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)v, (T*)out, N, HW, C, v_stride,
                                          out_cstride, out_coff));
    return check_launch("tt_broadcast_rows");
}

extern "C" int tt_ew(const void* a, const void* b, const void* g, void* out, long long R, int C, int a_stride,
                     int a_coff, int b_stride, int b_coff, int g_stride, int g_coff, int o_stride, int o_coff,
                     int op, int act, int dtype, void* stream) {
    TT_REQUIRE(a && out && op >= 0 && op <= 3, "tt_ew: bad args");
    TT_REQUIRE(op == 3 || b, "tt_ew: op %d needs b", op);
    TT_REQUIRE(op != 2 || g, "tt_ew: op 2 needs g");
    TT_DISPATCH(dtype, hipLaunchKernelGGL(ew_kernel<T>, dim3(grid_for(R * C)), dim3(256), 0, (hipStream_t)stream,
                                          (const T*)a, (const T*)b, (const T*)g, (T*)out, R, C, a_stride, a_coff,
                                          b_stride, b_coff, g_stride, g_coff, o_stride, o_coff, op, act));
    return check_launch("tt_ew");
}

extern "C" int tt_concat_rows(float* out, long long R, int out_stride, int nseg, const float* const* srcs,
                              const int* strides, const int* widths, const int* coffs, const int* divs,
                              const int* mods, void* stream) {
    TT_REQUIRE(out && R > 0 && nseg >= 1 && nseg <= 8 && strides && widths && coffs && divs && mods && srcs,
               "tt_concat_rows: bad args (1..8 pieces)");
    CatArgs a;
    a.nseg = nseg;
    int total_c = 0;
    for (int s = 0; s < nseg; ++s) {
        TT_REQUIRE(widths[s] > 0 && divs[s] >= 1 && mods[s] >= 0 && coffs[s] >= 0, "tt_concat_rows: piece %d", s);
        TT_REQUIRE(s == 0 || coffs[s] == coffs[s - 1] + widths[s - 1],
                   "tt_concat_rows: pieces must be contiguous in output-column order (piece %d)", s);
        a.seg[s] = CatSeg{srcs[s], strides[s], widths[s], coffs[s], divs[s], mods[s]};
        total_c += widths[s];
    }
    TT_REQUIRE(coffs[0] + total_c <= out_stride, "tt_concat_rows: pieces exceed the output row");
    hipLaunchKernelGGL(concat_rows_kernel, dim3(grid_for(R * total_c)), dim3(256), 0, (hipStream_t)stream, a, out, R,
                       out_stride, total_c);
    return check_launch("tt_concat_rows");
}
